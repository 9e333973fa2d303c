"""``cond_stage_model`` of the reference yaml: ``ldm.modules.encoders.modules.FrozenCLIPEmbedder``
(diffmodels/base_diffusion_makeup.yaml:109-110), used by ``get_learned_conditioning`` (diffmk/makeup_teacher.py:33-42)
and ``get_unconditional_conditioning`` (diffmk/diffusion_makeup.py:400).

UPSTREAM: CLIPTokenizer(text, truncation=True, max_length=77, padding='max_length') -> CLIPTextModel(input_ids)
.last_hidden_state.  Here the tokenizer is the same transformers class loaded from a LOCAL directory (vocab.json +
merges.txt; nothing is downloaded — none ships with this repository), and the transformer runs in libmkd
(``mkd_clip_encode``).  Without tokenizer files, token ids can be passed directly (``encode_tokens``)."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .engine import MkdEngine


def load_tokenizer(path: str):
    """transformers.CLIPTokenizer from local files only."""
    if not os.path.isdir(path):
        raise FileNotFoundError(f'CLIP tokenizer directory not found: {path} (vocab.json + merges.txt are needed; no network)')
    from transformers import CLIPTokenizer
    return CLIPTokenizer.from_pretrained(path, local_files_only=True)


class FrozenCLIPEmbedder:
    def __init__(self, engine: MkdEngine, tokenizer: Optional[Callable] = None, max_length: int = 77):
        if engine.clip_cfg is None:
            raise ValueError('engine has no text encoder configured (MkdEngine.configure_clip)')
        self.engine, self.tokenizer, self.max_length = engine, tokenizer, max_length
        self._cache: Dict[Tuple[str, ...], torch.Tensor] = {}

    def tokenize(self, text: Sequence[str]) -> torch.Tensor:
        if self.tokenizer is None:
            raise NotImplementedError('no CLIP tokenizer files available offline: set cond_stage_config.params.tokenizer_path to a '
                                      "local directory, or pass token ids (batch['txt_tokens'] / encode_tokens)")
        enc = self.tokenizer(list(text), truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding='max_length', return_tensors='pt')
        return enc['input_ids']

    def encode_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        return self.engine.encode_tokens(tokens)

    def encode(self, text: Sequence[str]) -> torch.Tensor:
        key = tuple(text)
        if key not in self._cache:            # the harness encodes two constant prompts ('makeup transfer', '') for every batch
            if len(self._cache) > 64:
                self._cache.clear()
            uniq = sorted(set(key))
            z = self.encode_tokens(self.tokenize(uniq))
            idx = torch.tensor([uniq.index(t) for t in key], device=z.device)
            self._cache[key] = z.index_select(0, idx)
        return self._cache[key]

    forward = encode
    __call__ = encode
