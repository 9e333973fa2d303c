"""fp32 CPU restatement of the text conditioning stage (test infrastructure only).

The reference's ``cond_stage_config`` is ``ldm.modules.encoders.modules.FrozenCLIPEmbedder``
(``diffmodels/base_diffusion_makeup.yaml:109-110``), reached through ``get_learned_conditioning`` at
``diffmk/makeup_teacher.py:33-42`` (prompt 'makeup transfer', ``diffdata/datasets.py:772``) and
``get_unconditional_conditioning`` at ``diffmk/diffusion_makeup.py:400``.  UPSTREAM it is
``transformers.CLIPTextModel(...)(input_ids=tokens).last_hidden_state`` on 77 max-length-padded tokens.
State-dict names are the checkpoint's (``cond_stage_model.transformer.text_model.*``, transformers-4.x layout).

PINNED, unlike the two nets: ``transformers`` IS importable in this container, so ``tests/test_oracle.py`` checks this
restatement against ``transformers.CLIPTextModel`` itself on seeded random weights (the pretrained weights and the
tokenizer vocabulary are not available offline).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
PREFIX = 'cond_stage_model.transformer.text_model.'


@dataclass
class ClipConfig:
    vocab_size: int = 49408
    max_positions: int = 77
    width: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    ln_eps: float = 1e-5


FULL = ClipConfig()
SMALL = ClipConfig(vocab_size=512, max_positions=77, width=128, layers=2, heads=2, intermediate=256)


def param_spec(cfg: ClipConfig, prefix: str = PREFIX) -> Dict[str, tuple]:
    W, I = cfg.width, cfg.intermediate
    d = {f'{prefix}embeddings.token_embedding.weight': (cfg.vocab_size, W),
         f'{prefix}embeddings.position_embedding.weight': (cfg.max_positions, W)}
    for l in range(cfg.layers):
        L = f'{prefix}encoder.layers.{l}'
        for n in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            d[f'{L}.self_attn.{n}.weight'] = (W, W)
            d[f'{L}.self_attn.{n}.bias'] = (W,)
        for n in ('layer_norm1', 'layer_norm2'):
            d[f'{L}.{n}.weight'] = (W,)
            d[f'{L}.{n}.bias'] = (W,)
        d[f'{L}.mlp.fc1.weight'] = (I, W); d[f'{L}.mlp.fc1.bias'] = (I,)
        d[f'{L}.mlp.fc2.weight'] = (W, I); d[f'{L}.mlp.fc2.bias'] = (W,)
    d[f'{prefix}final_layer_norm.weight'] = (W,)
    d[f'{prefix}final_layer_norm.bias'] = (W,)
    return d


def init_state_dict(cfg: ClipConfig, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in sorted(param_spec(cfg).items()):
        if 'layer_norm' in name:
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g) if name.endswith('weight') else 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif 'embedding' in name:
            sd[name] = 0.5 * torch.randn(shape, generator=g)
        else:
            sd[name] = torch.randn(shape, generator=g) / shape[1] ** 0.5
    return sd


def quick_gelu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(1.702 * x)


def encode_tokens(sd: Dict[str, Tensor], cfg: ClipConfig, tokens: Tensor, prefix: str = PREFIX) -> Tensor:
    """tokens [B, T] int -> last_hidden_state [B, T, width].  Pre-LN blocks; causal mask only (no padding mask)."""
    B, T = tokens.shape
    W, H = cfg.width, cfg.heads
    dh = W // H
    x = sd[f'{prefix}embeddings.token_embedding.weight'][tokens.long()] + sd[f'{prefix}embeddings.position_embedding.weight'][:T]
    mask = torch.full((T, T), float('-inf')).triu(1)
    for l in range(cfg.layers):
        L = f'{prefix}encoder.layers.{l}'
        h = F.layer_norm(x, (W,), sd[f'{L}.layer_norm1.weight'], sd[f'{L}.layer_norm1.bias'], cfg.ln_eps)
        q = F.linear(h, sd[f'{L}.self_attn.q_proj.weight'], sd[f'{L}.self_attn.q_proj.bias']) * dh ** -0.5
        k = F.linear(h, sd[f'{L}.self_attn.k_proj.weight'], sd[f'{L}.self_attn.k_proj.bias'])
        v = F.linear(h, sd[f'{L}.self_attn.v_proj.weight'], sd[f'{L}.self_attn.v_proj.bias'])
        q, k, v = (t.view(B, T, H, dh).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, W)
        x = x + F.linear(a, sd[f'{L}.self_attn.out_proj.weight'], sd[f'{L}.self_attn.out_proj.bias'])
        h = F.layer_norm(x, (W,), sd[f'{L}.layer_norm2.weight'], sd[f'{L}.layer_norm2.bias'], cfg.ln_eps)
        h = quick_gelu(F.linear(h, sd[f'{L}.mlp.fc1.weight'], sd[f'{L}.mlp.fc1.bias']))
        x = x + F.linear(h, sd[f'{L}.mlp.fc2.weight'], sd[f'{L}.mlp.fc2.bias'])
    return F.layer_norm(x, (W,), sd[f'{prefix}final_layer_norm.weight'], sd[f'{prefix}final_layer_norm.bias'], cfg.ln_eps)
