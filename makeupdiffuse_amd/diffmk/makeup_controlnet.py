"""``diffmk.makeup_controlnet`` — the earlier ControlNet variant's cond contract (reference
diffmk/makeup_controlnet.py:11-32 ``BaseModel`` and :102-167 ``MakeupDoubleControlModel``).

What differs from ``diffmk.makeup_diffuse.BaseMakeUpDiffuse.get_input`` (reference diffmk/makeup_diffuse.py:42-57) is the BATCH
contract, not the conditioning it produces: the control images arrive channels-LAST (``b h w c``, as UPSTREAM ControlLDM datasets
deliver them) under ``control_src_key`` (source face) and ``control_key`` (reference makeup) and are rearranged to ``b c h w``
(reference :158-159, :165-166); the cond dict is ``c_crossattn=[txt]``, ``c_concat=[cat(src, ref, 1)]`` — source first (:167)."""
from __future__ import annotations

from typing import Optional

import torch

from .cddim import MKDDIMSampler  # noqa: F401  (the reference module re-exports it, :5)
from .makeup_diffuse import BaseMakeUpDiffuse


class BaseModel(BaseMakeUpDiffuse):
    """reference diffmk/makeup_controlnet.py:11-32: ``control_src_key`` names the source-face entry of the batch."""

    def __init__(self, control_src_key: str, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.control_src_key = control_src_key

    def _control_image(self, batch: dict, key: str, bs: Optional[int]) -> torch.Tensor:
        """``batch[key]`` [b, h, w, c] -> [b, c, h, w] float, contiguous, on the model's device (reference :24-28 / :153-166)."""
        x = batch[key]
        if bs is not None:
            x = x[:bs]
        if x.dim() != 4:
            raise ValueError(f'batch[{key!r}] must be [b, h, w, c], got {tuple(x.shape)}')
        x = x.to(self.device).permute(0, 3, 1, 2)               # einops.rearrange(x, 'b h w c -> b c h w')
        return x.to(memory_format=torch.contiguous_format).float()

    @torch.no_grad()
    def get_input(self, batch: dict, k, bs: Optional[int] = None, *args, **kwargs):
        """-> (None, c): the latent of ``batch[first_stage_key]`` that the reference also returns needs the first-stage ENCODER,
        which is outside the sampling path; sampling starts from noise (SURVEY.md finding 5)."""
        src = self._control_image(batch, self.control_src_key, bs)
        ref = self._control_image(batch, self.control_key, bs)
        c = dict(c_crossattn=[self.get_cond_txt_coding(batch, bs)], c_concat=[torch.cat([src, ref], 1)])
        return None, c


class MakeupDoubleControlModel(BaseModel):
    """reference diffmk/makeup_controlnet.py:102-167 (same cond assembly as BaseModel, written out in full there)."""

    def get_origin_img_input(self, batch: dict, k: str, need_rearrange: bool = True) -> torch.Tensor:          # reference :106-114
        x = batch[k]
        if x.dim() == 3:
            x = x[..., None]
        x = x.to(self.device)
        if need_rearrange:
            x = x.permute(0, 3, 1, 2)
        return x.to(memory_format=torch.contiguous_format).float()
