"""``diffmk.makeups`` — the inference-side surface of the reference's DDIM-inversion fine-tune model
(reference diffmk/makeups.py), i.e. the only callers of ``MKDDIMSampler.reconstruct`` in the reference:

  * ``on_fit_start``        :44-47   builds the sampler, ``make_schedule(ddim_num_steps=iter_finetune)``
  * ``generate_image``      :119-127 ``reconstruct(x_latent=inv, cond=c, t_start=iter_finetune)`` -> ``decode_latent_code``
                                     -> ``(x + 1) / 2`` clamped to [0, 1]
  * ``decode_latent_code``  :260-262 ``first_stage_model.decode(z / scale_factor)``
  * ``log_images``          :265-286 the two reconstructions (source under the reference hint, reference under the source hint)

The losses (:80-245) are training code and out of scope; ``shared_step`` / ``p_losses`` raise.
The hint of this variant is ONE image (``c_concat_r`` / ``c_concat_s``, 3 channels): build the model from a
``control_stage_config`` with ``hint_channels: 3``.  The arithmetic runs in libmkd; nothing here has a CPU path."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .cddim import MKDDIMSampler
from .makeup_diffuse import BaseMakeUpDiffuse


class BaseModel(BaseMakeUpDiffuse):
    def __init__(self, src_msk_key: str = 'src_msk', ref_msk_key: str = 'ref_msk', src_img_key: str = 'src_img',
                 src_inv_key: str = 'src_inv', ref_img_key: str = 'ref_img', ref_inv_key: str = 'ref_inv', dataset_len: int = 0,
                 t0: int = 1000, inv_steps: int = 50, iter_finetune: int = 50, debug_dir: Optional[str] = None, *args, **kwargs):
        super().__init__(*args, src_img_key=src_img_key, ref_img_key=ref_img_key, **kwargs)
        self.src_inv_key, self.ref_inv_key = src_inv_key, ref_inv_key
        self.src_msk_key, self.ref_msk_key = src_msk_key, ref_msk_key
        self.dataset_len, self.debug_dir = dataset_len, debug_dir
        self.t0, self.inv_steps, self.iter_finetune = t0, inv_steps, iter_finetune
        self.ddim_sampler: Optional[MKDDIMSampler] = None

    def update_schedule(self) -> None:
        """:40-42: re-registers the LINEAR beta schedule with ``timesteps = t0`` (same linear_start / linear_end): the DDIM
        inversion and its fine-tune run on a t0-step DDPM chain."""
        self.register_schedule(given_betas=None, beta_schedule='linear', timesteps=self.t0, linear_start=self.linear_start,
                               linear_end=self.linear_end, cosine_s=8e-3)

    def on_fit_start(self) -> None:
        """:44-47."""
        self.update_schedule()
        self.ddim_sampler = MKDDIMSampler(self)
        self.ddim_sampler.make_schedule(ddim_num_steps=self.iter_finetune)

    def _sampler(self) -> MKDDIMSampler:
        if self.ddim_sampler is None:
            self.on_fit_start()
        return self.ddim_sampler

    @torch.no_grad()
    def get_input(self, batch: dict, k, bs: Optional[int] = None, *args, **kwargs):
        """:70-78 -> (src_inv, ref_inv, src_msk, ref_msk, c) with c_concat_s = [src_img], c_concat_r = [ref_img]."""
        src = self.get_origin_img_input(batch, self.src_img_key, bs)
        ref = self.get_origin_img_input(batch, self.ref_img_key, bs)
        msk = lambda key: batch[key].to(self.device) if key in batch else None
        c = dict(c_crossattn=[self.get_cond_txt_coding(batch, bs)], c_concat_s=[src], c_concat_r=[ref])
        return batch[self.src_inv_key].to(self.device), batch[self.ref_inv_key].to(self.device), msk(self.src_msk_key), msk(self.ref_msk_key), c

    @torch.no_grad()
    def generate_image(self, inv: torch.Tensor, c: dict, c_replace=None, c_type: str = 'c_concat_r') -> torch.Tensor:
        """:119-127.  Mutates ``c['c_concat']`` exactly as the reference does."""
        c['c_concat'] = c_replace if c_replace else c[c_type]
        z = self._sampler().reconstruct(x_latent=inv, cond=c, t_start=self.iter_finetune)
        img = (self.decode_latent_code(z) + 1.0) / 2.0
        return img.clamp(0, 1)

    @torch.no_grad()
    def log_images(self, batch: dict, **kwargs) -> Dict[str, torch.Tensor]:
        """:265-286: source latent re-generated under the reference hint and vice versa."""
        src_inv, ref_inv, _, _, c = self.get_input(batch, self.first_stage_key)
        log: Dict[str, torch.Tensor] = {}
        c['c_concat'] = c['c_concat_r']
        log['rec_src_ref'] = self.decode_first_stage(self._sampler().reconstruct(x_latent=src_inv, cond=c, t_start=self.iter_finetune))
        c['c_concat'] = c['c_concat_s']
        log['rec_ref_src'] = self.decode_first_stage(self._sampler().reconstruct(x_latent=ref_inv, cond=c, t_start=self.iter_finetune))
        log['ori_src'] = torch.cat(c['c_concat_s'], 0) * 2.0 - 1.0
        log['ori_ref'] = torch.cat(c['c_concat_r'], 0) * 2.0 - 1.0
        return log

    def shared_step(self, batch, **kwargs):
        raise NotImplementedError('training losses of diffmk/makeups.py:80-245 are outside the sampling hot path')

    p_losses = forward = shared_step
