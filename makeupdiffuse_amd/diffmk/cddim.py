"""``diffmk.cddim`` — drop-in for reference diffmk/cddim.py: MKDDIMSampler with denoising_step (:9-79) and
reconstruct (:81-100).  Same names, argument meaning and error behaviour; the arithmetic runs in libmkd."""
from __future__ import annotations

import torch

from ..ddim import *  # noqa: F401,F403  (the reference star-imports the stock sampler module the same way)
from ..ddim import DDIMSampler, np


class MKDDIMSampler(DDIMSampler):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    @torch.no_grad()
    def denoising_step(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                       temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                       unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None):
        """One reverse step -> (x_prev, pred_x0).  CFG batches [uncond; cond] through ONE apply_model."""
        return self._step(x, c, t, index, repeat_noise, use_original_steps, quantize_denoised, temperature, noise_dropout,
                          score_corrector, corrector_kwargs, unconditional_guidance_scale, unconditional_conditioning,
                          dynamic_threshold)

    @torch.no_grad()
    def reconstruct(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
                    use_original_steps=False, callback=None):
        """Reverse loop over ddim_timesteps[:t_start], newest first; returns the decoded latent."""
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        timesteps = timesteps[:t_start]
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        fast = getattr(self.model, 'sample_loop_fast', None)
        if (fast is not None and callback is None and not use_original_steps and total_steps > 0
                and float(self.ddim_sigmas[:total_steps].abs().max()) == 0.0):
            return fast(x_latent, cond, timesteps, self.ddim_alphas[:total_steps], self.ddim_alphas_prev[:total_steps],
                        self.ddim_sqrt_one_minus_alphas[:total_steps], unconditional_guidance_scale,
                        unconditional_conditioning)
        x_dec = x_latent
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.denoising_step(x_dec, cond, ts, index=index, use_original_steps=use_original_steps,
                                           unconditional_guidance_scale=unconditional_guidance_scale,
                                           unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        return x_dec
