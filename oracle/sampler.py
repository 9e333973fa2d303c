"""fp32 CPU restatement of the sampler side of the hot path (test infrastructure only).

Follows, line for line in meaning (not in text):
  * reference ``diffmk/cddim.py:9-79``  ``MKDDIMSampler.denoising_step``
  * reference ``diffmk/cddim.py:81-100`` ``MKDDIMSampler.reconstruct``
  * reference ``diffmk/makeup_diffuse.py:152-170`` ``apply_model``
  * reference ``diffmk/diffusion_makeup.py:391-410`` two-pass sampling via ``sample_log``
  * yaml ``diffmodels/base_diffusion_makeup.yaml:4-8`` linear schedule; the UPSTREAM
    ``make_schedule`` formulae restated in SURVEY.md App. B (KAT-pinned in tests).
PARITY UNPINNED at the ldm/cldm boundary (see ``oracle/__init__``).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import nets

Tensor = torch.Tensor


def linear_beta_schedule(timesteps=1000, linear_start=0.00085, linear_end=0.0120) -> np.ndarray:
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2


class Schedule:
    """alphas_cumprod tables + the DDIM sub-schedule (App. B)."""

    def __init__(self, timesteps=1000, linear_start=0.00085, linear_end=0.0120):
        betas = linear_beta_schedule(timesteps, linear_start, linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = timesteps
        self.alphas_cumprod64 = ac
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
        self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)
        self.sqrt_recip_alphas_cumprod = torch.tensor(np.sqrt(1.0 / ac), dtype=torch.float32)
        self.sqrt_recipm1_alphas_cumprod = torch.tensor(np.sqrt(1.0 / ac - 1.0), dtype=torch.float32)

    def make_ddim(self, ddim_num_steps: int, eta: float = 0.0):
        c = self.num_timesteps // ddim_num_steps
        ts = np.asarray(list(range(0, self.num_timesteps, c))) + 1
        ac = self.alphas_cumprod.numpy()      # upstream indexes the fp32 buffer (.cpu())
        a = ac[ts]
        a_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())
        sig = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
        self.ddim_timesteps = ts
        self.ddim_alphas = torch.tensor(a, dtype=torch.float32)
        self.ddim_alphas_prev = torch.tensor(a_prev, dtype=torch.float32)
        self.ddim_sigmas = torch.tensor(sig, dtype=torch.float32)
        self.ddim_sqrt_one_minus_alphas = torch.tensor(np.sqrt(1.0 - a), dtype=torch.float32)
        return self


def apply_model(sd, cfg: nets.NetConfig, x_noisy: Tensor, t: Tensor, cond: dict,
                control_scales: Optional[List[float]] = None, only_mid_control: bool = False) -> Tensor:
    """reference makeup_diffuse.py:152-170 (eps only)."""
    assert isinstance(cond, dict)
    cond_txt = torch.cat(cond['c_crossattn'], 1)
    if cond['c_concat'] is None:
        return nets.diffusion_model(sd, cfg, x_noisy, t, cond_txt, control=None,
                                    only_mid_control=only_mid_control)
    hint2 = torch.cat(cond['c_concat2'], 1) if cond.get('c_concat2') is not None else None
    control = nets.control_model(sd, cfg, x_noisy, torch.cat(cond['c_concat'], 1), t, cond_txt, hint2=hint2,
                                 alpha=cond.get('interp_alpha'))
    scales = control_scales if control_scales is not None else [1.0] * len(control)
    control = [c * s for c, s in zip(control, scales)]
    return nets.diffusion_model(sd, cfg, x_noisy, t, cond_txt, control=control,
                                only_mid_control=only_mid_control)


def predict_start_from_noise(sch: Schedule, x_t: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """ldm DDPM.predict_start_from_noise used at makeup_diffuse.py:169."""
    a = sch.sqrt_recip_alphas_cumprod[t].view(-1, 1, 1, 1)
    b = sch.sqrt_recipm1_alphas_cumprod[t].view(-1, 1, 1, 1)
    return a * x_t - b * noise


def cat_cond(uncond, c):
    """CFG conditioning batching, uncond FIRST (cddim.py:18-38)."""
    if isinstance(c, dict):
        assert isinstance(uncond, dict)
        out = {}
        for k in c:
            if isinstance(c[k], list):
                out[k] = [torch.cat([uncond[k][i], c[k][i]]) for i in range(len(c[k]))]
            else:
                out[k] = torch.cat([uncond[k], c[k]])
        return out
    if isinstance(c, list):
        assert isinstance(uncond, list)
        return [torch.cat([uncond[i], c[i]]) for i in range(len(c))]
    return torch.cat([uncond, c])


def denoising_step(eps_fn: Callable, sch: Schedule, x: Tensor, c, t: Tensor, index: int,
                   unconditional_guidance_scale: float = 1.0, unconditional_conditioning=None,
                   temperature: float = 1.0, noise: Optional[Tensor] = None):
    """cddim.py:9-79 for parameterization 'eps', no corrector/quantize/threshold."""
    b = x.shape[0]
    if unconditional_conditioning is None or unconditional_guidance_scale == 1.0:
        e_t = eps_fn(x, t, c)
    else:
        x_in = torch.cat([x] * 2)
        t_in = torch.cat([t] * 2)
        c_in = cat_cond(unconditional_conditioning, c)
        e_u, e_c = eps_fn(x_in, t_in, c_in).chunk(2)
        e_t = e_u + unconditional_guidance_scale * (e_c - e_u)
    a_t = torch.full((b, 1, 1, 1), float(sch.ddim_alphas[index]))
    a_prev = torch.full((b, 1, 1, 1), float(sch.ddim_alphas_prev[index]))
    sigma_t = torch.full((b, 1, 1, 1), float(sch.ddim_sigmas[index]))
    s1m = torch.full((b, 1, 1, 1), float(sch.ddim_sqrt_one_minus_alphas[index]))
    pred_x0 = (x - s1m * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    nz = sigma_t * (noise if noise is not None else torch.zeros_like(x)) * temperature
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + nz
    return x_prev, pred_x0


def reconstruct(eps_fn: Callable, sch: Schedule, x_latent: Tensor, cond, t_start: int,
                unconditional_guidance_scale: float = 1.0, unconditional_conditioning=None,
                callback=None) -> Tensor:
    """cddim.py:81-100."""
    timesteps = sch.ddim_timesteps[:t_start]
    time_range = np.flip(timesteps)
    total = timesteps.shape[0]
    x_dec = x_latent
    for i, step in enumerate(time_range):
        index = total - i - 1
        ts = torch.full((x_latent.shape[0],), int(step), dtype=torch.long)
        x_dec, _ = denoising_step(eps_fn, sch, x_dec, cond, ts, index,
                                  unconditional_guidance_scale, unconditional_conditioning)
        if callback:
            callback(i)
    return x_dec


def sample(eps_fn: Callable, sch: Schedule, x_T: Tensor, cond, ddim_steps: int, eta: float = 0.0,
           unconditional_guidance_scale: float = 1.0, unconditional_conditioning=None,
           intermediates: Optional[list] = None) -> Tensor:
    """UPSTREAM DDIMSampler.sample/ddim_sampling reached from sample_log
    (diffusion_makeup.py:393-408): the full reverse loop from x_T; eta=0 path only
    draws no noise."""
    assert eta == 0.0, 'oracle restates the deterministic (eta=0) path the reference uses'
    sch.make_ddim(ddim_steps, eta)
    out = reconstruct(eps_fn, sch, x_T, cond, ddim_steps, unconditional_guidance_scale,
                      unconditional_conditioning,
                      callback=None)
    return out


def make_eps_fn(sd, cfg, control_scales=None, only_mid_control=False):
    def fn(x, t, c):
        return apply_model(sd, cfg, x, t, c, control_scales, only_mid_control)
    return fn
