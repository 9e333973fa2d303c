"""DDPM/DDIM schedule tables (host, numpy/torch; device-agnostic).

What the reference gets from the un-vendored ldm package: the "linear" beta schedule configured by
diffmodels/base_diffusion_makeup.yaml:4-8 and DDIMSampler.make_schedule as called at
diffmk/makeups.py:47 / diffmk/pre_dataset.py:68.  Formulae: SURVEY.md App. B (KAT-pinned in tests)."""
from __future__ import annotations

import numpy as np
import torch


def make_beta_schedule(schedule: str = 'linear', n_timestep: int = 1000, linear_start: float = 0.00085,
                       linear_end: float = 0.0120) -> np.ndarray:
    if schedule != 'linear':
        raise NotImplementedError(f"beta schedule '{schedule}' (the reference config uses 'linear')")
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2


def make_ddim_timesteps(ddim_discr_method: str, num_ddim_timesteps: int, num_ddpm_timesteps: int) -> np.ndarray:
    if ddim_discr_method != 'uniform':
        raise NotImplementedError(f"ddim discretisation '{ddim_discr_method}'")
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


def make_ddim_sampling_parameters(alphacums: np.ndarray, ddim_timesteps: np.ndarray, eta: float):
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


class DDIMSchedule:
    """All the tables MKDDIMSampler / the model expose as attributes (SURVEY.md §8b sampler<->model contract)."""

    def __init__(self, timesteps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.0120,
                 beta_schedule: str = 'linear'):
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.num_timesteps = int(timesteps)
        self.betas = f32(betas)
        self.alphas_cumprod = f32(ac)
        self.alphas_cumprod_prev = f32(np.append(1.0, ac[:-1]))
        self.sqrt_alphas_cumprod = f32(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = f32(np.sqrt(1.0 / ac))
        self.sqrt_recipm1_alphas_cumprod = f32(np.sqrt(1.0 / ac - 1.0))

    def make_ddim(self, ddim_num_steps: int, ddim_discretize: str = 'uniform', ddim_eta: float = 0.0) -> 'DDIMSchedule':
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.num_timesteps)
        ac = self.alphas_cumprod.cpu().numpy()
        sig, a, ap = make_ddim_sampling_parameters(ac, self.ddim_timesteps, ddim_eta)
        f32 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float32)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = f32(sig), f32(a), f32(ap)
        self.ddim_sqrt_one_minus_alphas = f32(np.sqrt(1.0 - a))
        acp = self.alphas_cumprod_prev.numpy()
        self.ddim_sigmas_for_original_num_steps = f32(ddim_eta * np.sqrt((1 - acp) / (1 - ac) * (1 - ac / acp)))
        return self
