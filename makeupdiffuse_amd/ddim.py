"""Stock DDIM sampler (host logic): what `from ldm.models.diffusion.ddim import *` gives the reference at
diffmk/cddim.py:2 — DDIMSampler with make_schedule / sample / ddim_sampling / p_sample_ddim, `noise_like`
and `np`.  Only the eps parameterisation and the options the reference exercises are implemented; anything
else raises instead of silently diverging."""
from __future__ import annotations

import numpy as np
import torch

from .schedule import make_ddim_sampling_parameters, make_ddim_timesteps

__all__ = ['DDIMSampler', 'noise_like', 'np', 'torch']


def noise_like(shape, device, repeat=False):
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def _cat_cond(uncond, c):
    """CFG batching, unconditional FIRST (diffmk/cddim.py:18-38)."""
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


class DDIMSampler:
    def __init__(self, model, schedule='linear', **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor):
            attr = attr.detach().clone().to(torch.float32).to(getattr(self.model, 'device', 'cpu'))
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize='uniform', ddim_eta=0.0, verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        acn = ac.detach().cpu().to(torch.float32).numpy()
        self.register_buffer('betas', self.model.betas)
        self.register_buffer('alphas_cumprod', ac)
        self.register_buffer('alphas_cumprod_prev', self.model.alphas_cumprod_prev)
        self.register_buffer('sqrt_alphas_cumprod', torch.tensor(np.sqrt(acn)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', torch.tensor(np.sqrt(1.0 - acn)))
        self.register_buffer('sqrt_recip_alphas_cumprod', torch.tensor(np.sqrt(1.0 / acn)))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', torch.tensor(np.sqrt(1.0 / acn - 1)))
        sig, a, ap = make_ddim_sampling_parameters(acn, self.ddim_timesteps, ddim_eta)
        self.register_buffer('ddim_sigmas', torch.tensor(np.asarray(sig), dtype=torch.float32))
        self.register_buffer('ddim_alphas', torch.tensor(np.asarray(a), dtype=torch.float32))
        self.ddim_alphas_prev = np.asarray(ap)
        self.register_buffer('ddim_sqrt_one_minus_alphas', torch.tensor(np.sqrt(1.0 - a), dtype=torch.float32))
        acp = self.model.alphas_cumprod_prev.detach().cpu().numpy()
        self.register_buffer('ddim_sigmas_for_original_num_steps',
                             torch.tensor(ddim_eta * np.sqrt((1 - acp) / (1 - acn) * (1 - acn / acp)), dtype=torch.float32))

    # -- full sampling from noise (reached from ControlLDM.sample_log, diffmk/diffusion_makeup.py:393-408) --
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, eta=0.0, temperature=1.0, noise_dropout=0.0,
               x_T=None, log_every_t=100, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               verbose=True, **kwargs):
        for k in ('mask', 'x0', 'score_corrector', 'corrector_kwargs', 'dynamic_threshold', 'ucg_schedule'):
            if kwargs.get(k) is not None:
                raise NotImplementedError(f'DDIMSampler.sample option {k} is not on the MakeupDiffuse path')
        if kwargs.get('quantize_x0', False):
            raise NotImplementedError('quantize_x0')
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        return self.ddim_sampling(conditioning, size, callback=callback, x_T=x_T, log_every_t=log_every_t,
                                  temperature=temperature, noise_dropout=noise_dropout,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, log_every_t=100, temperature=1.0, noise_dropout=0.0,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, timesteps=None):
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        if timesteps is None:
            timesteps = self.ddim_timesteps
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        fast = getattr(self.model, 'sample_loop_fast', None)
        if fast is not None and callback is None:
            # the whole loop runs inside libmkd (mkd_sample / mkd_sample_eta); no per-step host work.  eta > 0: the draws of the
            # stochastic branch (cddim.py:74-78) are taken here, one per step with sigma_t != 0 in loop order - the generator is
            # consumed exactly as by the step-by-step loop below - and handed over as one [steps, ...] tensor
            sig = self.ddim_sigmas[:total_steps]
            kw = {}
            if float(sig.abs().max()) != 0.0:
                draws = []
                for i in range(total_steps):
                    if float(sig[total_steps - i - 1]) != 0.0:
                        nz = noise_like(tuple(shape), device, False)
                        if noise_dropout > 0.0:
                            nz = torch.nn.functional.dropout(nz, p=noise_dropout)
                    else:
                        nz = torch.zeros(tuple(shape), device=device)
                    draws.append(nz)
                kw = dict(sigmas=sig, noise=torch.stack(draws), temperature=temperature)
            img = fast(img, cond, timesteps, self.ddim_alphas[:total_steps], self.ddim_alphas_prev[:total_steps],
                       self.ddim_sqrt_one_minus_alphas[:total_steps], unconditional_guidance_scale,
                       unconditional_conditioning, **kw)
            intermediates['x_inter'].append(img)
            return img, intermediates
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              noise_dropout=noise_dropout,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1.0, noise_dropout=0.0, score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, dynamic_threshold=None):
        return self._step(x, c, t, index, repeat_noise, use_original_steps, quantize_denoised, temperature, noise_dropout,
                          score_corrector, corrector_kwargs, unconditional_guidance_scale, unconditional_conditioning,
                          dynamic_threshold)

    # One DDIM step; shared by p_sample_ddim and MKDDIMSampler.denoising_step (same arithmetic, SURVEY.md finding 6).
    def _step(self, x, c, t, index, repeat_noise, use_original_steps, quantize_denoised, temperature, noise_dropout,
              score_corrector, corrector_kwargs, unconditional_guidance_scale, unconditional_conditioning,
              dynamic_threshold):
        b, device = x.shape[0], x.device
        if getattr(self.model, 'parameterization', 'eps') != 'eps':
            raise NotImplementedError("only parameterization 'eps' (yaml :50) is supported")
        if score_corrector is not None:
            raise NotImplementedError('score_corrector is unused by the reference path')
        if quantize_denoised:
            raise NotImplementedError('quantize_denoised needs first_stage_model.quantize (VAE: SURVEY §8f)')
        if dynamic_threshold is not None:
            raise NotImplementedError()
        cfg_on = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.0)
        if not cfg_on:
            e_c, e_u = self.model.apply_model(x, t, c), None
        else:
            x_in = torch.cat([x] * 2)
            t_in = torch.cat([t] * 2)
            binder = getattr(self.model, 'cfg_conditioning', None)
            c_in = binder(unconditional_conditioning, c) if binder is not None else _cat_cond(unconditional_conditioning, c)
            e_u, e_c = self.model.apply_model(x_in, t_in, c_in).chunk(2)
        alphas = self.model.alphas_cumprod if use_original_steps else self.ddim_alphas
        alphas_prev = self.model.alphas_cumprod_prev if use_original_steps else self.ddim_alphas_prev
        s1m = self.model.sqrt_one_minus_alphas_cumprod if use_original_steps else self.ddim_sqrt_one_minus_alphas
        sigmas = self.ddim_sigmas_for_original_num_steps if use_original_steps else self.ddim_sigmas
        a_t, a_prev, sigma_t, s1m_t = float(alphas[index]), float(alphas_prev[index]), float(sigmas[index]), float(s1m[index])
        noise = None
        if sigma_t != 0.0:
            noise = noise_like(x.shape, device, repeat_noise)
            if noise_dropout > 0.0:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        step_fn = getattr(self.model, 'ddim_step', None)
        if step_fn is not None and x.is_cuda:
            return step_fn(x, e_c, e_u, unconditional_guidance_scale, a_t, a_prev, sigma_t, s1m_t, noise, temperature)
        # host tensors (plumbing with a stand-in model, e.g. CPU tests): same formulae in torch
        e_t = e_c if e_u is None else e_u + unconditional_guidance_scale * (e_c - e_u)
        pred_x0 = (x - s1m_t * e_t) / a_t ** 0.5
        dir_xt = (1.0 - a_prev - sigma_t ** 2) ** 0.5 * e_t
        x_prev = a_prev ** 0.5 * pred_x0 + dir_xt
        if noise is not None:
            x_prev = x_prev + sigma_t * noise * temperature
        return x_prev, pred_x0
