"""Generates tests/golden/*.npz from the CPU oracle (run once, commit the output).

    python tests/make_golden.py

The reference ships no golden vectors and cannot be imported here (ldm/cldm absent), so these
fixtures pin the ORACLE (regression) and give the GPU tests a data-only expectation; they do not pin
the oracle to the reference (parity unpinned, see oracle/__init__.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets, sampler  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

SMALL = dict(model_channels=64, channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2, context_dim=64,
             hint_widths=(16, 16, 32, 32, 32, 32, 64))


VSMALL = dict(z_channels=4, embed_dim=4, ch=32, ch_mult=(1, 2), num_res_blocks=1, out_ch=3)


def small_cfg():
    return nets.NetConfig(**SMALL)


def inputs(B, h, w, cfg, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, h, w, generator=g)
    hint = torch.rand(B, cfg.hint_channels, 8 * h, 8 * w, generator=g)
    ctx = torch.randn(B, 77, cfg.context_dim, generator=g)
    uctx = torch.randn(B, 77, cfg.context_dim, generator=g)
    return x, hint, ctx, uctx


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    cfg = small_cfg()
    sd = nets.init_state_dict(cfg, seed=11)
    B, h, w = 2, 8, 8
    x, hint, ctx, uctx = inputs(B, h, w, cfg, 5)
    t = torch.tensor([481, 21])
    cond = {'c_crossattn': [ctx], 'c_concat': [hint]}
    n_ctrl = len(nets.encoder_spec(cfg)) + 1
    scales = [0.5 + 0.1 * i for i in range(n_ctrl)]
    eps = sampler.apply_model(sd, cfg, x, t, cond)
    eps_scaled = sampler.apply_model(sd, cfg, x, t, cond, control_scales=scales)
    eps_mid = sampler.apply_model(sd, cfg, x, t, cond, only_mid_control=True)
    eps_noctl = sampler.apply_model(sd, cfg, x, t, {'c_crossattn': [ctx], 'c_concat': None})
    sch = sampler.Schedule()
    fn = sampler.make_eps_fn(sd, cfg)
    x5 = sampler.sample(fn, sch, x, cond, 5)
    ucond = {'c_crossattn': [uctx], 'c_concat': [hint]}
    x5_cfg = sampler.sample(fn, sch, x, cond, 5, unconditional_guidance_scale=9.0, unconditional_conditioning=ucond)
    # what the reference's test harness runs (diffmk/diffusion_makeup.py:308-309, 391-410): 50 steps, plain and CFG 9 with uc_cat = c_cat
    x50 = sampler.sample(fn, sampler.Schedule(), x, cond, 50)
    x50_cfg = sampler.sample(fn, sampler.Schedule(), x, cond, 50, unconditional_guidance_scale=9.0, unconditional_conditioning=ucond)
    # MKDDIMSampler surface (diffmk/cddim.py:9-100): reconstruct(t_start < S), one denoising_step, apply_model(return_all=True)
    s10 = sampler.Schedule().make_ddim(10)
    rec4 = sampler.reconstruct(fn, s10, x, cond, 4)
    rec4_cfg = sampler.reconstruct(fn, s10, x, cond, 4, unconditional_guidance_scale=9.0, unconditional_conditioning=ucond)
    ts = torch.full((B,), int(s10.ddim_timesteps[6]), dtype=torch.long)
    step_prev, step_x0 = sampler.denoising_step(fn, s10, x, cond, ts, 6)
    stepc_prev, stepc_x0 = sampler.denoising_step(fn, s10, x, cond, ts, 6, unconditional_guidance_scale=9.0, unconditional_conditioning=ucond)
    x_recon = sampler.predict_start_from_noise(sch, x, t, eps)
    # first stage (decode_first_stage / generate_image / decode_latent_code: makeup_diffuse.py:172-177, makeups.py:119-127,260-262)
    from oracle import vae
    vcfg = vae.VaeConfig(**VSMALL)
    vsd = vae.init_state_dict(vcfg, seed=5)
    img50 = vae.decode_first_stage(vsd, vcfg, x50)
    img_rec4 = vae.decode_first_stage(vsd, vcfg, rec4)
    np.savez_compressed(os.path.join(OUT, 'small_eps.npz'), seed_weights=11, x=x.numpy(), hint=hint.numpy(), ctx=ctx.numpy(),
                        uctx=uctx.numpy(), t=t.numpy(), scales=np.array(scales, dtype=np.float32), eps=eps.numpy(),
                        eps_scaled=eps_scaled.numpy(), eps_mid=eps_mid.numpy(), eps_noctl=eps_noctl.numpy(),
                        x5=x5.numpy(), x5_cfg=x5_cfg.numpy(), x50=x50.numpy(), x50_cfg=x50_cfg.numpy(), rec4=rec4.numpy(),
                        rec4_cfg=rec4_cfg.numpy(), step_prev=step_prev.numpy(), step_x0=step_x0.numpy(), stepc_prev=stepc_prev.numpy(),
                        stepc_x0=stepc_x0.numpy(), x_recon=x_recon.numpy(), seed_vae=5, img50=img50.numpy(), img_rec4=img_rec4.numpy())
    # schedule KATs as data
    s50 = sampler.Schedule().make_ddim(50)
    s20 = sampler.Schedule().make_ddim(20)
    np.savez_compressed(os.path.join(OUT, 'schedule.npz'), alphas_cumprod=s50.alphas_cumprod.numpy(),
                        ts50=s50.ddim_timesteps, a50=s50.ddim_alphas.numpy(), ap50=s50.ddim_alphas_prev.numpy(),
                        s1m50=s50.ddim_sqrt_one_minus_alphas.numpy(), ts20=s20.ddim_timesteps, a20=s20.ddim_alphas.numpy(),
                        ap20=s20.ddim_alphas_prev.numpy(), s1m20=s20.ddim_sqrt_one_minus_alphas.numpy())
    print('wrote', os.listdir(OUT))


if __name__ == '__main__':
    main()
