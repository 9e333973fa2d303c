#!/usr/bin/env python
"""Round 3: where the bench step's time outside the 50-step loop goes (batch 8, 256x256): mkd_prepare, mkd_sample, mkd_decode, each
timed with an event pair over 5 repetitions."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makeupdiffuse_amd.engine import MkdEngine, NetConfig, VaeConfig  # noqa: E402
from makeupdiffuse_amd.schedule import DDIMSchedule  # noqa: E402

eng = MkdEngine(NetConfig()); eng.configure_vae(VaeConfig()); eng.init_random(0)
g = torch.Generator().manual_seed(0)
B = int(os.environ.get('B', '8'))
hint = torch.rand(B, 6, 256, 256, generator=g).cuda(); ctx = torch.randn(B, 77, 768, generator=g).cuda(); x = torch.randn(B, 4, 32, 32, generator=g).cuda()
sch = DDIMSchedule().make_ddim(50)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


t_prep, _ = timed(lambda: eng.prepare(hint, ctx))
t_samp, lat = timed(lambda: eng.sample(x, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=True))
t_dec, _ = timed(lambda: eng.decode(lat))
t_all, _ = timed(lambda: (eng.prepare(hint, ctx), eng.decode(eng.sample(x, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=True)))[1])
print(f'batch {B}: prepare {t_prep:.3f} ms, sample {t_samp:.2f} ms ({t_samp / 50:.3f} per step), decode {t_dec:.3f} ms, whole step {t_all:.2f} ms '
      f'(sum of parts {t_prep + t_samp + t_dec:.2f})')
eng.close()
