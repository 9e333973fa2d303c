#!/usr/bin/env python
"""Experiment: how much concurrency is left on the chip?  Two independent batch-8 sampling loops (two contexts, private streams,
each its captured step graph) in flight at once vs one after the other."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makeupdiffuse_amd.engine import MkdEngine, NetConfig  # noqa: E402
from makeupdiffuse_amd.schedule import DDIMSchedule  # noqa: E402

B = int(os.environ.get('B', 8)); N = int(os.environ.get('N', 2)); STEPS = 20
dev = torch.device('cuda:0')
engs = []
for i in range(N):
    e = MkdEngine(NetConfig(), dev); e.init_random(0); engs.append(e)
g = torch.Generator().manual_seed(0)
hint = torch.rand(B, 6, 256, 256, generator=g).cuda(); ctx = torch.randn(B, 77, 768, generator=g).cuda()
x = torch.randn(B, 4, 32, 32, generator=g).cuda()
sch = DDIMSchedule().make_ddim(STEPS)
args = (sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas)
streams = [torch.cuda.Stream() for _ in range(N)]
for e, s in zip(engs, streams):
    with torch.cuda.stream(s):
        e.prepare(hint, ctx); e.sample(x, *args, use_graph=bool(int(os.environ.get('GRAPH', 1))))
torch.cuda.synchronize()


def run(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e, s in list(zip(engs, streams))[:k]:
        with torch.cuda.stream(s):
            e.sample(x, *args, use_graph=bool(int(os.environ.get('GRAPH', 1))))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    for k in range(1, N + 1):
        t = min(run(k) for _ in range(3))
        print(f'{k} loop(s) of batch {B} in flight: {t:.1f} ms for {STEPS} steps = {t / STEPS:.3f} ms per step, {k * B * 50 / (t / STEPS * 50) * 1e3 / 50:.1f} images/s at 50 steps', flush=True)
