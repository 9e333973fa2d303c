#!/usr/bin/env python
"""Experiment: aggregate throughput of N independent engines (batch 8/N each, own streams, own host thread) vs one
engine with batch 8.  Decides whether sub-batch 'lanes' inside one context are worth building."""
import sys, os, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd.engine import MkdEngine, NetConfig
from makeupdiffuse_amd.schedule import DDIMSchedule

def run(nl, total=8, steps=50, reps=2):
    dev = torch.device('cuda:0')
    b = total // nl
    engs, streams, ins = [], [], []
    for i in range(nl):
        e = MkdEngine(NetConfig(), dev); e.init_random(0)
        engs.append(e); streams.append(torch.cuda.Stream())
        g = torch.Generator().manual_seed(i)
        ins.append((torch.randn(b, 4, 32, 32, generator=g).to(dev), torch.rand(b, 6, 256, 256, generator=g).to(dev),
                    torch.randn(b, 77, 768, generator=g).to(dev)))
    sch = DDIMSchedule().make_ddim(steps)
    def work(i):
        with torch.cuda.stream(streams[i]):
            x, h, c = ins[i]
            engs[i].prepare(h, c)
            engs[i].sample(x, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=True)
    best = 1e9
    for r in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nl)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if r: best = min(best, dt)
    print(f'lanes={nl} batch/lane={b}: {best*1e3:.1f} ms per {total} images -> {total/best:.2f} img/s', flush=True)
    for e in engs: e.close()

for nl in (1, 2, 4):
    run(nl)
