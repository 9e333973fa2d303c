#!/usr/bin/env python
"""Round 3 diagnostic: where the ~11 us of a 32x32-latent GroupNorm go.  Needs the instrumented experiment build (thread 0 of every
workgroup stamps wall_clock64 - 100 MHz - at entry, after its slab loads have landed, after the statistics, after the apply loop has issued
its stores, after the stores are acknowledged); built outside the tree from tools/patches as libmkd_gntrace.so.

    MKD_LIB_PATH=makeupdiffuse_amd/libmkd_gntrace.so python tools/exp_r3_gn_trace.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makeupdiffuse_amd import lib as mlib  # noqa: E402

lib = mlib.load()
raw = ctypes.CDLL(os.environ['MKD_LIB_PATH'])
raw.mkd_exp_gn_trace.argtypes = [ctypes.c_void_p]
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = 'cuda'
for B, hw, C in ((4, 1024, 320), (8, 1024, 320), (4, 1024, 640), (4, 256, 640), (4, 64, 1280)):
    g = torch.Generator().manual_seed(0)
    src = torch.randn(B, hw, C, generator=g).to(dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    y = torch.empty(B, hw, C, device=dev, dtype=torch.bfloat16)
    trace = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    assert raw.mkd_exp_gn_trace(P(trace)) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    times = []
    for it in range(20):
        x = (src * 1.0001).to(torch.bfloat16)            # the producer: another kernel writes the tensor just before
        e0.record()
        assert lib.mkd_groupnorm(P(x), C, P(gamma), P(beta), 1e-5, 1, P(y), C, B, hw, C, 32, None) == 0
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    t = trace.cpu().view(-1, 8)
    t = t[t[:, 0] > 0][:, :5].double() * 10.0            # ns
    n = t.shape[0]
    ph = torch.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]], 1) / 1e3
    span = (t[:, 4].max() - t[:, 0].min()) / 1e3
    start_spread = (t[:, 0].max() - t[:, 0].min()) / 1e3
    med = ph.median(0).values; mx = ph.max(0).values
    print(f'B={B} HW={hw} C={C}: {n} workgroups, event-timed {sorted(times)[len(times) // 2]:.1f} us; first entry -> last store acknowledged {span:.2f} us '
          f'(entries spread over {start_spread:.2f} us); per workgroup median (max) us: loads {med[0]:.2f} ({mx[0]:.2f}), statistics {med[1]:.2f} ({mx[1]:.2f}), '
          f'apply + store issue {med[2]:.2f} ({mx[2]:.2f}), store drain {med[3]:.2f} ({mx[3]:.2f})', flush=True)
raw.mkd_exp_gn_trace(None)
