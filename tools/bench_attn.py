#!/usr/bin/env python
"""Times the attention kernel on the shapes of one eps evaluation (self: T x T, cross: T x 77)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load()
DEV = 'cuda:0'
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(8, 4096, 4096, 8, 40), (8, 1024, 1024, 8, 40), (8, 1024, 1024, 8, 80), (8, 256, 256, 8, 80), (8, 256, 256, 8, 160), (8, 64, 64, 8, 160),
          (8, 4096, 77, 8, 40), (8, 1024, 77, 8, 40), (8, 256, 77, 8, 80), (8, 64, 77, 8, 160)]
for (B, Tq, Tk, H, dh) in shapes:
    d = H * dh
    q = torch.randn(B * Tq, d, device=DEV).bfloat16(); k = torch.randn(B * Tk, d, device=DEV).bfloat16(); v = torch.randn(B * Tk, d, device=DEV).bfloat16()
    o = torch.empty(B * Tq, d, device=DEV, dtype=torch.bfloat16)
    run = lambda: lib.mkd_attention(P(q), d, P(k), d, P(v), d, P(o), d, B, Tq, Tk, H, dh, dh ** -0.5, None)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f'B={B} Tq={Tq} Tk={Tk} heads={H} dh={dh}: {us:8.1f} us  {4.0 * B * H * Tq * Tk * dh / us * 1e-6:7.1f} TF/s', flush=True)
