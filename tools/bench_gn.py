#!/usr/bin/env python
"""Times GroupNorm(32)+SiLU on the (pixels, channels) shapes of one eps evaluation at batch 8."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load(); DEV = 'cuda:0'; P = lambda t: C.c_void_p(t.data_ptr())
B = 8
for (hw, Cc) in [(1024, 320), (1024, 640), (1024, 960), (256, 320), (256, 640), (256, 960), (256, 1280), (256, 1920), (64, 640), (64, 1280), (64, 1920), (64, 2560), (16, 1280), (16, 2560), (4096, 320), (4096, 640)]:
    x = torch.randn(B, hw, Cc, device=DEV).bfloat16(); y = torch.empty_like(x)
    ga = torch.ones(Cc, device=DEV); be = torch.zeros(Cc, device=DEV)
    ws = torch.empty(1 << 20, device=DEV)
    run = lambda: lib.mkd_groupnorm(P(x), Cc, P(ga), P(be), 1e-5, 1, P(y), Cc, B, hw, Cc, 32, None)
    for _ in range(3): assert run() == 0, lib.mkd_last_error()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'hw={hw} C={Cc}: {us:7.1f} us  {2 * x.numel() * 2 / us * 1e-6:6.2f} TB/s', flush=True)
