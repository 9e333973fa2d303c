#!/usr/bin/env python
"""Times mkd_decode (first-stage decoder) for a batch of latents."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd.engine import MkdEngine, NetConfig, VaeConfig
eng = MkdEngine(NetConfig()); eng.configure_vae(VaeConfig()); eng.init_random(0)
for B, h in ((8, 32), (4, 32), (8, 64)):
    z = torch.randn(B, 4, h, h, device='cuda')
    for _ in range(2): eng.decode(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.decode(z)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'decode B={B} latent {h}x{h}: {dt * 1e3:.2f} ms  ({eng.decode_flops() / dt * 1e-12:.0f} TFLOP/s)', flush=True)
