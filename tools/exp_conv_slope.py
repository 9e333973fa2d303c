#!/usr/bin/env python
"""GPU box: time of one 3x3 convolution as a function of its depth (Cin) at fixed M, N and tile configuration - the intercept is the
per-launch constant, the slope the main loop's rate.   python tools/exp_conv_slope.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import tune_gemm as T  # noqa: E402
from makeupdiffuse_amd import lib as mlib  # noqa: E402


def main():
    lib = mlib.load()
    pool = torch.randn(T.POOL_BYTES // 2, device=T.DEV, dtype=torch.bfloat16) * 0.02
    for (B, H, N) in ((8, 32, 320), (32, 32, 320), (8, 32, 64), (8, 16, 640)):
        for cfg in (9, 38, 6, 42, 8, 40, 43, 36):
            pts = []
            for cin in (64, 320, 640, 1280, 2560):
                M, K = B * H * H, 9 * cin
                shape = (M, N, K, 1, 1, 0, H, H, cin, H, H)
                A = torch.randn(B * H * H * cin, device=T.DEV, dtype=torch.bfloat16)
                out = torch.empty(M * N, device=T.DEV, dtype=torch.bfloat16)
                t = T.time_cfg(lib, shape, cfg, 1, pool, A, out, iters=20)
                if t is None:
                    break
                pts.append((cin, t))
            if len(pts) < 3:
                continue
            (c0, t0), (c1, t1) = pts[1], pts[-1]
            slope = (t1 - t0) / (c1 - c0)                      # us per input channel
            icpt = t0 - slope * c0
            gf_per_c = 2.0 * B * H * H * N * 9 / 1e9            # GFLOP per input channel
            wgs = -(-B * H * H // T.TILE_M[cfg]) * -(-N // T.TILE_N[cfg])
            print(f'B={B} {H}x{H} N={N} cfg {cfg:2d} ({T.TILE_M[cfg]}x{T.TILE_N[cfg]}) workgroups {wgs:4d}: ' +
                  ' '.join(f'{c}:{t:.1f}' for c, t in pts) + f'  | intercept {icpt:.1f} us, marginal {gf_per_c / slope * 1e-3:.0f} TFLOP/s', flush=True)
    lib.mkd_gemm_force_tile(-1)


if __name__ == '__main__':
    main()
