#!/usr/bin/env python
"""GPU box: the whole (tile configuration x split-K) landscape of a few convolution shapes of the batch-8 evaluation, best first -
where does the LDS-staged kernel stand against the gather kernel, shape by shape?   python tools/exp_conv_shapes.py [top_n]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import tune_gemm as T  # noqa: E402
from makeupdiffuse_amd import lib as mlib  # noqa: E402

# (M, N, K, conv, stride, up, Hin, Win, Cin, Hout, Wout)
SHAPES = [
    (8192, 320, 2880, 1, 1, 0, 32, 32, 320, 32, 32),
    (2048, 640, 5760, 1, 1, 0, 16, 16, 640, 16, 16),
    (4096, 320, 2880, 1, 1, 0, 32, 32, 320, 32, 32),
    (1024, 640, 5760, 1, 1, 0, 16, 16, 640, 16, 16),
    (512, 1280, 11520, 1, 1, 0, 8, 8, 1280, 8, 8),
    (4096, 320, 5760, 1, 1, 0, 32, 32, 640, 32, 32),
    (128, 1280, 11520, 1, 1, 0, 4, 4, 1280, 4, 4),
]


def main():
    topn = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    lib = mlib.load()
    pool = torch.randn(T.POOL_BYTES // 2, device=T.DEV, dtype=torch.bfloat16) * 0.02
    for shape in SHAPES:
        M, N, K, conv, stride, up, Hin, Win, Cin, Hout, Wout = shape
        A = torch.randn((M // (Hout * Wout)) * Hin * Win * Cin, device=T.DEV, dtype=torch.bfloat16)
        out = torch.empty(M * N, device=T.DEV, dtype=torch.bfloat16)
        lib.mkd_gemm_force_tile(-1)
        t_def = T.time_cfg(lib, shape, -1, 0, pool, A, out)
        trials = []
        for cfg in range(len(T.TILE_M)):
            patch = 6 <= cfg <= 11 or 38 <= cfg <= 40 or cfg in (42, 43)
            tiles = -(-M // T.TILE_M[cfg]) * -(-N // T.TILE_N[cfg])
            units = Cin // 64 if patch else (K + 63) // 64
            for s in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
                if s > 1 and (units // s < (1 if patch else 2) or tiles * s > 2048 or tiles >= 512):
                    continue
                t = T.time_cfg(lib, shape, cfg, s, pool, A, out)
                if t is not None:
                    trials.append((t, cfg, s, tiles * s))
        trials.sort()
        gf = 2.0 * M * N * K / 1e9
        print(f'M={M} N={N} K={K} {Hin}x{Win}: table {t_def:.1f} us ({gf / t_def * 1e-3:.0f} TF/s)')
        for t, cfg, s, wgs in trials[:topn]:
            print(f'    cfg {cfg:2d} ({T.TILE_M[cfg]}x{T.TILE_N[cfg]}{" patch" if (6 <= cfg <= 11 or 38 <= cfg <= 40) else ""}) splitk {s:2d} workgroups {wgs:4d}: {t:6.1f} us  {gf / t * 1e-3:5.0f} TF/s')
        best_patch = [x for x in trials if 6 <= x[1] <= 11 or 38 <= x[1] <= 40][:3]
        for t, cfg, s, wgs in best_patch:
            print(f'    best patch: cfg {cfg} splitk {s} workgroups {wgs}: {t:.1f} us')
        sys.stdout.flush()
    lib.mkd_gemm_force_tile(-1)


if __name__ == '__main__':
    main()
