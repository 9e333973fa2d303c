#!/usr/bin/env python
"""Experiment: is the 2-stream sampling loop bound by the bytes staged through the CUs' vector-memory path (L2 -> LDS) rather than by
each kernel's isolated latency?  Re-plan every GEMM of the evaluation with the tile configuration that stages the FEWEST bytes
subject to a minimum number of workgroups, and compare the wall time per evaluation with the isolated-time-tuned table."""
import collections
import csv
import re
import sys
import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makeupdiffuse_amd import lib as mlib  # noqa: E402
from makeupdiffuse_amd.engine import MkdEngine, NetConfig  # noqa: E402
from makeupdiffuse_amd.schedule import DDIMSchedule  # noqa: E402

TILE_M = [256, 128, 128, 128, 64, 64, 256, 256, 128, 128, 64, 64, 64, 64, 64, 128, 64, 32, 64, 32, 32, 32, 64, 64, 64, 64, 32, 32, 128, 64]
TILE_N = [128, 128, 128, 64, 128, 64, 128, 64, 128, 64, 128, 64, 64, 128, 160, 160, 160, 64, 32, 32, 32, 32, 32, 32, 64, 64, 64, 64, 64, 128]
KEYS = ('M', 'N', 'K', 'conv', 'stride', 'up', 'Hin', 'Win', 'Cin', 'Hout', 'Wout', 'splitk')


def wall(eng, x_T, steps=20, reps=4):
    sch = DDIMSchedule().make_ddim(steps)
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.sample(x_T, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / steps)
    return best


def main():
    lib = mlib.load()
    B = 8
    eng = MkdEngine(NetConfig()); eng.init_random(0)
    g = torch.Generator().manual_seed(0)
    hint = torch.rand(B, 6, 256, 256, generator=g).cuda(); ctx = torch.randn(B, 77, 768, generator=g).cuda()
    x = torch.randn(B, 4, 32, 32, generator=g).cuda(); t = torch.full((B,), 500).cuda()
    lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, -1, 0)
    eng.prepare(hint, ctx)
    eng.eps_profile(x, t, csv_path='/tmp/ops.csv')
    shapes = collections.OrderedDict()
    cur_bytes = 0
    for r in csv.DictReader(open('/tmp/ops.csv')):
        if not r['kind'].startswith('gemm_'):
            continue
        kv = dict(re.findall(r'(\w+)=(-?\d+)', r['label']))
        key = tuple(int(kv[k]) for k in KEYS)
        shapes[key] = shapes.get(key, 0) + 1
    print(f'{len(shapes)} shapes; table wall {wall(eng, x):.3f} ms/eval', flush=True)

    def staged(M, N, K, cfg, patch):
        tm, tn = TILE_M[cfg], TILE_N[cfg]
        mt, nt = -(-M // tm), -(-N // tn)
        a = mt * nt * tm * K * 2 / (9 if patch else 1) * (1.3 if patch else 1.0)      # patch: the haloed tile is staged once per 9 taps
        return a + mt * nt * tn * K * 2
    for minblocks in (32, 64, 128, 192, 256):
        lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, -1, 0)
        tot = 0
        for sh, n in shapes.items():
            M, N, K, conv, stride, up, Hin, Win, Cin, Hout, Wout, sk = sh
            best = None
            for cfg in range(20):
                if not lib.mkd_gemm_cfg_supported(cfg, M, N, K, conv, Hin, Win, Cin, Hout, Wout, stride, up):
                    continue
                patch = 6 <= cfg <= 11
                if patch and not (conv and stride == 1 and up == 0 and Cin % 64 == 0):
                    continue
                tiles = -(-M // TILE_M[cfg]) * -(-N // TILE_N[cfg])
                units = Cin // 64 if patch else (K + 63) // 64
                s = min(sk, units)
                if tiles * s < min(minblocks, 0.9 * (-(-M // 32) * -(-N // 32))):
                    continue
                b = staged(M, N, K, cfg, patch)
                if best is None or b < best[0]:
                    best = (b, cfg, s)
            if best:
                lib.mkd_gemm_set_override(M, N, K, conv, stride, up, best[1], best[2])
                tot += best[0] * n
        eng.prepare(hint, ctx)
        print(f'min blocks {minblocks}: staged {tot / 1e9:.1f} GB/eval, wall {wall(eng, x):.3f} ms/eval', flush=True)
    lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, -1, 0)


if __name__ == '__main__':
    main()
