#!/usr/bin/env python
"""Stand-alone timing of the fused transformer head (csrc/kernels_tfm.hip tfm_head_kernel, d = 320: GroupNorm statistics launch +
one launch of GroupNorm apply + proj_in + LayerNorm 1 . q|k|v) against the back-to-back chain of the FOUR launches it replaces in
the engine's plan (GroupNorm, proj_in GEMM, LayerNorm kernel, q|k|v GEMM).  Both captured in one graph (n block evaluations in a
row) and replayed; the block's weights rotate through a pool larger than the 256 MiB Infinity Cache, so every evaluation streams
its 0.8 MB from HBM as in the sampling loop.

    python tools/bench_tfm_head.py [--rows 4096,8192,16384,32768] [--pool 360]
"""
import argparse
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib      # noqa: E402
from tools.bench_tfm_tail import timeit, stream, P      # noqa: E402

lib = mlib.load()
DEV = 'cuda:0'
D = 320
ORDER = ['gn_g', 'gn_b', 'pi_w', 'pi_b', 'n1_g', 'n1_b', 'q_w', 'k_w', 'v_w']


def weights(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    d = D
    return {'gn_g': 1 + 0.2 * r(d), 'gn_b': 0.2 * r(d), 'pi_w': r(d, d) / math.sqrt(d), 'pi_b': 0.1 * r(d), 'n1_g': 1 + 0.2 * r(d),
            'n1_b': 0.2 * r(d), 'q_w': r(d, d) / math.sqrt(d), 'k_w': r(d, d) / math.sqrt(d), 'v_w': r(d, d) / math.sqrt(d)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', default='4096,8192,16384,32768')
    ap.add_argument('--pool', type=int, default=360)
    ap.add_argument('--n', type=int, default=96)
    args = ap.parse_args()
    d = D
    handles, unf = [], []
    for i in range(args.pool):
        w = weights(100 + i % 8)              # (8 distinct weight sets, a separate device copy per pool entry)
        dev = {k: w[k].to(DEV).float().contiguous() for k in ORDER}
        h = C.c_void_p()
        mlib.check(lib.mkd_tfm_head_create(d, *[P(dev[k]) for k in ORDER], C.byref(h)), 'create')
        handles.append(h)
        wqkv = torch.cat([dev['q_w'], dev['k_w'], dev['v_w']], 0).bfloat16().contiguous()
        unf.append((dev['gn_g'], dev['gn_b'], dev['pi_w'].bfloat16().contiguous(), dev['pi_b'], dev['n1_g'], dev['n1_b'], wqkv))
    torch.cuda.synchronize()
    print(f'# pool {args.pool} blocks x 0.8 MB packed weights; {args.n} evaluations per graph; d = {d}')
    for M in [int(x) for x in args.rows.split(',')]:
        T = 1024 if M % 1024 == 0 and M <= 16384 else 4096
        T = min(T, M)
        B = M // T
        g = torch.Generator().manual_seed(M)
        x = torch.randn(M, d, generator=g).to(DEV).bfloat16()
        h0f = torch.empty(M, d, device=DEV, dtype=torch.bfloat16); qkvf = torch.empty(M, 3 * d, device=DEV, dtype=torch.bfloat16)
        gn = torch.empty_like(h0f); h0u = torch.empty_like(h0f); y = torch.empty_like(h0f); qkvu = torch.empty_like(qkvf)

        def fused(i):
            assert lib.mkd_tfm_head_run(handles[i % args.pool], P(x), d, 1e-6, P(h0f), P(qkvf), B, T, stream()) == 0, lib.mkd_last_error()

        def chain(i):
            u = unf[i % args.pool]; s = stream()
            assert lib.mkd_groupnorm(P(x), d, P(u[0]), P(u[1]), 1e-6, 0, P(gn), d, B, T, d, 32, s) == 0, lib.mkd_last_error()
            assert lib.mkd_gemm_bf16(P(gn), d, P(u[2]), d, P(u[3]), None, 0, 1, None, 0, 1.0, 0, P(h0u), d, 0, M, d, d, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, s) == 0
            assert lib.mkd_layernorm_ld(P(h0u), d, P(u[4]), P(u[5]), 1e-5, P(y), M, d, s) == 0
            assert lib.mkd_gemm_bf16(P(y), d, P(u[6]), d, None, None, 0, 1, None, 0, 1.0, 0, P(qkvu), 3 * d, 0, M, 3 * d, d, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, s) == 0

        for i in range(args.pool):          # (every handle sizes its partials workspace on first use: not inside a capture)
            fused(i)
        torch.cuda.synchronize()
        fl = 2.0 * M * 4.0 * d * d
        rounds = [(timeit(fused, args.n), timeit(chain, args.n)) for _ in range(3)]
        tf = min(r[0] for r in rounds); tu = min(r[1] for r in rounds)
        fused(0); chain(0); torch.cuda.synchronize()
        diff = ((qkvf.float() - qkvu.float()).norm() / qkvu.float().norm()).item()
        print(f'M={M:6d} (B={B} T={T})  {fl / 1e9:6.2f} GF   stats+head {tf:6.1f} us {fl / tf * 1e-6:6.1f} TF/s   4 launches {tu:6.1f} us   fused/chain {tf / tu:5.2f}   '
              f'rounds {[(round(a, 1), round(b, 1)) for a, b in rounds]}   q|k|v fused vs chain rel-L2 {diff:.2e}', flush=True)
    for h in handles:
        lib.mkd_tfm_head_destroy(h)


if __name__ == '__main__':
    main()
