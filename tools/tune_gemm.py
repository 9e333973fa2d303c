#!/usr/bin/env python
"""Offline GEMM tuner (run on the GPU box): times every tile config x split-K for each distinct GEMM/conv shape of
one eps evaluation and writes makeupdiffuse_amd/csrc/gemm_tuned.inc (shape -> config).  Weights rotate through a
pool larger than the Infinity Cache so they are read cold, as in the real evaluation; activations stay hot.

    python tools/tune_gemm.py --batch 8 --res 256 [--batch 16 ...] --out gpurun_out/tune.json
"""
import argparse
import collections
import csv
import ctypes as C
import json
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makeupdiffuse_amd import lib as mlib  # noqa: E402
from makeupdiffuse_amd.engine import MkdEngine, NetConfig  # noqa: E402

DEV = 'cuda:0'
TILE_M = [256, 128, 128, 128, 64, 64, 256, 256, 128, 128, 64, 64, 64, 64, 64, 128, 64, 32, 64, 32, 32, 32, 64, 64, 64, 64, 32, 32, 128, 64, 64, 128, 64, 64, 128, 128, 64, 64, 128, 64, 128, 256, 256, 128, 256, 256, 128, 128, 128, 256, 256]
TILE_N = [128, 128, 128, 64, 128, 64, 128, 64, 128, 64, 128, 64, 64, 128, 160, 160, 160, 64, 32, 32, 32, 32, 32, 32, 64, 64, 64, 64, 64, 128, 64, 64, 128, 32, 128, 64, 128, 64, 64, 128, 128, 64, 128, 128, 64, 128, 128, 64, 160, 64, 256]
POOL_BYTES = 640 << 20


def shapes_of(eng, batch, res):
    h = res // 8
    g = torch.Generator().manual_seed(0)
    eng.prepare(torch.rand(batch, 6, res, res, generator=g), torch.randn(batch, 77, 768, generator=g))
    path = '/tmp/ops_shapes.csv'
    eng.eps_profile(torch.randn(batch, 4, h, h, generator=g), torch.full((batch,), 500), csv_path=path)
    out = collections.Counter()
    for r in csv.DictReader(open(path)):
        if not r['kind'].startswith('gemm_'):
            continue
        kv = dict(re.findall(r'(\w+)=(-?\d+)', r['label']))
        key = tuple(int(kv[k]) for k in ('M', 'N', 'K', 'conv', 'stride', 'up', 'Hin', 'Win', 'Cin', 'Hout', 'Wout'))
        out[key] += 1
    return out


def vae_shapes(batch, res):
    """GEMM / conv shapes of the first-stage decoder (ch 128, mult 1,2,4,4, 2 res blocks) for `batch` images."""
    out = collections.Counter()
    h = res // 8
    ch, mult, nrb = 128, (1, 2, 4, 4), 2
    def conv(H, cin, cout, up=0):
        Ho = H << up
        out[(batch * Ho * Ho, cout, 9 * cin, 1, 1, up, H, H, cin, Ho, Ho)] += 1
    def lin(M, N, K):
        out[(M, N, K, 0, 0, 0, 0, 0, 0, 0, 0)] += 1
    def res_(H, cin, cout):
        conv(H, cin, cout); conv(H, cout, cout)
        if cin != cout:
            lin(batch * H * H, cout, cin)
    bi = ch * mult[-1]; H = h
    res_(H, bi, bi); res_(H, bi, bi)
    T = H * H
    lin(batch * T, 2 * bi, bi); lin(batch * T, bi, bi)
    for _ in range(batch):
        lin(bi, T, bi); lin(T, T, bi); lin(T, bi, T)
    for lvl in reversed(range(4)):
        bo = ch * mult[lvl]
        for j in range(nrb + 1):
            res_(H, bi, bo); bi = bo
        if lvl:
            conv(H, bi, bi, up=1); H *= 2
    return out


def time_cfg(lib, shape, cfg, splitk, pool, A, out, iters=12):
    M, N, K, conv, stride, up, Hin, Win, Cin, Hout, Wout = shape
    wbytes = N * K * 2
    ncopy = max(1, min(POOL_BYTES // wbytes, 4096))
    lib.mkd_gemm_force_tile(cfg)
    batch = M // max(1, Hout * Wout) if conv else 0
    lda = Cin if conv else K

    def run(i):
        wptr = pool.data_ptr() + (i % ncopy) * wbytes
        return lib.mkd_gemm_bf16(C.c_void_p(A.data_ptr()), lda, C.c_void_p(wptr), K, None, None, 0, 1, None, 0, 1.0, 0,
                                 C.c_void_p(out.data_ptr()), N, 0, M, N, K, conv, batch, Hin, Win, Cin, Hout, Wout, stride, up,
                                 splitk, None)
    for i in range(3):
        if run(i) != 0:
            return None
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(3 + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, action='append')
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--out', default='gpurun_out/tune.json')
    ap.add_argument('--cfgs', default=None, help='comma list: time only these tile configs (merge with earlier sweeps via gen_tuned_table.py); '
                    'shapes where none beats the HEURISTIC default are skipped')
    ap.add_argument('--vae', action='store_true', help='tune the first-stage decoder shapes instead of the eps plan')
    args = ap.parse_args()
    lib = mlib.load()
    shapes = collections.Counter()
    if args.vae:
        for b in (args.batch or [8]):
            shapes.update(vae_shapes(b, args.res))
    else:
        eng = MkdEngine(NetConfig())
        eng.init_random(0)
        for b in (args.batch or [8]):
            shapes.update(shapes_of(eng, b, args.res))
        eng.close()
        del eng
    torch.cuda.empty_cache()
    pool = torch.randn(POOL_BYTES // 2, device=DEV, dtype=torch.bfloat16) * 0.02
    results = {}
    total_best = total_default = 0.0
    for si, (shape, count) in enumerate(sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2])):
        M, N, K, conv, stride, up, Hin, Win, Cin, Hout, Wout = shape
        nA = (M // max(1, Hout * Wout)) * Hin * Win * Cin if conv else M * K
        A = torch.randn(nA, device=DEV, dtype=torch.bfloat16)
        out = torch.empty(M * N, device=DEV, dtype=torch.bfloat16)
        nk = (K + 63) // 64
        lib.mkd_gemm_force_tile(-1)
        t_def = time_cfg(lib, shape, -1, 0, pool, A, out, iters=6 if M * N * K > 4e11 else 12)
        if t_def is None:          # (a shape this harness cannot launch alone, e.g. a convolution with a folded second input: tuned through its base shape)
            continue
        best = (None, None, 1e30)
        trials = []
        only = [int(c) for c in args.cfgs.split(',')] if args.cfgs else None
        for cfg in (only if only else range(len(TILE_M))):
            if N % 128 and TILE_N[cfg] == 128 and N < 128:
                continue
            patch = 6 <= cfg <= 11 or 38 <= cfg <= 40 or cfg in (42, 43)
            if patch and not (conv and stride == 1 and up == 0 and Cin % 64 == 0):
                continue
            tiles = -(-M // TILE_M[cfg]) * -(-N // TILE_N[cfg])
            units = Cin // 64 if patch else nk
            for s in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24):
                if s > 1 and (units // s < (1 if patch else 2) or tiles * s > 2048 or tiles >= 512):
                    continue
                if s > 1 and s * M * N * 4 > (256 << 20):
                    continue
                t = time_cfg(lib, shape, cfg, s, pool, A, out, iters=4 if M * N * K > 4e11 else 12)
                if t is None:
                    continue
                trials.append((cfg, s, round(t, 2)))
                if t < best[2]:
                    best = (cfg, s, t)
        gf = 2.0 * M * N * K / 1e9
        if only:
            if best[0] is None or best[2] > 0.97 * t_def:
                total_best += t_def * count; total_default += t_def * count
                print(f'[{si + 1}/{len(shapes)}] M={M} N={N} K={K} conv={conv} x{count}: table {t_def:.1f} us stays (best new {best[2]:.1f})', flush=True)
                continue
        results['_'.join(map(str, shape[:6]))] = {'shape': shape, 'count': count, 'best_cfg': best[0], 'best_splitk': best[1],
                                                  'best_us': best[2], 'default_us': t_def, 'tflops': gf / best[2] * 1e-3,
                                                  'n_trials': len(trials)}      # (raw per-trial timings are not kept: only the winner is used)
        total_best += best[2] * count
        total_default += t_def * count
        print(f'[{si + 1}/{len(shapes)}] M={M} N={N} K={K} conv={conv} s={stride} up={up} x{count}: default {t_def:.1f} us -> '
              f'best cfg {best[0]} splitk {best[1]} {best[2]:.1f} us ({gf / best[2] * 1e-3:.0f} TF/s)', flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(results, open(args.out, 'w'), indent=1)
    lib.mkd_gemm_force_tile(-1)
    print(f'sum over shapes x count: default {total_default / 1e3:.2f} ms -> tuned {total_best / 1e3:.2f} ms')


if __name__ == '__main__':
    main()
