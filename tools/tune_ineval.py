#!/usr/bin/env python
"""In-eval GEMM tuner (run on the GPU box).

tools/tune_gemm.py times each layer shape in isolation (same activations every launch, hot in cache).  Inside one eps
evaluation a kernel's input was just written by the previous kernel and its neighbours differ, and the two disagree by up
to +-10 % per shape.  This tuner measures every candidate (tile config, split-K) IN PLACE: it installs the candidate as a
run-time override for every shape it is valid for, re-plans, and reads the per-launch HIP-event times of one serial eps
evaluation (mkd_eps_profile).  A shape keeps a candidate only when it beats the current table in the same run by > 3 %.

    python tools/tune_ineval.py --batch 8 --res 256 --out gpurun_out/ineval_b8_r256.json
"""
import argparse
import collections
import csv
import json
import os
import re
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makeupdiffuse_amd import lib as mlib  # noqa: E402
from makeupdiffuse_amd.engine import MkdEngine, NetConfig  # noqa: E402
from makeupdiffuse_amd.schedule import DDIMSchedule  # noqa: E402

TILE_M = [256, 128, 128, 128, 64, 64, 256, 256, 128, 128, 64, 64, 64, 64, 64, 128, 64, 32, 64, 32, 32, 32, 64, 64, 64, 64, 32, 32, 128, 64, 64, 128, 64, 64, 128, 128, 64, 64, 128, 64, 128, 256, 256, 128, 256, 256, 128, 128, 128, 256, 256]
TILE_N = [128, 128, 128, 64, 128, 64, 128, 64, 128, 64, 128, 64, 64, 128, 160, 160, 160, 64, 32, 32, 32, 32, 32, 32, 64, 64, 64, 64, 64, 128, 64, 64, 128, 32, 128, 64, 128, 64, 64, 128, 128, 64, 128, 128, 64, 128, 128, 64, 160, 64, 256]
SPLITS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24)
KEYS = ('M', 'N', 'K', 'conv', 'stride', 'up', 'Hin', 'Win', 'Cin', 'Hout', 'Wout')


def profile(eng, x, t, reps, path='/tmp/ineval_ops.csv'):
    """-> {shape: (launches per eval, median us per launch)} for the GEMM launches of one serial eps evaluation."""
    acc = collections.defaultdict(list)
    count = {}
    for _ in range(reps):
        eng.eps_profile(x, t, csv_path=path)
        per = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            if not r['kind'].startswith('gemm_'):
                continue
            kv = dict(re.findall(r'(\w+)=(-?\d+)', r['label']))
            key = tuple(int(kv[k]) for k in KEYS)
            per[key][0] += 1
            per[key][1] += float(r['ms']) * 1e3
        for k, (n, us) in per.items():
            acc[k].append(us / n)
            count[k] = n
    return {k: (count[k], statistics.median(v)) for k, v in acc.items()}


def valid(lib, shape, cfg, s):
    M, N, K, conv, stride, up, Hin, Win, Cin, Hout, Wout = shape
    if not lib.mkd_gemm_cfg_supported(cfg, M, N, K, conv, Hin, Win, Cin, Hout, Wout, stride, up):
        return False
    patch = 6 <= cfg <= 11 or 38 <= cfg <= 40 or cfg in (42, 43)
    tiles = -(-M // TILE_M[cfg]) * -(-N // TILE_N[cfg])
    units = Cin // 64 if patch else (K + 63) // 64
    if s == 1:
        return True
    if units // s < (1 if patch else 2) or tiles * s > 2048 or tiles >= 512:
        return False
    return s * M * N * 4 <= (256 << 20)


def wall_ms_per_eval(eng, x_T, steps=20, reps=3):
    sch = DDIMSchedule().make_ddim(steps)
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.sample(x_T, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / steps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--cfgs', default=None, help='comma-separated tile configurations to try (default: all)')
    ap.add_argument('--out', default='gpurun_out/ineval.json')
    args = ap.parse_args()
    lib = mlib.load()
    eng = MkdEngine(NetConfig()); eng.init_random(0)
    g = torch.Generator().manual_seed(0)
    h = args.res // 8
    hint = torch.rand(args.batch, 6, args.res, args.res, generator=g).cuda()
    ctx = torch.randn(args.batch, 77, 768, generator=g).cuda()
    x = torch.randn(args.batch, 4, h, h, generator=g).cuda()
    t = torch.full((args.batch,), 500).cuda()

    def replan():
        eng.prepare(hint, ctx)

    lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, -1, 0)
    replan()
    base = profile(eng, x, t, 2 * args.reps + 1)
    wall0 = wall_ms_per_eval(eng, x)
    shapes = sorted(base, key=lambda k: -base[k][0] * base[k][1])
    print(f'{len(shapes)} GEMM shapes, {sum(n * us for n, us in base.values()) / 1e3:.3f} ms of serial GEMM time per eval, '
          f'wall {wall0:.3f} ms/eval', flush=True)
    trials = collections.defaultdict(list)
    t_start = time.time()
    for cfg in ([int(c) for c in args.cfgs.split(',')] if args.cfgs else range(len(TILE_M))):
        for s in SPLITS:
            todo = [sh for sh in shapes if valid(lib, sh, cfg, s)]
            if not todo:
                continue
            lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, -1, 0)
            for sh in todo:
                lib.mkd_gemm_set_override(*sh[:6], cfg, s)
            replan()
            res = profile(eng, x, t, args.reps)
            for sh in todo:
                if sh in res:
                    trials[sh].append((cfg, s, round(res[sh][1], 2)))
        print(f'cfg {cfg} done ({time.time() - t_start:.0f} s)', flush=True)
    # pick, apply, verify
    lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, -1, 0)
    picks = {}
    for sh in shapes:
        n, us0 = base[sh]
        if not trials[sh]:
            continue
        cfg, s, us = min(trials[sh], key=lambda r: r[2])
        if us < 0.97 * us0 and (us0 - us) * n > 1.0:
            picks[sh] = (cfg, s, us)
            lib.mkd_gemm_set_override(*sh[:6], cfg, s)
    replan()
    after = profile(eng, x, t, 2 * args.reps + 1)
    wall1 = wall_ms_per_eval(eng, x)
    out = {}
    for sh, (cfg, s, us) in picks.items():
        n, us0 = base[sh]
        us1 = after[sh][1]
        keep = us1 < 0.985 * us0                      # must hold up when all picks are applied together
        print(f'M={sh[0]} N={sh[1]} K={sh[2]} conv={sh[3]} s={sh[4]} up={sh[5]} x{n}: table {us0:.1f} us -> cfg {cfg} splitk {s} '
              f'{us:.1f} us (together {us1:.1f}){"" if keep else "  DROPPED"}', flush=True)
        if keep:
            out['_'.join(map(str, sh[:6]))] = {'shape': list(sh), 'count': n, 'best_cfg': cfg, 'best_splitk': s, 'best_us': us1,
                                               'default_us': us0, 'trials': sorted(trials[sh], key=lambda r: r[2])[:6]}
    tot0 = sum(n * us for n, us in base.values()); tot1 = sum(n * us for n, us in after.values())
    print(f'serial GEMM time per eval: {tot0 / 1e3:.3f} -> {tot1 / 1e3:.3f} ms; wall {wall0:.3f} -> {wall1:.3f} ms/eval; '
          f'{len(out)} shapes changed', flush=True)
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(out, open(args.out, 'w'), indent=1)
    lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, -1, 0)
    eng.close()


if __name__ == '__main__':
    main()
