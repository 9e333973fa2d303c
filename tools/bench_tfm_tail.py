#!/usr/bin/env python
"""Stand-alone timing of the fused row-local transformer tail (csrc/kernels_tfm.hip, d = 320) against the back-to-back chain of the
SEVEN launches it replaces in the engine's default plan (to_out + residual GEMM, LayerNorm-on-the-fly to_q GEMM,
77-key cross-attention, to_out + residual GEMM, LayerNorm kernel, GEGLU GEMM, merged [FF2 . proj_out | proj_out] GEMM), same
shapes and epilogues, tuned tile table.  Both are captured in ONE graph (n block evaluations in a row) and replayed; the block's
weights rotate through a pool larger than the 256 MiB Infinity Cache, so every block evaluation streams its 3.3 MB from HBM as in
the sampling loop.  Prints us per block evaluation and the executed TFLOP/s.

    python tools/bench_tfm_tail.py [--rows 4096,8192,16384,32768] [--pool 96]
"""
import argparse
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib      # noqa: E402
from tests.test_gpu_tfm_tail import ORDER, block_weights      # noqa: E402

lib = mlib.load()
DEV = 'cuda:0'
P = lambda t: C.c_void_p(None if t is None else t.data_ptr())
D, HEADS, TK = 320, 8, 77


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(n):
                fn(i)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


class Unfused:
    """the block's weights in the forms mkd_ctx::finalize builds for the unfused plan"""

    def __init__(self, w):
        d = D
        dev = {k: w[k].to(DEV).float().contiguous() for k in ORDER}
        self.wo1 = dev['to_out1_w'].bfloat16(); self.bo1 = dev['to_out1_b']
        self.wo2 = dev['to_out2_w'].bfloat16(); self.bo2 = dev['to_out2_b']
        self.wq = torch.empty(d, d, device=DEV, dtype=torch.bfloat16); self.sq = torch.empty(d, device=DEV); self.bq = torch.empty(d, device=DEV)
        assert lib.mkd_fold_layernorm(P(dev['to_q2_w']), P(dev['norm2_g']), P(dev['norm2_b']), None, d, d, P(self.wq), 0, 1, P(self.sq), P(self.bq), None) == 0
        self.g3, self.b3 = dev['norm3_g'], dev['norm3_b']
        wv, wg = dev['ff0_w'][:4 * d], dev['ff0_w'][4 * d:]
        self.wff = torch.stack([wv, wg], 1).reshape(8 * d, d).bfloat16().contiguous()        # rows (v0, g0, v1, g1, ...)
        self.bff = torch.stack([dev['ff0_b'][:4 * d], dev['ff0_b'][4 * d:]], 1).reshape(8 * d).contiguous()
        pw = dev['proj_out_w'].double()
        self.wm = torch.cat([pw @ dev['ff2_w'].double(), pw], 1).float().bfloat16().contiguous()      # [d, 5d]
        self.bm = (pw @ dev['ff2_b'].double() + dev['proj_out_b'].double()).float()
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', default='4096,8192,16384,32768')
    ap.add_argument('--pool', type=int, default=96)
    ap.add_argument('--n', type=int, default=48)
    args = ap.parse_args()
    d = D
    ws = [block_weights(d, seed=100 + i) for i in range(args.pool)]
    handles = []
    for w in ws:
        h = C.c_void_p()
        dev = {k: w[k].to(DEV).float().contiguous() for k in ORDER}
        mlib.check(lib.mkd_tfm_tail_create(d, *[P(dev[k]) for k in ORDER], C.byref(h)), 'create')
        handles.append(h)
    unf = [Unfused(w) for w in ws]
    torch.cuda.synchronize()
    print(f'# pool {args.pool} blocks x 3.3 MB packed weights; {args.n} block evaluations per graph; d = {d}, {HEADS} heads, {TK} context keys')
    for M in [int(x) for x in args.rows.split(',')]:
        T = 1024 if M % 1024 == 0 and M <= 16384 else 4096
        T = min(T, M)
        B = M // T
        g = torch.Generator().manual_seed(M)
        a1 = torch.randn(M, d, generator=g).to(DEV).bfloat16()
        h0 = torch.randn(M, d, generator=g).to(DEV).bfloat16()
        xin = torch.randn(M, d, generator=g).to(DEV).bfloat16()
        kv = torch.randn(B * TK, 2 * d, generator=g).to(DEV).bfloat16()
        out_f = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
        out_u = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
        for h in handles:
            mlib.check(lib.mkd_tfm_tail_set_context(h, P(kv), 2 * d, B, TK, None), 'ctx')
        h1 = torch.empty(M, d, device=DEV, dtype=torch.bfloat16); q2 = torch.empty_like(h1); a2 = torch.empty_like(h1); y = torch.empty_like(h1)
        cat5 = torch.empty(M, 5 * d, device=DEV, dtype=torch.bfloat16)
        h2 = cat5[:, 4 * d:]
        scale = 1.0 / math.sqrt(d // HEADS)

        def fused(i):
            rc = lib.mkd_tfm_tail_run(handles[i % args.pool], P(a1), d, P(h0), d, P(xin), d, P(out_f), d, M, T, stream())
            assert rc == 0, lib.mkd_last_error()

        def chain_all(i):
            u = unf[i % args.pool]; s = stream()
            assert lib.mkd_gemm_bf16(P(a1), d, P(u.wo1), d, P(u.bo1), None, 0, 1, P(h0), d, 1.0, 0, P(h1), d, 0, M, d, d, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, s) == 0
            assert lib.mkd_gemm_ln_bf16(P(h1), d, P(u.wq), d, P(u.bq), P(u.sq), None, 0, 1e-5, 0, P(q2), d, M, d, d, s) == 0
            assert lib.mkd_attention(P(q2), d, P(kv), 2 * d, C.c_void_p(kv.data_ptr() + 2 * d), 2 * d, P(a2), d, B, T, TK, HEADS, d // HEADS, scale, s) == 0
            assert lib.mkd_gemm_bf16(P(a2), d, P(u.wo2), d, P(u.bo2), None, 0, 1, P(h1), d, 1.0, 0, P(h2), 5 * d, 0, M, d, d, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, s) == 0
            assert lib.mkd_layernorm_ld(P(h2), 5 * d, P(u.g3), P(u.b3), 1e-5, P(y), M, d, s) == 0
            assert lib.mkd_gemm_bf16(P(y), d, P(u.wff), d, P(u.bff), None, 0, 1, None, 0, 1.0, 2, P(cat5), 5 * d, 0, M, 8 * d, d, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, s) == 0, lib.mkd_last_error()
            assert lib.mkd_gemm_bf16(P(cat5), 5 * d, P(u.wm), 5 * d, P(u.bm), None, 0, 1, P(xin), d, 1.0, 0, P(out_u), d, 0, M, d, 5 * d, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, s) == 0, lib.mkd_last_error()

        fl = 2.0 * M * (16.0 * d * d + 2.0 * TK * d)
        rounds = []
        for _ in range(3):                       # interleaved rounds in one process
            rounds.append((timeit(fused, args.n), timeit(chain_all, args.n)))
        tf = min(r[0] for r in rounds); tu = min(r[1] for r in rounds)
        fused(0); chain_all(0); torch.cuda.synchronize()
        diff = ((out_f.float() - out_u.float()).norm() / out_u.float().norm()).item()
        print(f'M={M:6d} (B={B} T={T})  {fl / 1e9:7.2f} GF   fused {tf:7.1f} us {fl / tf * 1e-6:6.1f} TF/s   7 launches {tu:7.1f} us {fl / tu * 1e-6:6.1f} TF/s   '
              f'fused/chain {tf / tu:5.2f}   rounds {[(round(a, 1), round(b, 1)) for a, b in rounds]}   fused vs chain rel-L2 {diff:.2e}', flush=True)
    for h in handles:
        lib.mkd_tfm_tail_destroy(h)


if __name__ == '__main__':
    main()
