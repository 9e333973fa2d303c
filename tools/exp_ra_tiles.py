#!/usr/bin/env python
"""Register-A GEMM tiles (configurations 44-49, gemm_ra_kernel) against the tuned plan of gemm_kernel / conv3x3_patch_kernel on the
heavy shapes of an evaluation: weights rotated through a pool (cold), 32 launches captured in one graph and replayed (device time)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load()
DEV = 'cuda:0'
P = lambda t: C.c_void_p(t.data_ptr())
# (M, N, K, conv, H, W, Cin)   conv: stride 1, pad 1
LIN = [(8192, 320, 320), (8192, 2560, 320), (8192, 320, 1600), (8192, 960, 320), (2048, 5120, 640), (2048, 640, 3200), (2048, 640, 640),
       (512, 10240, 1280), (512, 1280, 6400), (4096, 2560, 320), (4096, 320, 1600)]
CONV = [(8, 32, 32, 320, 320), (4, 32, 32, 640, 320), (4, 32, 32, 960, 320), (8, 16, 16, 640, 640), (4, 16, 16, 1280, 640), (8, 8, 8, 1280, 1280), (4, 8, 8, 2560, 1280)]
NPOOL = 12
RA = [int(v) for v in os.environ.get('MKD_EXP_CFGS', '44,45,46,47,48,49').split(',')]          # candidate tile configurations
SPLITS = [1, 2, 3, 4, 6, 8]


def timeit(fn, n=32):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(3): fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(n): fn(i)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (4 * n)


def bench(M, N, K, cv, A, lda, label):
    Ws = [torch.randn(N, K, device=DEV).bfloat16() / K ** 0.5 for _ in range(NPOOL)]
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=DEV)

    def run(splitk):
        def f(i):
            rc = lib.mkd_gemm_bf16(P(A), lda, P(Ws[i % NPOOL]), K, P(bias), None, 0, 1, None, 0, 1.0, 0, P(out), N, 0, M, N, K, *cv, splitk,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, lib.mkd_last_error()
        return f
    lib.mkd_gemm_force_tile(-1)
    t_tab = timeit(run(0))
    ref = out.clone()
    best = None
    res = []
    for cfg in RA:
        lib.mkd_gemm_force_tile(cfg)
        for sk in SPLITS:
            if sk > 1 and (K // 64) // sk < 4:
                continue
            try:
                t = timeit(run(sk))
            except AssertionError:
                continue
            res.append((t, cfg, sk))
    lib.mkd_gemm_force_tile(-1)
    res.sort()
    fl = 2.0 * M * N * K
    top = ' '.join(f'[{c} s{k}: {t:.1f}]' for t, c, k in res[:4])
    print(f'{label:34s} {fl / 1e9:6.1f} GF  tuned {t_tab:6.1f} us {fl / t_tab * 1e-6:6.0f} TF/s | best RA {res[0][0]:6.1f} us {fl / res[0][0] * 1e-6:6.0f} TF/s  x{t_tab / res[0][0]:.2f} | {top}', flush=True)


for (M, N, K) in LIN:
    A = torch.randn(M, K, device=DEV).bfloat16()
    bench(M, N, K, (0, 0, 0, 0, 0, 0, 0, 1, 0), A, K, f'linear {M}x{N}x{K}')
for (B, H, W, Cin, Cout) in CONV:
    x = torch.randn(B, H, W, Cin, device=DEV).bfloat16()
    M = B * H * W
    bench(M, Cout, 9 * Cin, (1, B, H, W, Cin, H, W, 1, 0), x, Cin, f'conv3x3 B{B} {H}x{W} {Cin}->{Cout}')
