#!/usr/bin/env python
"""Headroom check (measurement only, nothing of it is on the product path): the linear-GEMM shapes that take most of an evaluation,
timed through libmkd's gemm_kernel (tuned table) and through torch.matmul (hipBLASLt / rocBLAS, the vendor's kernels) on the same
bf16 operands, weights rotated through a pool so that they come from HBM as in the sampling loop.  Prints us per launch and TFLOP/s."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load()
DEV = 'cuda:0'
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(8192, 320, 320), (4096, 320, 320), (2048, 640, 640), (1024, 640, 640), (512, 1280, 1280), (256, 1280, 1280),
          (8192, 2560, 320), (4096, 2560, 320), (2048, 5120, 640), (512, 10240, 1280), (8192, 320, 1600), (2048, 640, 3200), (512, 1280, 6400),
          (8192, 960, 320), (2048, 1920, 640), (512, 3840, 1280), (616, 2560, 768)]
NPOOL = 24


def timeit(fn, n=48):
    """n launches captured in ONE graph and replayed: device time per launch without the host's per-call cost (torch.matmul alone
    takes ~18 us of host time per call, more than most of these kernels run)."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(6): fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(n): fn(i)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


for (M, N, K) in shapes:
    A = torch.randn(M, K, device=DEV).bfloat16()
    Ws = [torch.randn(N, K, device=DEV).bfloat16() / K ** 0.5 for _ in range(NPOOL)]
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=DEV)

    def ours(i):
        rc = lib.mkd_gemm_bf16(P(A), K, P(Ws[i % NPOOL]), K, P(bias), None, 0, 1, None, 0, 1.0, 0, P(out), N, 0, M, N, K, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.mkd_last_error()

    def vendor(i):
        torch.matmul(A, Ws[i % NPOOL].t(), out=out)

    t_o, t_v = timeit(ours), timeit(vendor)
    ref = (A.float() @ Ws[(48 + 5) % NPOOL].float().t())
    fl = 2.0 * M * N * K
    print(f'M={M:5d} N={N:5d} K={K:5d}  {fl / 1e9:6.2f} GF   libmkd {t_o:7.2f} us {fl / t_o * 1e-6:7.1f} TF/s   vendor {t_v:7.2f} us {fl / t_v * 1e-6:7.1f} TF/s   ratio {t_o / t_v:5.2f}', flush=True)
