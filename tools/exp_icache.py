#!/usr/bin/env python
"""GPU box, under rocprofv3 --pmc SQC_ICACHE_*: eager launches of ONE-tile GEMMs (one K-step) for a small and a large tile configuration, each
launched 50 times back to back, then alternating with a different kernel in between - do instruction-cache misses explain the
fixed cost per launch that grows with the tile?      rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES -- python3 tools/exp_icache.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib  # noqa: E402

lib = mlib.load()
DEV = 'cuda:0'
P = lambda t: C.c_void_p(None if t is None else t.data_ptr())
bf = lambda *s: (torch.randn(*s, device=DEV) * 0.1).to(torch.bfloat16)
g1 = torch.ones(2560, device=DEV); b1 = torch.zeros(2560, device=DEV)
xl = bf(256, 1280); yl = torch.empty_like(xl)


def gemm(M, N, K, cfg):
    A = bf(M, K); W = bf(N, K); out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    lib.mkd_gemm_force_tile(cfg)
    for _ in range(50):
        lib.mkd_gemm_bf16(P(A), K, P(W), K, None, None, 0, 1, None, 0, 1.0, 0, P(out), N, 0, M, N, K, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, None)
    torch.cuda.synchronize()
    for _ in range(50):      # a different kernel between two launches
        lib.mkd_gemm_bf16(P(A), K, P(W), K, None, None, 0, 1, None, 0, 1.0, 0, P(out), N, 0, M, N, K, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, None)
        lib.mkd_layernorm(P(xl), P(g1), P(b1), 1e-5, P(yl), 256, 1280, None)
    torch.cuda.synchronize()
    lib.mkd_gemm_force_tile(-1)


gemm(32, 32, 64, 19)
gemm(128, 128, 64, 1)
gemm(2048, 2048, 64, 1)
gemm(256, 128, 64, 0)
