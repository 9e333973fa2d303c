#!/usr/bin/env python
"""GPU box: what does ONE dependent launch of each kernel family cost at a size where it has next to no work?  N launches captured
into a graph (one stream, every node reads what the previous one wrote), replayed; compare with tools/micro/launch_floor.hip's
1.56 us for an empty kernel.    python tools/exp_floor_graph.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib  # noqa: E402

lib = mlib.load()
DEV = 'cuda:0'
P = lambda t: C.c_void_p(None if t is None else t.data_ptr())
N = 200


def graph_time(name, fn_pair):
    """fn_pair(stream, i): enqueue launch i (ping-pong buffers by i & 1) on `stream`."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(4):
            fn_pair(C.c_void_p(s.cuda_stream), i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(N):
            fn_pair(C.c_void_p(s.cuda_stream), i)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f'{name:64s} {best * 1e3 / N:6.2f} us per node', flush=True)


bf = lambda *s: (torch.randn(*s, device=DEV) * 0.1).to(torch.bfloat16)
g1 = torch.ones(2560, device=DEV); b1 = torch.zeros(2560, device=DEV)


def ln(rows, d):
    x = [bf(rows, d), bf(rows, d)]
    graph_time(f'layernorm rows={rows} d={d}', lambda st, i: lib.mkd_layernorm(P(x[i & 1]), P(g1), P(b1), 1e-5, P(x[1 - (i & 1)]), rows, d, st))


def gn(B, hw, Cc):
    x = [bf(B, hw, Cc), bf(B, hw, Cc)]
    graph_time(f'groupnorm B={B} HW={hw} C={Cc}', lambda st, i: lib.mkd_groupnorm(P(x[i & 1]), Cc, P(g1), P(b1), 1e-5, 1, P(x[1 - (i & 1)]), Cc, B, hw, Cc, 32, st))


def gemm(M, N_, K, cfg=-1, splitk=1):
    A = [bf(M, K), bf(M, K)] if N_ == K else None
    W = bf(N_, K) * 0.1
    if A is None:
        a0 = bf(M, K); out = torch.empty(M, N_, device=DEV, dtype=torch.bfloat16)
        fn = lambda st, i: lib.mkd_gemm_bf16(P(a0), K, P(W), K, None, None, 0, 1, None, 0, 1.0, 0, P(out), N_, 0, M, N_, K, 0, 0, 0, 0, 0, 0, 0, 1, 0, splitk, st)
    else:
        fn = lambda st, i: lib.mkd_gemm_bf16(P(A[i & 1]), K, P(W), K, None, None, 0, 1, None, 0, 1.0, 0, P(A[1 - (i & 1)]), N_, 0, M, N_, K, 0, 0, 0, 0, 0, 0, 0, 1, 0, splitk, st)
    lib.mkd_gemm_force_tile(cfg)
    graph_time(f'gemm M={M} N={N_} K={K} cfg={cfg} splitk={splitk}' + (' (chained)' if A is not None else ' (same input)'), fn)
    lib.mkd_gemm_force_tile(-1)


def attn(B, T, Tk, heads, dh):
    d = heads * dh
    q = bf(B * T, d); kv = bf(B * Tk, 2 * d); o = torch.empty_like(q)
    graph_time(f'attention B={B} Tq={T} Tk={Tk} heads={heads} dh={dh}',
               lambda st, i: lib.mkd_attention(P(q), d, P(kv), 2 * d, C.c_void_p(kv.data_ptr() + 2 * d), 2 * d, P(o), d, B, T, Tk, heads, dh, dh ** -0.5, st))


ln(4, 64); ln(256, 1280); ln(8192, 320)
gn(8, 16, 1280); gn(8, 64, 1280); gn(8, 256, 640); gn(8, 1024, 320); gn(4, 1024, 320)
for cfg in (19, 5, 3, 1):
    gemm(64, 64, 64, cfg)
gemm(128, 1280, 1280); gemm(256, 1280, 1280); gemm(512, 1280, 1280); gemm(1024, 640, 640); gemm(4096, 320, 320); gemm(8192, 320, 320)
gemm(8192, 2560, 320); gemm(8192, 320, 1600); gemm(128, 1280, 6400)
attn(8, 16, 16, 8, 160); attn(8, 64, 77, 8, 160); attn(8, 256, 77, 8, 80); attn(8, 1024, 77, 8, 40); attn(8, 1024, 1024, 8, 40); attn(4, 1024, 1024, 8, 40)
