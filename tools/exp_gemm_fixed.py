#!/usr/bin/env python
"""GPU box: the fixed cost of one GEMM launch per tile configuration, in a replayed graph (clean device timeline): one tile / a full
grid, K = 64 .. 1024.  Intercept = launch + prologue + epilogue, slope = one K-step.    python tools/exp_gemm_fixed.py"""
import os
import sys
sys.argv = ['x']
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'exp_floor_graph.py')).read()
exec(src[:src.index("ln(4, 64); ln(256, 1280)")])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tune_gemm as T  # noqa: E402  (tile sizes)

for cfg in (19, 5, 3, 1, 0, 14, 36, 21, 24):
    tm, tn = T.TILE_M[cfg], T.TILE_N[cfg]
    for (M, N_) in ((tm, tn), (tm * 16, tn * 16)):
        for K in (64, 128, 256, 512, 1024):
            gemm(M, N_, K, cfg)
