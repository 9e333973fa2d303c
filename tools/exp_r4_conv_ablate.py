#!/usr/bin/env python
"""What the LDS-staged 3x3 convolution kernel waits for: the heavy shapes timed alone (cold weights rotating through a pool, graph
replay) on experiment builds without the weight DMA / the patch DMA / both / the MFMAs (tools/build_variant.sh convexpN
-DMKD_CONV_EXP=N; wrong results on purpose).  MKD_LIB_PATH selects the build."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load(); P = lambda t: C.c_void_p(None if t is None else t.data_ptr())
DEV = 'cuda:0'
SHAPES = [(8, 32, 32, 320, 320, 7), (8, 32, 32, 320, 320, 9), (8, 32, 32, 320, 320, 38), (8, 16, 16, 640, 640, 8), (8, 16, 16, 640, 640, 38), (8, 64, 64, 320, 320, 6), (8, 64, 64, 320, 320, 7), (8, 64, 64, 320, 320, 40), (8, 64, 64, 320, 320, 42), (4, 32, 32, 320, 320, 9), (8, 8, 8, 1280, 1280, 9)]      # B, H, W, Cin, Cout, tile config
for (B, H, W, Cin, Cout, cfg) in SHAPES:
    M = B * H * W
    pool = max(2, (300 << 20) // (Cout * 9 * Cin * 2))
    ws = [torch.randn(Cout, 9 * Cin, device=DEV).bfloat16() * 0.02 for _ in range(pool)]
    x = torch.randn(M, Cin, device=DEV).bfloat16(); y = torch.empty(M, Cout, device=DEV, dtype=torch.bfloat16)
    bias = torch.zeros(Cout, device=DEV)
    lib.mkd_gemm_force_tile(cfg)
    st = torch.cuda.Stream()
    def run(i):
        rc = lib.mkd_gemm_bf16(P(x), Cin, P(ws[i % pool]), 9 * Cin, P(bias), None, 0, 1, None, 0, 1.0, 0, P(y), Cout, 0, M, Cout, 9 * Cin, 1, B, H, W, Cin, H, W, 1, 0, 1,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.mkd_last_error()
    with torch.cuda.stream(st):
        for i in range(3): run(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(32): run(i)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * 32)
    print(f'{os.environ.get("MKD_LIB_PATH", "libmkd.so").split("/")[-1]:24s} M={M} {Cin}->{Cout} cfg {cfg}: {us:7.1f} us  {2.0 * M * Cout * 9 * Cin / us * 1e-6:6.1f} TF/s', flush=True)
    lib.mkd_gemm_force_tile(-1)
