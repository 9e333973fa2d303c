#!/usr/bin/env python
"""Phase trace of the fused transformer tail (csrc/kernels_tfm.hip): needs a -DMKD_TFM_TRACE build (tools/build_variant.sh tfmtrace
-DMKD_TFM_TRACE; MKD_LIB_PATH=makeupdiffuse_amd/libmkd_tfmtrace.so).  wall_clock64 stamps (100 MHz) by lane 0 of every wave at the
stage boundaries; prints the median over workgroups of every interval for wave 0 and wave 7, in microseconds."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
from tests.test_gpu_tfm_tail import ORDER, block_weights
lib = mlib.load()
lib.mkd_debug_tfm_trace.argtypes = [C.c_void_p]
DEV = 'cuda:0'
P = lambda t: C.c_void_p(t.data_ptr())
NAMES = {1: 'a1 tile load + barrier', 2: 'S1 k-loop', 3: 'S1 epilogue', 4: 'S1 barrier', 5: 'S2 stats + k-loop', 6: 'S2 epilogue', 7: 'S2 barrier',
         8: 'S3 cross-attention', 9: 'S3 barrier', 10: 'S4 k-loop', 11: 'S4 epilogue', 12: 'S4 barrier', 13: 'chunk 0 (G k-loop, epilogue, barrier, M) + G(1) k-loop',
         14: 'G(1) epilogue', 15: 'G(1) barrier', 16: 'M(1) k-loop', 17: 'chunks 2..4', 18: 'M tail k-loop', 19: 'output epilogue'}
d = 320
for M, T in [(4096, 1024), (8192, 1024), (32768, 4096)]:
    B = M // T
    pool = 24
    hs = []
    for i in range(pool):
        w = block_weights(d, seed=i)
        dev = {k: w[k].to(DEV).float().contiguous() for k in ORDER}
        h = C.c_void_p(); mlib.check(lib.mkd_tfm_tail_create(d, *[P(dev[k]) for k in ORDER], C.byref(h)), 'create'); hs.append(h)
    g = torch.Generator().manual_seed(1)
    a1 = torch.randn(M, d, generator=g).to(DEV).bfloat16(); h0 = torch.randn(M, d, generator=g).to(DEV).bfloat16()
    xin = torch.randn(M, d, generator=g).to(DEV).bfloat16(); kv = torch.randn(B * 77, 2 * d, generator=g).to(DEV).bfloat16()
    out = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    for h in hs:
        mlib.check(lib.mkd_tfm_tail_set_context(h, P(kv), 2 * d, B, 77, None), 'ctx')
    nwg = M // 64
    tr = torch.zeros(nwg, 8, 32, dtype=torch.int64, device=DEV)
    lib.mkd_debug_tfm_trace(P(tr))
    junk = torch.empty(64 << 20, device=DEV)
    for i in range(pool):                     # every block's weights cold, the traced one last
        junk.normal_()
        mlib.check(lib.mkd_tfm_tail_run(hs[i], P(a1), d, P(h0), d, P(xin), d, P(out), d, M, T, None), 'run')
    torch.cuda.synchronize()
    t = tr.cpu().double() / 100.0           # us
    t0 = t[:, :, 0].min()
    print(f'--- M = {M} ({nwg} workgroups): kernel span {float(t[:, :, 19].max() - t0):.1f} us; first-stamp spread {float(t[:, :, 0].max() - t0):.1f} us; '
          f'per-workgroup duration median {float((t[:, :, 19].max(1).values - t[:, :, 0].min(1).values).median()):.1f} us')
    for wv in (0, 7):
        parts = []
        for i in range(1, 20):
            dt = (t[:, wv, i] - t[:, wv, i - 1]).median().item()
            parts.append(f'{NAMES[i]} {dt:.2f}')
        print(f'  wave {wv}: ' + ' | '.join(parts))
    lib.mkd_debug_tfm_trace(None)
    for h in hs:
        lib.mkd_tfm_tail_destroy(h)
