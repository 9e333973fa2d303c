#!/usr/bin/env python
"""Per-phase cycle sums of the streaming self-attention kernel (needs tools/build_variant.sh attntrace -DMKD_ATTN_TRACE and
MKD_LIB_PATH): clock64 deltas accumulated per wave over all key tiles: barrier 1 (previous tile consumed), registers -> LDS stores,
barrier 2, prefetch issue, QK^T MFMAs (to the first use of the scores), softmax, P.V + loop tail."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load(); P = lambda t: C.c_void_p(t.data_ptr())
NAMES = ['barrier 1', 'LDS stores', 'barrier 2', 'prefetch issue', 'QK^T', 'softmax', 'P.V + tail']
# (LDS-DMA kernel, dh 40 from 1024 keys: 'barrier 1' = wait for this thread's DMA, 'LDS stores' = the barrier, 'barrier 2' unused, 'prefetch issue' = DMA issue; tiles of 128 keys)
for (B, T, H, dh) in ((8, 4096, 8, 40), (8, 1024, 8, 40), (8, 1024, 8, 80)):
    d = H * dh
    q = torch.randn(B * T, d, device='cuda').bfloat16(); k = torch.randn(B * T, d, device='cuda').bfloat16(); v = torch.randn(B * T, d, device='cuda').bfloat16(); o = torch.empty_like(q)
    nwg = (T // 128) * B * H
    tr = torch.zeros(nwg, 8, 8, dtype=torch.int64, device='cuda')
    assert lib.mkd_debug_attn_trace(P(tr)) == 0
    for _ in range(2):
        assert lib.mkd_attention(P(q), d, P(k), d, P(v), d, P(o), d, B, T, T, H, dh, dh ** -0.5, None) == 0
    torch.cuda.synchronize()
    t = tr.cpu().double()
    tot = t[:, :, :7].sum(-1)
    ntile = T // (128 if dh == 40 and os.environ.get('MKD_ATTN_DMA', '1') != '0' else 64)
    print(f'B={B} T={T} dh={dh}: cycles per wave per tile {tot.mean().item() / ntile:.0f}: ' +
          ' | '.join(f'{NAMES[i]} {t[:, :, i].mean().item() / ntile:.0f}' for i in range(7)))
    lib.mkd_debug_attn_trace(None)
