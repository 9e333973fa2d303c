import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from gpu_util import *
lib = L()
for (B, Tq, Tk, heads, dh) in [(1, 1024, 1024, 2, 40), (1, 1024, 1024, 8, 40), (1, 1024, 1024, 2, 80), (1, 1024, 1024, 2, 64), (1, 1024, 1024, 1, 160), (2, 2048, 2048, 1, 40), (1, 1024, 1024, 1, 32), (1,1024,1024,1,16)]:
    g = torch.Generator().manual_seed(1)
    d = heads * dh
    q = bf(torch.randn(B * Tq, d, generator=g)); k = bf(torch.randn(B*Tk, d, generator=g)); v = bf(torch.randn(B*Tk, d, generator=g))
    o = torch.zeros(B * Tq, d, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_attention(P(q), d, P(k), d, P(v), d, P(o), d, B, Tq, Tk, heads, dh, dh ** -0.5, None) == 0
    sync()
    qf = q.float().view(B, Tq, heads, dh).transpose(1, 2); kf = k.float().view(B, Tk, heads, dh).transpose(1, 2); vf = v.float().view(B, Tk, heads, dh).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * dh ** -0.5, -1) @ vf).transpose(1, 2).reshape(B * Tq, d)
    err = (o.float() - ref).abs()
    rows = err.max(1).values
    bad = (rows > 0.05).nonzero().flatten()
    print((B, Tq, Tk, heads, dh), 'rel', rel_l2(o, ref), 'bad rows', bad.numel(), 'wave hist', torch.bincount((bad % 128) // 16, minlength=8).tolist(), flush=True)
