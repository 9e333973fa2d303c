import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load(); DEV='cuda:0'
P = lambda t: C.c_void_p(t.data_ptr())
def run(M, N, K, cfg, iters=10):
    A = (torch.rand(M, K, device=DEV)*2-1).to(torch.bfloat16); W = (torch.rand(N, K, device=DEV)*2-1).to(torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    lib.mkd_gemm_force_tile(cfg)
    f = lambda: lib.mkd_gemm_bf16(P(A), K, P(W), K, None, None, 0, 1, None, 0, 1.0, 0, P(out), N, 0, M, N, K, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, None)
    for _ in range(3): assert f() == 0, lib.mkd_last_error()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/iters
    print(f'M={M} N={N} K={K} cfg={cfg}: {us:9.1f} us  {2.0*M*N*K/us*1e-6:7.1f} TF/s', flush=True)
    lib.mkd_gemm_force_tile(-1)
for (M,N,K) in [(4096,4096,4096),(8192,8192,8192),(8192,2560,320),(8192,320,2880)]:
    for cfg in (0,1,2,3,5):
        run(M,N,K,cfg)
