import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load(); DEV='cuda:0'
P = lambda t: C.c_void_p(None if t is None else t.data_ptr())
def timeit(name, fn, n=300):
    for _ in range(20): assert fn() == 0, lib.mkd_last_error()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:60s} {e0.elapsed_time(e1)*1e3/n:7.2f} us', flush=True)
SC, OF = float(sys.argv[1]), float(sys.argv[2])
for (M,N,K,act) in [(8192,2560,320,2),(8192,960,320,0),(2048,5120,640,2),(512,3840,1280,0),(8192,320,320,0)]:
    A = (torch.randn(M,K,device=DEV)*SC+OF).to(torch.bfloat16); W = (torch.randn(N,K,device=DEV)*0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV); s = torch.randn(N, device=DEV)
    slots = max(1, (K+127)//128)
    stats = torch.rand(slots, M, 2, device=DEV) + 1.0
    Nout = N//2 if act==2 else N
    out = torch.empty(M, Nout, device=DEV, dtype=torch.bfloat16)
    timeit(f'plain  M={M} N={N} K={K} act={act}', lambda: lib.mkd_gemm_bf16(P(A), K, P(W), K, P(bias), None, 0, 1, None, 0, 1.0, act, P(out), Nout, 0, M, N, K, 0,0,0,0,0,0,0,1,0, 1, None))
    timeit(f'LN     M={M} N={N} K={K} act={act} slots={slots}', lambda: lib.mkd_gemm_ln_bf16(P(A), K, P(W), K, P(bias), P(s), P(stats), slots, 1e-5, act, P(out), Nout, M, N, K, None))
