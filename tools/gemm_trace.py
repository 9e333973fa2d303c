import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd import lib as mlib
lib = mlib.load(); DEV='cuda:0'; P=lambda t: C.c_void_p(t.data_ptr())
M,N,K = 256,1280,1280
A=torch.randn(M,K,device=DEV).bfloat16(); W=torch.randn(N,K,device=DEV).bfloat16(); R=torch.randn(M,N,device=DEV).bfloat16(); b=torch.randn(N,device=DEV)
out=torch.empty(M,N,device=DEV,dtype=torch.bfloat16)
for mode in ('cold', 'hot', 'hotA-coldW'):
    for cfg, s in [(5,1),(12,1)]:
        lib.mkd_gemm_force_tile(cfg)
        for i in range(2):
            if mode == 'cold':
                big=torch.randn(128<<20,device=DEV); torch.cuda.synchronize()
            if mode == 'hotA-coldW':
                big=torch.randn(128<<20,device=DEV); torch.cuda.synchronize(); A2 = A.clone(); A.copy_(A2); torch.cuda.synchronize()
            print(f'{mode}:', file=sys.stderr, end=' ', flush=True)
            lib.mkd_gemm_bf16(P(A),K,P(W),K,P(b),None,0,1,P(R),N,1.0,0,P(out),N,0,M,N,K,0,0,0,0,0,0,0,0,0,s,None)
            torch.cuda.synchronize()
