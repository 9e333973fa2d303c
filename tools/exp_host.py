import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd.engine import MkdEngine, NetConfig
dev = torch.device('cuda:0')
e = MkdEngine(NetConfig(), dev); e.init_random(0)
g = torch.Generator().manual_seed(0)
b = 8
x = torch.randn(b, 4, 32, 32, generator=g).to(dev); h = torch.rand(b, 6, 256, 256, generator=g).to(dev); c = torch.randn(b, 77, 768, generator=g).to(dev)
t = torch.full((b,), 500, device=dev)
e.prepare(h, c)
out = torch.empty_like(x)
for _ in range(3): e.eps(x, t, out)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): e.eps(x, t, out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue per eval: {(t1-t0)/n*1e3:.2f} ms; total per eval incl. GPU: {(t2-t0)/n*1e3:.2f} ms; launches {e.eps_launches()}')
