import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makeupdiffuse_amd.engine import MkdEngine, NetConfig, VaeConfig
from makeupdiffuse_amd.schedule import DDIMSchedule
eng = MkdEngine(NetConfig())
if os.environ.get("WITH_VAE", "1") == "1": eng.configure_vae(VaeConfig())
eng.init_random(0)
g = torch.Generator().manual_seed(0)
hint = torch.rand(8, 6, 256, 256, generator=g).cuda(); ctx = torch.randn(8, 77, 768, generator=g).cuda(); x = torch.randn(8, 4, 32, 32, generator=g).cuda()
sch = DDIMSchedule().make_ddim(50)
def T(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for i in range(3):
    _, tp = T(lambda: eng.prepare(hint, ctx))
    lat, ts = T(lambda: eng.sample(x, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=bool(int(os.environ.get("GRAPH", "0")))))
    img, td = T(lambda: eng.decode(lat)) if eng.vae_cfg is not None else (None, 0.0)
    print(f'step {i}: prepare {tp:.1f} ms, sample {ts:.1f} ms, decode {td:.1f} ms', flush=True)
