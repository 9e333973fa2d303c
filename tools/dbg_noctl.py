import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makeupdiffuse_amd.engine import MkdEngine, NetConfig
from oracle import nets
SMALL = dict(model_channels=64, channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2, context_dim=64, hint_widths=(16, 16, 32, 32, 32, 32, 64))
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'small_eps.npz'))
ocfg = nets.NetConfig(**SMALL)
sd = nets.init_state_dict(ocfg, seed=int(g['seed_weights']))
eng = MkdEngine(NetConfig(**SMALL)); eng.load_state_dict(sd)
G = {k: torch.from_numpy(g[k]) for k in g.files if k != 'seed_weights'}
eng.prepare(None, G['ctx'], latent_hw=(8, 8))
ref = G['eps_noctl']
for i in range(4):
    out = eng.eps(G['x'], G['t']).cpu()
    err = (out - ref).abs()
    print(i, 'rel', float((out - ref).norm() / ref.norm()), 'per-sample max err', err.flatten(1).max(1).values.tolist(), flush=True)
eng.prepare(G['hint'], G['ctx'])
out = eng.eps(G['x'], G['t']).cpu(); print('ctl rel', float((out - G['eps']).norm() / G['eps'].norm()))
eng.prepare(None, G['ctx'], latent_hw=(8, 8))
out = eng.eps(G['x'], G['t']).cpu(); print('noctl again rel', float((out - ref).norm() / ref.norm()))
eng.eps_profile(G['x'], G['t'], csv_path='gpurun_out/noctl_plan.csv')
