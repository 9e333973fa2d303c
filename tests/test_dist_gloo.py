"""CPU, world_size 2, gloo: the N>1 path's host logic — shard the batch, run per-rank, ONE all-gather, global order
restored, result independent of the world size.  The per-rank "sampler" is the CPU oracle here."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from makeupdiffuse_amd import dist as mdist
    from oracle import nets, sampler
    r, w, _ = mdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    cfg = nets.NetConfig(model_channels=64, channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2, context_dim=64,
                         hint_widths=(16, 16, 32, 32, 32, 32, 64))
    sd = nets.init_state_dict(cfg, seed=11)
    lo, hi = mdist.shard_range(n_total, rank, world)
    xs, hs, cs = [], [], []
    for i in range(lo, hi):                      # per-sample seeds -> shard-invariant inputs
        g = torch.Generator().manual_seed(1234 + i)
        xs.append(torch.randn(1, 4, 8, 8, generator=g)); hs.append(torch.rand(1, 6, 64, 64, generator=g))
        cs.append(torch.randn(1, 77, 64, generator=g))
    x, h, c = torch.cat(xs), torch.cat(hs), torch.cat(cs)
    lat = sampler.sample(sampler.make_eps_fn(sd, cfg), sampler.Schedule(), x, {'c_crossattn': [c], 'c_concat': [h]}, 2)
    full = mdist.gather_shards(lat, n_total)
    t = mdist.max_over_ranks(float(rank + 1))
    assert t == float(world)
    mdist.barrier()
    np.save(os.path.join(out_dir, f'full_w{world}_r{rank}.npy'), full.numpy())
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gather_equals_single_process(tmp_path):
    n_total = 3                                   # ragged: rank 0 gets 2 samples, rank 1 gets 1
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    _worker(0, 1, _free_port(), n_total, str(tmp_path))
    one = np.load(tmp_path / 'full_w1_r0.npy')
    a = np.load(tmp_path / 'full_w2_r0.npy')
    b = np.load(tmp_path / 'full_w2_r1.npy')
    assert one.shape == (n_total, 4, 8, 8)
    assert np.array_equal(a, b)                   # every rank holds the same gathered batch
    # equals the unsharded run, in global order; fp32 torch-CPU kernels round differently per batch size (measured
    # 3.7e-6 abs on values up to 6.6), hence the tolerance
    assert np.allclose(a, one, rtol=1e-4, atol=1e-4)


def test_spawn_ranks_self_launch_rendezvous_and_single_json_line(tmp_path):
    """The launcher bench.py uses for a plain `python bench.py --gpus N` (no torchrun): N fresh children, env-based rendezvous on
    127.0.0.1, rank 0's stdout is the only stdout, the exit code is the worst child's."""
    import subprocess
    import sys
    child = tmp_path / 'child.py'
    child.write_text(
        'import json, os, sys\n'
        f'sys.path.insert(0, {ROOT!r})\n'
        'import torch\n'
        'from makeupdiffuse_amd import dist as mdist\n'
        "rank, world, local = mdist.init_from_env(backend='gloo')\n"
        'lo, hi = mdist.shard_range(5, rank, world)\n'
        'full = mdist.gather_shards(torch.arange(lo, hi, dtype=torch.float32)[:, None], 5)\n'
        't = mdist.max_over_ranks(float(rank))\n'
        'mdist.barrier()\n'
        "print(json.dumps({'rank': rank, 'ranks_seen': torch.distributed.get_world_size(), 'backend': torch.distributed.get_backend(),\n"
        "                  'full': full[:, 0].tolist(), 'max': t, 'extra': os.environ.get('MKD_TEST_EXTRA')}), flush=True)\n"
        "sys.exit(3 if (rank == 1 and os.environ.get('MKD_TEST_FAIL')) else 0)\n")
    launcher = tmp_path / 'launch.py'
    launcher.write_text(
        'import sys\n'
        f'sys.path.insert(0, {ROOT!r})\n'
        'from makeupdiffuse_amd import dist as mdist\n'
        f"sys.exit(mdist.spawn_ranks([sys.executable, {str(child)!r}], 2, {{'MKD_TEST_EXTRA': 'x'}}, timeout=300))\n")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, str(launcher)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]        # (gloo itself prints a connection banner on stdout)
    assert len(lines) == 1, r.stdout                       # rank 0 only
    import json
    out = json.loads(lines[0])
    assert out == {'rank': 0, 'ranks_seen': 2, 'backend': 'gloo', 'full': [0.0, 1.0, 2.0, 3.0, 4.0], 'max': 1.0, 'extra': 'x'}
    r = subprocess.run([sys.executable, str(launcher)], capture_output=True, text=True, timeout=600, env=dict(env, MKD_TEST_FAIL='1'))
    assert r.returncode == 3


def test_bench_self_launch_is_decided_before_any_gpu_call():
    """bench.py must branch into the launcher before importing the engine touches a device (a process that has initialised HIP
    must not be the parent of the ranks' rendezvous, and must never exec)."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main = src[src.index('def main():'):]
    assert main.index('spawn_ranks') < main.index('init_from_env') < main.index('torch.cuda.is_available()')
    assert 'os.exec' not in src and 'execv' not in src


def test_group_of_one_runs_the_real_collectives(tmp_path):
    """WORLD_SIZE=1 with an explicit backend (MKD_DIST_BACKEND): init_from_env still builds the process group and gather_shards /
    max_over_ranks / barrier go through torch.distributed - the path the -m gpu suite repeats with backend nccl (RCCL) on the card."""
    import subprocess
    import sys
    from makeupdiffuse_amd import dist as mdist
    child = tmp_path / 'one.py'
    child.write_text(
        'import json, sys\n'
        f'sys.path.insert(0, {ROOT!r})\n'
        'import torch\n'
        'from makeupdiffuse_amd import dist as mdist\n'
        'rank, world, local = mdist.init_from_env()\n'
        'x = torch.arange(6, dtype=torch.float32).reshape(3, 2)\n'
        'g = mdist.gather_shards(x, 3)\n'
        'mdist.barrier()\n'
        "print(json.dumps({'init': torch.distributed.is_initialized(), 'backend': torch.distributed.get_backend(), 'world': world,\n"
        "                  'same_storage': g.data_ptr() == x.data_ptr(), 'equal': bool(torch.equal(g, x)), 'max': mdist.max_over_ranks(2.5)}), flush=True)\n")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(mdist.free_port()), MKD_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, str(child)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out == {'init': True, 'backend': 'gloo', 'world': 1, 'same_storage': False, 'equal': True, 'max': 2.5}
    # without an explicit backend a single process stays group-free (no rendezvous, the shard IS the batch)
    env.pop('MKD_DIST_BACKEND')
    child2 = tmp_path / 'none.py'
    child2.write_text(
        'import sys\n'
        f'sys.path.insert(0, {ROOT!r})\n'
        'import torch\n'
        'from makeupdiffuse_amd import dist as mdist\n'
        'mdist.init_from_env()\n'
        'x = torch.zeros(2, 1)\n'
        'assert not torch.distributed.is_initialized() and mdist.gather_shards(x, 2) is x\n')
    r = subprocess.run([sys.executable, str(child2)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
