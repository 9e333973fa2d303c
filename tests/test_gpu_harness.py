"""runs/test.py end to end on the device: yaml -> model -> (folder dataset | synthetic) -> test_step -> PNG grids."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('with_tokenizer', [False, True])
def test_runs_test_py_writes_png_grids_from_a_pair_folder(tmp_path, with_tokenizer):
    from PIL import Image
    data = tmp_path / 'data'
    os.makedirs(data / 'images' / 'non-makeup'); os.makedirs(data / 'images' / 'makeup')
    rng = np.random.default_rng(1)
    for d, n in (('non-makeup', 's1.png'), ('makeup', 'r1.png'), ('non-makeup', 's2.png'), ('makeup', 'r2.png')):
        Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(data / 'images' / d / n)
    (data / 'test_0412.txt').write_text('non-makeup/s1.png makeup/r1.png\nnon-makeup/s2.png makeup/r2.png\n')
    out = tmp_path / 'out'
    extra = []
    if with_tokenizer:       # prompts 'makeup transfer' / '' -> local CLIPTokenizer files -> mkd_clip_encode (synthetic vocabulary)
        import json
        words = ['<|startoftext|>', '<|endoftext|>'] + [c + sfx for c in 'makeuptrnsf' for sfx in ('', '</w>')]
        tk = tmp_path / 'tok'; os.makedirs(tk)
        (tk / 'vocab.json').write_text(json.dumps({w: i for i, w in enumerate(dict.fromkeys(words))}))
        (tk / 'merges.txt').write_text('#version: 0.2\n')
        extra = ['--tokenizer', str(tk)]
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'runs', 'test.py'), '--data-root', str(data), '--res', '64',
                        '--batch-size', '2', '--ddim-steps', '4', '--out', str(out)] + extra,
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    root = out / 'makeupdiffuse_mi355x'
    names = sorted(os.listdir(root))
    assert names == ['control_ref_0000.png', 'control_src_0000.png', 'samples_0000.png', 'samples_cfg_scale_9.00_0000.png'], names
    g = np.asarray(Image.open(root / 'samples_0000.png'))
    assert g.shape == (64 + 4, 2 * 66 + 2, 3) and g.dtype == np.uint8
    src = np.asarray(Image.open(root / 'control_src_0000.png'))
    exp = np.asarray(Image.open(data / 'images' / 'non-makeup' / 's1.png'))
    assert np.abs(src[2:66, 2:66].astype(int) - exp.astype(int)).max() <= 1     # (x*2-1 -> clamp -> +1)/2*255 truncation
    pairs = (out / 'test_pairs_rank0.txt').read_text().splitlines()
    assert pairs == ['0000-1 non-makeup/s1.png makeup/r1.png', '0000-2 non-makeup/s2.png makeup/r2.png']
    assert g.std() > 1.0                                                       # a decoded image, not a constant


def test_each_pair_of_a_batched_run_equals_its_own_single_pair_run(tmp_path):
    """PNG names and std > 1 cannot see wrong conditioning, a swapped src/ref or state leaking from one batch into the
    next.  Three pairs, fixed per-pair start noise (--seed): the latents of the run with batch size 2 (batches [0, 1] and [2],
    i.e. two consecutive batches through one model) must equal the batch-size-1 run pair by pair, for both sampling passes."""
    import torch
    from PIL import Image
    data = tmp_path / 'data'
    os.makedirs(data / 'images' / 'non-makeup'); os.makedirs(data / 'images' / 'makeup')
    rng = np.random.default_rng(3)
    lines = []
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(data / 'images' / 'non-makeup' / f's{i}.png')
        Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(data / 'images' / 'makeup' / f'r{i}.png')
        lines.append(f'non-makeup/s{i}.png makeup/r{i}.png')
    (data / 'test_0412.txt').write_text('\n'.join(lines) + '\n')
    lat = {}
    for bs in (2, 1):
        out = tmp_path / f'out{bs}'
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'runs', 'test.py'), '--data-root', str(data), '--res', '64', '--batch-size', str(bs),
                            '--ddim-steps', '4', '--seed', '100', '--out', str(out)], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        files = sorted(f for f in os.listdir(out) if f.startswith('latents_'))
        assert files == (['latents_0000.pt', 'latents_0002.pt'] if bs == 2 else ['latents_0000.pt', 'latents_0001.pt', 'latents_0002.pt'])
        lat[bs] = [torch.load(out / f, weights_only=True) for f in files]
    keys = ('samples_latent', 'samples_cfg_scale_9.00_latent')
    two = {k: torch.cat([d[k] for d in lat[2]]) for k in keys}
    one = {k: torch.cat([d[k] for d in lat[1]]) for k in keys}
    for k in keys:
        assert two[k].shape == one[k].shape == (3, 4, 8, 8)
        for i in range(3):
            r_ = float((two[k][i] - one[k][i]).norm() / one[k][i].norm())
            print(f'{k} pair {i}: batched vs single rel-L2 {r_:.3e}')
            assert r_ <= (8e-2 if 'cfg' in k else 1.5e-2), (k, i, r_)      # measured 3.7e-3 plain / 3.5e-2 CFG 9 (guidance amplifies the bf16 batch-shape noise)
        # and the pairs really differ from each other (conditioning is per pair)
        assert float((one[k][0] - one[k][1]).norm() / one[k][1].norm()) > 0.1
    # the second batch of the batched run ([2]) is the same single pair as latents_0002 of the other run: same batch shape -> bit-exact
    for k in keys:
        assert torch.equal(lat[2][1][k], lat[1][2][k]), k


def test_bench_self_launches_two_ranks_on_one_card_in_rehearsal_mode():
    """`python bench.py --gpus 2` with no torchrun and ONE visible device: bench.py starts its own two ranks (fresh processes,
    gloo, both on card 0), runs the sharded path with the final all-gather and rank 0 prints one JSON line that says so."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--ddim-steps', '2',
                        '--batch', '2', '--no-cpu-baseline'], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['backend'] == 'gloo' and out['rehearsal_single_device'] is True
    assert out['config']['global_batch'] == 4 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['roofline']['frac'] > 0 and len(out['devices']) == 1
    assert out['roofline_hbm']['bound'] == 'hbm' and 0 < out['roofline_hbm']['frac'] < 1 and out['roofline_hbm']['bytes_per_launch'] > 0


def test_rccl_group_of_one_on_the_card():
    """SURVEY.md §8e on the hardware that IS here: a fresh process with WORLD_SIZE=1 and MKD_DIST_BACKEND=nccl initialises the RCCL
    process group (dist.init_from_env), runs gather_shards / max_over_ranks / barrier on DEVICE tensors - the real collectives, not
    the single-process shortcut - and a 2-step `bench.py --gpus 1` under the same environment reports backend nccl, one rank.
    (N > 1 over xGMI stays unmeasured: no multi-GPU node is reachable from here.)"""
    import json
    from makeupdiffuse_amd import dist as mdist
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(mdist.free_port()), MKD_DIST_BACKEND='nccl',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    code = (
        "import torch, torch.distributed as d\n"
        "from makeupdiffuse_amd import dist as m\n"
        "r, w, l = m.init_from_env()\n"
        "assert (r, w, l) == (0, 1, 0) and d.is_initialized() and d.get_backend() == 'nccl' and d.get_world_size() == 1\n"
        "x = torch.arange(24, dtype=torch.float32, device='cuda:0').reshape(3, 2, 4)\n"
        "g = m.gather_shards(x, 3)\n"
        "assert g.is_cuda and g.data_ptr() != x.data_ptr() and torch.equal(g, x)\n"
        "assert m.max_over_ranks(1.25, torch.device('cuda:0')) == 1.25\n"
        "m.barrier(); torch.cuda.synchronize(); d.destroy_process_group()\n"
        "print('RCCL_OK', torch.cuda.get_device_name(0))\n")
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    env['MASTER_PORT'] = str(mdist.free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '0', '--ddim-steps', '2',
                        '--batch', '2', '--no-cpu-baseline', '--live-pmc', '0'], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['backend'] == 'nccl' and out['ranks_seen'] == 1 and out['n_gpus'] == 1 and out['rehearsal_single_device'] is False
    assert out['value'] > 0 and out['roofline']['traffic_source']['collected_live'] is False


def test_bench_measures_its_hbm_counters_live():
    """`roofline.traffic` / `roofline_hbm.traffic` of the bench line are measured BY the bench run when rocprofv3 is there: two --pmc child
    passes (FETCH_SIZE, WRITE_SIZE) of a 2-step eager run of the same workload, after the timed region.  The GroupNorm class moves at
    least its algorithmic bytes and not more than a few times that."""
    import json
    import shutil
    if not shutil.which('rocprofv3'):
        pytest.skip('rocprofv3 not installed')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--ddim-steps', '2', '--batch', '2',
                        '--no-cpu-baseline', '--decode', '0', '--live-pmc', '1'], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    for key in ('roofline', 'roofline_hbm'):
        src = out[key]['traffic_source']
        assert src['collected_live'] is True and src['dispatches'] > 0, (key, src, r.stderr[-3000:])
        assert out[key]['traffic'] > 0
    h = out['roofline_hbm']
    assert 0.2 * h["bytes_per_launch"] < h["traffic"] < 8 * h["bytes_per_launch"], h
