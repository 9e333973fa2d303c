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
