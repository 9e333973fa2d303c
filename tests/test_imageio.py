"""Host-side data contract of the test harness (no GPU): grid layout, PNG writer, pair-folder dataset, checkpoint files."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from makeupdiffuse_amd import imageio as mio  # noqa: E402
from makeupdiffuse_amd.config import create_model, load_state_dict  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_make_grid_layout_matches_documented_torchvision_behaviour():
    # 5 images of 3x4x6, 3 per row, padding 2 -> 2 rows; cell = (4+2, 6+2); grid = 3 x (2*6+2) x (3*8+2)
    imgs = torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).reshape(5, 3, 4, 6) + 1.0
    g = mio.make_grid(imgs, nrow=3)
    assert tuple(g.shape) == (3, 14, 26)
    for k in range(5):
        y, x = divmod(k, 3)
        assert torch.equal(g[:, y * 6 + 2: y * 6 + 6, x * 8 + 2: x * 8 + 8], imgs[k])
    mask = torch.ones_like(g, dtype=torch.bool)
    for k in range(5):
        y, x = divmod(k, 3)
        mask[:, y * 6 + 2: y * 6 + 6, x * 8 + 2: x * 8 + 8] = False
    assert float(g[mask].abs().max()) == 0.0                      # padding and the empty 6th cell are pad_value 0
    assert torch.equal(mio.make_grid(imgs[:1], nrow=3), imgs[0])  # a single image is returned unpadded
    assert tuple(mio.make_grid(torch.zeros(2, 1, 4, 4), nrow=2).shape) == (3, 8, 14)   # gray -> 3 channels


def test_png_roundtrip_and_rescale(tmp_path):
    from PIL import Image
    imgs = torch.rand(2, 3, 8, 8) * 2 - 1
    p = mio.save_grid_png(imgs, str(tmp_path / 'a' / 'g.png'), nrow=4)
    arr = np.asarray(Image.open(p))
    assert arr.shape == (8 + 4, 2 * 10 + 2, 3) and arr.dtype == np.uint8
    exp = (((imgs[1] + 1) / 2).permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    assert np.array_equal(arr[2:10, 12:20], exp)
    assert arr[0, 0, 0] == 127                                     # pad value 0 in (-1,1) -> (0+1)/2*255 truncated


def test_pair_folder_dataset_fields(tmp_path):
    from PIL import Image
    os.makedirs(tmp_path / 'images' / 'non-makeup'); os.makedirs(tmp_path / 'images' / 'makeup')
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8); b = rng.integers(0, 256, (20, 12, 3), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / 'images' / 'non-makeup' / 'x1.png'); Image.fromarray(b).save(tmp_path / 'images' / 'makeup' / 'y2.png')
    (tmp_path / 'test_0412.txt').write_text('non-makeup/x1.png makeup/y2.png\n\n')
    ds = mio.PairFolderDataset(str(tmp_path), dim=(16, 16))
    assert len(ds) == 1
    it = ds[0]
    assert it['img_name'] == 'x1&y2' and it['txt'] == 'makeup transfer'
    assert tuple(it['src_img'].shape) == (3, 16, 16) and tuple(it['ref_img'].shape) == (3, 16, 16)
    assert torch.equal(it['src_img'], torch.from_numpy(a.astype(np.float32) / 255).permute(2, 0, 1))   # no resize needed: exact
    assert 0.0 <= float(it['ref_img'].min()) and float(it['ref_img'].max()) <= 1.0
    assert torch.allclose(it['nonmakeup_img'], it['src_img'] * 2 - 1)
    batch = mio.collate([it, it])
    assert tuple(batch['src_img'].shape) == (2, 3, 16, 16) and batch['img_name'] == ['x1&y2', 'x1&y2']


def test_checkpoint_files_are_read_without_unpickling_code(tmp_path):
    """.safetensors and tensor-only .ckpt (weights_only=True) both come back as the same flat dict; a model created
    from the yaml keeps the two nets' + decoder's keys pending until .cuda() and reports the rest as unexpected."""
    from safetensors.torch import save_file
    sd = {'model.diffusion_model.out.2.bias': torch.randn(4), 'control_model.input_hint_block.0.bias': torch.randn(16),
          'first_stage_model.decoder.conv_out.bias': torch.randn(3), 'cond_stage_model.transformer.x': torch.randn(2),
          'first_stage_model.encoder.conv_in.bias': torch.randn(128)}
    save_file(sd, str(tmp_path / 'm.safetensors'))
    torch.save({'state_dict': sd, 'epoch': 3}, tmp_path / 'm.ckpt')
    a = load_state_dict(str(tmp_path / 'm.safetensors')); b = load_state_dict(str(tmp_path / 'm.ckpt'))
    assert set(a) == set(b) == set(sd) and all(torch.equal(a[k], b[k]) for k in sd)
    model = create_model(os.path.join(ROOT, 'diffmodels', 'test_diffusion_makeup.yaml'))
    missing, unexpected = model.load_state_dict(a)
    assert missing == [] and sorted(unexpected) == ['cond_stage_model.transformer.x', 'first_stage_model.encoder.conv_in.bias']


def test_test_pairs_bookkeeping(tmp_path):
    model = create_model(os.path.join(ROOT, 'diffmodels', 'test_diffusion_makeup.yaml'))
    model.test_pairs_file = str(tmp_path / 'pairs.txt')
    model.on_test_epoch_start()
    model.test_pairs.append(['0003-1', 'non-makeup/a.png', 'makeup/b.png'])
    model.on_test_batch_end(None, None, 3)
    assert (tmp_path / 'pairs.txt').read_text() == '0003-1 non-makeup/a.png makeup/b.png\n'
    imgs = {'samples': torch.rand(2, 3, 8, 8) * 2 - 1, 'samples_latent': torch.randn(2, 4, 1, 1), 'alpha': torch.zeros(2)}
    model.saved_dir = str(tmp_path)
    out = model.save_local(imgs, 7)
    assert [os.path.basename(p) for p in out] == ['samples_0007.png']
    assert os.path.exists(os.path.join(str(tmp_path), model.model_name, 'samples_0007.png'))
