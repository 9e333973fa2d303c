"""Image-side data contract of the test harness (host only; none of this is on the device hot path).

  * ``make_grid`` / ``save_grid_png``  — what ``save_local`` does with every entry of the log dict
    (reference diffmk/diffusion_makeup.py:344-358: torchvision ``make_grid(images, nrow)``, ``(g+1)/2``, HWC uint8, PNG).
    torchvision is not installed, so the grid layout is restated from its documented behaviour
    (padding 2, pad value 0, ``xmaps = min(nrow, N)``, single-channel images repeated to 3 channels).
  * ``PairFolderDataset`` — the batch-dict fields the sampler path reads from ``TestFixed_Dataset``
    (reference diffdata/datasets.py:728-784): ``src_img`` / ``ref_img`` in [0,1] CHW, ``txt`` = 'makeup transfer',
    ``img_name`` = '<src>&<ref>'.  The reference additionally crops by face landmarks / masks (``PreProcess``), which
    needs data files this repository does not have; here images are resized to ``dim`` directly.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def make_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """[N,C,H,W] (or [C,H,W] / [H,W]) -> [3 or C, gh, gw] laid out row-major, `nrow` images per row."""
    t = images
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() == 3:
        if t.shape[0] == 1:
            t = torch.cat((t, t, t), 0)
        t = t.unsqueeze(0)
    if t.dim() != 4:
        raise ValueError('make_grid expects a 2-, 3- or 4-D tensor')
    if t.shape[1] == 1:
        t = torch.cat((t, t, t), 1)
    n = t.shape[0]
    if n == 1:
        return t[0]
    xmaps = min(int(nrow), n)
    ymaps = int(math.ceil(n / xmaps))
    ch, cw = t.shape[2] + padding, t.shape[3] + padding
    grid = t.new_full((t.shape[1], ch * ymaps + padding, cw * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * ch + padding: y * ch + ch, x * cw + padding: x * cw + cw] = t[k]
            k += 1
    return grid


def grid_to_uint8(grid: torch.Tensor, rescale: bool = True) -> np.ndarray:
    """(-1,1) CHW -> HWC uint8, the arithmetic of save_local (:350-354; truncating cast like ndarray.astype)."""
    g = grid.detach().float().cpu()
    if rescale:
        g = (g + 1.0) / 2.0
    g = g.permute(1, 2, 0).numpy()
    return (g * 255).astype(np.uint8)


def save_grid_png(images: torch.Tensor, path: str, nrow: int, rescale: bool = True) -> str:
    from PIL import Image
    arr = grid_to_uint8(make_grid(images, nrow=nrow), rescale)
    os.makedirs(os.path.split(path)[0] or '.', exist_ok=True)
    Image.fromarray(arr).save(path)
    return path


def read_pairs(path: str) -> List[Tuple[str, str]]:
    """Lines of '<non-makeup file> <makeup file>' (datasets.py:739-742)."""
    out = []
    with open(path, 'r') as f:
        for line in f:
            parts = line.strip().split(' ')
            if len(parts) >= 2 and parts[0]:
                out.append((parts[0], parts[1]))
    return out


class PairFolderDataset:
    """root/images/<name> + a pairs file -> dicts with the keys get_input reads."""

    def __init__(self, root: str, pairs_file: str = 'test_0412.txt', dim: Sequence[int] = (256, 256), prompt: str = 'makeup transfer'):
        self.root = root
        self.pairs = read_pairs(pairs_file if os.path.isabs(pairs_file) else os.path.join(root, pairs_file))
        self.dim = tuple(dim)
        self.prompt = prompt

    def __len__(self) -> int:
        return len(self.pairs)

    def _load(self, name: str) -> torch.Tensor:
        from PIL import Image
        img = Image.open(os.path.join(self.root, 'images', name)).convert('RGB')
        if self.dim and img.size != (self.dim[1], self.dim[0]):
            img = img.resize((self.dim[1], self.dim[0]), Image.BILINEAR)
        a = np.asarray(img, dtype=np.float32) / 255.0
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()

    def __getitem__(self, i: int) -> Dict[str, object]:
        s, r = self.pairs[i]
        src, ref = self._load(s), self._load(r)
        base = lambda n: os.path.basename(n).split('.')[0]
        return {'src_img': src, 'ref_img': ref, 'nonmakeup_img': src * 2 - 1, 'makeup_img': ref * 2 - 1,
                'txt': self.prompt, 'img_name': f'{base(s)}&{base(r)}'}


def collate(items: Sequence[Dict[str, object]]) -> Dict[str, object]:
    out: Dict[str, object] = {}
    for k in items[0]:
        v = [it[k] for it in items]
        out[k] = torch.stack(v) if isinstance(v[0], torch.Tensor) else v
    return out
