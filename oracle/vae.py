"""fp32 CPU restatement of the first-stage DECODER (test infrastructure only).

The reference decodes latents with ``self.decode_first_stage(z)`` (``diffmk/diffusion_makeup.py:396,409``,
``diffmk/makeups.py:260-262``): ``z / scale_factor`` -> UPSTREAM ``ldm.models.autoencoder.AutoencoderKL.decode``
(post_quant_conv 1x1 -> ``Decoder``), configured by ``diffmodels/base_diffusion_makeup.yaml:86-107``
(embed_dim 4, z_channels 4, ch 128, ch_mult 1,2,4,4, 2 res blocks, no attention resolutions, out_ch 3).
The autoencoder class itself is not in the reference (un-vendored ``ldm``); this restates the published SD-1.x
decoder with upstream state-dict names (``first_stage_model.decoder.*``).  PARITY UNPINNED (see oracle/__init__).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
PREFIX = 'first_stage_model.'


@dataclass
class VaeConfig:
    z_channels: int = 4
    embed_dim: int = 4
    ch: int = 128
    ch_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    out_ch: int = 3


FULL = VaeConfig()


def _res(p, cin, cout):
    d = {f'{p}.norm1.weight': (cin,), f'{p}.norm1.bias': (cin,), f'{p}.conv1.weight': (cout, cin, 3, 3), f'{p}.conv1.bias': (cout,),
         f'{p}.norm2.weight': (cout,), f'{p}.norm2.bias': (cout,), f'{p}.conv2.weight': (cout, cout, 3, 3), f'{p}.conv2.bias': (cout,)}
    if cin != cout:
        d[f'{p}.nin_shortcut.weight'] = (cout, cin, 1, 1)
        d[f'{p}.nin_shortcut.bias'] = (cout,)
    return d


def param_spec(cfg: VaeConfig, prefix: str = PREFIX) -> Dict[str, tuple]:
    d = {f'{prefix}post_quant_conv.weight': (cfg.z_channels, cfg.embed_dim, 1, 1), f'{prefix}post_quant_conv.bias': (cfg.z_channels,)}
    D = f'{prefix}decoder.'
    bi = cfg.ch * cfg.ch_mult[-1]
    d[f'{D}conv_in.weight'] = (bi, cfg.z_channels, 3, 3)
    d[f'{D}conv_in.bias'] = (bi,)
    d.update(_res(f'{D}mid.block_1', bi, bi))
    for n in ('q', 'k', 'v', 'proj_out'):
        d[f'{D}mid.attn_1.{n}.weight'] = (bi, bi, 1, 1)
        d[f'{D}mid.attn_1.{n}.bias'] = (bi,)
    d[f'{D}mid.attn_1.norm.weight'] = (bi,)
    d[f'{D}mid.attn_1.norm.bias'] = (bi,)
    d.update(_res(f'{D}mid.block_2', bi, bi))
    for lvl in reversed(range(len(cfg.ch_mult))):
        bo = cfg.ch * cfg.ch_mult[lvl]
        for j in range(cfg.num_res_blocks + 1):
            d.update(_res(f'{D}up.{lvl}.block.{j}', bi, bo))
            bi = bo
        if lvl != 0:
            d[f'{D}up.{lvl}.upsample.conv.weight'] = (bi, bi, 3, 3)
            d[f'{D}up.{lvl}.upsample.conv.bias'] = (bi,)
    d[f'{D}norm_out.weight'] = (bi,)
    d[f'{D}norm_out.bias'] = (bi,)
    d[f'{D}conv_out.weight'] = (cfg.out_ch, bi, 3, 3)
    d[f'{D}conv_out.bias'] = (cfg.out_ch,)
    return d


def init_state_dict(cfg: VaeConfig, seed: int = 0, norm_jitter: float = 0.2) -> Dict[str, Tensor]:
    """Seeded synthetic decoder weights; GroupNorm gamma = 1 + norm_jitter * N(0,1), beta = norm_jitter * N(0,1)
    (random, so that a fixture sees a swapped or dropped norm parameter; 0 gives gamma 1 / beta 0)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in sorted(param_spec(cfg).items()):
        if len(shape) == 1:
            if 'norm' in name:
                t = torch.ones(shape) if name.endswith('weight') else torch.zeros(shape)
                if norm_jitter:
                    gn = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
                    t = t + norm_jitter * torch.randn(shape, generator=gn)
                sd[name] = t
            else:
                sd[name] = 0.02 * torch.randn(shape, generator=g)
        else:
            fan = 1
            for s in shape[1:]:
                fan *= s
            sd[name] = torch.randn(shape, generator=g) / fan ** 0.5
    return sd


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[f'{p}.weight'], sd[f'{p}.bias'], eps=1e-6)


def _resblock(sd, p, x):
    h = F.conv2d(F.silu(_gn(sd, f'{p}.norm1', x)), sd[f'{p}.conv1.weight'], sd[f'{p}.conv1.bias'], padding=1)
    h = F.conv2d(F.silu(_gn(sd, f'{p}.norm2', h)), sd[f'{p}.conv2.weight'], sd[f'{p}.conv2.bias'], padding=1)
    if f'{p}.nin_shortcut.weight' in sd:
        x = F.conv2d(x, sd[f'{p}.nin_shortcut.weight'], sd[f'{p}.nin_shortcut.bias'])
    return x + h


def _attn(sd, p, x):
    b, c, h, w = x.shape
    g = _gn(sd, f'{p}.norm', x)
    q = F.conv2d(g, sd[f'{p}.q.weight'], sd[f'{p}.q.bias']).reshape(b, c, h * w).permute(0, 2, 1)
    k = F.conv2d(g, sd[f'{p}.k.weight'], sd[f'{p}.k.bias']).reshape(b, c, h * w)
    v = F.conv2d(g, sd[f'{p}.v.weight'], sd[f'{p}.v.bias']).reshape(b, c, h * w)
    wgt = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)          # [b, hw(q), hw(k)]
    o = torch.bmm(v, wgt.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + F.conv2d(o, sd[f'{p}.proj_out.weight'], sd[f'{p}.proj_out.bias'])


def decode(sd: Dict[str, Tensor], cfg: VaeConfig, z: Tensor, prefix: str = PREFIX) -> Tensor:
    """AutoencoderKL.decode: post_quant_conv -> Decoder.  (decode_first_stage divides z by scale_factor first.)"""
    D = f'{prefix}decoder.'
    h = F.conv2d(z, sd[f'{prefix}post_quant_conv.weight'], sd[f'{prefix}post_quant_conv.bias'])
    h = F.conv2d(h, sd[f'{D}conv_in.weight'], sd[f'{D}conv_in.bias'], padding=1)
    h = _resblock(sd, f'{D}mid.block_1', h)
    h = _attn(sd, f'{D}mid.attn_1', h)
    h = _resblock(sd, f'{D}mid.block_2', h)
    for lvl in reversed(range(len(cfg.ch_mult))):
        for j in range(cfg.num_res_blocks + 1):
            h = _resblock(sd, f'{D}up.{lvl}.block.{j}', h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')
            h = F.conv2d(h, sd[f'{D}up.{lvl}.upsample.conv.weight'], sd[f'{D}up.{lvl}.upsample.conv.bias'], padding=1)
    h = F.silu(_gn(sd, f'{D}norm_out', h))
    return F.conv2d(h, sd[f'{D}conv_out.weight'], sd[f'{D}conv_out.bias'], padding=1)


def decode_first_stage(sd, cfg: VaeConfig, z: Tensor, scale_factor: float = 0.18215) -> Tensor:
    return decode(sd, cfg, z / scale_factor)
