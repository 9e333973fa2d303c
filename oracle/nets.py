"""fp32 CPU restatement of the two networks on the hot path (test infrastructure only).

The reference does not ship these networks: it instantiates them from the
un-vendored ``cldm.cldm.ControlNet`` / ``cldm.cldm.ControlledUnetModel``
(reference ``diffmodels/base_diffusion_makeup.yaml:52-84``) and calls them at
``diffmk/makeup_diffuse.py:161-168``.  This file restates the published
architecture as plain functions over a flat ``state_dict``
that uses the upstream parameter names, so a real
checkpoint could be dropped in.  PARITY UNPINNED (see ``oracle/__init__``).

Everything is NCHW fp32 ``torch`` on the CPU, no nn.Module, no autograd.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass
class NetConfig:
    """Hyper-parameters read from yaml ``control_stage_config`` / ``unet_config`` (:52-84)."""
    in_channels: int = 4
    out_channels: int = 4
    hint_channels: int = 6
    model_channels: int = 320
    attention_resolutions: Sequence[int] = (4, 2, 1)
    num_res_blocks: int = 2
    channel_mult: Sequence[int] = (1, 2, 4, 4)
    num_heads: int = 8
    transformer_depth: int = 1
    context_dim: int = 768
    # hint block widths (cldm: 16,16,32,32,96,96,256 -> model_channels); kept
    # configurable so reduced-size fixtures stay small.
    hint_widths: Sequence[int] = (16, 16, 32, 32, 96, 96, 256)

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.model_channels


FULL = NetConfig()


# ----------------------------------------------------------------------------
# structure walk shared by the param-spec, the oracle forward and the tests
# ----------------------------------------------------------------------------
@dataclass
class BlockSpec:
    kind: str                 # 'conv_in' | 'res' | 'down' | 'up'
    cin: int = 0
    cout: int = 0
    attn: bool = False        # SpatialTransformer after the ResBlock
    up: bool = False          # Upsample at the end of this output block
    ds: int = 1               # downsample factor of the block's *output*


def encoder_spec(cfg: NetConfig) -> List[BlockSpec]:
    """input_blocks[0..] of UNet and ControlNet (App. A.1)."""
    mc = cfg.model_channels
    blocks = [BlockSpec('conv_in', cfg.in_channels, mc, ds=1)]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            blocks.append(BlockSpec('res', ch, mult * mc, attn=ds in cfg.attention_resolutions, ds=ds))
            ch = mult * mc
        if level != len(cfg.channel_mult) - 1:
            ds *= 2
            blocks.append(BlockSpec('down', ch, ch, ds=ds))
    return blocks


def decoder_spec(cfg: NetConfig) -> List[BlockSpec]:
    """output_blocks[0..] of the UNet (App. A.1)."""
    mc = cfg.model_channels
    enc = encoder_spec(cfg)
    chans = [b.cout for b in enc]
    ch = enc[-1].cout
    ds = enc[-1].ds
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            up = bool(level and i == cfg.num_res_blocks)
            out.append(BlockSpec('res', ch + ich, mc * mult, attn=ds in cfg.attention_resolutions, up=up, ds=ds))
            ch = mc * mult
            if up:
                ds //= 2
    return out


def _res_params(p: str, cin: int, cout: int, temb: int) -> Dict[str, tuple]:
    d = {
        f'{p}.in_layers.0.weight': (cin,), f'{p}.in_layers.0.bias': (cin,),
        f'{p}.in_layers.2.weight': (cout, cin, 3, 3), f'{p}.in_layers.2.bias': (cout,),
        f'{p}.emb_layers.1.weight': (cout, temb), f'{p}.emb_layers.1.bias': (cout,),
        f'{p}.out_layers.0.weight': (cout,), f'{p}.out_layers.0.bias': (cout,),
        f'{p}.out_layers.3.weight': (cout, cout, 3, 3), f'{p}.out_layers.3.bias': (cout,),
    }
    if cin != cout:
        d[f'{p}.skip_connection.weight'] = (cout, cin, 1, 1)
        d[f'{p}.skip_connection.bias'] = (cout,)
    return d


def _st_params(p: str, ch: int, ctx: int, depth: int) -> Dict[str, tuple]:
    d = {
        f'{p}.norm.weight': (ch,), f'{p}.norm.bias': (ch,),
        f'{p}.proj_in.weight': (ch, ch, 1, 1), f'{p}.proj_in.bias': (ch,),
        f'{p}.proj_out.weight': (ch, ch, 1, 1), f'{p}.proj_out.bias': (ch,),
    }
    for k in range(depth):
        t = f'{p}.transformer_blocks.{k}'
        for a, kd in (('attn1', ch), ('attn2', ctx)):
            d[f'{t}.{a}.to_q.weight'] = (ch, ch)
            d[f'{t}.{a}.to_k.weight'] = (ch, kd)
            d[f'{t}.{a}.to_v.weight'] = (ch, kd)
            d[f'{t}.{a}.to_out.0.weight'] = (ch, ch)
            d[f'{t}.{a}.to_out.0.bias'] = (ch,)
        for n in ('norm1', 'norm2', 'norm3'):
            d[f'{t}.{n}.weight'] = (ch,)
            d[f'{t}.{n}.bias'] = (ch,)
        d[f'{t}.ff.net.0.proj.weight'] = (8 * ch, ch)
        d[f'{t}.ff.net.0.proj.bias'] = (8 * ch,)
        d[f'{t}.ff.net.2.weight'] = (ch, 4 * ch)
        d[f'{t}.ff.net.2.bias'] = (ch,)
    return d


def param_spec(cfg: NetConfig, which: str, prefix: str = '') -> Dict[str, tuple]:
    """name -> shape for ``which`` in {'unet','control'} with upstream names (App. A.5)."""
    assert which in ('unet', 'control')
    mc, temb = cfg.model_channels, cfg.time_embed_dim
    d: Dict[str, tuple] = {}
    d[f'{prefix}time_embed.0.weight'] = (temb, mc)
    d[f'{prefix}time_embed.0.bias'] = (temb,)
    d[f'{prefix}time_embed.2.weight'] = (temb, temb)
    d[f'{prefix}time_embed.2.bias'] = (temb,)
    enc = encoder_spec(cfg)
    for i, b in enumerate(enc):
        p = f'{prefix}input_blocks.{i}'
        if b.kind == 'conv_in':
            d[f'{p}.0.weight'] = (b.cout, b.cin, 3, 3)
            d[f'{p}.0.bias'] = (b.cout,)
        elif b.kind == 'res':
            d.update(_res_params(f'{p}.0', b.cin, b.cout, temb))
            if b.attn:
                d.update(_st_params(f'{p}.1', b.cout, cfg.context_dim, cfg.transformer_depth))
        else:
            d[f'{p}.0.op.weight'] = (b.cout, b.cin, 3, 3)
            d[f'{p}.0.op.bias'] = (b.cout,)
    ch = enc[-1].cout
    mid_attn = True
    d.update(_res_params(f'{prefix}middle_block.0', ch, ch, temb))
    if mid_attn:
        d.update(_st_params(f'{prefix}middle_block.1', ch, cfg.context_dim, cfg.transformer_depth))
    d.update(_res_params(f'{prefix}middle_block.2', ch, ch, temb))
    if which == 'unet':
        for i, b in enumerate(decoder_spec(cfg)):
            p = f'{prefix}output_blocks.{i}'
            d.update(_res_params(f'{p}.0', b.cin, b.cout, temb))
            k = 1
            if b.attn:
                d.update(_st_params(f'{p}.1', b.cout, cfg.context_dim, cfg.transformer_depth))
                k = 2
            if b.up:
                d[f'{p}.{k}.conv.weight'] = (b.cout, b.cout, 3, 3)
                d[f'{p}.{k}.conv.bias'] = (b.cout,)
        d[f'{prefix}out.0.weight'] = (mc,)
        d[f'{prefix}out.0.bias'] = (mc,)
        d[f'{prefix}out.2.weight'] = (cfg.out_channels, mc, 3, 3)
        d[f'{prefix}out.2.bias'] = (cfg.out_channels,)
    else:
        widths = [cfg.hint_channels] + list(cfg.hint_widths) + [mc]
        for j in range(8):
            d[f'{prefix}input_hint_block.{2 * j}.weight'] = (widths[j + 1], widths[j], 3, 3)
            d[f'{prefix}input_hint_block.{2 * j}.bias'] = (widths[j + 1],)
        for i, b in enumerate(enc):
            d[f'{prefix}zero_convs.{i}.0.weight'] = (b.cout, b.cout, 1, 1)
            d[f'{prefix}zero_convs.{i}.0.bias'] = (b.cout,)
        d[f'{prefix}middle_block_out.0.weight'] = (ch, ch, 1, 1)
        d[f'{prefix}middle_block_out.0.bias'] = (ch,)
    return d


UNET_PREFIX = 'model.diffusion_model.'
CONTROL_PREFIX = 'control_model.'


def full_param_spec(cfg: NetConfig) -> Dict[str, tuple]:
    d = param_spec(cfg, 'unet', UNET_PREFIX)
    d.update(param_spec(cfg, 'control', CONTROL_PREFIX))
    return d


def init_state_dict(cfg: NetConfig, seed: int = 0, dtype=torch.float32, device='cpu',
                    gain: float = 1.0, norm_jitter: float = 0.2) -> SD:
    """Seeded synthetic weights, SURVEY.md §8(d): N(0, 1/fan_in) for every matrix/conv
    INCLUDING the tensors upstream zero-initialises (finding 8), small random biases.
    Norm affine parameters are RANDOM (gamma = 1 + norm_jitter * N(0,1), beta = norm_jitter * N(0,1)),
    each from its own name-seeded stream: a fixture built from this initialiser fails when two norms are
    swapped (norm1/norm2/norm3, in_layers.0/out_layers.0) or a beta is dropped.  norm_jitter = 0 gives
    the upstream initial values gamma = 1 / beta = 0 (what bench.py's synthetic weights use).
    Generated tensor-by-tensor in name order on ``device``."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd: SD = {}
    for name, shape in sorted(full_param_spec(cfg).items()):
        is_norm = ('.norm' in name or 'in_layers.0' in name or 'out_layers.0' in name
                   or name.endswith('out.0.weight') or name.endswith('out.0.bias'))
        if len(shape) == 1:
            if is_norm:
                base = 1.0 if name.endswith('weight') else 0.0
                t = torch.full(shape, base, dtype=torch.float32, device=device)
                if norm_jitter:
                    # own generator per tensor: the matrices keep the values they had before norm_jitter existed
                    gn = torch.Generator(device=device)
                    gn.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
                    t = t + norm_jitter * torch.randn(shape, generator=gn, dtype=torch.float32, device=device)
                t = t.to(dtype)
            else:
                t = 0.02 * torch.randn(shape, generator=g, dtype=torch.float32, device=device).to(dtype)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = (gain / math.sqrt(fan_in)) * torch.randn(shape, generator=g, dtype=torch.float32,
                                                          device=device).to(dtype)
        sd[name] = t
    return sd


# ----------------------------------------------------------------------------
# op semantics (App. A.2)
# ----------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def time_embed(sd: SD, p: str, t: Tensor, mc: int) -> Tensor:
    e = timestep_embedding(t, mc)
    e = F.linear(e, sd[f'{p}time_embed.0.weight'], sd[f'{p}time_embed.0.bias'])
    e = F.silu(e)
    return F.linear(e, sd[f'{p}time_embed.2.weight'], sd[f'{p}time_embed.2.bias'])


def resblock(sd: SD, p: str, x: Tensor, emb: Tensor) -> Tensor:
    h = F.group_norm(x.float(), 32, sd[f'{p}.in_layers.0.weight'], sd[f'{p}.in_layers.0.bias'], eps=1e-5)
    h = F.conv2d(F.silu(h), sd[f'{p}.in_layers.2.weight'], sd[f'{p}.in_layers.2.bias'], padding=1)
    e = F.linear(F.silu(emb), sd[f'{p}.emb_layers.1.weight'], sd[f'{p}.emb_layers.1.bias'])
    h = h + e[:, :, None, None]
    h = F.group_norm(h.float(), 32, sd[f'{p}.out_layers.0.weight'], sd[f'{p}.out_layers.0.bias'], eps=1e-5)
    h = F.conv2d(F.silu(h), sd[f'{p}.out_layers.3.weight'], sd[f'{p}.out_layers.3.bias'], padding=1)
    if f'{p}.skip_connection.weight' in sd:
        x = F.conv2d(x, sd[f'{p}.skip_connection.weight'], sd[f'{p}.skip_connection.bias'])
    return x + h


def attention(sd: SD, p: str, x: Tensor, context: Optional[Tensor], heads: int) -> Tensor:
    ctx = x if context is None else context
    q = F.linear(x, sd[f'{p}.to_q.weight'])
    k = F.linear(ctx, sd[f'{p}.to_k.weight'])
    v = F.linear(ctx, sd[f'{p}.to_v.weight'])
    b, n, d = q.shape
    dh = d // heads
    q = q.view(b, n, heads, dh).transpose(1, 2)
    k = k.view(b, -1, heads, dh).transpose(1, 2)
    v = v.view(b, -1, heads, dh).transpose(1, 2)
    sim = torch.einsum('bhid,bhjd->bhij', q, k) * (dh ** -0.5)
    o = torch.einsum('bhij,bhjd->bhid', sim.softmax(dim=-1), v)
    o = o.transpose(1, 2).reshape(b, n, d)
    return F.linear(o, sd[f'{p}.to_out.0.weight'], sd[f'{p}.to_out.0.bias'])


def transformer_block(sd: SD, p: str, x: Tensor, context: Tensor, heads: int) -> Tensor:
    d = x.shape[-1]
    x = attention(sd, f'{p}.attn1', F.layer_norm(x, (d,), sd[f'{p}.norm1.weight'], sd[f'{p}.norm1.bias']),
                  None, heads) + x
    x = attention(sd, f'{p}.attn2', F.layer_norm(x, (d,), sd[f'{p}.norm2.weight'], sd[f'{p}.norm2.bias']),
                  context, heads) + x
    y = F.layer_norm(x, (d,), sd[f'{p}.norm3.weight'], sd[f'{p}.norm3.bias'])
    y = F.linear(y, sd[f'{p}.ff.net.0.proj.weight'], sd[f'{p}.ff.net.0.proj.bias'])
    a, gate = y.chunk(2, dim=-1)
    y = F.linear(a * F.gelu(gate), sd[f'{p}.ff.net.2.weight'], sd[f'{p}.ff.net.2.bias'])
    return y + x


def spatial_transformer(sd: SD, p: str, x: Tensor, context: Tensor, heads: int, depth: int) -> Tensor:
    b, c, h, w = x.shape
    x_in = x
    y = F.group_norm(x, 32, sd[f'{p}.norm.weight'], sd[f'{p}.norm.bias'], eps=1e-6)
    y = F.conv2d(y, sd[f'{p}.proj_in.weight'], sd[f'{p}.proj_in.bias'])
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    for k in range(depth):
        y = transformer_block(sd, f'{p}.transformer_blocks.{k}', y, context, heads)
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    y = F.conv2d(y, sd[f'{p}.proj_out.weight'], sd[f'{p}.proj_out.bias'])
    return y + x_in


def _enc_block(sd: SD, cfg: NetConfig, p: str, b: BlockSpec, h: Tensor, emb: Tensor, ctx: Tensor) -> Tensor:
    if b.kind == 'conv_in':
        return F.conv2d(h, sd[f'{p}.0.weight'], sd[f'{p}.0.bias'], padding=1)
    if b.kind == 'down':
        return F.conv2d(h, sd[f'{p}.0.op.weight'], sd[f'{p}.0.op.bias'], stride=2, padding=1)
    h = resblock(sd, f'{p}.0', h, emb)
    if b.attn:
        h = spatial_transformer(sd, f'{p}.1', h, ctx, cfg.num_heads, cfg.transformer_depth)
    return h


def _middle(sd: SD, cfg: NetConfig, p: str, h: Tensor, emb: Tensor, ctx: Tensor) -> Tensor:
    h = resblock(sd, f'{p}middle_block.0', h, emb)
    h = spatial_transformer(sd, f'{p}middle_block.1', h, ctx, cfg.num_heads, cfg.transformer_depth)
    return resblock(sd, f'{p}middle_block.2', h, emb)


def hint_block(sd: SD, p: str, hint: Tensor) -> Tensor:
    """input_hint_block (App. A.3): 8 conv3x3, SiLU between, strides 1,1,2,1,2,1,2,1."""
    strides = (1, 1, 2, 1, 2, 1, 2, 1)
    h = hint
    for j, s in enumerate(strides):
        h = F.conv2d(h, sd[f'{p}input_hint_block.{2 * j}.weight'], sd[f'{p}input_hint_block.{2 * j}.bias'],
                     stride=s, padding=1)
        if j != 7:
            h = F.silu(h)
    return h


def control_model(sd: SD, cfg: NetConfig, x: Tensor, hint: Tensor, timesteps: Tensor, context: Tensor,
                  prefix: str = CONTROL_PREFIX, hint2: Optional[Tensor] = None, alpha: Optional[Tensor] = None) -> List[Tensor]:
    """cldm ControlNet.forward as called at reference makeup_diffuse.py:164-165 -> 13 residuals.

    hint2 / alpha: BUILD-DEFINED makeup interpolation (the reference has no code for it, only README.md:23-25):
    the two hint embeddings E(src||ref1), E(src||ref2) are blended per sample, (1-alpha) E1 + alpha E2, before
    they enter the ControlNet ('blend the two cached hint embeddings')."""
    emb = time_embed(sd, prefix, timesteps, cfg.model_channels)
    guided = hint_block(sd, prefix, hint)
    if hint2 is not None:
        a = alpha.view(-1, 1, 1, 1).to(guided.dtype)
        guided = (1.0 - a) * guided + a * hint_block(sd, prefix, hint2)
    outs = []
    h = x
    for i, b in enumerate(encoder_spec(cfg)):
        h = _enc_block(sd, cfg, f'{prefix}input_blocks.{i}', b, h, emb, context)
        if i == 0:
            h = h + guided
        outs.append(F.conv2d(h, sd[f'{prefix}zero_convs.{i}.0.weight'], sd[f'{prefix}zero_convs.{i}.0.bias']))
    h = _middle(sd, cfg, prefix, h, emb, context)
    outs.append(F.conv2d(h, sd[f'{prefix}middle_block_out.0.weight'], sd[f'{prefix}middle_block_out.0.bias']))
    return outs


def diffusion_model(sd: SD, cfg: NetConfig, x: Tensor, timesteps: Tensor, context: Tensor,
                    control: Optional[List[Tensor]] = None, only_mid_control: bool = False,
                    prefix: str = UNET_PREFIX) -> Tensor:
    """cldm ControlledUnetModel.forward as called at reference makeup_diffuse.py:161-168.
    Pops from ``control`` like upstream does (mutates the caller's list)."""
    emb = time_embed(sd, prefix, timesteps, cfg.model_channels)
    hs = []
    h = x
    for i, b in enumerate(encoder_spec(cfg)):
        h = _enc_block(sd, cfg, f'{prefix}input_blocks.{i}', b, h, emb, context)
        hs.append(h)
    h = _middle(sd, cfg, prefix, h, emb, context)
    if control is not None:
        h = h + control.pop()
    for i, b in enumerate(decoder_spec(cfg)):
        if only_mid_control or control is None:
            h = torch.cat([h, hs.pop()], dim=1)
        else:
            h = torch.cat([h, hs.pop() + control.pop()], dim=1)
        p = f'{prefix}output_blocks.{i}'
        h = resblock(sd, f'{p}.0', h, emb)
        k = 1
        if b.attn:
            h = spatial_transformer(sd, f'{p}.1', h, context, cfg.num_heads, cfg.transformer_depth)
            k = 2
        if b.up:
            h = F.interpolate(h, scale_factor=2, mode='nearest')
            h = F.conv2d(h, sd[f'{p}.{k}.conv.weight'], sd[f'{p}.{k}.conv.bias'], padding=1)
    h = F.group_norm(h.float(), 32, sd[f'{prefix}out.0.weight'], sd[f'{prefix}out.0.bias'], eps=1e-5)
    return F.conv2d(F.silu(h), sd[f'{prefix}out.2.weight'], sd[f'{prefix}out.2.bias'], padding=1)
