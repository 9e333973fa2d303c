"""Python host wrapper of the libmkd engine (plumbing: torch supplies device memory and the stream).

``MkdEngine`` is what ``diffmk.makeup_diffuse`` model classes delegate ``apply_model`` / ``sample_log``
to.  Everything here fails loudly without a GPU or without libmkd.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib


@dataclass
class NetConfig:
    """control_stage_config / unet_config params of diffmodels/base_diffusion_makeup.yaml:52-84."""
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
    hint_widths: Sequence[int] = (16, 16, 32, 32, 96, 96, 256)

    @classmethod
    def from_yaml_params(cls, control: dict, unet: dict) -> 'NetConfig':
        for k in ('model_channels', 'attention_resolutions', 'num_res_blocks', 'channel_mult', 'num_heads',
                  'transformer_depth', 'context_dim', 'in_channels'):
            if k in control and k in unet and list(np.atleast_1d(control[k])) != list(np.atleast_1d(unet[k])):
                raise ValueError(f'control_stage_config and unet_config disagree on {k}')
        if not unet.get('use_spatial_transformer', True) or unet.get('legacy', False):
            raise NotImplementedError('only use_spatial_transformer=True, legacy=False is supported')
        return cls(in_channels=unet.get('in_channels', 4), out_channels=unet.get('out_channels', 4),
                   hint_channels=control.get('hint_channels', 6), model_channels=unet['model_channels'],
                   attention_resolutions=tuple(unet['attention_resolutions']),
                   num_res_blocks=unet['num_res_blocks'], channel_mult=tuple(unet['channel_mult']),
                   num_heads=unet['num_heads'], transformer_depth=unet.get('transformer_depth', 1),
                   context_dim=unet['context_dim'],
                   hint_widths=tuple(control.get('hint_widths', (16, 16, 32, 32, 96, 96, 256))))

    def to_c(self) -> _lib.NetConfigC:
        c = _lib.NetConfigC()
        c.in_channels, c.out_channels, c.hint_channels = self.in_channels, self.out_channels, self.hint_channels
        c.model_channels, c.num_res_blocks = self.model_channels, self.num_res_blocks
        c.n_levels = len(self.channel_mult)
        for i, m in enumerate(self.channel_mult):
            c.channel_mult[i] = m
        c.n_attention_resolutions = len(self.attention_resolutions)
        for i, a in enumerate(self.attention_resolutions):
            c.attention_resolutions[i] = a
        c.num_heads, c.transformer_depth, c.context_dim = self.num_heads, self.transformer_depth, self.context_dim
        for i, hw in enumerate(self.hint_widths):
            c.hint_widths[i] = hw
        return c

    @property
    def n_control(self) -> int:
        n = 1
        for level in range(len(self.channel_mult)):
            n += self.num_res_blocks + (1 if level != len(self.channel_mult) - 1 else 0)
        return n + 1


@dataclass
class VaeConfig:
    """first_stage_config.params.ddconfig of diffmodels/base_diffusion_makeup.yaml:86-107 (decoder half)."""
    z_channels: int = 4
    embed_dim: int = 4
    ch: int = 128
    ch_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    out_ch: int = 3

    @classmethod
    def from_yaml_params(cls, first_stage_params: dict) -> 'VaeConfig':
        dd = first_stage_params.get('ddconfig', first_stage_params)
        if dd.get('attn_resolutions'):
            raise NotImplementedError('decoder attention resolutions other than the mid block are not supported')
        return cls(z_channels=dd.get('z_channels', 4), embed_dim=first_stage_params.get('embed_dim', 4), ch=dd.get('ch', 128),
                   ch_mult=tuple(dd.get('ch_mult', (1, 2, 4, 4))), num_res_blocks=dd.get('num_res_blocks', 2),
                   out_ch=dd.get('out_ch', 3))

    def to_c(self) -> _lib.VaeConfigC:
        c = _lib.VaeConfigC()
        c.z_channels, c.embed_dim, c.ch, c.n_levels = self.z_channels, self.embed_dim, self.ch, len(self.ch_mult)
        for i, m in enumerate(self.ch_mult):
            c.ch_mult[i] = m
        c.num_res_blocks, c.out_ch = self.num_res_blocks, self.out_ch
        return c


@dataclass
class ClipConfig:
    """cond_stage_config FrozenCLIPEmbedder (diffmodels/base_diffusion_makeup.yaml:109-110): UPSTREAM default
    openai/clip-vit-large-patch14 text tower (vocab 49408, 77 positions, width 768, 12 layers x 12 heads, MLP 3072, quick-GELU)."""
    vocab_size: int = 49408
    max_positions: int = 77
    width: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    ln_eps: float = 1e-5

    @classmethod
    def from_yaml_params(cls, cond_stage_params: Optional[dict]) -> 'ClipConfig':
        p = dict(cond_stage_params or {})
        known = {k: p[k] for k in ('vocab_size', 'max_positions', 'width', 'layers', 'heads', 'intermediate', 'ln_eps') if k in p}
        if 'max_length' in p:
            known['max_positions'] = int(p['max_length'])
        return cls(**known)

    def to_c(self) -> _lib.ClipConfigC:
        c = _lib.ClipConfigC()
        c.vocab_size, c.max_positions, c.width, c.layers = self.vocab_size, self.max_positions, self.width, self.layers
        c.heads, c.intermediate, c.ln_eps = self.heads, self.intermediate, self.ln_eps
        return c


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


class MkdEngine:
    """Owns one mkd_ctx on the current CUDA(HIP) device."""

    UNET_PREFIX = 'model.diffusion_model.'
    CONTROL_PREFIX = 'control_model.'

    def __init__(self, cfg: NetConfig, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise _lib.MkdError('MkdEngine needs a HIP device: the hot path has no CPU implementation')
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_ctx_create(C.byref(cfg.to_c()), C.byref(self._ctx)), 'mkd_ctx_create')
        self._keep: list = []          # tensors the prepared plan points at
        self.vae_cfg: Optional[VaeConfig] = None
        self.clip_cfg: Optional[ClipConfig] = None
        self._prepared_key = None
        self.batch = 0
        self.latent_hw: Tuple[int, int] = (0, 0)

    def close(self):
        if self._ctx:
            self.lib.mkd_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ---------------------------------------------------------------------------------
    def expected_params(self) -> Dict[str, Tuple[int, ...]]:
        n = self.lib.mkd_param_total(self._ctx)
        out = {}
        shp = (C.c_int64 * 4)()
        for i in range(n):
            name = self.lib.mkd_param_name(self._ctx, i).decode()
            nd = self.lib.mkd_param_shape(self._ctx, i, shp)
            out[name] = tuple(int(shp[j]) for j in range(nd))
        return out

    def param_count(self, which: str) -> int:
        return int(self.lib.mkd_param_count(self._ctx, 0 if which == 'unet' else 1))

    def load_weight(self, name: str, tensor: torch.Tensor) -> None:
        t = tensor.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
        with torch.cuda.device(self.device):
            if t.is_cuda:
                torch.cuda.current_stream().synchronize()
            _lib.check(self.lib.mkd_load_weight(self._ctx, name.encode(), C.c_void_p(t.data_ptr()), t.dim(), shape),
                       f'mkd_load_weight({name})')

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> List[str]:
        """Loads every key under model.diffusion_model. / control_model. (upstream names); other keys
        (first_stage_model.*, cond_stage_model.*, teacher_model*) are returned as 'unused'."""
        expected = self.expected_params()
        unused = []
        for k, v in sd.items():
            if k in expected:
                self.load_weight(k, v)
            else:
                unused.append(k)
        missing = [k for k in expected if k not in sd]
        core_missing = [k for k in missing if not k.startswith(('first_stage_model.', 'cond_stage_model.'))]
        if core_missing and strict:
            raise _lib.MkdError(f'{len(core_missing)} weights missing from state_dict, e.g. {core_missing[:3]}')
        if not core_missing:
            self.finalize()
        if getattr(self, 'vae_cfg', None) is not None and not [k for k in missing if k.startswith('first_stage_model.')]:
            self.finalize_vae()
        if getattr(self, 'clip_cfg', None) is not None and not [k for k in missing if k.startswith('cond_stage_model.')]:
            self.finalize_clip()
        return unused

    def init_random(self, seed: int = 0, gain: float = 1.0, norm_jitter: float = 0.0) -> None:
        """Seeded synthetic weights generated ON the device (bench / property tests): N(0, 1/fan_in) for
        every matrix/conv including upstream's zero-initialised ones, small biases, norm gamma = 1 + norm_jitter * N(0,1) and
        beta = norm_jitter * N(0,1) (0: the upstream initial values gamma 1 / beta 0)."""
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        for name, shape in self.expected_params().items():
            is_norm = ('.norm' in name or 'layer_norm' in name or 'in_layers.0' in name or 'out_layers.0' in name
                       or name.endswith('out.0.weight') or name.endswith('out.0.bias'))
            if len(shape) == 1:
                if is_norm:
                    t = (torch.ones if name.endswith('weight') else torch.zeros)(shape, device=self.device)
                    if norm_jitter:
                        t = t + norm_jitter * torch.randn(shape, generator=g, device=self.device)
                else:
                    t = 0.02 * torch.randn(shape, generator=g, device=self.device)
            else:
                fan_in = int(np.prod(shape[1:]))
                t = (gain / fan_in ** 0.5) * torch.randn(shape, generator=g, device=self.device)
            self.load_weight(name, t)
        self.finalize()

    def finalize(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_weights_finalize(self._ctx), 'mkd_weights_finalize')

    # ---- first-stage decoder -------------------------------------------------------------------------
    def configure_vae(self, vcfg: VaeConfig) -> None:
        """Adds the first_stage_model.{post_quant_conv,decoder}.* entries to expected_params(); load them like the rest."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_vae_configure(self._ctx, C.byref(vcfg.to_c())), 'mkd_vae_configure')
        self.vae_cfg = vcfg

    def finalize_vae(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_vae_finalize(self._ctx), 'mkd_vae_finalize')

    def decode(self, z: torch.Tensor, scale_factor: float = 0.18215) -> torch.Tensor:
        """decode_first_stage: z [B,4,h,w] -> images [B,3,8h,8w] fp32 (unclamped)."""
        z = _f32c(z, self.device)
        B, _, h, w = z.shape
        up = 2 ** (len(self.vae_cfg.ch_mult) - 1)
        out = torch.empty((B, self.vae_cfg.out_ch, h * up, w * up), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_decode(self._ctx, C.c_void_p(z.data_ptr()), B, h, w, float(scale_factor),
                                           C.c_void_p(out.data_ptr()), C.c_void_p(_stream())), 'mkd_decode')
        return out

    # ---- CLIP text encoder ----------------------------------------------------------------------------
    CLIP_PREFIX = 'cond_stage_model.transformer.text_model.'

    def configure_clip(self, ccfg: ClipConfig) -> None:
        """Adds the cond_stage_model.transformer.text_model.* entries to expected_params(); load them like the rest."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_clip_configure(self._ctx, C.byref(ccfg.to_c())), 'mkd_clip_configure')
        self.clip_cfg = ccfg

    def finalize_clip(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_clip_finalize(self._ctx), 'mkd_clip_finalize')

    def encode_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """FrozenCLIPEmbedder.forward after tokenisation: ids [B, T<=77] -> last_hidden_state [B, T, width] fp32."""
        if getattr(self, 'clip_cfg', None) is None:
            raise _lib.MkdError('text encoder not configured (configure_clip)')
        if tokens.dim() != 2 or tokens.dtype not in (torch.int32, torch.int64):
            raise ValueError('tokens must be an integer [B, T] tensor')
        if tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= self.clip_cfg.vocab_size):
            raise ValueError(f'token id outside [0, {self.clip_cfg.vocab_size})')      # torch.nn.Embedding raises here too
        tok = tokens.to(device=self.device, dtype=torch.int32).contiguous()
        B, T = tok.shape
        out = torch.empty((B, T, self.clip_cfg.width), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_clip_encode(self._ctx, C.c_void_p(tok.data_ptr()), B, T, C.c_void_p(out.data_ptr()),
                                                C.c_void_p(_stream())), 'mkd_clip_encode')
        return out

    def set_option(self, name: str, value: float) -> None:
        """Per-context plan switch (include/mkd.h: mkd_ctx_set_option); takes effect at the next prepare()."""
        _lib.check(self.lib.mkd_ctx_set_option(self._ctx, name.encode(), float(value)), f'mkd_ctx_set_option({name})')

    def get_option(self, name: str) -> float:
        v = C.c_double()
        _lib.check(self.lib.mkd_ctx_get_option(self._ctx, name.encode(), C.byref(v)), f'mkd_ctx_get_option({name})')
        return v.value

    def debug_poison(self) -> None:
        """Tests only: NaN-fill everything one eps evaluation produces (see mkd_debug_poison)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_debug_poison(self._ctx), 'mkd_debug_poison')

    def decode_flops(self) -> float:
        return float(self.lib.mkd_decode_flops(self._ctx))

    # ---- conditioning / eval -----------------------------------------------------------------------
    def prepare(self, hint: Optional[torch.Tensor], context: torch.Tensor, latent_hw: Optional[Tuple[int, int]] = None,
                control_scales: Optional[Sequence[float]] = None, only_mid_control: bool = False,
                hint2: Optional[torch.Tensor] = None, alpha: Optional[torch.Tensor] = None) -> None:
        """hint [B,6,8h,8w] in [0,1] or None (c_concat is None); context [B,77,ctx_dim]."""
        B = context.shape[0]
        ctx = _f32c(context, self.device)
        if ctx.shape[1] != 77 or ctx.shape[2] != self.cfg.context_dim:
            raise ValueError(f'context must be [B,77,{self.cfg.context_dim}], got {tuple(ctx.shape)}')
        hint_t = None
        if hint is not None:
            hint_t = _f32c(hint, self.device)
            if hint_t.shape[0] != B or hint_t.shape[1] != self.cfg.hint_channels or hint_t.shape[2] % 8 or hint_t.shape[3] % 8:
                raise ValueError(f'hint must be [B,{self.cfg.hint_channels},8h,8w], got {tuple(hint_t.shape)}')
            h, w = hint_t.shape[2] // 8, hint_t.shape[3] // 8
            if latent_hw is not None and tuple(latent_hw) != (h, w):
                raise ValueError('latent_hw disagrees with the hint size')
        else:
            if latent_hw is None:
                raise ValueError('latent_hw is required when hint is None')
            h, w = latent_hw
        scales = None
        if control_scales is not None:
            if len(control_scales) != self.cfg.n_control:
                raise ValueError(f'control_scales must have {self.cfg.n_control} entries')
            scales = (C.c_float * len(control_scales))(*[float(s) for s in control_scales])
        hint2_t = alpha_t = None
        if hint2 is not None:
            if hint_t is None or alpha is None:
                raise ValueError('interpolation needs hint, hint2 and alpha')
            hint2_t = _f32c(hint2, self.device)
            alpha_t = _f32c(alpha, self.device).reshape(-1)
            if tuple(hint2_t.shape) != tuple(hint_t.shape) or alpha_t.shape[0] != B:
                raise ValueError('hint2 must match hint and alpha must have one entry per sample')
        with torch.cuda.device(self.device):
            if hint2_t is None:
                _lib.check(self.lib.mkd_prepare(self._ctx, B, h, w, C.c_void_p(_ptr(hint_t)), C.c_void_p(ctx.data_ptr()),
                                                scales, int(bool(only_mid_control)), C.c_void_p(_stream())), 'mkd_prepare')
            else:
                _lib.check(self.lib.mkd_prepare_interp(self._ctx, B, h, w, C.c_void_p(hint_t.data_ptr()), C.c_void_p(hint2_t.data_ptr()),
                                                       C.c_void_p(alpha_t.data_ptr()), C.c_void_p(ctx.data_ptr()), scales,
                                                       int(bool(only_mid_control)), C.c_void_p(_stream())), 'mkd_prepare_interp')
        self._keep = [hint_t, ctx, hint2_t, alpha_t]
        self.batch, self.latent_hw = B, (h, w)

    def eps(self, x: torch.Tensor, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One apply_model evaluation on the prepared conditioning. x [B,4,h,w] fp32, t [B] int64."""
        x = _f32c(x, self.device)
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        if tuple(x.shape) != (self.batch, self.cfg.in_channels, *self.latent_hw) or t.shape[0] != self.batch:
            raise ValueError(f'x/t do not match the prepared batch {self.batch} x {self.latent_hw}: {tuple(x.shape)}')
        if out is None:
            out = torch.empty((self.batch, self.cfg.out_channels, *self.latent_hw), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_eps(self._ctx, C.c_void_p(x.data_ptr()), C.c_void_p(t.data_ptr()),
                                        C.c_void_p(out.data_ptr()), C.c_void_p(_stream())), 'mkd_eps')
        return out

    def ddim_step(self, x, eps_c, eps_u, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None,
                  temperature: float = 1.0, want_x0: bool = True):
        x = _f32c(x, self.device); eps_c = _f32c(eps_c, self.device)
        eps_u = None if eps_u is None else _f32c(eps_u, self.device)
        noise = None if noise is None else _f32c(noise, self.device)
        x_prev = torch.empty_like(x)
        x0 = torch.empty_like(x) if want_x0 else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_ddim_step(C.c_void_p(x.data_ptr()), C.c_void_p(eps_c.data_ptr()), C.c_void_p(_ptr(eps_u)),
                                              float(cfg_scale), float(a_t), float(a_prev), float(sigma_t),
                                              float(sqrt_one_minus_at), C.c_void_p(_ptr(noise)), float(temperature),
                                              C.c_void_p(x_prev.data_ptr()), C.c_void_p(_ptr(x0)), x.numel(),
                                              C.c_void_p(_stream())), 'mkd_ddim_step')
        return x_prev, x0

    def sample(self, x_T: torch.Tensor, timesteps: Sequence[int], alphas: Sequence[float], alphas_prev: Sequence[float],
               sqrt_one_minus_alphas: Sequence[float], cfg_scale: float = 1.0, use_graph: bool = False,
               sigmas: Optional[Sequence[float]] = None, noise: Optional[torch.Tensor] = None, temperature: float = 1.0) -> torch.Tensor:
        """Whole reverse loop in one call (cddim.py:81-100). Prepared batch must be B or 2B (CFG).  eta > 0 (cddim.py:74-78):
        ``sigmas`` like the other tables and ``noise`` [n_steps, B, 4, h, w], row k = the draw of the k-th executed step."""
        x_T = _f32c(x_T, self.device)
        cfg_on = float(cfg_scale) != 1.0
        want_b = self.batch // 2 if cfg_on else self.batch
        if (x_T.dim() != 4 or tuple(x_T.shape[1:]) != (self.cfg.in_channels, *self.latent_hw) or x_T.shape[0] != want_b
                or (cfg_on and self.batch % 2)):
            # libmkd copies batch * C * h * w floats using the PREPARED h, w: a smaller latent would be read out of bounds
            raise ValueError(f'x_T {tuple(x_T.shape)} does not match the prepared conditioning: expected '
                             f'({want_b}, {self.cfg.in_channels}, {self.latent_hw[0]}, {self.latent_hw[1]})'
                             + (' (CFG: prepared batch is [uncond; cond])' if cfg_on else ''))
        n = len(timesteps)
        if n <= 0 or not (len(alphas) == len(alphas_prev) == len(sqrt_one_minus_alphas) == n):
            raise ValueError('timesteps / alphas / alphas_prev / sqrt_one_minus_alphas must be non-empty and equally long')
        ts = (C.c_int64 * n)(*[int(v) for v in timesteps])
        a = (C.c_float * n)(*[float(v) for v in alphas])
        ap = (C.c_float * n)(*[float(v) for v in alphas_prev])
        s1 = (C.c_float * n)(*[float(v) for v in sqrt_one_minus_alphas])
        out = torch.empty_like(x_T)
        if sigmas is not None and any(float(v) != 0.0 for v in sigmas):
            if len(sigmas) != n:
                raise ValueError('sigmas must be as long as timesteps')
            if noise is None or tuple(noise.shape) != (n, *x_T.shape):
                raise ValueError(f'eta > 0 needs noise of shape {(n, *x_T.shape)} (one draw per executed step)')
            noise = _f32c(noise, self.device)
            sg = (C.c_float * n)(*[float(v) for v in sigmas])
            with torch.cuda.device(self.device):
                _lib.check(self.lib.mkd_sample_eta(self._ctx, C.c_void_p(x_T.data_ptr()), x_T.shape[0], n, ts, a, ap, s1, sg,
                                                   C.c_void_p(noise.data_ptr()), float(temperature), float(cfg_scale),
                                                   C.c_void_p(out.data_ptr()), int(use_graph), C.c_void_p(_stream())), 'mkd_sample_eta')
                if use_graph:
                    torch.cuda.synchronize(self.device)          # the replayed loop reads `noise` after this call returns: keep it alive
            return out
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_sample(self._ctx, C.c_void_p(x_T.data_ptr()), x_T.shape[0], n, ts, a, ap, s1,
                                           float(cfg_scale), C.c_void_p(out.data_ptr()), int(use_graph),
                                           C.c_void_p(_stream())), 'mkd_sample')
        return out

    def eps_profile(self, x: torch.Tensor, t: torch.Tensor, csv_path: Optional[str] = None) -> Dict[str, Dict[str, float]]:
        """One eps with HIP events around every launch group -> {kernel class: {ms, flops, launches, bytes, ms_b2b}}: ``bytes`` =
        algorithmic HBM bytes of the memory-bound classes, ``ms_b2b`` = the class's launches replayed back to back between one
        event pair (no per-launch event overhead)."""
        x = _f32c(x, self.device)
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty((self.batch, self.cfg.out_channels, *self.latent_hw), device=self.device, dtype=torch.float32)
        n = self.lib.mkd_kind_count()
        ms = (C.c_double * n)(); fl = (C.c_double * n)(); ln = (C.c_int * n)(); by = (C.c_double * n)(); b2b = (C.c_double * n)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mkd_eps_profile2(self._ctx, C.c_void_p(x.data_ptr()), C.c_void_p(t.data_ptr()),
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(_stream()), ms, fl, ln, by, b2b,
                                                 None if csv_path is None else csv_path.encode()), 'mkd_eps_profile2')
        return {self.lib.mkd_kind_name(k).decode(): {'ms': ms[k], 'flops': fl[k], 'launches': ln[k], 'bytes': by[k], 'ms_b2b': b2b[k]}
                for k in range(n)}

    # ---- introspection -----------------------------------------------------------------------------
    def eps_flops(self) -> float:
        return float(self.lib.mkd_eps_flops(self._ctx))

    def eps_launches(self) -> int:
        return int(self.lib.mkd_eps_launches(self._ctx))

    def step_launches(self, use_graph: bool = True, cfg: bool = False) -> int:
        """Kernel launches of one DDIM step inside ``sample`` (time embedding hoisted out of the loop), as that loop is run."""
        return int(self.lib.mkd_step_launches_ex(self._ctx, int(use_graph), int(cfg)))

    def device_bytes(self) -> int:
        return int(self.lib.mkd_device_bytes(self._ctx))
