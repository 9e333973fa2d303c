"""``diffmk.makeup_diffuse`` — inference-side drop-in for the reference model classes on the hot path.

Mirrors, for the DDIM sampling path only:
  * ``apply_model``                 reference diffmk/makeup_diffuse.py:152-170
  * ``get_input`` cond assembly     reference diffmk/makeup_diffuse.py:42-57, diffmk/makeup_controlnet.py:137-167
  * ``sample_log`` / ``log_results``reference diffmk/diffusion_makeup.py:360-411 (UPSTREAM ControlLDM.sample_log)
Training losses, teachers, PL hooks and the VAE/CLIP stages are out of scope for this path;
calling them raises NotImplementedError rather than returning something different from the reference.
The arithmetic runs in libmkd (HIP); there is no CPU implementation behind these classes."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from ..ddim import DDIMSampler
from ..engine import ClipConfig, MkdEngine, NetConfig, VaeConfig
from ..lib import MkdError
from ..schedule import DDIMSchedule


class _Held:
    """Cache key over tensors that HOLDS them: identity + in-place version.  An address-based key would match a new tensor
    that the caching allocator placed where a freed one used to be (the next batch of a test loop: same shapes, same
    allocation order) and silently reuse the previous batch's conditioning."""

    def __init__(self, tensors, extra=()):
        self.tensors = tuple(tensors)
        self.versions = tuple(None if t is None else t._version for t in self.tensors)
        self.extra = tuple(extra)

    def matches(self, tensors, extra=()) -> bool:
        tensors = tuple(tensors)
        return (len(tensors) == len(self.tensors) and tuple(extra) == self.extra
                and all(a is b and (a is None or a._version == v) for a, b, v in zip(tensors, self.tensors, self.versions)))


class BaseMakeUpDiffuse:
    """ControlLDM-shaped inference model: ControlNet(hint = src ‖ ref) -> 13 residuals -> ControlledUnet."""

    def __init__(self, control_stage_config: dict, unet_config: dict, first_stage_config: Optional[dict] = None,
                 cond_stage_config: Optional[dict] = None, linear_start: float = 0.00085, linear_end: float = 0.0120,
                 timesteps: int = 1000, beta_schedule: str = 'linear', scale_factor: float = 0.18215,
                 only_mid_control: bool = False, parameterization: str = 'eps', channels: int = 4, image_size: int = 64,
                 conditioning_key: str = 'crossattn', first_stage_key: str = 'jpg', cond_stage_key: str = 'txt',
                 control_key: str = 'ref_img', src_key: str = 'src_img', src_img_key: str = 'src_img',
                 ref_img_key: str = 'ref_img', use_ema: bool = False, **unused_training_params):
        if parameterization != 'eps':
            raise NotImplementedError("parameterization must be 'eps' (base_diffusion_makeup.yaml:50)")
        self.net_config = NetConfig.from_yaml_params(dict(control_stage_config.get('params', control_stage_config)),
                                                     dict(unet_config.get('params', unet_config)))
        self.first_stage_config, self.cond_stage_config = first_stage_config, cond_stage_config
        self.vae_config = None
        if first_stage_config is not None:
            self.vae_config = VaeConfig.from_yaml_params(dict(first_stage_config.get('params', first_stage_config)))
        self.clip_config = None
        if cond_stage_config is not None:
            self.clip_config = ClipConfig.from_yaml_params(cond_stage_config.get('params'))
        self.extra_params = dict(unused_training_params)      # w_idt_src, lambda_lip, teacher_type, ... (training only)
        self.parameterization = parameterization
        self.only_mid_control = bool(only_mid_control)
        self.control_scales: List[float] = [1.0] * self.net_config.n_control
        self.scale_factor, self.channels, self.image_size = scale_factor, channels, image_size
        self.conditioning_key = conditioning_key
        self.first_stage_key, self.cond_stage_key, self.control_key = first_stage_key, cond_stage_key, control_key
        self.src_img_key = src_img_key if src_img_key else src_key
        self.ref_img_key = ref_img_key if ref_img_key else control_key
        self.linear_start, self.linear_end = linear_start, linear_end
        self.device = torch.device('cpu')
        self.register_schedule(beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start, linear_end=linear_end)
        self.engine: Optional[MkdEngine] = None
        self._pending_sd: Optional[Dict[str, torch.Tensor]] = None
        self._bound = None
        self._cfg_cache = None
        self._cat_cache = None
        self.cond_stage_model = None          # callable(list[str]) -> [B,77,768]; built on .cuda() when cond_stage_config is set
        self.training = False

    _SCHEDULE_TABLES = ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod',
                        'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod')

    def register_schedule(self, given_betas=None, beta_schedule: str = 'linear', timesteps: int = 1000, linear_start: float = 1e-4,
                          linear_end: float = 2e-2, cosine_s: float = 8e-3) -> None:
        """UPSTREAM DDPM.register_schedule as the reference calls it (diffmk/makeups.py:40-42, ``update_schedule``): (re)builds
        the beta / alphas_cumprod tables for ``timesteps`` steps.  Samplers built afterwards see the new ``num_timesteps``."""
        if given_betas is not None:
            raise NotImplementedError('given_betas (the reference passes None, diffmk/makeups.py:41)')
        sch = DDIMSchedule(int(timesteps), linear_start, linear_end, beta_schedule)
        self.schedule = sch
        self.num_timesteps = sch.num_timesteps
        self.linear_start, self.linear_end = linear_start, linear_end
        for n in self._SCHEDULE_TABLES:
            setattr(self, n, getattr(sch, n).to(self.device))

    # ---- nn.Module-ish surface used by runs/test.py --------------------------------------------------------
    def cpu(self):
        return self

    def eval(self):
        self.training = False
        return self

    def cuda(self, device=None):
        return self.to(torch.device('cuda', torch.cuda.current_device() if device is None else device))

    def to(self, device):
        device = torch.device(device)
        if device.type == 'cuda':
            if self.engine is None:
                self.engine = MkdEngine(self.net_config, device)
                if self.vae_config is not None:
                    self.engine.configure_vae(self.vae_config)
                if self.clip_config is not None:
                    from ..clip import FrozenCLIPEmbedder, load_tokenizer
                    self.engine.configure_clip(self.clip_config)
                    params = dict((self.cond_stage_config or {}).get('params') or {})
                    tok_dir = params.get('tokenizer_path') or params.get('version')
                    tok = load_tokenizer(tok_dir) if tok_dir and os.path.isdir(str(tok_dir)) else None
                    self.cond_stage_model = FrozenCLIPEmbedder(self.engine, tok, max_length=self.clip_config.max_positions)
                if self._pending_sd is not None:
                    self.engine.load_state_dict(self._pending_sd, strict=True)
                    self._pending_sd = None
            for n in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
                      'sqrt_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod'):
                setattr(self, n, getattr(self, n).to(device))
            self.device = device
        return self

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Accepts an upstream-named checkpoint dict (runs/test.py:59-60).  Keys outside the two nets
        (first_stage_model.*, cond_stage_model.*, teacher_model*) are reported back as unexpected."""
        sd = state_dict.get('state_dict', state_dict)
        if self.engine is not None:
            unused = self.engine.load_state_dict(sd, strict=strict)
        else:
            self._pending_sd = {k: v for k, v in sd.items()
                                if k.startswith(MkdEngine.UNET_PREFIX) or k.startswith(MkdEngine.CONTROL_PREFIX)
                                or k.startswith('first_stage_model.post_quant_conv.') or k.startswith('first_stage_model.decoder.')
                                or (k.startswith(MkdEngine.CLIP_PREFIX) and not k.endswith('position_ids'))}
            unused = [k for k in sd if k not in self._pending_sd]
        return [], unused

    def _require_engine(self) -> MkdEngine:
        if self.engine is None:
            raise MkdError('model is not on a HIP device: call .cuda() first (the hot path has no CPU implementation)')
        return self.engine

    # ---- conditioning -------------------------------------------------------------------------------------
    def get_origin_img_input(self, batch: dict, k: str, bs: Optional[int] = None) -> torch.Tensor:
        x = batch[k]
        if bs is not None:
            x = x[:bs]
        return x.to(self.device).to(memory_format=torch.contiguous_format).float()

    def get_learned_conditioning(self, txt: Sequence[str]) -> torch.Tensor:
        if self.cond_stage_model is None:
            raise NotImplementedError('no cond_stage_config in the yaml and no cond_stage_model set: '
                                      "put a precomputed [B,77,768] embedding under batch['txt_emb']")
        return self.cond_stage_model(list(txt)).to(self.device).float()

    def get_cond_txt_coding(self, batch: dict, bs: Optional[int] = None) -> torch.Tensor:
        if 'txt_emb' in batch:
            c = batch['txt_emb']
            return (c if bs is None else c[:bs]).to(self.device).float()
        if 'txt_tokens' in batch and self.cond_stage_model is not None:      # ids from an external tokenizer
            tk = batch['txt_tokens']
            return self.cond_stage_model.encode_tokens(tk if bs is None else tk[:bs]).float()
        txt = batch[self.cond_stage_key]
        return self.get_learned_conditioning(txt if bs is None else txt[:bs])

    def get_unconditional_conditioning(self, N: int) -> torch.Tensor:
        if getattr(self, 'uncond_embedding', None) is not None:
            u = self.uncond_embedding.to(self.device).float()
            return u.expand(N, -1, -1).contiguous() if u.shape[0] == 1 else u[:N]
        return self.get_learned_conditioning([''] * N)

    @torch.no_grad()
    def get_input(self, batch: dict, k, bs: Optional[int] = None, *args, **kwargs):
        """-> (None, c) with c_concat = [cat(src_img, ref_img, 1)] (source first) and c_crossattn = [text]."""
        src = self.get_origin_img_input(batch, self.src_img_key, bs)
        ref = self.get_origin_img_input(batch, self.ref_img_key, bs)
        c = {'c_crossattn': [self.get_cond_txt_coding(batch, bs)], 'src_img': src, 'ref_img': ref,
             'c_concat': [torch.cat((src, ref), 1)]}
        return None, c

    def _bind(self, hint: Optional[torch.Tensor], ctx: torch.Tensor, latent_hw) -> MkdEngine:
        eng = self._require_engine()
        extra = (tuple(latent_hw), tuple(self.control_scales), self.only_mid_control)
        if self._bound is None or not self._bound.matches((hint, ctx), extra):
            eng.prepare(hint, ctx, latent_hw=tuple(latent_hw), control_scales=self.control_scales,
                        only_mid_control=self.only_mid_control)
            self._bound = _Held((hint, ctx), extra)
        return eng

    def _bind_cond(self, cond: dict, latent_hw) -> MkdEngine:
        """cat(c_crossattn, 1) / cat(c_concat, 1) of a cond dict (reference diffmk/makeup_diffuse.py:159,165) bound to the engine.
        A list with several entries is concatenated ONCE per distinct set of entries: the key holds the list ELEMENTS (a fresh
        torch.cat result would never match itself and the hint block / K-V caches would be rebuilt every DDIM step)."""
        xs = list(cond['c_crossattn'])
        hs = None if cond.get('c_concat') is None else list(cond['c_concat'])
        if len(xs) == 1 and (hs is None or len(hs) == 1):
            return self._bind(None if hs is None else hs[0], xs[0], latent_hw)
        parts = xs + (hs or [])
        shape = (len(xs), -1 if hs is None else len(hs))
        if self._cat_cache is None or not self._cat_cache[0].matches(parts, shape):
            self._cat_cache = (_Held(parts, shape), torch.cat(xs, 1) if len(xs) > 1 else xs[0],
                               None if hs is None else (torch.cat(hs, 1) if len(hs) > 1 else hs[0]))
        return self._bind(self._cat_cache[2], self._cat_cache[1], latent_hw)

    def reset_conditioning_cache(self) -> None:
        """Drop the prepared-conditioning and CFG-merge caches (and the tensors they hold)."""
        self._bound = None
        self._cfg_cache = None
        self._cat_cache = None

    def cfg_conditioning(self, uncond: dict, cond: dict) -> dict:
        """[uncond; cond] batching (cddim.py:18-38), cached per (uncond, cond) pair so that a step-by-step
        caller does not rebuild — and libmkd does not re-prepare — identical conditioning every step."""
        def flat(c):
            out, names = [], []
            for k in sorted(c):
                v = c[k]
                if isinstance(v, list):
                    out += list(v); names.append((k, len(v)))
                elif isinstance(v, torch.Tensor):
                    out.append(v); names.append((k, -1))
            return out, names
        (tu, nu), (tc, nc) = flat(uncond), flat(cond)
        if self._cfg_cache is None or not self._cfg_cache[0].matches(tu + tc, (tuple(nu), tuple(nc))):
            merged = {}
            for k in cond:
                if isinstance(cond[k], list):
                    merged[k] = [torch.cat([uncond[k][i], cond[k][i]]) for i in range(len(cond[k]))]
                elif isinstance(cond[k], torch.Tensor):
                    merged[k] = torch.cat([uncond[k], cond[k]])
                else:
                    merged[k] = cond[k]
            self._cfg_cache = (_Held(tu + tc, (tuple(nu), tuple(nc))), merged)
        return self._cfg_cache[1]

    # ---- the eps model ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def apply_model(self, x_noisy: torch.Tensor, t: torch.Tensor, cond: dict, return_all: bool = False, *args, **kwargs):
        assert isinstance(cond, dict)
        eng = self._bind_cond(cond, x_noisy.shape[2:])
        eps = eng.eps(x_noisy, t)
        if not return_all:
            return eps
        return eps, self.predict_start_from_noise(x_t=x_noisy, t=t, noise=eps)

    def predict_start_from_noise(self, x_t: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        a = self.sqrt_recip_alphas_cumprod.to(x_t.device)[t].view(-1, 1, 1, 1)
        b = self.sqrt_recipm1_alphas_cumprod.to(x_t.device)[t].view(-1, 1, 1, 1)
        return a * x_t - b * noise

    # hooks the samplers use to stay on the device ----------------------------------------------------------------
    def ddim_step(self, x, e_c, e_u, scale, a_t, a_prev, sigma_t, s1m_t, noise, temperature):
        return self._require_engine().ddim_step(x, e_c, e_u, scale, a_t, a_prev, sigma_t, s1m_t, noise, temperature)

    # hipGraph replay of the sampling loop (one captured step, five steps per graph): the configuration bench.py measures.  False: eager
    sample_use_graph = True

    def sample_loop_fast(self, x_latent, cond, timesteps, alphas, alphas_prev, sqrt_one_minus_alphas,
                         unconditional_guidance_scale=1.0, unconditional_conditioning=None, sigmas=None, noise=None, temperature=1.0):
        cfg_on = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.0)
        c = self.cfg_conditioning(unconditional_conditioning, cond) if cfg_on else cond
        eng = self._bind_cond(c, x_latent.shape[2:])
        return eng.sample(x_latent, [int(v) for v in timesteps], [float(v) for v in alphas], [float(v) for v in alphas_prev],
                          [float(v) for v in sqrt_one_minus_alphas],
                          cfg_scale=float(unconditional_guidance_scale) if cfg_on else 1.0, use_graph=bool(self.sample_use_graph),
                          sigmas=None if sigmas is None else [float(v) for v in sigmas], noise=noise, temperature=float(temperature))

    # ---- sampling drivers ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_log(self, cond: dict, batch_size: int, ddim: bool, ddim_steps: int, **kwargs):
        """UPSTREAM ControlLDM.sample_log: latent shape from the hint, x_T ~ N(0, I) unless given."""
        if not ddim:
            raise NotImplementedError('only the DDIM sampler is on the MakeupDiffuse test path')
        sampler = DDIMSampler(self)
        _, _, h, w = cond['c_concat'][0].shape
        shape = (self.channels, h // 8, w // 8)
        return sampler.sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)

    def decode_first_stage(self, z: torch.Tensor) -> torch.Tensor:
        """z / scale_factor -> post_quant_conv -> Decoder (UPSTREAM AutoencoderKL.decode), on the device via mkd_decode."""
        eng = self._require_engine()
        if eng.vae_cfg is None:
            raise NotImplementedError('no first_stage_config in the yaml: the decoder was not configured')
        return eng.decode(z, self.scale_factor)

    def decode_latent_code(self, z: torch.Tensor, predict_cids: bool = False, force_not_quantize: bool = False) -> torch.Tensor:
        """reference diffmk/makeups.py:260-262: ``first_stage_model.decode(z / scale_factor)`` (unclamped)."""
        return self.decode_first_stage(z)

    @torch.no_grad()
    def generate_image(self, z: torch.Tensor, format: bool = False) -> torch.Tensor:
        """reference diffmk/makeup_diffuse.py:172-177: decode, clamp to [-1, 1], optionally map to [0, 1]."""
        img = self.decode_first_stage(z).clamp(-1, 1)
        if format:
            img = (img + 1.0) / 2.0
        return img

    @property
    def has_first_stage(self) -> bool:
        return self.engine is not None and self.engine.vae_cfg is not None


class TestDiffuseModel(BaseMakeUpDiffuse):
    """Reference Test* harness classes (diffmk/diffusion_makeup.py:308-411, diffmk/makeup_diffuse.py:413-464):
    adds the sampling settings and ``log_results``' two DDIM passes."""
    __test__ = False          # (pytest: a class named Test*, not a test case)

    def __init__(self, saved_dir: str = './results', model_name: str = 'makeupdiffuse', img_name_key: str = 'img_name',
                 unconditional_guidance_scale: float = 9, ddim_steps: int = 50, ddim_eta: float = 0.0, sample: bool = True,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.unconditional_guidance_scale = unconditional_guidance_scale
        self.ddim_steps, self.ddim_eta, self.sample = ddim_steps, ddim_eta, sample
        self.saved_dir, self.model_name, self.img_name_key = saved_dir, model_name, img_name_key
        self.clamp = True
        self.rescale = True
        self.save_images = True                    # save_local after every test_step, as the reference does
        self.test_pairs: list = []
        self.test_pairs_file = 'test_0412_pairs.txt'

    def on_test_epoch_start(self) -> None:
        self.eval()
        self.test_pairs = []

    def on_test_batch_end(self, outputs=None, batch=None, batch_idx: int = 0, dataloader_idx: int = 0) -> None:
        """'num-num nonmakeup makeup' bookkeeping file (diffusion_makeup.py:327-331), rewritten after every batch."""
        with open(self.test_pairs_file, 'w') as f:
            for tp in self.test_pairs:
                f.write('%s %s %s\n' % (tp[0], tp[1], tp[2]))

    @torch.no_grad()
    def log_results(self, batch: dict, batch_idx: int, x_T: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """The sampler calls of reference log_results (:391-410): a plain 50-step pass and a CFG pass whose
        unconditional branch keeps the SAME hint (uc_cat = c_cat, :401).  Returns latents (and decoded images
        once a first_stage_model is attached); the teacher / reconstruction rows are out of scope."""
        use_ddim = self.ddim_steps is not None
        log: Dict[str, torch.Tensor] = {}
        _, c = self.get_input(batch, self.first_stage_key)
        c_cat, c_txt = c['c_concat'][0], c['c_crossattn'][0]
        src, ref = torch.chunk(c_cat, 2, dim=1)
        log['control_src'] = src * 2.0 - 1.0
        log['control_ref'] = ref * 2.0 - 1.0
        names = batch.get(self.img_name_key)
        if names is not None:                      # :379-384
            for i, nm in enumerate(names):
                self.test_pairs.append(['%04d-%d' % (batch_idx, i + 1), 'non-makeup/%s.png' % nm.split('&')[0],
                                        'makeup/%s.png' % nm.split('&')[1]])
        b = c_cat.shape[0]
        extra = {} if x_T is None else {'x_T': x_T}
        cond = {'c_concat': [c_cat], 'c_crossattn': [c_txt]}
        if self.sample:
            samples, _ = self.sample_log(cond=cond, batch_size=b, ddim=use_ddim, ddim_steps=self.ddim_steps,
                                         eta=self.ddim_eta, **extra)
            log['samples_latent'] = samples
            if self.has_first_stage:
                log['samples'] = self.decode_first_stage(samples)
        if self.unconditional_guidance_scale > 1.0:
            uc_full = {'c_concat': [c_cat], 'c_crossattn': [self.get_unconditional_conditioning(b)]}
            samples_cfg, _ = self.sample_log(cond=cond, batch_size=b, ddim=use_ddim, ddim_steps=self.ddim_steps,
                                             eta=self.ddim_eta, unconditional_guidance_scale=self.unconditional_guidance_scale,
                                             unconditional_conditioning=uc_full, **extra)
            name = f'samples_cfg_scale_{self.unconditional_guidance_scale:.2f}'
            log[name + '_latent'] = samples_cfg
            if self.has_first_stage:
                log[name] = self.decode_first_stage(samples_cfg)
        return log

    @torch.no_grad()
    def interpolate(self, batch: dict, alphas, x_T: Optional[torch.Tensor] = None, ref2_key: str = 'ref_img2',
                    unconditional_guidance_scale: float = 1.0) -> Dict[str, torch.Tensor]:
        """Makeup interpolation between two references (README.md:23-25 shows the figure; the reference has no code, so the
        definition is this build's: the ControlNet hint embeddings E(src||ref1), E(src||ref2) are blended per sample with
        weight alpha before the 50-step loop).  batch holds src_img, ref_img, `ref2_key`, txt_emb for
        N pairs; every pair is sampled at every alpha -> latents [N*len(alphas), 4, h, w] ordered pair-major."""
        src = self.get_origin_img_input(batch, self.src_img_key)
        r1 = self.get_origin_img_input(batch, self.ref_img_key)
        r2 = self.get_origin_img_input(batch, ref2_key)
        ctx = self.get_cond_txt_coding(batch)
        A = torch.as_tensor(list(alphas), dtype=torch.float32)
        n, k = src.shape[0], A.numel()
        rep = lambda t: t.repeat_interleave(k, 0)
        h1, h2 = rep(torch.cat((src, r1), 1)), rep(torch.cat((src, r2), 1))
        ctxr, al = rep(ctx), A.repeat(n).to(self.device)
        eng = self._require_engine()
        h, w = src.shape[2] // 8, src.shape[3] // 8
        if x_T is None:
            x_T = torch.randn(n, self.channels, h, w, device=self.device)
        x_T = rep(x_T.to(self.device))           # the same start noise for every alpha of a pair
        sch = self.schedule
        sch.make_ddim(self.ddim_steps, ddim_eta=0.0)
        if unconditional_guidance_scale != 1.0:
            u = self.get_unconditional_conditioning(n * k)
            eng.prepare(torch.cat([h1, h1]), torch.cat([u, ctxr]), hint2=torch.cat([h2, h2]), alpha=torch.cat([al, al]),
                        control_scales=self.control_scales, only_mid_control=self.only_mid_control)
        else:
            eng.prepare(h1, ctxr, hint2=h2, alpha=al, control_scales=self.control_scales, only_mid_control=self.only_mid_control)
        self.reset_conditioning_cache()
        lat = eng.sample(x_T, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas,
                         cfg_scale=float(unconditional_guidance_scale), use_graph=True)
        out = {'samples_latent': lat, 'alpha': al}
        if self.has_first_stage:
            out['samples'] = self.decode_first_stage(lat)
        return out

    def test_step(self, batch: dict, batch_idx: int, x_T: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """reference diffusion_makeup.py:332-341.  x_T (not in the reference, which always draws fresh noise): fixed start
        noise for both sampling passes, so that runs can be compared."""
        images = self.log_results(batch, batch_idx, x_T=x_T)
        for k in images:
            if isinstance(images[k], torch.Tensor):
                images[k] = images[k].detach().cpu()
                if self.clamp and not k.endswith('_latent'):
                    images[k] = torch.clamp(images[k], -1.0, 1.0)
        if self.save_images:
            self.save_local(images, batch_idx)
        return images

    def save_local(self, images: Dict[str, torch.Tensor], batch_idx: int) -> List[str]:
        """One PNG grid per image-valued log entry under saved_dir/model_name (diffusion_makeup.py:344-358; the
        reference's nrow = number of log entries is kept).  Latents (4 channels) are not images and are skipped."""
        from ..imageio import save_grid_png
        root = os.path.join(self.saved_dir, self.model_name)
        nrow = len(images)
        written = []
        for k, v in images.items():
            if not isinstance(v, torch.Tensor) or v.dim() != 4 or v.shape[1] not in (1, 3):
                continue
            written.append(save_grid_png(v, os.path.join(root, '{}_{:04}.png'.format(k, batch_idx)), nrow, self.rescale))
        return written
