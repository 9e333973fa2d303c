"""CPU: the host-side mirror of the reference interface (sampler API, config surface, schedule tables, sharding
arithmetic) and the C ABI's symbol table.  No compute is run through libmkd here (no GPU)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from makeupdiffuse_amd import dist as mdist
from makeupdiffuse_amd import lib as mlib
from makeupdiffuse_amd.config import create_model, load_yaml
from makeupdiffuse_amd.schedule import DDIMSchedule
from oracle import nets, sampler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(model_channels=64, channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2, context_dim=64,
             hint_widths=(16, 16, 32, 32, 32, 32, 64))


# ---- C ABI ---------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    """include/mkd.h <-> libmkd.so <-> the ctypes table: same set of entry points."""
    from makeupdiffuse_amd import build as mbuild
    path = mbuild.build(verbose=False)
    so = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, 'include', 'mkd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(mkd_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(mlib.SIGNATURES), declared ^ set(mlib.SIGNATURES)
    for name in declared:
        getattr(so, name)
    so.mkd_abi_version.restype = ctypes.c_int
    assert so.mkd_abi_version() == mlib.ABI_VERSION


def test_no_gpu_is_a_loud_error_not_a_fallback():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from makeupdiffuse_amd.engine import MkdEngine, NetConfig
    with pytest.raises(mlib.MkdError):
        MkdEngine(NetConfig(**SMALL))
    lib = mlib.load()
    cfg = NetConfig(**SMALL).to_c()
    h = ctypes.c_void_p()
    assert lib.mkd_ctx_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b'no CPU path' in lib.mkd_last_error() or b'HIP' in lib.mkd_last_error()


def test_system_scope_signal_switch_is_refused_loudly():
    """with ROC_SYSTEM_SCOPE_SIGNAL=0 a replay of the step graph never completed (profiles/exp_r3_rt_env2.txt: the
    cross-queue join barriers wait on completion signals that need system scope).  mkd_ctx_create refuses the setting - before it
    looks for a device, so this runs on the CPU; nothing is launched."""
    import subprocess
    code = ("import ctypes, sys; sys.path.insert(0, %r); from makeupdiffuse_amd import lib as m; from makeupdiffuse_amd.engine import NetConfig; "
            "l = m.load(); c = NetConfig(model_channels=64, channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2, context_dim=64, "
            "hint_widths=(16, 16, 32, 32, 32, 32, 64)).to_c(); h = ctypes.c_void_p(); rc = l.mkd_ctx_create(ctypes.byref(c), ctypes.byref(h)); "
            "print(rc, l.mkd_last_error().decode())" % ROOT)
    for val, refused in (('0', True), ('1', False)):
        env = dict(os.environ, ROC_SYSTEM_SCOPE_SIGNAL=val)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120).stdout
        rc = int(out.split()[0])
        assert (rc == -4 and 'ROC_SYSTEM_SCOPE_SIGNAL' in out) == refused, out


def test_product_does_not_import_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for base in ('makeupdiffuse_amd', 'diffmk', 'runs'):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith('.py'):
                    assert not pat.search(open(os.path.join(dp, f)).read()), f'{dp}/{f} imports oracle/'


# ---- schedule / config surface -----------------------------------------------------------------------------------
def test_product_schedule_equals_oracle_schedule():
    for S in (20, 50):
        a = DDIMSchedule().make_ddim(S)
        b = sampler.Schedule().make_ddim(S)
        assert np.array_equal(a.ddim_timesteps, b.ddim_timesteps)
        for n in ('ddim_alphas', 'ddim_alphas_prev', 'ddim_sigmas', 'ddim_sqrt_one_minus_alphas', 'alphas_cumprod',
                  'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod'):
            assert torch.equal(getattr(a, n), getattr(b, n)), n


def test_yaml_surface_own_and_reference_style():
    m = create_model(os.path.join(ROOT, 'diffmodels', 'test_diffusion_makeup.yaml'))
    assert type(m).__name__ == 'TestDoubleControlModel'
    assert m.ddim_steps == 50 and m.unconditional_guidance_scale == 9 and m.ddim_eta == 0.0
    assert m.net_config.model_channels == 320 and tuple(m.net_config.channel_mult) == (1, 2, 4, 4)
    assert m.net_config.hint_channels == 6 and m.net_config.context_dim == 768 and m.net_config.n_control == 13
    assert m.parameterization == 'eps' and abs(m.scale_factor - 0.18215) < 1e-9 and m.only_mid_control is False
    assert len(m.control_scales) == 13 and m.num_timesteps == 1000
    cfg = load_yaml(os.path.join(ROOT, 'diffmodels', 'base_diffusion_makeup.yaml'))
    assert cfg['model']['target'] == 'diffmk.diffusion_makeup.BaseDoubleControlModel'
    with pytest.raises(mlib.MkdError):          # no engine until .cuda(): loud
        m.apply_model(torch.zeros(1, 4, 8, 8), torch.zeros(1, dtype=torch.long),
                      {'c_crossattn': [torch.zeros(1, 77, 768)], 'c_concat': [torch.zeros(1, 6, 64, 64)]})


def test_get_input_builds_source_first_hint():
    m = create_model(os.path.join(ROOT, 'diffmodels', 'test_diffusion_makeup.yaml'))
    batch = {'src_img': torch.full((2, 3, 16, 16), 0.25), 'ref_img': torch.full((2, 3, 16, 16), 0.75),
             'txt_emb': torch.randn(2, 77, 768)}
    _, c = m.get_input(batch, 'jpg')
    hint = c['c_concat'][0]
    assert hint.shape == (2, 6, 16, 16) and hint.dtype == torch.float32
    assert torch.all(hint[:, :3] == 0.25) and torch.all(hint[:, 3:] == 0.75)       # makeup_diffuse.py:56 (src, ref)
    assert c['c_crossattn'][0].shape == (2, 77, 768)
    with pytest.raises(NotImplementedError):
        m.get_input({'src_img': batch['src_img'], 'ref_img': batch['ref_img'], 'txt': ['makeup transfer'] * 2}, 'jpg')


# ---- sampler host logic against the oracle, with a stand-in eps model ----------------------------------------------
class _OracleBackedModel:
    """Duck-typed `model` the samplers need whose apply_model is the CPU oracle: lets the host
    logic of DDIMSampler / MKDDIMSampler be checked on CPU.  Test scaffolding only."""

    def __init__(self):
        self.cfg = nets.NetConfig(**SMALL)
        self.sd = nets.init_state_dict(self.cfg, seed=11)
        sch = DDIMSchedule()
        self.num_timesteps = sch.num_timesteps
        self.parameterization = 'eps'
        self.device = torch.device('cpu')
        for n in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_one_minus_alphas_cumprod'):
            setattr(self, n, getattr(sch, n))
        self.calls = []

    def apply_model(self, x, t, c):
        self.calls.append(x.shape[0])
        return sampler.apply_model(self.sd, self.cfg, x, t, c)


@pytest.fixture(scope='module')
def gold():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'small_eps.npz'))
    return {k: torch.from_numpy(g[k]) for k in g.files if k != 'seed_weights'}


def test_mkddim_reconstruct_and_step_match_oracle(gold):
    from diffmk.cddim import MKDDIMSampler
    torch.set_num_threads(4)
    model = _OracleBackedModel()
    s = MKDDIMSampler(model)
    s.make_schedule(ddim_num_steps=5, verbose=False)
    cond = {'c_crossattn': [gold['ctx']], 'c_concat': [gold['hint']]}
    out = s.reconstruct(gold['x'], cond, t_start=5)
    assert torch.allclose(out, gold['x5'], rtol=1e-4, atol=1e-5)
    seen = []
    out3 = s.reconstruct(gold['x'], cond, t_start=3, callback=seen.append)
    assert seen == [0, 1, 2]
    sch = sampler.Schedule().make_ddim(5)
    ref3 = sampler.reconstruct(sampler.make_eps_fn(model.sd, model.cfg), sch, gold['x'], cond, 3)
    assert torch.allclose(out3, ref3, rtol=1e-4, atol=1e-5)
    # CFG: ONE apply_model call on 2B, uncond first (cddim.py:18-40)
    model.calls.clear()
    uc = {'c_crossattn': [gold['uctx']], 'c_concat': [gold['hint']]}
    outc = s.reconstruct(gold['x'], cond, t_start=5, unconditional_guidance_scale=9.0, unconditional_conditioning=uc)
    assert model.calls == [4] * 5
    assert torch.allclose(outc, gold['x5_cfg'], rtol=1e-3, atol=1e-4)
    t = torch.full((2,), int(s.ddim_timesteps[4]), dtype=torch.long)
    xp, x0 = s.denoising_step(gold['x'], cond, t, index=4)
    rp, r0 = sampler.denoising_step(sampler.make_eps_fn(model.sd, model.cfg), sch, gold['x'], cond, t, 4)
    assert torch.allclose(xp, rp, rtol=1e-5, atol=1e-6) and torch.allclose(x0, r0, rtol=1e-5, atol=1e-6)


def test_sampler_rejects_what_the_reference_path_never_uses(gold):
    from diffmk.cddim import MKDDIMSampler
    model = _OracleBackedModel()
    s = MKDDIMSampler(model)
    s.make_schedule(ddim_num_steps=5, verbose=False)
    cond = {'c_crossattn': [gold['ctx']], 'c_concat': [gold['hint']]}
    t = torch.full((2,), 801, dtype=torch.long)
    with pytest.raises(NotImplementedError):
        s.denoising_step(gold['x'], cond, t, index=4, dynamic_threshold=0.5)        # cddim.py:70-71
    with pytest.raises(NotImplementedError):
        s.denoising_step(gold['x'], cond, t, index=4, quantize_denoised=True)
    with pytest.raises(AssertionError):                                              # cddim.py:21 dict/dict check
        s.denoising_step(gold['x'], cond, t, index=4, unconditional_guidance_scale=2.0,
                         unconditional_conditioning=[gold['uctx']])


def test_ddim_sampler_full_loop_from_x_T(gold):
    from makeupdiffuse_amd.ddim import DDIMSampler
    torch.set_num_threads(4)
    model = _OracleBackedModel()
    s = DDIMSampler(model)
    cond = {'c_crossattn': [gold['ctx']], 'c_concat': [gold['hint']]}
    out, inter = s.sample(5, 2, (4, 8, 8), cond, verbose=False, eta=0.0, x_T=gold['x'])
    assert torch.allclose(out, gold['x5'], rtol=1e-4, atol=1e-5)
    assert len(inter['x_inter']) >= 2


# ---- model-class host logic with a recording stand-in engine (no GPU) ---------------------------------------------------
class _RecordingEngine:
    """Stands for MkdEngine on the CPU: records prepare() calls, answers eps/decode with tagged tensors."""
    vae_cfg = object()

    def __init__(self):
        self.prepared = []

    def prepare(self, hint, ctx, **kw):
        self.prepared.append((None if hint is None else float(hint.sum()), float(ctx.sum())))

    def eps(self, x, t):
        return torch.full_like(x, self.prepared[-1][1])

    def decode(self, z, scale_factor):
        return z[:, :3] * 4.0 / scale_factor * 0.18215


def test_conditioning_cache_is_keyed_on_the_tensor_objects_not_their_addresses():
    """a (data_ptr, version, shape) key matches a NEW tensor that the allocator placed where a freed one was;
    the cache must hold the tensors and compare identity + version."""
    from makeupdiffuse_amd.diffmk.makeup_diffuse import BaseMakeUpDiffuse, _Held
    m = create_model(os.path.join(ROOT, 'diffmodels', 'test_diffusion_makeup.yaml'))
    eng = _RecordingEngine()
    m._require_engine = lambda: eng
    x, t = torch.zeros(1, 4, 8, 8), torch.zeros(1, dtype=torch.long)

    def cond(v):
        return {'c_crossattn': [torch.full((1, 77, 768), float(v))], 'c_concat': [torch.full((1, 6, 64, 64), float(v))]}
    c1 = cond(1)
    m.apply_model(x, t, c1); m.apply_model(x, t, c1)
    assert len(eng.prepared) == 1                                  # same objects, unchanged: bound once
    c1['c_crossattn'][0].add_(1.0)                                  # in-place edit bumps ._version
    m.apply_model(x, t, c1)
    assert len(eng.prepared) == 2
    # a different tensor of the same shape re-binds even when it sits at the SAME address (storage shared on purpose here)
    alias = {'c_crossattn': [c1['c_crossattn'][0].view(1, 77, 768)], 'c_concat': [c1['c_concat'][0].view(1, 6, 64, 64)]}
    assert alias['c_concat'][0].data_ptr() == c1['c_concat'][0].data_ptr() and alias['c_concat'][0] is not c1['c_concat'][0]
    out = m.apply_model(x, t, alias)
    assert len(eng.prepared) == 3 and float(out[0, 0, 0, 0]) == eng.prepared[-1][1]
    # the held tensors cannot be freed (hence their addresses cannot be recycled) while the entry is live
    import weakref
    c2 = cond(5)
    ref = weakref.ref(c2['c_concat'][0])
    m.apply_model(x, t, c2)
    del c2
    assert ref() is not None
    m.reset_conditioning_cache()
    assert ref() is None
    # CFG merge cache: same rule
    u, c = cond(0), cond(3)
    a = m.cfg_conditioning(u, c); b = m.cfg_conditioning(u, c)
    assert a is b and a['c_concat'][0].shape[0] == 2 and float(a['c_crossattn'][0][0].sum()) == 0.0     # uncond first
    c_new = cond(4)
    assert m.cfg_conditioning(u, c_new) is not a
    h = _Held((u['c_concat'][0], None), ('k',))
    assert h.matches((u['c_concat'][0], None), ('k',)) and not h.matches((u['c_concat'][0], None), ('j',))
    assert not h.matches((u['c_concat'][0].clone(), None), ('k',))


def test_generate_image_and_decode_latent_code_follow_the_reference():
    """makeup_diffuse.py:172-177 (decode -> clamp(-1, 1) -> optional (x + 1) / 2) and makeups.py:260-262 (unclamped decode)."""
    m = create_model(os.path.join(ROOT, 'diffmodels', 'test_diffusion_makeup.yaml'))
    eng = _RecordingEngine()
    m._require_engine = lambda: eng
    z = torch.linspace(-1, 1, 2 * 4 * 2 * 2).reshape(2, 4, 2, 2)
    raw = m.decode_latent_code(z)
    assert torch.equal(raw, m.decode_first_stage(z)) and float(raw.abs().max()) > 1.0
    a = m.generate_image(z)
    assert torch.equal(a, raw.clamp(-1, 1))
    b = m.generate_image(z, format=True)
    assert torch.equal(b, (raw.clamp(-1, 1) + 1.0) / 2.0) and float(b.min()) >= 0.0 and float(b.max()) <= 1.0


def test_makeups_generate_image_call_shape_with_oracle_backed_model(gold):
    """reference diffmk/makeups.py:44-47,119-127 on the CPU with the oracle as eps model / decoder: reconstruct(x_latent=inv,
    cond=c, t_start=iter_finetune) -> decode_latent_code -> (x + 1) / 2 clamped to [0, 1]; c['c_concat'] chosen by c_type."""
    from diffmk.makeups import BaseModel
    from oracle import vae
    torch.set_num_threads(4)
    ocfg = nets.NetConfig(**SMALL)
    sd = nets.init_state_dict(ocfg, seed=11)
    vcfg = vae.VaeConfig(z_channels=4, embed_dim=4, ch=32, ch_mult=(1, 2), num_res_blocks=1, out_ch=3)
    vsd = vae.init_state_dict(vcfg, seed=5)
    ctrl = dict(SMALL, hint_channels=6, num_res_blocks=2, in_channels=4, use_spatial_transformer=True, legacy=False)
    m = BaseModel(control_stage_config={'params': ctrl}, unet_config={'params': dict(ctrl, out_channels=4)}, iter_finetune=5)
    m.apply_model = lambda x, t, c, *a, **k: sampler.apply_model(sd, ocfg, x, t, c)     # stand-ins for the device engine
    m.decode_first_stage = lambda z: vae.decode_first_stage(vsd, vcfg, z)
    m.sample_loop_fast = None
    m.ddim_step = None
    m.on_fit_start()
    assert m.ddim_sampler.ddim_timesteps.shape[0] == 5
    c = dict(c_crossattn=[gold['ctx']], c_concat_r=[gold['hint']], c_concat_s=[gold['hint'].flip(1)])
    img = m.generate_image(gold['x'], c, c_type='c_concat_r')
    assert c['c_concat'] is c['c_concat_r']
    want = ((vae.decode_first_stage(vsd, vcfg, gold['x5']) + 1.0) / 2.0).clamp(0, 1)
    assert torch.allclose(img, want, rtol=1e-4, atol=1e-5) and float(img.min()) >= 0 and float(img.max()) <= 1
    img2 = m.generate_image(gold['x'], c, c_replace=c['c_concat_s'])
    assert c['c_concat'] is c['c_concat_s'] and not torch.allclose(img2, img)
    with pytest.raises(NotImplementedError):
        m.shared_step({})


# ---- batch sharding ------------------------------------------------------------------------------------------------
def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 64, 352):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        mdist.shard_range(8, 2, 2)


# ---- generated code hygiene (no GPU needed: hipcc cross-compiles) ------------------------------------------------------
@pytest.mark.timeout(600)
def test_hot_kernels_use_no_scratch(tmp_path):
    """A by-value argument block that escapes to the stack silently costs 1.3-1.9x (seen once): every MFMA kernel must
    compile to private_segment_fixed_size 0 and without VGPR spills."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    csrc = os.path.join(ROOT, 'makeupdiffuse_amd', 'csrc')
    for src in ('kernels_gemm.hip', 'kernels_conv.hip', 'kernels_attn.hip', 'kernels_norm.hip', 'kernels_tfm.hip'):
        out = tmp_path / (src + '.s')
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-I', csrc, '-S',
                               '--cuda-device-only', os.path.join(csrc, src), '-o', str(out)], stderr=subprocess.DEVNULL)
        txt = out.read_text()
        sizes = re.findall(r'^\s+\.private_segment_fixed_size:\s+(\d+)', txt, flags=re.M)
        spills = re.findall(r'^\s+\.vgpr_spill_count:\s+(\d+)', txt, flags=re.M)
        assert sizes and all(int(v) == 0 for v in sizes), (src, sizes)
        assert all(int(v) == 0 for v in spills), (src, spills)


# ---- round 3: the earlier variant's batch contract, update_schedule, multi-entry cond lists -----------------------------------
def _ctrl_params():
    return dict(SMALL, hint_channels=6, num_res_blocks=2, in_channels=4, use_spatial_transformer=True, legacy=False)


def test_makeup_double_control_model_takes_channels_last_batches():
    """reference diffmk/makeup_controlnet.py:137-167: control images arrive [b, h, w, c] under control_src_key / control_key, are
    rearranged 'b h w c -> b c h w', and the cond dict is the one the newer class builds from CHW src_img / ref_img
    (diffmk/makeup_diffuse.py:42-57): 6-channel hint, SOURCE FIRST, c_crossattn = [txt]."""
    from diffmk.makeup_controlnet import BaseModel, MakeupDoubleControlModel
    from diffmk.makeup_diffuse import BaseMakeUpDiffuse
    ctrl = _ctrl_params()
    g = torch.Generator().manual_seed(2)
    src = torch.rand(3, 16, 24, 3, generator=g); ref = torch.rand(3, 16, 24, 3, generator=g); txt = torch.randn(3, 77, 64, generator=g)
    for cls in (MakeupDoubleControlModel, BaseModel):
        m = cls('source', control_stage_config={'params': ctrl}, unet_config={'params': dict(ctrl, out_channels=4)}, control_key='hint')
        assert m.control_src_key == 'source' and m.control_key == 'hint'
        z, c = m.get_input({'source': src, 'hint': ref, 'txt_emb': txt, 'jpg': torch.zeros(3, 16, 24, 3)}, 'jpg')
        hint = c['c_concat'][0]
        assert z is None and sorted(c) == ['c_concat', 'c_crossattn'] and len(c['c_concat']) == 1 and len(c['c_crossattn']) == 1
        assert hint.shape == (3, 6, 16, 24) and hint.dtype == torch.float32 and hint.is_contiguous()
        assert torch.equal(hint[:, :3], src.permute(0, 3, 1, 2)) and torch.equal(hint[:, 3:], ref.permute(0, 3, 1, 2))      # :167 (src, ref)
        assert torch.equal(c['c_crossattn'][0], txt)
        # the newer class on the SAME images delivered channels-first gives the same conditioning
        n = BaseMakeUpDiffuse(control_stage_config={'params': ctrl}, unet_config={'params': dict(ctrl, out_channels=4)})
        _, c2 = n.get_input({'src_img': src.permute(0, 3, 1, 2), 'ref_img': ref.permute(0, 3, 1, 2), 'txt_emb': txt}, 'jpg')
        assert torch.equal(c2['c_concat'][0], hint) and torch.equal(c2['c_crossattn'][0], c['c_crossattn'][0])
        _, c3 = m.get_input({'source': src, 'hint': ref, 'txt_emb': txt}, 'jpg', bs=2)                                       # :148-156 bs
        assert c3['c_concat'][0].shape[0] == 2 and c3['c_crossattn'][0].shape[0] == 2
        with pytest.raises(ValueError):
            m.get_input({'source': src[0], 'hint': ref, 'txt_emb': txt}, 'jpg')
    d = MakeupDoubleControlModel('source', control_stage_config={'params': ctrl}, unet_config={'params': dict(ctrl, out_channels=4)})
    assert d.get_origin_img_input({'k': src}, 'k').shape == (3, 3, 16, 24)                    # :106-114 rearranges by default
    assert d.get_origin_img_input({'k': src}, 'k', need_rearrange=False).shape == (3, 16, 24, 3)


def test_update_schedule_reregisters_the_linear_schedule_with_t0_steps():
    """reference diffmk/makeups.py:40-47: on_fit_start -> update_schedule -> register_schedule(beta_schedule='linear',
    timesteps=t0, same linear_start / linear_end) -> MKDDIMSampler.make_schedule(ddim_num_steps=iter_finetune).  Known answers
    from the closed form (SURVEY.md App. B formulae at 600 steps): betas = linspace(sqrt(.00085), sqrt(.012), 600)^2."""
    from diffmk.makeups import BaseModel
    ctrl = _ctrl_params()
    m = BaseModel(control_stage_config={'params': ctrl}, unet_config={'params': dict(ctrl, out_channels=4)}, t0=600, iter_finetune=40)
    assert m.num_timesteps == 1000 and m.alphas_cumprod.shape[0] == 1000
    m.on_fit_start()
    assert m.num_timesteps == 600 and m.alphas_cumprod.shape[0] == 600 and m.ddim_sampler.ddpm_num_timesteps == 600
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 600, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    assert abs(float(m.betas[0]) - 0.00085) < 1e-9 and abs(float(m.betas[-1]) - 0.012) < 1e-8
    assert np.allclose(m.alphas_cumprod.numpy(), ac.astype(np.float32), rtol=0, atol=1e-7)
    assert abs(float(m.alphas_cumprod[0]) - 0.99915) < 1e-6 and abs(float(m.alphas_cumprod[-1]) - ac[-1]) < 1e-7
    assert ac[-1] > 0.0046600985 * 3                         # a 600-step chain ends far less noisy than the 1000-step one (App. B value)
    ts = m.ddim_sampler.ddim_timesteps
    assert ts.shape[0] == 40 and int(ts[0]) == 1 and int(ts[1]) == 16 and int(ts[-1]) == 586      # arange(0, 600, 600 // 40) + 1
    assert np.allclose(m.ddim_sampler.ddim_alphas.numpy(), ac[ts].astype(np.float32), atol=1e-7)
    assert abs(float(m.ddim_sampler.ddim_alphas_prev[0]) - ac[0]) < 1e-7 and abs(float(m.ddim_sampler.ddim_alphas_prev[1]) - ac[1]) < 1e-7
    assert float(m.ddim_sampler.ddim_sigmas.abs().max()) == 0.0
    m.t0 = 1000
    m.update_schedule()
    assert m.num_timesteps == 1000 and torch.equal(m.alphas_cumprod, DDIMSchedule().alphas_cumprod)
    with pytest.raises(NotImplementedError):
        m.register_schedule(given_betas=betas)


def test_multi_entry_cond_lists_are_concatenated_once_per_distinct_set():
    """apply_model cats c_crossattn / c_concat lists (reference diffmk/makeup_diffuse.py:159,165); a cache keyed on the
    CAT RESULT never matches, so every DDIM step would re-run mkd_prepare.  The key holds the list elements."""
    m = create_model(os.path.join(ROOT, 'diffmodels', 'test_diffusion_makeup.yaml'))
    eng = _RecordingEngine()
    m._require_engine = lambda: eng
    x, t = torch.zeros(1, 4, 8, 8), torch.zeros(1, dtype=torch.long)
    a, b = torch.full((1, 40, 768), 1.0), torch.full((1, 37, 768), 2.0)
    s, r = torch.full((1, 3, 64, 64), 0.25), torch.full((1, 3, 64, 64), 0.75)
    cond = {'c_crossattn': [a, b], 'c_concat': [s, r]}
    for _ in range(3):
        m.apply_model(x, t, cond)
    assert len(eng.prepared) == 1
    m.apply_model(x, t, {'c_crossattn': [a, b], 'c_concat': [s, r]})          # new lists, same elements
    assert len(eng.prepared) == 1
    m.apply_model(x, t, {'c_crossattn': [a, b], 'c_concat': [r, s]})          # another order is another hint
    assert len(eng.prepared) == 2
    b.add_(1.0)
    m.apply_model(x, t, {'c_crossattn': [a, b], 'c_concat': [r, s]})
    assert len(eng.prepared) == 3
    import weakref
    ref = weakref.ref(s)
    del s, cond
    m.reset_conditioning_cache()
    m.apply_model(x, t, {'c_crossattn': [a], 'c_concat': [r]})
    assert len(eng.prepared) == 4
