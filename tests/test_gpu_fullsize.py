"""BASELINE-size (1.22 G parameter nets, batch 8, 256x256 and 512x512) checks through properties that need no oracle run at
that size (SURVEY.md §8c identities), plus one oracle comparison at 512x512 (B = 1).  Weights: seeded, generated on the device."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from makeupdiffuse_amd.engine import MkdEngine, NetConfig  # noqa: E402
from makeupdiffuse_amd.schedule import DDIMSchedule  # noqa: E402

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.fixture(scope='module')
def full():
    eng = MkdEngine(NetConfig())
    eng.init_random(0, norm_jitter=0.2)
    g = torch.Generator().manual_seed(7)
    B = 8
    inp = dict(x=torch.randn(B, 4, 32, 32, generator=g), hint=torch.rand(B, 6, 256, 256, generator=g),
               ctx=torch.randn(B, 77, 768, generator=g), t=torch.tensor([981, 901, 701, 501, 401, 301, 101, 1]))
    yield eng, inp
    eng.close()


def test_full_size_batch8_identities(full):
    """config 2 (batch 8, 256x256): repeatability; per-sample independence (batch 8 == the same samples evaluated as 8 x batch 1
    and as 3 + 5, i.e. other lane splits and other tile tables, within the bf16 budget); control_scales = 0 == c_concat None
    BIT FOR BIT (the combine epilogue computes 0 * (zero_conv + b) + skip); only_mid_control leaves the 12 skip residuals out."""
    eng, I = full
    eng.prepare(I['hint'], I['ctx'])
    a = eng.eps(I['x'], I['t'])
    assert torch.isfinite(a).all() and torch.equal(a, eng.eps(I['x'], I['t']))
    parts = []
    for lo, hi in ((0, 3), (3, 8)):
        eng.prepare(I['hint'][lo:hi], I['ctx'][lo:hi])
        parts.append(eng.eps(I['x'][lo:hi], I['t'][lo:hi]))
    assert rel(torch.cat(parts), a) <= 2e-2
    singles = []
    for i in range(8):
        eng.prepare(I['hint'][i:i + 1], I['ctx'][i:i + 1])
        singles.append(eng.eps(I['x'][i:i + 1], I['t'][i:i + 1]))
    assert rel(torch.cat(singles), a) <= 2e-2
    eng.prepare(None, I['ctx'], latent_hw=(32, 32))
    noctl = eng.eps(I['x'], I['t'])
    eng.prepare(I['hint'], I['ctx'], control_scales=[0.0] * 13)
    assert torch.equal(eng.eps(I['x'], I['t']), noctl)
    assert rel(noctl, a) > 5e-2                                    # the control branch does matter with these weights
    eng.prepare(I['hint'], I['ctx'], only_mid_control=True)
    mid = eng.eps(I['x'], I['t'])
    eng.prepare(I['hint'], I['ctx'], control_scales=[0.0] * 12 + [1.0])
    assert torch.equal(eng.eps(I['x'], I['t']), mid)               # only the middle residual is injected either way


def test_full_size_fused_transformer_tail_equals_the_seven_launch_plan(full):
    """the d = 320 transformer blocks' row-local tail as ONE kernel (csrc/kernels_tfm.hip; the ops of the two net
    calls of /root/reference/diffmk/makeup_diffuse.py:164-168 after the self-attention product).  Same folded / merged weights as the
    7-launch plan, other accumulation order and LayerNorm 3 folded instead of a bf16 LayerNorm pass: eps agrees within the bf16 budget
    of two plans of the same nets at batch 8 (256x256: M = 8192 rows, decoder lanes 4096) and at 512x512 (32768 rows); forced on at batch 1 (1024
    rows, below the shape policy's threshold) it still equals the unfused plan; 10-step latents, graph replay == eager."""
    eng, I = full
    import os
    def engine(mode):
        old = os.environ.get('MKD_TFM_TAIL')
        os.environ['MKD_TFM_TAIL'] = str(mode)
        try:
            e = MkdEngine(NetConfig())
        finally:
            if old is None: os.environ.pop('MKD_TFM_TAIL')
            else: os.environ['MKD_TFM_TAIL'] = old
        e.init_random(0, norm_jitter=0.2)
        return e
    off, on = engine(0), engine(1)
    try:
        for e in (eng, off, on):
            e.prepare(I['hint'], I['ctx'])
        a_def, a_off, a_on = eng.eps(I['x'], I['t']), off.eps(I['x'], I['t']), on.eps(I['x'], I['t'])
        assert torch.equal(a_def, a_on)                      # batch 8 is inside the default policy
        assert off.eps_launches() - on.eps_launches() == 80, (off.eps_launches(), on.eps_launches())      # 10 fused blocks x (7 -> 1 tail, 4 -> 2 head)
        r8 = rel(a_on, a_off)
        sch = DDIMSchedule().make_ddim(10)
        args = (sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas)
        l_on, l_off = on.sample(I['x'], *args, use_graph=True), off.sample(I['x'], *args, use_graph=True)
        assert torch.equal(l_on, on.sample(I['x'], *args, use_graph=False))
        rl = rel(l_on, l_off)
        for e in (off, on):
            e.prepare(I['hint'][:1], I['ctx'][:1])
        r1 = rel(on.eps(I['x'][:1], I['t'][:1]), off.eps(I['x'][:1], I['t'][:1]))
        assert off.eps_launches() - on.eps_launches() == 42      # batch 1: no decoder lanes, 7 blocks x ((6 - 1) + (3 - 2)): LayerNorms 1 and 3 are on the fly there
        g = torch.Generator().manual_seed(11)
        x = torch.randn(2, 4, 64, 64, generator=g); hint = torch.rand(2, 6, 512, 512, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
        t = torch.tensor([901, 101])
        for e in (off, on):
            e.prepare(hint, ctx)
        r512 = rel(on.eps(x, t), off.eps(x, t))
        print(f'fused tail vs 7 launches: eps batch 8 rel-L2 {r8:.3e}, batch 1 {r1:.3e}, 512x512 batch 2 {r512:.3e}; 10-step latents {rl:.3e}')
        # measured on MI355X: eps 1.27e-2 / 1.14e-2 / 1.28e-2, 10-step latents 3.2e-3 - the size of the difference between any two
        # bf16 plans of these nets (batch 8 vs 3 + 5: same budget above); against the fp32 oracle both plans sit at 1.48e-2 / 1.49e-2
        # (tests/test_gpu_engine.py::test_full_size_eps_vs_oracle)
        assert max(r8, r1, r512) <= 2e-2 and rl <= 1e-2
        # the head of the same blocks (GroupNorm apply + proj_in + LayerNorm 1 . q|k|v as one launch behind a statistics launch) is its
        # own switch: off, the 10 blocks run GroupNorm, proj_in, LayerNorm, q|k|v again (measured 8.6e-3 between the two)
        n_on = None
        for hd in (1, 0):
            on.set_option('tfm_head', hd)
            on.prepare(I['hint'], I['ctx'])
            a = on.eps(I['x'], I['t'])
            if hd: assert torch.equal(a, a_on); n_on = on.eps_launches()
            else:
                rh = rel(a_on, a)
                print(f'fused head vs 4 launches: eps batch 8 rel-L2 {rh:.3e}')
                assert on.eps_launches() - n_on == 20 and rh <= 2e-2
        # ... and the ResBlocks' 1x1 skip_connection folded into conv2's K loop where conv2 runs on the gather kernel (the decoder
        # lanes' shapes): off, the skip GEMMs (and their split-K reduces) are launches of their own again
        on.set_option('tfm_head', 1)
        on.set_option('skip_fold', 0)
        on.prepare(I['hint'], I['ctx'])
        a = on.eps(I['x'], I['t'])
        rs = rel(a_on, a)
        print(f'folded skip vs separate skip GEMMs: eps batch 8 rel-L2 {rs:.3e}, launches {n_on} vs {on.eps_launches()}')
        assert on.eps_launches() - n_on >= 20 and rs <= 2e-2
        on.set_option('skip_fold', 1)
    finally:
        off.close(); on.close()


def test_full_size_cfg_and_loop_properties(full):
    """50-step loop at batch 8: finite, repeatable, graph replay == step-by-step launch; guidance scale 1 with an uncond batch
    == the cond-only loop (cddim.py:15-16 vs :18-40); CFG batches uncond first."""
    eng, I = full
    sch = DDIMSchedule().make_ddim(50)
    args = (sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas)
    eng.prepare(I['hint'], I['ctx'])
    a = eng.sample(I['x'], *args, use_graph=False)
    b = eng.sample(I['x'], *args, use_graph=True)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    g = torch.Generator().manual_seed(9)
    uctx = torch.randn(8, 77, 768, generator=g)
    eng.prepare(torch.cat([I['hint'], I['hint']]), torch.cat([uctx, I['ctx']]))
    c = eng.sample(I['x'], *args, cfg_scale=1.0000001, use_graph=True)     # CFG path, u + s (c - u) with s ~ 1
    cos = torch.nn.functional.cosine_similarity(c.flatten().float().cpu(), a.flatten().float().cpu(), dim=0).item()
    assert cos >= 0.99, cos                                                 # 50 steps of bf16 drift between two batch shapes
    d = eng.sample(I['x'], *args, cfg_scale=9.0, use_graph=True)
    assert torch.isfinite(d).all() and rel(d, a) > 1e-2


def test_full_size_512_batch8_properties(full):
    """BASELINE config 4 (batch 8, 512x512 -> 64x64 latents: the tuned-table entries of that geometry, 4096-token attention):
    repeatable, batch 8 == 3 + 5 (other lane splits / tile choices) within the bf16 budget, hipGraph replay == eager launches over
    the metric's 50 steps."""
    eng, _ = full
    g = torch.Generator().manual_seed(11)
    B = 8
    x = torch.randn(B, 4, 64, 64, generator=g); hint = torch.rand(B, 6, 512, 512, generator=g)
    ctx = torch.randn(B, 77, 768, generator=g); t = torch.tensor([981, 901, 701, 501, 401, 301, 101, 1])
    eng.prepare(hint, ctx)
    a = eng.eps(x, t)
    assert torch.isfinite(a).all() and torch.equal(a, eng.eps(x, t))
    sch = DDIMSchedule().make_ddim(50)          # the metric's 50 steps: 10 five-step graphs against 50 x ~720 eager launches
    args = (sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas)
    e = eng.sample(x, *args, use_graph=False)
    assert torch.isfinite(e).all() and torch.equal(e, eng.sample(x, *args, use_graph=True))
    parts = []
    for lo, hi in ((0, 3), (3, 8)):
        eng.prepare(hint[lo:hi], ctx[lo:hi])
        parts.append(eng.eps(x[lo:hi], t[lo:hi]))
    r = rel(torch.cat(parts), a)
    print(f'512x512 batch 8 vs 3 + 5: rel-L2 {r:.3e}')
    assert r <= 2e-2


def test_full_size_interpolation_batch44_properties(full):
    """BASELINE config 5, one GPU's share (4 sources x 11 alpha = 44 images per step, build-defined blend of the two cached hint
    embeddings): repeatable, 44 == 22 + 22, hipGraph replay == eager over 50 steps, alpha = 0 / 1 rows == the single-reference path
    bit for bit."""
    eng, _ = full
    g = torch.Generator().manual_seed(13)
    n, k = 4, 11
    src = torch.rand(n, 3, 256, 256, generator=g); r1 = torch.rand(n, 3, 256, 256, generator=g); r2 = torch.rand(n, 3, 256, 256, generator=g)
    rep = lambda v: v.repeat_interleave(k, 0)
    h1, h2 = rep(torch.cat([src, r1], 1)), rep(torch.cat([src, r2], 1))
    ctx = rep(torch.randn(n, 77, 768, generator=g)); x = rep(torch.randn(n, 4, 32, 32, generator=g))
    alpha = torch.linspace(0, 1, k).repeat(n)
    t = torch.full((n * k,), 601)
    eng.prepare(h1, ctx, hint2=h2, alpha=alpha)
    a = eng.eps(x, t)
    assert torch.isfinite(a).all() and torch.equal(a, eng.eps(x, t))
    sch = DDIMSchedule().make_ddim(50)
    args = (sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas)
    e = eng.sample(x, *args, use_graph=False)
    assert torch.isfinite(e).all() and torch.equal(e, eng.sample(x, *args, use_graph=True))
    parts = []
    for lo, hi in ((0, 22), (22, 44)):
        eng.prepare(h1[lo:hi], ctx[lo:hi], hint2=h2[lo:hi], alpha=alpha[lo:hi])
        parts.append(eng.eps(x[lo:hi], t[lo:hi]))
    r = rel(torch.cat(parts), a)
    print(f'interpolation batch 44 vs 22 + 22: rel-L2 {r:.3e}')
    assert r <= 2e-2
    eng.prepare(h1, ctx)
    only1 = eng.eps(x, t)
    eng.prepare(h2, ctx)
    only2 = eng.eps(x, t)
    assert torch.equal(a[0::k], only1[0::k]) and torch.equal(a[k - 1::k], only2[k - 1::k])
    assert rel(a[5::k], only1[5::k]) > 1e-3


@pytest.mark.timeout(1200)
def test_full_size_512_eps_vs_oracle():
    """config 4 geometry (512x512 -> 64x64 latent, 4096-token self-attention in the 8-wave kernel), B = 1, vs the fp32 oracle."""
    from oracle import nets, sampler
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = nets.FULL
    sd = nets.init_state_dict(cfg, seed=0)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(1, 4, 64, 64, generator=gen); hint = torch.rand(1, 6, 512, 512, generator=gen)
    ctx = torch.randn(1, 77, 768, generator=gen); t = torch.tensor([301])
    ref = sampler.apply_model(sd, cfg, x, t, {'c_crossattn': [ctx], 'c_concat': [hint]})
    eng = MkdEngine(NetConfig())
    eng.load_state_dict(sd)
    del sd
    eng.prepare(hint, ctx)
    out = eng.eps(x, t)
    r = rel(out, ref)
    cos = torch.nn.functional.cosine_similarity(out.flatten().float().cpu(), ref.flatten(), dim=0).item()
    print(f'512x512 eps: rel-L2 {r:.4e} cos {cos:.6f}')
    assert torch.isfinite(out).all() and r <= 2e-2 and cos >= 0.9995, (r, cos)
    # 543.40 GMAC per sample per eval at 64x64 latents, minus the cached hint block (7.47) and cross-attention
    # K/V projections (2.16): 533.77 executed
    assert abs(eng.eps_flops() / 2e9 - 533.77) < 0.2, eng.eps_flops() / 2e9
    eng.close()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('steps', [20, 50])
def test_full_size_cfg9_trajectory_and_image_vs_oracle(steps):
    """The loop the reference runs, end to end at full size (B = 1, 256x256, eta 0, guidance scale 9 with the unconditional batch
    first and the same hint: diffusion_makeup.py:308-309,391-410): 20 DDIM steps (BASELINE config 1) and the reference's default 50.
    The whole trajectory of the 1.22 G-parameter nets and the decoded image vs the fp32 CPU oracle.  SURVEY.md §8c states the budget
    for a bf16 trajectory: cosine >= 0.99, PSNR >= 30 dB.  2 x steps oracle evaluations (0.5-1.5 minutes of CPU)."""
    import time
    from oracle import nets, sampler, vae as ovae
    from makeupdiffuse_amd.engine import VaeConfig
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = nets.FULL
    sd = nets.init_state_dict(cfg, seed=0)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 32, 32, generator=gen); hint = torch.rand(1, 6, 256, 256, generator=gen)
    ctx = torch.randn(1, 77, 768, generator=gen); uctx = torch.randn(1, 77, 768, generator=gen)
    scale = 9.0
    t0 = time.time()
    ref = sampler.sample(sampler.make_eps_fn(sd, cfg), sampler.Schedule(), x, {'c_crossattn': [ctx], 'c_concat': [hint]}, steps,
                         unconditional_guidance_scale=scale, unconditional_conditioning={'c_crossattn': [uctx], 'c_concat': [hint]})
    t_oracle = time.time() - t0
    vcfg = ovae.FULL
    vsd = ovae.init_state_dict(vcfg, seed=0)
    eng = MkdEngine(NetConfig())
    eng.configure_vae(VaeConfig())
    eng.load_state_dict({**sd, **vsd})
    del sd
    sch = DDIMSchedule().make_ddim(steps)
    eng.prepare(torch.cat([hint, hint]), torch.cat([uctx, ctx]))
    out = eng.sample(x, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, cfg_scale=scale,
                     use_graph=True)
    cos = torch.nn.functional.cosine_similarity(out.flatten().float().cpu(), ref.flatten(), dim=0).item()
    r = rel(out, ref)
    img_ref = ovae.decode_first_stage(vsd, vcfg, ref)
    img = eng.decode(out).float().cpu()
    peak = float(img_ref.max() - img_ref.min())
    psnr = 10.0 * torch.log10(torch.tensor(peak * peak) / ((img - img_ref) ** 2).mean()).item()
    cos_img = torch.nn.functional.cosine_similarity(img.flatten(), img_ref.flatten(), dim=0).item()
    print(f'full-size {steps}-step CFG-9 trajectory: latent rel-L2 {r:.4e} cos {cos:.6f}; decoded image PSNR {psnr:.1f} dB (peak-to-peak {peak:.2f}) '
          f'cos {cos_img:.6f}; oracle {t_oracle:.0f} s')
    assert torch.isfinite(out).all() and cos >= 0.99 and psnr >= 30.0, (cos, psnr)
    eng.close()


def _traj_stats(out, ref):
    out = out.float().cpu(); ref = ref.float().cpu()
    cos = torch.nn.functional.cosine_similarity(out.flatten(), ref.flatten(), dim=0).item()
    return rel(out, ref), cos


@pytest.mark.timeout(1500)
def test_full_size_512_ten_step_trajectory_vs_oracle():
    """config 4 geometry as a TRAJECTORY: B = 1, 512x512 (64x64 latent, 4096-token self-attention), 10 DDIM steps from x_T, eta 0, no
    guidance, the 1.22 G-parameter nets, graph replay - against oracle/sampler.sample on the same weights / latents (10 oracle
    evaluations at 543 GMAC: about a minute of CPU).  Limits are ~3x the drift measured on MI355X (printed)."""
    import time
    from oracle import nets, sampler
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = nets.FULL
    sd = nets.init_state_dict(cfg, seed=0)
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(1, 4, 64, 64, generator=gen); hint = torch.rand(1, 6, 512, 512, generator=gen); ctx = torch.randn(1, 77, 768, generator=gen)
    steps = 10
    t0 = time.time()
    ref = sampler.sample(sampler.make_eps_fn(sd, cfg), sampler.Schedule(), x, {'c_crossattn': [ctx], 'c_concat': [hint]}, steps)
    t_oracle = time.time() - t0
    eng = MkdEngine(NetConfig())
    eng.load_state_dict(sd)
    del sd
    sch = DDIMSchedule().make_ddim(steps)
    eng.prepare(hint, ctx)
    out = eng.sample(x, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=True)
    r, cos = _traj_stats(out, ref)
    print(f'512x512 10-step trajectory (B = 1): latent rel-L2 {r:.4e} cos {cos:.6f}; oracle {t_oracle:.0f} s')
    assert torch.isfinite(out).all() and r <= 1.2e-2 and cos >= 0.9999, (r, cos)          # measured on MI355X: 3.9e-3 / 0.999992
    assert torch.equal(out, eng.sample(x, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas, use_graph=False))
    eng.close()


@pytest.mark.timeout(1500)
def test_full_size_interpolation_ten_step_trajectory_vs_oracle():
    """config 5 as a TRAJECTORY at full size: one source, two references, alpha = 0.4 (build-defined blend of the two cached hint
    embeddings; parity unpinned by construction - the reference only shows a figure, README.md:23-25), 10 DDIM
    steps, against the oracle's restatement of the same definition.  Also alpha 0 / 1 trajectories == the single-reference loops
    bit for bit, and the second reference does move the latent."""
    import time
    from oracle import nets, sampler
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = nets.FULL
    sd = nets.init_state_dict(cfg, seed=0)
    gen = torch.Generator().manual_seed(22)
    x = torch.randn(1, 4, 32, 32, generator=gen); src = torch.rand(1, 3, 256, 256, generator=gen)
    r1 = torch.rand(1, 3, 256, 256, generator=gen); r2 = torch.rand(1, 3, 256, 256, generator=gen); ctx = torch.randn(1, 77, 768, generator=gen)
    h1, h2 = torch.cat([src, r1], 1), torch.cat([src, r2], 1)
    alpha = torch.tensor([0.4])
    steps = 10
    t0 = time.time()
    ref = sampler.sample(sampler.make_eps_fn(sd, cfg), sampler.Schedule(), x,
                         {'c_crossattn': [ctx], 'c_concat': [h1], 'c_concat2': [h2], 'interp_alpha': alpha}, steps)
    t_oracle = time.time() - t0
    eng = MkdEngine(NetConfig())
    eng.load_state_dict(sd)
    del sd
    sch = DDIMSchedule().make_ddim(steps)
    args = (sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas)
    eng.prepare(h1, ctx, hint2=h2, alpha=alpha)
    out = eng.sample(x, *args, use_graph=True)
    r, cos = _traj_stats(out, ref)
    print(f'full-size interpolation 10-step trajectory (alpha 0.4): latent rel-L2 {r:.4e} cos {cos:.6f}; oracle {t_oracle:.0f} s')
    assert torch.isfinite(out).all() and r <= 1.2e-2 and cos >= 0.9999, (r, cos)          # measured on MI355X: 3.9e-3 / 0.999992
    ends = []
    for a, h in ((0.0, h1), (1.0, h2)):
        eng.prepare(h1, ctx, hint2=h2, alpha=torch.tensor([a]))
        e = eng.sample(x, *args, use_graph=True)
        eng.prepare(h, ctx)
        assert torch.equal(e, eng.sample(x, *args, use_graph=True)), f'alpha = {a} must reduce to the single-reference loop'
        ends.append(e)
    d01 = rel(ends[1], ends[0])
    print(f'alpha 0 vs alpha 1 latents: rel-L2 {d01:.3e}; alpha 0.4 vs alpha 0: {rel(out, ends[0]):.3e}, vs alpha 1: {rel(out, ends[1]):.3e}')
    assert d01 > 1e-3 and not torch.equal(out, ends[0]) and not torch.equal(out, ends[1]), 'the second reference must move the trajectory'
    eng.close()


def test_full_size_grouped_encoder_equals_two_chains_bit_for_bit(monkeypatch):
    """MKD_ENC_GROUP at BASELINE size (batch 8, 256x256, the 1.22 G-parameter nets): every tile configuration the tuned table picks
    for the encoder phase - LDS-staged convolutions, split-K with its reduce, slab-fed GroupNorm, in-block K splits, on-the-fly
    LayerNorm - run as 2-problem grouped launches gives the eps and the 5-step latent of the two-chain plan BIT FOR BIT, after
    NaN-poisoning, with 140+ launches fewer per step.  Third engine: the default plan without the XCD-aware tile order of the weight-heavy
    layers (MKD_XCD_AUTO_RATIO=0) - a permutation of the tiles, same bits."""
    g = torch.Generator().manual_seed(7)
    B = 8
    x = torch.randn(B, 4, 32, 32, generator=g); hint = torch.rand(B, 6, 256, 256, generator=g)
    ctx = torch.randn(B, 77, 768, generator=g); t = torch.tensor([981, 901, 701, 501, 401, 301, 101, 1])
    sch = DDIMSchedule().make_ddim(5)
    args = (sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas)
    res = {}
    from makeupdiffuse_amd import lib as _mlib
    grouped = bool(_mlib.load().mkd_grouped_launches_available())      # (an experiment build since round 4: -DMKD_PAIR_N=2)
    for group in (0, 1, 2):
        if group == 1 and not grouped:
            continue
        monkeypatch.setenv('MKD_ENC_GROUP', str(group & 1))
        monkeypatch.setenv('MKD_XCD_AUTO_RATIO', '0' if group == 2 else '2')          # (2: two chains, launch order everywhere)
        eng = MkdEngine(NetConfig())
        eng.init_random(0, norm_jitter=0.2)
        eng.prepare(hint, ctx)
        eng.debug_poison()
        e = eng.eps(x, t)
        eng.debug_poison()
        lat = eng.sample(x, *args, use_graph=True)
        assert torch.isfinite(e).all() and torch.isfinite(lat).all()
        res[group] = (e, lat, eng.step_launches())
        eng.close()
    if grouped:
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1]), 'the XCD-aware tile order of the weight-heavy layers changed a bit'
    if grouped:
        print(f'launches per step at batch 8: two chains {res[0][2]}, grouped {res[1][2]}')
        # (round 4: the d = 320 blocks' row-local tail is ONE launch in both plans - kernels_tfm.hip - so pairing saves 130, not 146)
        assert res[1][2] <= res[0][2] - 120


@pytest.mark.timeout(1500)
def test_full_size_batch8_eps_vs_oracle_at_256_and_512():
    """BASELINE configs 2 and 4 AT batch 8 against the fp32 CPU oracle on identical weights - one eps evaluation of
    8 samples at 256x256 (M = 8192 rows: the fused transformer tail, decoder lanes of 4) and at 512x512 (32768 rows, 4096-token
    self-attention with MFMA row sums, full-chip GroupNorm), per-sample timesteps.  SURVEY.md section 8c budget for one evaluation:
    rel-L2 <= 2e-2, cosine >= 0.9995; measured on MI355X (printed): 256x256 1.42e-2 / 0.99990 (worst sample 1.49e-2), 512x512 1.40e-2 /
    0.99990 (1.51e-2) - the same distance as the batch-1 evaluations (1.48e-2), i.e. the bf16 storage format, not the batch plan."""
    import time
    from oracle import nets, sampler
    import torch.nn.functional as F
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = nets.FULL
    sd = nets.init_state_dict(cfg, seed=0)
    eng = MkdEngine(NetConfig())
    eng.load_state_dict(sd)
    t = torch.tensor([981, 901, 701, 501, 401, 301, 101, 1])
    worst = {}
    for res in (256, 512):
        gen = torch.Generator().manual_seed(30 + res)
        h = res // 8
        x = torch.randn(8, 4, h, h, generator=gen); hint = torch.rand(8, 6, res, res, generator=gen); ctx = torch.randn(8, 77, 768, generator=gen)
        t0 = time.time()
        ref = sampler.apply_model(sd, cfg, x, t, {'c_crossattn': [ctx], 'c_concat': [hint]})
        t_or = time.time() - t0
        eng.prepare(hint, ctx)
        out = eng.eps(x, t).float().cpu()
        assert torch.isfinite(out).all()
        r = rel(out, ref); c = F.cosine_similarity(out.flatten(), ref.flatten(), dim=0).item()
        per = [rel(out[i], ref[i]) for i in range(8)]
        print(f'batch 8 eps vs oracle at {res}x{res}: rel-L2 {r:.4e} cos {c:.6f}; per sample max {max(per):.4e}; oracle {t_or:.0f} s; launches {eng.eps_launches()}')
        worst[res] = (r, c, max(per))
        assert r <= 2e-2 and c >= 0.9995 and max(per) <= 2.5e-2, (res, r, c, per)
    eng.close()
