"""CPU: pins the oracle: schedule KATs of App. B, parameter counts, committed golden vectors,
and the algebraic identities of the path.  The reference ships no vectors of its own (parity unpinned)."""
import os

import numpy as np
import pytest
import torch

from oracle import nets, sampler

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SMALL = dict(model_channels=64, channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2, context_dim=64,
             hint_widths=(16, 16, 32, 32, 32, 32, 64))


def test_param_counts_match_published_sizes():
    n = lambda d: sum(int(np.prod(s)) for s in d.values())
    assert abs(n(nets.param_spec(nets.FULL, 'unet')) / 1e6 - 859.52) < 0.01        # SD-1.5 UNet
    assert abs(n(nets.param_spec(nets.FULL, 'control')) / 1e6 - 361.28) < 0.01     # ControlNet (6-ch hint: +432 vs 3-ch)


def test_state_dict_names_follow_upstream_layout():
    spec = nets.full_param_spec(nets.FULL)
    for k in ('model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight',
              'model.diffusion_model.output_blocks.2.1.conv.weight', 'model.diffusion_model.output_blocks.5.2.conv.weight',
              'model.diffusion_model.out.2.weight', 'control_model.input_hint_block.0.weight', 'control_model.zero_convs.11.0.bias',
              'control_model.middle_block_out.0.weight', 'model.diffusion_model.input_blocks.3.0.op.weight',
              'model.diffusion_model.middle_block.1.proj_out.weight', 'model.diffusion_model.input_blocks.4.0.skip_connection.weight'):
        assert k in spec, k
    assert spec['control_model.input_hint_block.0.weight'] == (16, 6, 3, 3)       # reference runs/train.py:60-62
    assert spec['model.diffusion_model.output_blocks.0.0.in_layers.2.weight'] == (1280, 2560, 3, 3)


def test_schedule_known_answers():
    """SURVEY.md App. B (derived from yaml :4-8 with the upstream formulae)."""
    s = sampler.Schedule()
    ac = s.alphas_cumprod64
    assert abs(ac[0] - 0.99915) < 1e-12
    assert abs(ac[1] - 0.9982960278384514) < 1e-12
    assert abs(ac[999] - 0.004660098513077238) < 1e-12
    assert abs(1 / np.sqrt(ac[999]) - 14.648813544520891) < 1e-9
    assert abs(np.sqrt(1 / ac[999] - 1) - 14.614641229333639) < 1e-9
    s.make_ddim(50)
    assert list(s.ddim_timesteps[:3]) == [1, 21, 41] and list(s.ddim_timesteps[-2:]) == [961, 981]
    assert abs(float(s.ddim_alphas[0]) - 0.99829603) < 1e-7 and abs(float(s.ddim_alphas[49]) - 0.00577550) < 1e-7
    assert abs(float(s.ddim_alphas_prev[0]) - 0.99915) < 1e-7 and abs(float(s.ddim_alphas_prev[49]) - 0.00728173) < 1e-7
    assert abs(float(s.ddim_sqrt_one_minus_alphas[49]) - 0.99710807) < 1e-7
    assert float(s.ddim_sigmas.abs().max()) == 0.0
    s.make_ddim(20)
    assert list(s.ddim_timesteps[:3]) == [1, 51, 101] and list(s.ddim_timesteps[-2:]) == [901, 951]
    assert abs(float(s.ddim_alphas[19]) - 0.00815500) < 1e-7 and abs(float(s.ddim_alphas_prev[19]) - 0.01400490) < 1e-7
    assert abs(float(s.ddim_sqrt_one_minus_alphas[19]) - 0.99591415) < 1e-7


def test_schedule_matches_golden_file():
    g = np.load(os.path.join(GOLD, 'schedule.npz'))
    s = sampler.Schedule().make_ddim(50)
    assert np.array_equal(s.ddim_timesteps, g['ts50'])
    assert np.array_equal(s.ddim_alphas.numpy(), g['a50']) and np.array_equal(s.ddim_alphas_prev.numpy(), g['ap50'])
    assert np.array_equal(s.alphas_cumprod.numpy(), g['alphas_cumprod'])


@pytest.fixture(scope='module')
def small():
    g = np.load(os.path.join(GOLD, 'small_eps.npz'))
    cfg = nets.NetConfig(**SMALL)
    sd = nets.init_state_dict(cfg, seed=int(g['seed_weights']))
    return cfg, sd, {k: torch.from_numpy(g[k]) for k in g.files if k != 'seed_weights'}


def test_oracle_reproduces_golden(small):
    cfg, sd, g = small
    torch.set_num_threads(4)
    cond = {'c_crossattn': [g['ctx']], 'c_concat': [g['hint']]}
    eps = sampler.apply_model(sd, cfg, g['x'], g['t'], cond)
    assert torch.allclose(eps, g['eps'], rtol=1e-4, atol=1e-5)
    x5 = sampler.sample(sampler.make_eps_fn(sd, cfg), sampler.Schedule(), g['x'], cond, 5)
    assert torch.allclose(x5, g['x5'], rtol=1e-3, atol=1e-4)


def test_identity_zero_zero_convs_equals_no_control(small):
    """all zero-convs = 0  =>  eps == UNet(control=None)   (makeup_diffuse.py:160-162 vs :164-168)."""
    cfg, sd, g = small
    sd0 = dict(sd)
    for k in sd0:
        if 'zero_convs' in k or 'middle_block_out' in k:
            sd0[k] = torch.zeros_like(sd0[k])
    a = sampler.apply_model(sd0, cfg, g['x'], g['t'], {'c_crossattn': [g['ctx']], 'c_concat': [g['hint']]})
    b = sampler.apply_model(sd0, cfg, g['x'], g['t'], {'c_crossattn': [g['ctx']], 'c_concat': None})
    assert torch.allclose(a, b, rtol=0, atol=1e-6)
    # and with random zero-convs the control branch really matters (fixtures are not degenerate, finding 8)
    assert (g['eps'] - g['eps_noctl']).abs().max() > 1e-2


def test_identity_only_mid_control_ignores_skip_residuals(small):
    cfg, sd, g = small
    ctrl = nets.control_model(sd, cfg, g['x'], g['hint'], g['t'], g['ctx'])
    junk = [torch.randn_like(c) for c in ctrl[:-1]] + [ctrl[-1]]
    a = nets.diffusion_model(sd, cfg, g['x'], g['t'], g['ctx'], control=list(ctrl), only_mid_control=True)
    b = nets.diffusion_model(sd, cfg, g['x'], g['t'], g['ctx'], control=junk, only_mid_control=True)
    assert torch.equal(a, b)
    assert torch.allclose(a, g['eps_mid'], rtol=1e-4, atol=1e-5)
    assert len(ctrl) == len(nets.encoder_spec(cfg)) + 1


def test_identity_cfg_scale_one_is_cond_only(small):
    cfg, sd, g = small
    sch = sampler.Schedule().make_ddim(5)
    fn = sampler.make_eps_fn(sd, cfg)
    cond = {'c_crossattn': [g['ctx']], 'c_concat': [g['hint']]}
    uc = {'c_crossattn': [g['uctx']], 'c_concat': [g['hint']]}
    t = torch.full((2,), int(sch.ddim_timesteps[4]), dtype=torch.long)
    a, _ = sampler.denoising_step(fn, sch, g['x'], cond, t, 4, 1.0, uc)
    b, _ = sampler.denoising_step(fn, sch, g['x'], cond, t, 4)
    assert torch.equal(a, b)


def test_identity_per_sample_independence(small):
    """B=2 == two B=1 evaluations: what makes batch sharding exact."""
    cfg, sd, g = small
    for i in range(2):
        one = sampler.apply_model(sd, cfg, g['x'][i:i + 1], g['t'][i:i + 1],
                                  {'c_crossattn': [g['ctx'][i:i + 1]], 'c_concat': [g['hint'][i:i + 1]]})
        assert torch.allclose(one, g['eps'][i:i + 1], rtol=1e-4, atol=1e-5)


def test_cfg_batches_uncond_first():
    u = {'c_crossattn': [torch.zeros(1, 2, 3)], 'c_concat': [torch.zeros(1, 1)]}
    c = {'c_crossattn': [torch.ones(1, 2, 3)], 'c_concat': [torch.ones(1, 1)]}
    m = sampler.cat_cond(u, c)
    assert m['c_crossattn'][0][0].sum() == 0 and m['c_crossattn'][0][1].sum() == 6     # cddim.py:25-31


def test_fixture_weights_expose_a_miswired_norm_parameter(small):
    """with gamma = 1 / beta = 0 everywhere a swapped norm1/norm2/norm3 or in_layers.0/out_layers.0, or a dropped
    beta, is invisible to every golden file.  The fixture initialiser draws them at random; each such mistake must now move
    the golden eps far outside the GPU tests' 2e-2 budget."""
    cfg, sd, g = small
    cond = {'c_crossattn': [g['ctx']], 'c_concat': [g['hint']]}
    norms = [k for k in sd if k.endswith('.weight') and sd[k].dim() == 1 and ('norm' in k or 'in_layers.0' in k or 'out_layers.0' in k or k.endswith('out.0.weight'))]
    assert len(norms) > 40 and all(float((sd[k] - 1).abs().max()) > 0.05 for k in norms)
    assert all(float(sd[k[:-6] + 'bias'].abs().max()) > 0.05 for k in norms)

    def rel_after(edit):
        sd2 = dict(sd)
        edit(sd2)
        out = sampler.apply_model(sd2, cfg, g['x'], g['t'], cond)
        return float((out - g['eps']).norm() / g['eps'].norm())

    def swap(a, b):
        def f(d):
            for s in ('.weight', '.bias'):
                d[a + s], d[b + s] = d[b + s], d[a + s]
        return f
    T = 'model.diffusion_model.input_blocks.1.1.transformer_blocks.0'
    R = 'model.diffusion_model.output_blocks.4.0'
    assert rel_after(swap(T + '.norm1', T + '.norm2')) > 0.05
    assert rel_after(swap(T + '.norm2', T + '.norm3')) > 0.05
    assert sd[R + '.in_layers.0.weight'].shape != sd[R + '.out_layers.0.weight'].shape        # (concat input): swap a same-width pair instead
    R2 = 'control_model.input_blocks.2.0'
    assert rel_after(swap(R2 + '.in_layers.0', R2 + '.out_layers.0')) > 0.05
    def drop_beta(d):
        d[R + '.out_layers.0.bias'] = torch.zeros_like(d[R + '.out_layers.0.bias'])
    assert rel_after(drop_beta) > 0.02
