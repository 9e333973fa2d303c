"""-m gpu: the fused row-local transformer tail (csrc/kernels_tfm.hip) against a plain torch fp32 chain of the same ops
(UPSTREAM cldm BasicTransformerBlock after the self-attention product + SpatialTransformer.proj_out; the two
net calls it serves: /root/reference/diffmk/makeup_diffuse.py:164-168), called through the C ABI."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import DEV, L, P, bf, rel_l2, sync
from makeupdiffuse_amd import lib as mlib

pytestmark = pytest.mark.gpu


def block_weights(d, seed, norm_jitter=0.2):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    w = {
        'to_out1_w': r(d, d) / math.sqrt(d), 'to_out1_b': 0.1 * r(d),
        'norm2_g': 1 + norm_jitter * r(d), 'norm2_b': norm_jitter * r(d),
        'to_q2_w': r(d, d) / math.sqrt(d),
        'to_out2_w': r(d, d) / math.sqrt(d), 'to_out2_b': 0.1 * r(d),
        'norm3_g': 1 + norm_jitter * r(d), 'norm3_b': norm_jitter * r(d),
        'ff0_w': r(8 * d, d) / math.sqrt(d), 'ff0_b': 0.1 * r(8 * d),
        'ff2_w': r(d, 4 * d) / math.sqrt(4 * d), 'ff2_b': 0.1 * r(d),
        'proj_out_w': r(d, d) / math.sqrt(d), 'proj_out_b': 0.1 * r(d),
    }
    return w


ORDER = ['to_out1_w', 'to_out1_b', 'norm2_g', 'norm2_b', 'to_q2_w', 'to_out2_w', 'to_out2_b', 'norm3_g', 'norm3_b', 'ff0_w', 'ff0_b',
         'ff2_w', 'ff2_b', 'proj_out_w', 'proj_out_b']


def torch_chain(w, a1, h0, xin, kv, B, T, Tk, d, heads=8):
    """fp32 reference on the bf16-rounded INPUTS (a1, h0, x_in, K | V); weights fp32 as given."""
    dh = d // heads
    h1 = a1 @ w['to_out1_w'].T + w['to_out1_b'] + h0
    q = F.layer_norm(h1, (d,), w['norm2_g'], w['norm2_b'], 1e-5) @ w['to_q2_w'].T
    k = kv[:, :d].reshape(B, Tk, heads, dh).permute(0, 2, 1, 3)
    v = kv[:, d:].reshape(B, Tk, heads, dh).permute(0, 2, 1, 3)
    qh = q.reshape(B, T, heads, dh).permute(0, 2, 1, 3)
    att = torch.softmax(qh @ k.transpose(-1, -2) * dh ** -0.5, dim=-1) @ v
    a2 = att.permute(0, 2, 1, 3).reshape(B * T, d)
    h2 = a2 @ w['to_out2_w'].T + w['to_out2_b'] + h1
    u = F.layer_norm(h2, (d,), w['norm3_g'], w['norm3_b'], 1e-5) @ w['ff0_w'].T + w['ff0_b']
    gg = u[:, :4 * d] * F.gelu(u[:, 4 * d:])
    h3 = gg @ w['ff2_w'].T + w['ff2_b'] + h2
    return h3 @ w['proj_out_w'].T + w['proj_out_b'] + xin


def make_handle(w, d):
    dev = {k: w[k].to(DEV).float().contiguous() for k in ORDER}
    h = C.c_void_p()
    mlib.check(L().mkd_tfm_tail_create(d, *[P(dev[k]) for k in ORDER], C.byref(h)), 'mkd_tfm_tail_create')
    sync()
    return h


@pytest.mark.parametrize('B,T,Tk,pad', [(1, 64, 77, 0), (2, 256, 77, 0), (3, 192, 77, 64), (2, 1024, 77, 0), (1, 128, 16, 0), (2, 64, 80, 8), (1, 320, 1, 0)])
def test_tfm_tail_matches_the_torch_fp32_chain(B, T, Tk, pad):
    d = 320
    w = block_weights(d, seed=B * 1000 + T + Tk)
    g = torch.Generator().manual_seed(7 + T)
    M = B * T
    ld = d + pad
    # activations of the scale the block sees (h0 / x_in: O(1) residual streams with a per-channel offset; a1: attention output)
    a1 = bf(torch.randn(M, ld, generator=g))
    h0 = bf(torch.randn(M, ld, generator=g) + 0.5 * torch.randn(1, ld, generator=g))
    xin = bf(torch.randn(M, ld, generator=g))
    kv = bf(torch.randn(B * Tk, 2 * d, generator=g))
    out = torch.full((M, ld), float('nan'), device=DEV, dtype=torch.bfloat16)
    h = make_handle(w, d)
    try:
        mlib.check(L().mkd_tfm_tail_set_context(h, P(kv), 2 * d, B, Tk, None), 'set_context')
        mlib.check(L().mkd_tfm_tail_run(h, P(a1), ld, P(h0), ld, P(xin), ld, P(out), ld, M, T, None), 'run')
        sync()
        first = out.clone()
        out.fill_(float('nan'))
        mlib.check(L().mkd_tfm_tail_run(h, P(a1), ld, P(h0), ld, P(xin), ld, P(out), ld, M, T, None), 'run')
        sync()
    finally:
        L().mkd_tfm_tail_destroy(h)
    assert torch.equal(first[:, :d].view(torch.int16), out[:, :d].view(torch.int16)), 'not bit-repeatable'
    if pad:
        assert torch.isnan(out[:, d:].float()).all(), 'wrote outside its columns'
    ref = torch_chain(w, a1[:, :d].float().cpu(), h0[:, :d].float().cpu(), xin[:, :d].float().cpu(), kv.float().cpu(), B, T, Tk, d)
    got = out[:, :d].float().cpu()
    assert torch.isfinite(got).all()
    r = rel_l2(got, ref)
    mx = (got - ref).abs().max().item()
    print(f'tfm_tail B={B} T={T} Tk={Tk}: rel-L2 {r:.3e} max-abs {mx:.3e} (|ref|inf {ref.abs().max().item():.2f})')
    # five chained bf16 GEMMs with two LayerNorms between them: per-kernel budget 4e-3 each; measured ~3e-3 for the chain
    assert r <= 8e-3, f'rel-L2 {r:.3e}'
    assert mx <= ref.abs().max().item() * 2 ** -4


def test_tfm_tail_rejects_shapes_it_does_not_cover():
    d = 320
    w = block_weights(d, seed=1)
    h = make_handle(w, d)
    try:
        x = bf(torch.zeros(96, d))
        kv = bf(torch.zeros(77, 2 * d))
        assert L().mkd_tfm_tail_run(h, P(x), d, P(x), d, P(x), d, P(x), d, 96, 96, None) != 0      # no context yet
        mlib.check(L().mkd_tfm_tail_set_context(h, P(kv), 2 * d, 1, 77, None), 'set_context')
        assert L().mkd_tfm_tail_run(h, P(x), d, P(x), d, P(x), d, P(x), d, 96, 96, None) != 0      # T not a multiple of 64
        assert L().mkd_tfm_tail_set_context(h, P(kv), 2 * d, 1, 81, None) != 0                     # more than 80 keys
        hh = C.c_void_p()
        dev = {k: torch.zeros(1, device=DEV) for k in ORDER}
        assert L().mkd_tfm_tail_create(640, *[P(dev[k]) for k in ORDER], C.byref(hh)) != 0         # only d = 320 is built
    finally:
        L().mkd_tfm_tail_destroy(h)


@pytest.mark.parametrize('B,T,pad', [(1, 64, 0), (2, 1024, 0), (3, 256, 64), (2, 4096, 0)])
def test_tfm_head_matches_the_torch_fp32_chain(B, T, pad):
    """The head of the block as one kernel behind a GroupNorm statistics launch: GroupNorm(32, 1e-6) -> proj_in (1x1 conv = per-token
    linear) -> LayerNorm 1 -> to_q | to_k | to_v (no bias), against torch fp32 on the bf16-rounded input."""
    d = 320
    g = torch.Generator().manual_seed(B * 100 + T)
    r = lambda *s: torch.randn(*s, generator=g)
    w = {'gn_g': 1 + 0.2 * r(d), 'gn_b': 0.2 * r(d), 'pi_w': r(d, d) / math.sqrt(d), 'pi_b': 0.1 * r(d), 'n1_g': 1 + 0.2 * r(d), 'n1_b': 0.2 * r(d),
         'q_w': r(d, d) / math.sqrt(d), 'k_w': r(d, d) / math.sqrt(d), 'v_w': r(d, d) / math.sqrt(d)}
    order = ['gn_g', 'gn_b', 'pi_w', 'pi_b', 'n1_g', 'n1_b', 'q_w', 'k_w', 'v_w']
    dev = {k: w[k].to(DEV).float().contiguous() for k in order}
    ld = d + pad
    x = bf((r(B * T, ld) * (1 + 0.5 * r(1, ld)) + 0.3 * r(1, ld)))          # per-channel scale / offset: the group statistics matter
    h0 = torch.full((B * T, d), float('nan'), device=DEV, dtype=torch.bfloat16)
    qkv = torch.full((B * T, 3 * d), float('nan'), device=DEV, dtype=torch.bfloat16)
    h = C.c_void_p()
    mlib.check(L().mkd_tfm_head_create(d, *[P(dev[k]) for k in order], C.byref(h)), 'mkd_tfm_head_create')
    try:
        mlib.check(L().mkd_tfm_head_run(h, P(x), ld, 1e-6, P(h0), P(qkv), B, T, None), 'run')
        sync()
        first = (h0.clone(), qkv.clone())
        mlib.check(L().mkd_tfm_head_run(h, P(x), ld, 1e-6, P(h0), P(qkv), B, T, None), 'run')
        sync()
    finally:
        L().mkd_tfm_head_destroy(h)
    assert torch.equal(first[0].view(torch.int16), h0.view(torch.int16)) and torch.equal(first[1].view(torch.int16), qkv.view(torch.int16))
    xf = x[:, :d].float().cpu().view(B, T, d).permute(0, 2, 1)                    # [B, C, T]
    gn = F.group_norm(xf, 32, w['gn_g'], w['gn_b'], 1e-6).permute(0, 2, 1).reshape(B * T, d)
    h0_ref = gn @ w['pi_w'].T + w['pi_b']
    ln = F.layer_norm(h0_ref, (d,), w['n1_g'], w['n1_b'], 1e-5)
    qkv_ref = torch.cat([ln @ w['q_w'].T, ln @ w['k_w'].T, ln @ w['v_w'].T], 1)
    r0, r1 = rel_l2(h0, h0_ref), rel_l2(qkv, qkv_ref)
    print(f'tfm_head B={B} T={T}: h0 rel-L2 {r0:.3e}, qkv rel-L2 {r1:.3e}')
    assert torch.isfinite(h0.float()).all() and torch.isfinite(qkv.float()).all()
    assert r0 <= 4e-3 and r1 <= 8e-3
