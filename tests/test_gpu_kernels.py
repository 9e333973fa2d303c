"""-m gpu: every HIP kernel through the C ABI vs a plain torch fp32 reference of the same op
(inputs pre-rounded to bf16 so only accumulation order / output rounding differ)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, L, P, assert_close_bf16, bf, rel_l2, sync

pytestmark = pytest.mark.gpu


def gemm(A, W, bias=None, rowbias=None, rpb=1, R=None, scale=1.0, act=0, out_f32=False, conv=None, splitk=0,
         ldc=None, lda=None):
    lib = L()
    N, K = W.shape
    if conv is None:
        M = A.shape[0]
        lda_ = A.stride(0) if lda is None else lda
        cv = (0, 0, 0, 0, 0, 0, 0, 1, 0)
    else:
        B, Hin, Win, Cin, Hout, Wout, stride, up = conv
        M = B * Hout * Wout
        lda_ = Cin if lda is None else lda
        cv = (1, B, Hin, Win, Cin, Hout, Wout, stride, up)
    ldc_ = N if ldc is None else ldc
    out = torch.zeros((M, ldc_), device=DEV, dtype=torch.float32 if out_f32 else torch.bfloat16)
    rc = lib.mkd_gemm_bf16(P(A), lda_, P(W), K, P(bias), P(rowbias), 0 if rowbias is None else rowbias.stride(0), rpb,
                           P(R), 0 if R is None else R.stride(0), float(scale), act, P(out), ldc_, int(out_f32), M, N, K,
                           *cv, splitk, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    return out


@pytest.mark.parametrize('M,N,K,splitk', [(200, 320, 320, 0), (128, 1280, 1280, 0), (128, 1280, 1280, 5), (8, 1280, 320, 0),
                                          (77 * 2, 640, 768, 0), (1024, 64, 64, 1), (300, 192, 2560, 3), (4096, 320, 320, 1)])
def test_gemm_linear(M, N, K, splitk):
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)); W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g).to(DEV)
    out = gemm(A, W, bias=bias, splitk=splitk)
    ref = A.float() @ W.float().t() + bias
    assert_close_bf16(out, ref, what=f'gemm {M}x{N}x{K}')


@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 4, 5, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 41, 44, 45, 46, 47, 48, 49, 50])
@pytest.mark.parametrize('M,N,K,splitk', [(300, 320, 320, 1), (1000, 640, 1344, 1), (128, 1280, 2560, 4), (77, 64, 64, 1), (256, 1280, 1280, 1), (96, 320, 200, 1)])
def test_gemm_every_tile_config(cfg, M, N, K, splitk):
    """each gather-GEMM tile / pipeline-depth configuration, incl. K not a multiple of 64 and ragged M/N."""
    lib = L()
    g = torch.Generator().manual_seed(cfg * 7 + M + K)
    A = bf(torch.randn(M, K, generator=g)); W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g).to(DEV)
    lib.mkd_gemm_force_tile(cfg)
    try:
        out = gemm(A, W, bias=bias, splitk=splitk)
    finally:
        lib.mkd_gemm_force_tile(-1)
    assert_close_bf16(out, A.float() @ W.float().t() + bias, what=f'gemm cfg {cfg}')


def test_gemm_fused_geglu():
    g = torch.Generator().manual_seed(11)
    for (M, inner, K, splitk) in [(512, 1280, 320, 0), (200, 640, 1280, 2)]:
        A = bf(torch.randn(M, K, generator=g)); Wv = bf(torch.randn(inner, K, generator=g) / math.sqrt(K))
        Wg = bf(torch.randn(inner, K, generator=g) / math.sqrt(K))
        bv = torch.randn(inner, generator=g).to(DEV); bg = torch.randn(inner, generator=g).to(DEV)
        Wi = torch.stack([Wv, Wg], 1).reshape(2 * inner, K).contiguous()       # rows (v0, g0, v1, g1, ...)
        bi = torch.stack([bv, bg], 1).reshape(2 * inner).contiguous()
        lib = L()
        out = torch.zeros(M, inner, device=DEV, dtype=torch.bfloat16)
        rc = lib.mkd_gemm_bf16(P(A), K, P(Wi), K, P(bi), None, 0, 1, None, 0, 1.0, 2, P(out), inner, 0, M, 2 * inner, K,
                               0, 0, 0, 0, 0, 0, 0, 1, 0, splitk, None)
        assert rc == 0, lib.mkd_last_error()
        sync()
        ref = (A.float() @ Wv.float().t() + bv) * F.gelu(A.float() @ Wg.float().t() + bg)
        assert_close_bf16(out, ref, what='fused geglu')


@pytest.mark.parametrize('M,d,N2,cfg', [(300, 320, 960, -1), (128, 1280, 1280, -1), (2048, 640, 1920, -1), (512, 320, 320, 3),
                                         (100, 64, 128, 5), (8192, 320, 2560, -1), (512, 1280, 3840, -1)])
def test_gemm_fused_layernorm_chain(M, d, N2, cfg):
    """h = A.W1^T + b1 + R (producer GEMM emits partial row sums of the rounded h), then LN(h).W2^T + b2 through the
    folded-weight GEMM whose epilogue applies rstd*(acc - mu*rowsum(W2')) -- vs torch linear -> layer_norm -> linear.
    Rows get a large common offset so the mean-subtraction path is really exercised."""
    lib = L()
    g = torch.Generator().manual_seed(M + d + N2)
    A = bf(torch.randn(M, d, generator=g)); W1 = bf(torch.randn(d, d, generator=g) / math.sqrt(d))
    b1 = torch.randn(d, generator=g).to(DEV)
    R = bf(torch.randn(M, d, generator=g) + torch.randn(M, 1, generator=g) * 2.0)
    h = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
    stats = torch.full((64, M, 2), float('nan'), device=DEV)
    slots = C.c_int(0)
    rc = lib.mkd_gemm_rowstats_bf16(P(A), d, P(W1), d, P(b1), P(R), d, P(h), d, M, d, d, P(stats), 64, C.byref(slots), None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    href = A.float() @ W1.float().t() + b1 + R.float()
    assert_close_bf16(h, href, what='producer gemm')
    st = stats[:slots.value].sum(0)
    assert torch.isfinite(st).all()
    assert torch.allclose(st[:, 0], h.float().sum(1), rtol=1e-4, atol=1e-2), 'row sums'
    assert torch.allclose(st[:, 1], (h.float() ** 2).sum(1), rtol=1e-4, atol=1e-2), 'row sums of squares'
    W2 = (torch.randn(N2, d, generator=g) / math.sqrt(d)).to(DEV)
    gamma = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV); beta = (0.2 * torch.randn(d, generator=g)).to(DEV)
    b2 = torch.randn(N2, generator=g).to(DEV)
    Wf = torch.empty(N2, d, device=DEV, dtype=torch.bfloat16); s = torch.empty(N2, device=DEV); bfold = torch.empty(N2, device=DEV)
    assert lib.mkd_fold_layernorm(P(W2), P(gamma), P(beta), P(b2), N2, d, P(Wf), 0, 1, P(s), P(bfold), None) == 0
    out = torch.zeros(M, N2, device=DEV, dtype=torch.bfloat16)
    lib.mkd_gemm_force_tile(cfg)
    try:
        rc = lib.mkd_gemm_ln_bf16(P(h), d, P(Wf), d, P(bfold), P(s), P(stats), slots.value, 1e-5, 0, P(out), N2, M, N2, d, None)
        assert rc == 0, lib.mkd_last_error()
        sync()
    finally:
        lib.mkd_gemm_force_tile(-1)
    ref = F.linear(F.layer_norm(h.float(), (d,), gamma, beta, 1e-5), W2, b2)
    # the A operand is the raw row (bf16) instead of the normalised row rounded to bf16: same input precision, but
    # the rounding is relative to |h| not |h - mean| -> allow 2x the generic kernel budget
    assert_close_bf16(out, ref, rel=8e-3, what='fused layernorm gemm')


@pytest.mark.parametrize('cfg', [0, 1, 3, 5, 14, 19, 21, 24, 34, 36, 41, 44, 45, 46, 47, 48, 49, 50])
@pytest.mark.parametrize('M,rpb', [(300, 20), (300, 100), (520, 64), (96, 16)])
def test_gemm_row_bias_on_ragged_and_straddling_tiles(cfg, M, rpb):
    """bias + per-sample row bias + scale + residual (the straight-line epilogue when all rows of a wave lie in one sample, the general
    one otherwise): samples that end inside a tile / inside a wave, a ragged last tile whose trailing waves own no row at all."""
    lib = L()
    g = torch.Generator().manual_seed(cfg * 13 + M + rpb)
    N, K = 320, 192
    A = bf(torch.randn(M, K, generator=g)); W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g).to(DEV)
    nb = (M + rpb - 1) // rpb
    rowbias = torch.randn(nb, N, generator=g).to(DEV)          # EXACTLY nb rows: a read past the last sample leaves the allocation
    R = bf(torch.randn(M, N, generator=g))
    lib.mkd_gemm_force_tile(cfg)
    try:
        out = gemm(A, W, bias=bias, rowbias=rowbias, rpb=rpb, R=R, scale=0.75)
    finally:
        lib.mkd_gemm_force_tile(-1)
    ref = (A.float() @ W.float().t() + bias + rowbias.repeat_interleave(rpb, 0)[:M]) * 0.75 + R.float()
    assert_close_bf16(out, ref, what=f'row bias, cfg {cfg}, M {M}, rows per sample {rpb}')


def test_gemm_epilogue_variants():
    g = torch.Generator().manual_seed(7)
    M, N, K, rpb = 512, 640, 640, 128
    A = bf(torch.randn(M, K, generator=g)); W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g).to(DEV)
    rowbias = torch.randn(M // rpb, N + 64, generator=g).to(DEV)[:, 32:32 + N]   # strided view (ldrb > N)
    R = bf(torch.randn(M, N, generator=g))
    base = A.float() @ W.float().t() + bias
    out = gemm(A, W, bias=bias, rowbias=rowbias, rpb=rpb)
    assert_close_bf16(out, base + rowbias.repeat_interleave(rpb, 0), what='rowbias')
    out = gemm(A, W, bias=bias, R=R, scale=0.5)
    assert_close_bf16(out, base * 0.5 + R.float(), what='scale+residual')
    out = gemm(A, W, bias=bias, act=1)
    assert_close_bf16(out, F.silu(base), what='silu')
    out = gemm(A, W, bias=bias, out_f32=True)
    assert out.dtype == torch.float32
    assert_close_bf16(out, base, rel=1e-5 * 50, what='f32 out')
    # strided output (write into a channel slice of a wider buffer) must not touch the rest
    out = gemm(A, W, bias=bias, ldc=N + 128)
    assert_close_bf16(out[:, :N], base, what='ldc')
    assert (out[:, N:] == 0).all()


@pytest.mark.parametrize('cfg', [14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 41, 44, 45, 46, 47, 48, 49, 50])
@pytest.mark.parametrize('B,H,W_,Cin,Cout,stride,up', [(2, 16, 16, 64, 320, 1, 0), (1, 8, 8, 128, 160, 1, 1), (2, 12, 20, 32, 96, 2, 0)])
def test_gemm_conv3x3_160_wide_tiles(cfg, B, H, W_, Cin, Cout, stride, up):
    lib = L()
    g = torch.Generator().manual_seed(cfg + B * H + Cin + Cout)
    x = torch.randn(B, Cin, H, W_, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g).to(DEV)
    xb = bf(x); wbf = bf(w)
    xn = xb.permute(0, 2, 3, 1).contiguous()
    wp = torch.empty(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_pack_conv_weight(P(wbf.float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
    xr = F.interpolate(xb.float(), scale_factor=2, mode='nearest') if up else xb.float()
    ref = F.conv2d(xr, wbf.float(), bias, stride=stride, padding=1)
    Hout, Wout = ref.shape[2], ref.shape[3]
    lib.mkd_gemm_force_tile(cfg)
    try:
        out = gemm(xn, wp, bias=bias, conv=(B, H, W_, Cin, Hout, Wout, stride, up), lda=Cin, splitk=1)
    finally:
        lib.mkd_gemm_force_tile(-1)
    assert_close_bf16(out.float().view(B, Hout, Wout, Cout).permute(0, 3, 1, 2), ref, what=f'conv3x3 cfg {cfg}')


@pytest.mark.parametrize('B,H,W_,Cin,Cout,stride,up,pad_ld,splitk', [
    (2, 16, 16, 64, 128, 1, 0, 0, 0), (2, 16, 16, 64, 64, 2, 0, 0, 0), (1, 8, 8, 128, 128, 1, 1, 0, 0),
    (2, 12, 20, 16, 32, 1, 0, 0, 1), (3, 8, 8, 320, 320, 1, 0, 64, 0), (2, 4, 4, 1280, 1280, 1, 0, 0, 0),
    (2, 32, 32, 96, 256, 2, 0, 0, 0), (1, 4, 4, 2560, 1280, 1, 0, 0, 7)])
def test_gemm_conv3x3(B, H, W_, Cin, Cout, stride, up, pad_ld, splitk):
    lib = L()
    g = torch.Generator().manual_seed(B * H + Cin + Cout)
    x = torch.randn(B, Cin, H, W_, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g).to(DEV)
    xb = bf(x); wbf = bf(w)
    ld = Cin + pad_ld
    xn = torch.zeros(B, H, W_, ld, device=DEV, dtype=torch.bfloat16)
    xn[..., :Cin] = xb.permute(0, 2, 3, 1)
    if pad_ld:
        xn[..., Cin:] = 99.0          # poison: must never be read
    wp = torch.empty(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)
    wf = wbf.float().contiguous()
    assert lib.mkd_pack_conv_weight(P(wf), P(wp), Cout, Cin, 3, 3, None) == 0
    xr = xb.float()
    if up:
        xr = F.interpolate(xr, scale_factor=2, mode='nearest')
    ref = F.conv2d(xr, wbf.float(), bias, stride=stride, padding=1)
    Hout, Wout = ref.shape[2], ref.shape[3]
    out = gemm(xn, wp, bias=bias, conv=(B, H, W_, Cin, Hout, Wout, stride, up), lda=ld, splitk=splitk)
    out = out.float().view(B, Hout, Wout, Cout).permute(0, 3, 1, 2)
    assert_close_bf16(out, ref, what='conv3x3')


@pytest.mark.parametrize('cfg', [6, 7, 8, 9, 10, 11, 38, 39, 40, 42, 43])
@pytest.mark.parametrize('B,H,W_,Cin,Cout,pad_ld,splitk', [(2, 32, 32, 64, 128, 0, 1), (8, 4, 4, 1280, 1280, 0, 5), (3, 8, 8, 320, 320, 64, 1),
                                                          (2, 16, 16, 128, 64, 0, 2), (1, 32, 32, 320, 320, 0, 1), (8, 8, 8, 640, 1280, 0, 3),
                                                          (5, 4, 4, 128, 192, 0, 1), (2, 64, 64, 64, 64, 0, 1)])
def test_conv3x3_lds_staged_tiles(cfg, B, H, W_, Cin, Cout, pad_ld, splitk):
    """every LDS-staged 3x3 tile configuration (kernels_conv.hip) vs torch conv2d, incl. ragged image groups, padded
    pixel stride, split-K over channel chunks, fused bias + per-sample row-bias + residual."""
    lib = L()
    g = torch.Generator().manual_seed(cfg * 131 + B * H + Cin)
    x = torch.randn(B, Cin, H, W_, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g).to(DEV)
    rowbias = torch.randn(B, Cout, generator=g).to(DEV)
    xb = bf(x); wbf = bf(w)
    ld = Cin + pad_ld
    xn = torch.full((B, H, W_, ld), 99.0, device=DEV, dtype=torch.bfloat16)
    xn[..., :Cin] = xb.permute(0, 2, 3, 1)
    wp = torch.empty(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_pack_conv_weight(P(wbf.float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
    R = bf(torch.randn(B * H * W_, Cout, generator=g))
    ref = F.conv2d(xb.float(), wbf.float(), bias, padding=1) + rowbias[:, :, None, None]
    ref = ref + R.float().view(B, H, W_, Cout).permute(0, 3, 1, 2)
    lib.mkd_gemm_force_tile(cfg)
    try:
        out = torch.zeros(B * H * W_, Cout, device=DEV, dtype=torch.bfloat16)
        rc = lib.mkd_gemm_bf16(P(xn), ld, P(wp), 9 * Cin, P(bias), P(rowbias), Cout, H * W_, P(R), Cout, 1.0, 0, P(out), Cout, 0,
                               B * H * W_, Cout, 9 * Cin, 1, B, H, W_, Cin, H, W_, 1, 0, splitk, None)
        if rc == -4:
            pytest.skip('tile configuration does not fit this geometry: ' + lib.mkd_last_error().decode())
        assert rc == 0, lib.mkd_last_error()
        sync()
    finally:
        lib.mkd_gemm_force_tile(-1)
    assert_close_bf16(out.float().view(B, H, W_, Cout).permute(0, 3, 1, 2), ref, what=f'patch conv cfg {cfg}')


@pytest.mark.parametrize('B,hw,C,silu,eps,pad', [(2, 64, 320, 1, 1e-5, 0), (3, 256, 640, 0, 1e-6, 0), (2, 16, 2560, 1, 1e-5, 0),
                                                (2, 1024, 320, 1, 1e-5, 0), (2, 64, 960, 1, 1e-5, 64), (2, 16, 64, 1, 1e-5, 0), (1, 4096, 960, 1, 1e-5, 0), (2, 4096, 320, 0, 1e-5, 0),
                                                (1, 64, 1920, 1, 1e-5, 0),
                                                # large tensors: ragged pixel counts, strided rows, every vector width
                                                (4, 1024, 640, 1, 1e-5, 64), (4, 1024, 960, 1, 1e-5, 0), (8, 1024, 320, 1, 1e-5, 320),
                                                (1, 1024, 2560, 0, 1e-6, 0), (3, 1600, 320, 1, 1e-5, 0), (2, 1024, 1920, 1, 1e-5, 0),
                                                (5, 1032, 1280, 1, 1e-5, 0)])
def test_groupnorm(B, hw, C, silu, eps, pad):
    lib = L()
    g = torch.Generator().manual_seed(C + hw)
    x = bf(torch.randn(B, hw, C + pad, generator=g) * 2 + 0.5)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV); beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    y = torch.zeros(B, hw, C, device=DEV, dtype=torch.bfloat16)
    rc = lib.mkd_groupnorm(P(x), C + pad, P(gamma), P(beta), eps, silu, P(y), C, B, hw, C, 32, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    xr = x[..., :C].float().permute(0, 2, 1)
    ref = F.group_norm(xr, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    assert_close_bf16(y.float().permute(0, 2, 1), ref, what='groupnorm')


@pytest.mark.parametrize('rows,d', [(100, 320), (37, 640), (16, 1280), (5, 64)])
def test_layernorm(rows, d):
    lib = L()
    g = torch.Generator().manual_seed(d)
    x = bf(torch.randn(rows, d, generator=g) * 3 + 1)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV); beta = (0.1 * torch.randn(d, generator=g)).to(DEV)
    y = torch.empty_like(x)
    assert lib.mkd_layernorm(P(x), P(gamma), P(beta), 1e-5, P(y), rows, d, None) == 0, lib.mkd_last_error()
    sync()
    assert_close_bf16(y, F.layer_norm(x.float(), (d,), gamma, beta, 1e-5), what='layernorm')


@pytest.mark.parametrize('B,Tq,Tk,heads,dh', [(2, 256, 256, 8, 40), (1, 1024, 1024, 2, 40), (2, 64, 64, 8, 160), (2, 16, 16, 8, 160),
                                            (2, 256, 77, 8, 80), (2, 100, 77, 4, 40), (1, 64, 77, 2, 32), (2, 16, 77, 8, 160),
                                            (1, 200, 130, 3, 64), (2, 1024, 77, 8, 40), (1, 64, 96, 2, 80), (1, 48, 65, 2, 160), (1, 64, 64, 2, 80),
                                            (1, 1100, 1100, 2, 40), (1, 1024, 1153, 1, 40)])      # (ragged queries / a 1-key last tile in the LDS-DMA kernel)
def test_attention(B, Tq, Tk, heads, dh):
    lib = L()
    g = torch.Generator().manual_seed(Tq + Tk + dh)
    d = heads * dh
    q = bf(torch.randn(B * Tq, d, generator=g)); kv = bf(torch.randn(B * Tk, 2 * d, generator=g))
    o = torch.zeros(B * Tq, d, device=DEV, dtype=torch.bfloat16)
    scale = dh ** -0.5
    k = kv[:, :d]; v = kv[:, d:]
    rc = lib.mkd_attention(P(q), d, P(k), 2 * d, C.c_void_p(kv.data_ptr() + 2 * d), 2 * d, P(o), d, B, Tq, Tk, heads, dh, scale, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    qf = q.float().view(B, Tq, heads, dh).transpose(1, 2)
    kf = k.float().reshape(B, Tk, heads, dh).transpose(1, 2)
    vf = v.float().reshape(B, Tk, heads, dh).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).transpose(1, 2).reshape(B * Tq, d)
    # P is rounded to bf16 before the PV product: allow 2x the generic kernel budget
    assert_close_bf16(o, ref, rel=8e-3, what='attention')


def test_attention_row_sums_on_the_matrix_cores():
    """Round 4: for >= 2048 keys at dh 40 the softmax denominator is row DH of O^T = V^T P^T (a column of ones in the padded V tile)
    instead of 16 VALU adds per tile.  4096 / 2304 (ragged last tile) keys against torch, plus a spiked key in a late tile (rescale
    path: the denominator row is rescaled with the rest of O)."""
    lib = L()
    for (B, Tq, Tk, heads) in ((1, 1024, 4096, 2), (2, 1152, 2304, 1)):
        dh = 40; d = heads * dh
        g = torch.Generator().manual_seed(Tk)
        q = torch.randn(B * Tq, d, generator=g); k = torch.randn(B * Tk, d, generator=g); v = torch.randn(B * Tk, d, generator=g)
        k[Tk - 100, :dh] = q[7, :dh] * 4.0                         # sample 0, head 0: query 7 meets its spike in the last tiles
        q, k, v = bf(q), bf(k), bf(v)
        o = torch.zeros(B * Tq, d, device=DEV, dtype=torch.bfloat16)
        assert lib.mkd_attention(P(q), d, P(k), d, P(v), d, P(o), d, B, Tq, Tk, heads, dh, dh ** -0.5, None) == 0, lib.mkd_last_error()
        sync()
        qf = q.float().view(B, Tq, heads, dh).transpose(1, 2); kf = k.float().view(B, Tk, heads, dh).transpose(1, 2)
        vf = v.float().view(B, Tk, heads, dh).transpose(1, 2)
        ref = (torch.softmax(qf @ kf.transpose(-1, -2) * dh ** -0.5, -1) @ vf).transpose(1, 2).reshape(B * Tq, d)
        assert_close_bf16(o, ref, rel=8e-3, what=f'attention, MFMA row sums, {Tk} keys')


def test_attention_spiked_scores():
    """online-softmax rescale path: one key far above the rest, placed in a late tile."""
    lib = L()
    B, Tq, Tk, heads, dh = 1, 64, 256, 1, 40
    g = torch.Generator().manual_seed(3)
    q = torch.randn(Tq, dh, generator=g); k = torch.randn(Tk, dh, generator=g); v = torch.randn(Tk, dh, generator=g)
    k[200] = q[5] * 4.0
    q, k, v = bf(q), bf(k), bf(v)
    o = torch.zeros(Tq, dh, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_attention(P(q), dh, P(k), dh, P(v), dh, P(o), dh, B, Tq, Tk, heads, dh, dh ** -0.5, None) == 0
    sync()
    ref = torch.softmax(q.float() @ k.float().t() * dh ** -0.5, -1) @ v.float()
    assert_close_bf16(o, ref, rel=8e-3, what='attention spike')


def test_geglu():
    lib = L()
    g = torch.Generator().manual_seed(1)
    rows, inner = 77, 1280
    x = bf(torch.randn(rows, 2 * inner, generator=g) * 2)
    y = torch.empty(rows, inner, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_geglu(P(x), P(y), rows, inner, None) == 0
    sync()
    a, gate = x.float().chunk(2, -1)
    assert_close_bf16(y, a * F.gelu(gate), what='geglu')


@pytest.mark.parametrize('Cin,Cout,stride,in_nchw,out_nchw,act', [(4, 320, 1, 1, 0, 0), (6, 16, 1, 1, 0, 1), (320, 4, 1, 0, 1, 0), (16, 32, 2, 0, 0, 1)])
def test_conv3x3_direct(Cin, Cout, stride, in_nchw, out_nchw, act):
    lib = L()
    g = torch.Generator().manual_seed(Cin + Cout)
    B, H, W_ = 2, 16, 24
    x = torch.randn(B, Cin, H, W_, generator=g)
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g).to(DEV)
    wp = torch.empty(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_pack_conv_weight(P(w.float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
    if in_nchw:
        xin = x.to(DEV).contiguous(); xr = xin
    else:
        xb = bf(x); xin = xb.permute(0, 2, 3, 1).contiguous(); xr = xb.float()
    ref = F.conv2d(xr, w.float(), bias, stride=stride, padding=1)
    if act:
        ref = F.silu(ref)
    Ho, Wo = ref.shape[2:]
    add = bf(torch.randn(B, Ho, Wo, Cout, generator=g)) if not out_nchw else None
    if add is not None:
        ref = ref + add.float().permute(0, 3, 1, 2)
    out = torch.zeros((B, Cout, Ho, Wo) if out_nchw else (B, Ho, Wo, Cout), device=DEV,
                      dtype=torch.float32 if out_nchw else torch.bfloat16)
    rc = lib.mkd_conv3x3_direct(P(xin), in_nchw, P(wp), P(bias), P(out), out_nchw, act, P(add), B, H, W_, Cin, Cout, stride, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    o = out if out_nchw else out.float().permute(0, 3, 1, 2)
    assert_close_bf16(o, ref, what='conv3x3_direct')


def test_ddim_step_matches_reference_formula():
    """cddim.py:39-40, 56-78 in fp32: bit-for-bit is not required (fma contraction), 1e-6 relative is."""
    lib = L()
    g = torch.Generator().manual_seed(0)
    n = 2 * 4 * 32 * 32
    x = torch.randn(n, generator=g).to(DEV); ec = torch.randn(n, generator=g).to(DEV); eu = torch.randn(n, generator=g).to(DEV)
    noise = torch.randn(n, generator=g).to(DEV)
    a_t, a_prev, sigma, s = 0.0057755, 0.00728173, 0.05, 9.0
    s1m = math.sqrt(1 - a_t)
    xp = torch.empty_like(x); x0 = torch.empty_like(x)
    rc = lib.mkd_ddim_step(P(x), P(ec), P(eu), s, a_t, a_prev, sigma, s1m, P(noise), 1.0, P(xp), P(x0), n, None)
    assert rc == 0
    sync()
    e = eu + s * (ec - eu)
    r0 = (x - s1m * e) / math.sqrt(a_t)
    rp = math.sqrt(a_prev) * r0 + math.sqrt(1 - a_prev - sigma ** 2) * e + sigma * noise
    assert torch.allclose(x0, r0, rtol=2e-6, atol=1e-5)
    assert torch.allclose(xp, rp, rtol=2e-6, atol=1e-5)
    # no CFG, no noise, no x0
    rc = lib.mkd_ddim_step(P(x), P(ec), None, 1.0, a_t, a_prev, 0.0, s1m, None, 1.0, P(xp), None, n, None)
    assert rc == 0
    sync()
    r0 = (x - s1m * ec) / math.sqrt(a_t)
    assert torch.allclose(xp, math.sqrt(a_prev) * r0 + math.sqrt(1 - a_prev) * ec, rtol=2e-6, atol=1e-5)


def test_patch_entry_on_a_geometry_it_does_not_fit_falls_back_with_a_big_enough_workspace():
    """the tuned table is keyed on (M, N, K) only.  A table / override entry that names an LDS-patch tile for a
    geometry the patch kernel rejects (here W = 12: neither < 16 nor a multiple of 16) makes launch_gemm re-plan with the generic
    tile and HEURISTIC split-K (4 here), while the workspace used to be sized from the entry's own split (1).  Plan-time and
    launch-time decisions now come from one function and the launch checks the slab capacity."""
    lib = L()
    B, H, W_, Cin, Cout = 2, 8, 12, 640, 128
    M, K = B * H * W_, 9 * Cin
    assert lib.mkd_gemm_cfg_supported(9, M, Cout, K, 1, H, W_, Cin, H, W_, 1, 0) == 0
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, Cin, H, W_, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g).to(DEV)
    xb = bf(x); wbf = bf(w)
    xn = xb.permute(0, 2, 3, 1).contiguous()
    wp = torch.empty(Cout, K, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_pack_conv_weight(P(wbf.float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
    ref = F.conv2d(xb.float(), wbf.float(), bias, padding=1)
    lib.mkd_gemm_set_override(M, Cout, K, 1, 1, 0, 9, 1)          # patch 128x64, split 1
    try:
        out = gemm(xn, wp, bias=bias, conv=(B, H, W_, Cin, H, W_, 1, 0))
    finally:
        lib.mkd_gemm_set_override(0, 0, 0, 0, 0, 0, 0, 0)
    assert_close_bf16(out.float().view(B, H, W_, Cout).permute(0, 3, 1, 2), ref, what='patch override on an unsupported geometry')


# ---- GroupNorm with producer-emitted statistics ---------------------------------------------------------------------------
def _ref_gstat(out_bhwc, cg, coff):
    """float64 (sum, sumsq) per (sample, consumer group) of a [B, hw, N] tensor sitting at consumer column coff."""
    B, hw, N = out_bhwc.shape
    o = out_bhwc.double()
    S = torch.zeros(B, 32, 2, dtype=torch.float64)
    for c in range(N):
        gidx = (coff + c) // cg
        S[:, gidx, 0] += o[:, :, c].sum(1).cpu()
        S[:, gidx, 1] += (o[:, :, c] ** 2).sum(1).cpu()
    return S


def _check_gstat(gst, out_bhwc, cg, coff, what):
    ref = _ref_gstat(out_bhwc, cg, coff)
    got = torch.stack([gst[..., 0].double().cpu() / 2 ** 24, gst[..., 1].double().cpu() / 2 ** 18], -1)
    n = out_bhwc.shape[1] * cg
    # fp32 partial sums of <= 128 values + fixed-point rounding: absolute error per element far below the GroupNorm eps
    err_s = ((got[..., 0] - ref[..., 0]).abs() / n).max().item()
    err_q = ((got[..., 1] - ref[..., 1]).abs() / n).max().item()
    scale = max(1.0, (ref[..., 1] / n).max().item())
    assert err_s <= 2e-6 * scale ** 0.5 and err_q <= 2e-6 * scale, f'{what}: mean error {err_s:.2e}, mean-square error {err_q:.2e} (scale {scale:.1f})'


@pytest.mark.parametrize('B,hw,C,silu,eps,pad', [(2, 64, 320, 1, 1e-5, 0), (3, 256, 640, 0, 1e-6, 0), (2, 16, 2560, 1, 1e-5, 0),
                                                (2, 1024, 320, 1, 1e-5, 0), (2, 64, 960, 1, 1e-5, 64), (2, 16, 64, 1, 1e-5, 0),
                                                (1, 4096, 960, 1, 1e-5, 0), (1, 64, 1920, 1, 1e-5, 0), (5, 12, 128, 1, 1e-5, 0), (8, 16, 1280, 0, 1e-6, 0)])
def test_groupnorm_split_stats_then_apply(B, hw, C, silu, eps, pad):
    """mkd_gn_colstats (stand-alone statistics) + mkd_gn_apply_stats == torch group_norm; statistics vs float64 sums; two
    column slices (a concat of two producers) accumulate into the same buffer; bit-repeatable."""
    lib = L()
    g = torch.Generator().manual_seed(C + hw)
    x = bf(torch.randn(B, hw, C + pad, generator=g) * 2 + 0.5)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV); beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    cg = C // 32
    split = (C // 3) // 8 * 8 or 8                       # first "producer" writes columns [0, split), the second the rest
    runs = []
    for _ in range(2):
        gst = torch.zeros(B, 32, 2, device=DEV, dtype=torch.int64)
        assert lib.mkd_gn_colstats(P(x), C + pad, B, hw, split, cg, 0, P(gst), None) == 0, lib.mkd_last_error()
        assert lib.mkd_gn_colstats(P(x[:, :, split:]), C + pad, B, hw, C - split, cg, split, P(gst), None) == 0, lib.mkd_last_error()
        sync()
        runs.append(gst)
    assert torch.equal(runs[0], runs[1])
    _check_gstat(runs[0], x[..., :C], cg, 0, 'colstats')
    y = torch.zeros(B, hw, C, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_gn_apply_stats(P(x), C + pad, P(gamma), P(beta), eps, silu, P(y), C, B, hw, C, P(runs[0]), None) == 0, lib.mkd_last_error()
    sync()
    ref = F.group_norm(x[..., :C].float().permute(0, 2, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    assert_close_bf16(y.float().permute(0, 2, 1), ref, what='groupnorm (split statistics)')


def _gemm_gn(A, W, bias, M, N, K, cv, splitk, gst, cg, coff, hw, R=None, ldc=None, lda=None):
    lib = L()
    ldc_ = N if ldc is None else ldc
    out = torch.zeros((M, ldc_), device=DEV, dtype=torch.bfloat16)
    rc = lib.mkd_gemm_gnstat_bf16(P(A), A.stride(0) if lda is None else lda, P(W), K, P(bias), None, 0, 1, P(R), 0 if R is None else R.stride(0),
                                  1.0, 0, P(out), ldc_, 0, M, N, K, *cv, splitk, P(gst), cg, coff, hw, None)
    return rc, out


@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 4, 5, 12, 13, 14, 15, 16, 17, 18, 19, 21, 24])
@pytest.mark.parametrize('B,hw,N,K,cg,coff,splitk', [(3, 64, 320, 320, 10, 0, 1), (2, 256, 640, 192, 30, 320, 1), (5, 16, 1280, 256, 80, 1280, 1),
                                                     (7, 12, 64, 64, 6, 128, 1), (2, 64, 320, 1280, 10, 0, 4)])
def test_gemm_epilogue_emits_groupnorm_statistics(cfg, B, hw, N, K, cg, coff, splitk):
    """Every gather-GEMM tile configuration (and the split-K reduce kernel): the statistics a GEMM adds for its output equal the
    float64 sums of the bf16 tensor it stored - tiles spanning several samples (hw 12 / 16), several tiles per sample, groups
    that straddle column tiles (cg 10 / 30 / 6), an output that is the second half of a concat (coff > 0) - and are bit-repeatable."""
    lib = L()
    M = B * hw
    g = torch.Generator().manual_seed(cfg * 13 + M + N)
    A = bf(torch.randn(M, K, generator=g)); W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g).to(DEV); R = bf(torch.randn(M, N, generator=g))
    lib.mkd_gemm_force_tile(cfg)
    try:
        res = []
        for _ in range(2):
            gst = torch.zeros(B, 32, 2, device=DEV, dtype=torch.int64)
            rc, out = _gemm_gn(A, W, bias, M, N, K, (0, 0, 0, 0, 0, 0, 0, 1, 0), splitk, gst, cg, coff, hw, R=R)
            assert rc == 0, lib.mkd_last_error()
            sync()
            res.append((gst, out))
    finally:
        lib.mkd_gemm_force_tile(-1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert_close_bf16(res[0][1], A.float() @ W.float().t() + bias + R.float(), what=f'gemm+gn cfg {cfg}')
    _check_gstat(res[0][0], res[0][1].view(B, hw, N), cg, coff, f'gemm statistics cfg {cfg}')
    used = torch.zeros(32, dtype=torch.bool); used[coff // cg:(coff + N - 1) // cg + 1] = True
    assert (res[0][0].cpu()[:, ~used] == 0).all()


@pytest.mark.parametrize('cfg', [6, 7, 8, 9, 10, 11, 3, 5, 25, 38, 40, 42])
@pytest.mark.parametrize('B,H,W_,Cin,Cout,cg,coff,splitk', [(2, 32, 32, 64, 320, 10, 0, 1), (8, 4, 4, 128, 1280, 40, 0, 2), (3, 8, 8, 320, 640, 30, 320, 1),
                                                           (5, 4, 4, 128, 192, 6, 0, 1), (2, 16, 16, 128, 64, 2, 0, 2)])
def test_conv_epilogue_emits_groupnorm_statistics(cfg, B, H, W_, Cin, Cout, cg, coff, splitk):
    """The LDS-staged 3x3 tiles (spatial tiles: one segment per image of the block, ragged image groups) and the gather conv, with
    and without split-K over channel chunks."""
    lib = L()
    M = B * H * W_
    g = torch.Generator().manual_seed(cfg * 17 + M + Cout)
    x = torch.randn(B, Cin, H, W_, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g).to(DEV)
    xb = bf(x); wbf = bf(w)
    xn = xb.permute(0, 2, 3, 1).contiguous()
    wp = torch.empty(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_pack_conv_weight(P(wbf.float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
    ref = F.conv2d(xb.float(), wbf.float(), bias, padding=1)
    lib.mkd_gemm_force_tile(cfg)
    try:
        res = []
        for _ in range(2):
            gst = torch.zeros(B, 32, 2, device=DEV, dtype=torch.int64)
            rc, out = _gemm_gn(xn, wp, bias, M, Cout, 9 * Cin, (1, B, H, W_, Cin, H, W_, 1, 0), splitk, gst, cg, coff, H * W_, lda=Cin)
            if rc == -4:
                pytest.skip('tile configuration does not fit this geometry')
            assert rc == 0, lib.mkd_last_error()
            sync()
            res.append((gst, out))
    finally:
        lib.mkd_gemm_force_tile(-1)
    assert torch.equal(res[0][0], res[1][0])
    assert_close_bf16(res[0][1].float().view(B, H, W_, Cout).permute(0, 3, 1, 2), ref, what=f'conv+gn cfg {cfg}')
    _check_gstat(res[0][0], res[0][1].view(B, H * W_, Cout), cg, coff, f'conv statistics cfg {cfg}')


@pytest.mark.parametrize('B,H,W_,Cin,Cout,splitk,res,rowb,silu,eps', [(2, 8, 8, 1280, 1280, 5, 1, 0, 1, 1e-5), (8, 4, 4, 640, 1280, 0, 0, 1, 1, 1e-5),
                                                                    (3, 16, 16, 320, 640, 3, 1, 1, 0, 1e-6), (2, 32, 32, 128, 320, 2, 0, 1, 1, 1e-5),
                                                                    (5, 4, 12, 256, 256, 4, 1, 1, 1, 1e-5)])
def test_splitk_conv_feeds_groupnorm_from_its_slabs(B, H, W_, Cin, Cout, splitk, res, rowb, silu, eps):
    """mkd_gemm_groupnorm_bf16: [split-K conv3x3 -> fp32 slabs][ONE kernel: reduce + epilogue + GroupNorm(+SiLU)] vs torch, and
    BIT-IDENTICAL to the three-kernel path (conv + split-K reduce, then mkd_groupnorm) on both outputs."""
    lib = L()
    hw, M, K = H * W_, B * H * W_, 9 * Cin
    g = torch.Generator().manual_seed(B * hw + Cin + Cout)
    x = torch.randn(B, Cin, H, W_, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g).to(DEV)
    rowbias = torch.randn(B, Cout, generator=g).to(DEV) if rowb else None
    R = bf(torch.randn(M, Cout, generator=g)) if res else None
    gamma = (1 + 0.2 * torch.randn(Cout, generator=g)).to(DEV); beta = (0.2 * torch.randn(Cout, generator=g)).to(DEV)
    xb = bf(x); wbf = bf(w)
    xn = xb.permute(0, 2, 3, 1).contiguous()
    wp = torch.empty(Cout, K, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_pack_conv_weight(P(wbf.float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
    raw = torch.zeros(M, Cout, device=DEV, dtype=torch.bfloat16); y = torch.zeros(M, Cout, device=DEV, dtype=torch.bfloat16)
    rc = lib.mkd_gemm_groupnorm_bf16(P(xn), Cin, P(wp), K, P(bias), P(rowbias), Cout if rowb else 0, hw, P(R), Cout if res else 0, 1.0, P(raw), Cout, 1,
                                     M, Cout, K, 1, B, H, W_, Cin, H, W_, 1, 0, splitk, hw, P(gamma), P(beta), eps, silu, P(y), Cout, None)
    if rc == -4:
        pytest.skip(lib.mkd_last_error().decode())
    assert rc == 0, lib.mkd_last_error()
    sync()
    conv = F.conv2d(xb.float(), wbf.float(), bias, padding=1)
    if rowb:
        conv = conv + rowbias[:, :, None, None]
    if res:
        conv = conv + R.float().view(B, H, W_, Cout).permute(0, 3, 1, 2)
    assert_close_bf16(raw.float().view(B, H, W_, Cout).permute(0, 3, 1, 2), conv, what='raw conv output')
    ref = F.group_norm(raw.float().view(B, hw, Cout).permute(0, 2, 1), 32, gamma, beta, eps)      # statistics of the STORED bf16 tensor
    if silu:
        ref = F.silu(ref)
    assert_close_bf16(y.float().view(B, hw, Cout).permute(0, 2, 1), ref, what='GroupNorm from slabs')
    # the separate kernels
    raw2 = gemm(xn, wp, bias=bias, rowbias=rowbias, rpb=hw, R=R, conv=(B, H, W_, Cin, H, W_, 1, 0), lda=Cin, splitk=splitk)
    y2 = torch.zeros_like(y)
    assert lib.mkd_groupnorm(P(raw2), Cout, P(gamma), P(beta), eps, silu, P(y2), Cout, B, hw, Cout, 32, None) == 0
    sync()
    assert torch.equal(raw, raw2), 'raw output differs from the split-K reduce kernel'
    assert torch.equal(y, y2), 'GroupNorm output differs from the two-kernel path'
    # without the raw output the normalised one must not change
    y3 = torch.zeros_like(y)
    junk = torch.full_like(raw, 7.0)
    rc = lib.mkd_gemm_groupnorm_bf16(P(xn), Cin, P(wp), K, P(bias), P(rowbias), Cout if rowb else 0, hw, P(R), Cout if res else 0, 1.0, P(junk), Cout, 0,
                                     M, Cout, K, 1, B, H, W_, Cin, H, W_, 1, 0, splitk, hw, P(gamma), P(beta), eps, silu, P(y3), Cout, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    assert torch.equal(y3, y) and (junk == 7.0).all()
    # ... and C may then be NULL, as mkd.h says (valid only when write_raw != 0)
    y4 = torch.zeros_like(y)
    rc = lib.mkd_gemm_groupnorm_bf16(P(xn), Cin, P(wp), K, P(bias), P(rowbias), Cout if rowb else 0, hw, P(R), Cout if res else 0, 1.0, None, Cout, 0,
                                     M, Cout, K, 1, B, H, W_, Cin, H, W_, 1, 0, splitk, hw, P(gamma), P(beta), eps, silu, P(y4), Cout, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    assert torch.equal(y4, y)


@pytest.mark.parametrize('cfg', [-1, 0, 1, 3, 4, 5, 14, 15, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29])
@pytest.mark.parametrize('M,d,N2,act', [(300, 320, 960, 0), (128, 1280, 1280, 0), (1024, 640, 5120, 2), (77, 64, 128, 0)])
def test_gemm_layernorm_on_the_fly(cfg, M, d, N2, act):
    """C = act(LN(A) . W^T + b) with the row statistics taken INSIDE the GEMM (row_stats = NULL: ones . A^T and diag(A . A^T) on the
    matrix cores), every tile configuration incl. the in-block K split, GEGLU epilogue, a stride on A, rows with a large common
    offset (the mean-subtraction path) - vs torch layer_norm -> linear."""
    lib = L()
    g = torch.Generator().manual_seed(M + d + N2 + 7 * cfg)
    lda = d + 64
    Ab = bf(torch.randn(M, lda, generator=g) + torch.randn(M, 1, generator=g) * 2.0)
    Ab[:, d:] = 55.0                                   # beyond the row: must not enter the statistics
    A = Ab[:, :d]
    W2 = (torch.randn(N2, d, generator=g) / math.sqrt(d)).to(DEV)
    gamma = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV); beta = (0.2 * torch.randn(d, generator=g)).to(DEV)
    b2 = torch.randn(N2, generator=g).to(DEV)
    Wf = torch.empty(N2, d, device=DEV, dtype=torch.bfloat16); s = torch.empty(N2, device=DEV); bfold = torch.empty(N2, device=DEV)
    if act == 2:          # GEGLU: (value, gate) rows interleaved, as the engine folds ff.net.0.proj
        inner = N2 // 2
        for half in range(2):
            assert lib.mkd_fold_layernorm(P(W2[half * inner:]), P(gamma), P(beta), P(b2[half * inner:]), inner, d, P(Wf), half, 2, P(s), P(bfold), None) == 0
    else:
        assert lib.mkd_fold_layernorm(P(W2), P(gamma), P(beta), P(b2), N2, d, P(Wf), 0, 1, P(s), P(bfold), None) == 0
    ncol = N2 // 2 if act == 2 else N2
    out = torch.zeros(M, ncol, device=DEV, dtype=torch.bfloat16)
    lib.mkd_gemm_force_tile(cfg)
    try:
        rc = lib.mkd_gemm_ln_bf16(P(Ab), lda, P(Wf), d, P(bfold), P(s), None, 0, 1e-5, act, P(out), ncol, M, N2, d, None)
        assert rc == 0, lib.mkd_last_error()
        sync()
    finally:
        lib.mkd_gemm_force_tile(-1)
    ref = F.linear(F.layer_norm(A.float(), (d,), gamma, beta, 1e-5), W2, b2)
    if act == 2:
        ref = ref[:, :inner] * F.gelu(ref[:, inner:])
    assert_close_bf16(out, ref, rel=8e-3, what=f'on-the-fly layernorm gemm cfg {cfg}')


@pytest.mark.parametrize('cfg', [1, 5, 19, 24])
@pytest.mark.parametrize('offset,std', [(50.0, 0.5), (100.0, 1.0), (-30.0, 0.25)])
def test_gemm_layernorm_on_the_fly_rows_with_mean_far_above_std(cfg, offset, std):
    """the on-the-fly form takes var = E[x^2] - mean^2 from single-pass fp32 MFMA sums and applies rstd * (acc - mu * s) to
    accumulators of RAW rows; rows with |mean| >> std lose precision to cancellation in both.  Rows at mean / std = 100 - 120 (far
    beyond what a residual stream after attention / FF carries) against torch layer_norm -> linear on the same bf16 inputs, and
    against the two-kernel path (layernorm_kernel, then the plain GEMM)."""
    lib = L()
    M, d, N2 = 256, 320, 320
    g = torch.Generator().manual_seed(int(abs(offset)) + cfg)
    A = bf(offset + std * torch.randn(M, d, generator=g))
    W2 = (torch.randn(N2, d, generator=g) / math.sqrt(d)).to(DEV)
    gamma = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV); beta = (0.2 * torch.randn(d, generator=g)).to(DEV)
    b2 = torch.randn(N2, generator=g).to(DEV)
    Wf = torch.empty(N2, d, device=DEV, dtype=torch.bfloat16); s = torch.empty(N2, device=DEV); bfold = torch.empty(N2, device=DEV)
    assert lib.mkd_fold_layernorm(P(W2), P(gamma), P(beta), P(b2), N2, d, P(Wf), 0, 1, P(s), P(bfold), None) == 0
    out = torch.zeros(M, N2, device=DEV, dtype=torch.bfloat16)
    lib.mkd_gemm_force_tile(cfg)
    try:
        assert lib.mkd_gemm_ln_bf16(P(A), d, P(Wf), d, P(bfold), P(s), None, 0, 1e-5, 0, P(out), N2, M, N2, d, None) == 0, lib.mkd_last_error()
        sync()
    finally:
        lib.mkd_gemm_force_tile(-1)
    ref = F.linear(F.layer_norm(A.float(), (d,), gamma, beta, 1e-5), W2, b2)
    # two-kernel path: LayerNorm kernel (bf16 out), then the plain GEMM on the unfolded weights
    y = torch.empty(M, d, device=DEV, dtype=torch.bfloat16); two = torch.zeros(M, N2, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_layernorm(P(A), P(gamma), P(beta), 1e-5, P(y), M, d, None) == 0
    W2b = bf(W2)
    assert lib.mkd_gemm_bf16(P(y), d, P(W2b), d, P(b2), None, 0, 1, None, 0, 1.0, 0, P(two), N2, 0, M, N2, d, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, None) == 0
    sync()
    r_fly, r_two = rel_l2(out, ref), rel_l2(two, ref)
    print(f'mean {offset} std {std} cfg {cfg}: on-the-fly rel-L2 {r_fly:.3e}, LayerNorm kernel + GEMM {r_two:.3e}')
    assert torch.isfinite(out.float()).all() and r_fly <= 6e-3, r_fly          # measured on MI355X: 2.0e-3 - 2.1e-3 (the two-kernel path, which rounds LN(x) to bf16 first: 2.4e-3)
    assert r_fly <= 3.0 * r_two + 2e-3


@pytest.mark.parametrize('mode', [1, 2])
def test_xcd_tile_order_is_a_permutation_of_the_tiles(mode):
    """mkd_gemm_set_xcd_mode: the workgroup -> tile remap (one contiguous run of the tile sequence per XCD) must visit every tile
    exactly once whatever the grid (tile counts not divisible by 8, split-K planes, the LDS-staged conv's spatial tiles): results
    bit-identical to the launch order."""
    lib = L()
    g = torch.Generator().manual_seed(77 + mode)
    cases = []
    for (M, N, K, splitk, cfg) in [(300, 320, 320, 1, 2), (1000, 640, 1344, 1, 3), (128, 1280, 2560, 4, 0), (96, 320, 200, 1, 1),
                                   (8192, 320, 320, 1, 14), (520, 1280, 640, 3, 21), (77, 64, 64, 1, 0)]:
        A = bf(torch.randn(M, K, generator=g)); W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
        bias = torch.randn(N, generator=g).to(DEV)
        cases.append((cfg, lambda A=A, W=W, bias=bias, splitk=splitk: gemm(A, W, bias=bias, splitk=splitk)))
    for (B, H, Wd, Cin, Cout, splitk, cfg) in [(3, 16, 16, 64, 128, 1, 9), (8, 32, 32, 128, 192, 2, 9), (5, 8, 8, 128, 64, 1, 11)]:
        x = bf(torch.randn(B, H, Wd, Cin, generator=g))
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        wp = torch.empty(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)
        assert lib.mkd_pack_conv_weight(P(bf(w).float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
        cases.append((cfg, lambda x=x, wp=wp, B=B, H=H, Wd=Wd, Cin=Cin, splitk=splitk: gemm(x, wp, conv=(B, H, Wd, Cin, H, Wd, 1, 0), splitk=splitk)))
    for cfg, run in cases:
        lib.mkd_gemm_force_tile(cfg)
        try:
            lib.mkd_gemm_set_xcd_mode(0)
            ref = run()
            lib.mkd_gemm_set_xcd_mode(mode)
            out = run()
        finally:
            lib.mkd_gemm_set_xcd_mode(0)
            lib.mkd_gemm_force_tile(-1)
        assert torch.equal(out, ref), f'xcd mode {mode}, cfg {cfg}: differs from the launch order'


@pytest.mark.parametrize('cfg,B,H,W_,Cin,K2,pad2,splitk', [(-1, 2, 8, 8, 64, 128, 0, 0), (5, 1, 16, 16, 64, 192, 64, 0), (3, 2, 16, 16, 128, 64, 0, 2),
                                                         (24, 3, 8, 8, 128, 256, 0, 0), (28, 2, 16, 16, 64, 128, 8, 0), (34, 1, 32, 32, 64, 64, 0, 0),
                                                         (19, 1, 4, 4, 128, 320, 0, 3)])
def test_conv3x3_with_folded_skip(cfg, B, H, W_, Cin, K2, pad2, splitk):
    """ResBlock tail as ONE implicit GEMM (round 4): conv3x3(h) + skip_connection(x) over K = 9 Cin + K2 on the gather kernel
    (GemmArgs::A2; UPSTREAM ResBlock._forward, reached from /root/reference/diffmk/makeup_diffuse.py:164-168) against torch
    conv2d + a 1x1 conv2d of the second input: plain, in-block K split, split-K and eight-wave tiles, ragged tiles, a padded stride of
    the second input (the decoder's concat buffers)."""
    lib = L()
    Cout = Cin
    g = torch.Generator().manual_seed(cfg * 7 + B * H + K2)
    h = bf(torch.randn(B, Cin, H, W_, generator=g)); x2 = bf(torch.randn(B, K2, H, W_, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)); ws = bf(torch.randn(Cout, K2, generator=g) / math.sqrt(K2))
    bias = torch.randn(Cout, generator=g).to(DEV)
    wp = torch.empty(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)
    assert lib.mkd_pack_conv_weight(P(w.float().contiguous()), P(wp), Cout, Cin, 3, 3, None) == 0
    sync()
    wf = torch.cat([wp, ws], 1).contiguous()
    hn = h.permute(0, 2, 3, 1).contiguous()
    ld2 = K2 + pad2
    x2n = torch.full((B, H, W_, ld2), float('nan'), device=DEV, dtype=torch.bfloat16)
    x2n[..., :K2] = x2.permute(0, 2, 3, 1)
    ref = F.conv2d(h.float(), w.float(), bias, padding=1) + F.conv2d(x2.float(), ws.float()[:, :, None, None])
    out = torch.zeros(B * H * W_, Cout, device=DEV, dtype=torch.bfloat16)
    lib.mkd_gemm_force_tile(cfg)
    try:
        rc = lib.mkd_conv3x3_fold_bf16(P(hn), Cin, P(wf), P(bias), P(x2n), ld2, K2, P(out), Cout, B, H, W_, Cin, Cout, splitk, None)
        assert rc == 0, lib.mkd_last_error()
        sync()
        lib.mkd_gemm_force_tile(9)                   # an LDS-staged tile cannot take the second input: refused, not computed wrongly
        assert lib.mkd_conv3x3_fold_bf16(P(hn), Cin, P(wf), P(bias), P(x2n), ld2, K2, P(out.clone()), Cout, B, H, W_, Cin, Cout, 0, None) != 0
    finally:
        lib.mkd_gemm_force_tile(-1)
    assert_close_bf16(out.float().view(B, H, W_, Cout).permute(0, 3, 1, 2), ref, what=f'conv3x3 + folded 1x1 skip, cfg {cfg}')
