"""Helpers for the -m gpu parity tests: call libmkd through its C ABI with torch-owned device memory."""
import ctypes as C

import torch

from makeupdiffuse_amd import lib as mlib

DEV = 'cuda:0'


def L():
    return mlib.load()


def P(t):
    return C.c_void_p(None if t is None else t.data_ptr())


def bf(t):
    return t.to(DEV).to(torch.bfloat16).contiguous()


def rel_l2(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def assert_close_bf16(out, ref, rel=4e-3, what=''):
    """SURVEY.md §8c per-kernel tolerance: rel-L2 <= 4e-3, max-abs <= 2^-6 * |ref|_inf (bf16 in, fp32 acc)."""
    out = out.float().cpu(); ref = ref.float().cpu()
    assert torch.isfinite(out).all(), f'{what}: non-finite output'
    r = rel_l2(out, ref)
    mx = (out - ref).abs().max().item()
    lim = ref.abs().max().item() * 2 ** -6
    assert r <= rel, f'{what}: rel-L2 {r:.3e} > {rel:.1e}'
    assert mx <= lim, f'{what}: max-abs {mx:.3e} > {lim:.3e}'


def sync():
    torch.cuda.synchronize()
