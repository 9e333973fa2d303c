#!/usr/bin/env python
"""Experiment: do the kernel BOUNDARIES of one stream (dispatch + cache write-back / invalidate of the 8 non-coherent L2s) slow a
kernel that is running on another stream?  A large bf16 matmul (L2-resident operands re-used across workgroups) is timed alone and
beside a chain of tiny dependent kernels (1 workgroup each, so they take no CUs worth mentioning)."""
import time
import torch

dev = torch.device('cuda:0')
torch.manual_seed(0)


def bench(n, chain_len, reps=20):
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16); b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    tiny = torch.zeros(64, device=dev)
    s_big, s_tiny = torch.cuda.Stream(), torch.cuda.Stream()
    out = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, b, out=out)
    torch.cuda.synchronize()
    res = {}
    for with_chain in (0, 1, 0, 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if with_chain:
            with torch.cuda.stream(s_tiny):
                for _ in range(chain_len):
                    tiny.add_(1.0)
        with torch.cuda.stream(s_big):
            e0.record()
            for _ in range(reps):
                torch.mm(a, b, out=out)
            e1.record()
        torch.cuda.synchronize()
        res.setdefault(with_chain, []).append(e0.elapsed_time(e1) / reps * 1e3)
    return res


for n in (1024, 2048, 4096):
    flops = 2 * n ** 3
    reps = 40 if n <= 2048 else 20
    # chain long enough to cover the matmuls: ~5 us per tiny kernel
    r = bench(n, chain_len=20000 if n == 4096 else 6000, reps=reps)
    alone, beside = min(r[0]), min(r[1])
    print(f'matmul {n}^3: alone {alone:.1f} us ({flops / alone * 1e-6:.0f} TFLOP/s), beside a chain of tiny dependent kernels {beside:.1f} us  (x{beside / alone:.2f})', flush=True)
