"""Builds makeupdiffuse_amd/libmkd.so (gfx950 only) with hipcc. In-tree, incremental.

    python -m makeupdiffuse_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj')
LIB = os.path.join(HERE, 'libmkd.so')
SOURCES = ['kernels_gemm.hip', 'kernels_conv.hip', 'kernels_norm.hip', 'kernels_attn.hip', 'kernels_tfm.hip', 'kernels_misc.hip', 'engine.hip']
HEADERS = ['mkd_common.h', 'gemm_device.h', 'gemm_tuned.inc', os.path.join('..', '..', 'include', 'mkd.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-unused-result',
         '-ffp-contract=fast']


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libmkd.so cannot be built (there is no CPU fallback)')


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + '.o')
        stamp = op + '.sha'
        dg = _digest([sp] + hdrs)
        old = open(stamp).read() if os.path.exists(stamp) else ''
        if force or not os.path.exists(op) or old != dg:
            cmd = [hipcc] + FLAGS + ['-c', sp, '-o', op]
            if verbose:
                print('[mkd build]', ' '.join(cmd), flush=True)
            jobs.append((subprocess.Popen(cmd), cmd, stamp, dg))          # translation units compile concurrently (<= 6 of them)
        objs.append(op)
    rebuilt = bool(jobs)
    failed = None
    for proc, cmd, stamp, dg in jobs:
        if proc.wait() != 0:
            failed = failed or cmd
        else:
            with open(stamp, 'w') as f:
                f.write(dg)
    if failed:
        raise subprocess.CalledProcessError(1, failed)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print('[mkd build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
