"""Multi-GPU: one process per GPU, batch-sharded replicas, ONE collective after the loop.

Every op on the path is per-sample, so ranks never talk inside the 50-step loop; the only exchange is the
all-gather of the finished latents/images.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the
CPU tests of this host logic."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; the first n_items % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f'bad rank/world {rank}/{world}')
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun env; initialises the default group when WORLD_SIZE > 1 - and also for a
    world of ONE when a backend is asked for explicitly (argument or MKD_DIST_BACKEND) with WORLD_SIZE set: the same RCCL
    init / all-gather / all-reduce / barrier code path, runnable on a one-GPU box."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    explicit = (backend or os.environ.get('MKD_DIST_BACKEND')) and 'WORLD_SIZE' in os.environ
    if (world > 1 or explicit) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('MKD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(argv, world: int, env_extra: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Start `world` FRESH child processes of `argv` (one per rank: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set, rendezvous on 127.0.0.1) and wait for them; returns the largest exit code.  The caller must not have touched the GPU:
    children are new processes (no fork of a HIP context, no exec from an initialised process).  Rank 0 inherits stdout, so its
    single JSON line is the launcher's output; every rank inherits stderr."""
    import subprocess
    import sys
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.update(env_extra or {})
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL, stderr=None))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait(timeout=timeout)))
    except BaseException:
        for p in procs:                    # exactly the processes started above
            if p.poll() is None:
                p.kill()
        raise
    finally:
        sys.stdout.flush()
    return rc


def gather_shards(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-rank batch shards (possibly ragged by one) into the full [n_total, ...] tensor, in
    global sample order, on every rank.  Single collective: shards are padded to the largest shard."""
    if not dist.is_initialized():
        if local.shape[0] != n_total:
            raise ValueError('single-process gather: shard is not the whole batch')
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not os.environ.get('MKD_DIST_BACKEND'):
        # a group of one that the CALLER initialised (e.g. torchrun --nproc-per-node 1): nothing to exchange, same storage back.
        # Only a group of one that was asked for explicitly (MKD_DIST_BACKEND: the RCCL-of-one rehearsal) runs the collective.
        if local.shape[0] != n_total:
            raise ValueError('world of one: shard is not the whole batch')
        return local
    lo, hi = shard_range(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f'rank {rank}: shard has {local.shape[0]} samples, expected {hi - lo}')
    cap = -(-n_total // world)
    pad = local
    if local.shape[0] < cap:
        pad = torch.cat([local, local.new_zeros((cap - local.shape[0],) + tuple(local.shape[1:]))])
    if pad.is_cuda and dist.get_backend(group) == 'gloo':
        # rehearsal mode (several ranks on one card): gloo gathers host tensors
        host = pad.cpu().contiguous()
        out_h = host.new_empty((world * cap,) + tuple(host.shape[1:]))
        dist.all_gather_into_tensor(out_h, host, group=group)
        out = out_h.to(pad.device)
    else:
        out = local.new_empty((world * cap,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = []
    for r in range(world):
        l, h = shard_range(n_total, r, world)
        parts.append(out[r * cap: r * cap + (h - l)])
    return torch.cat(parts)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    if dist.get_backend() == 'gloo':
        device = 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
