#!/usr/bin/env python
"""bench.py — makeup-transfer images/sec at 256x256, 50 DDIM steps (BASELINE.json metric), MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cfg] [--res 256|512] [--batch 8]

One "step" = one pass of the hot path over one batch: mkd_prepare (hint embedding + cross-attn K/V caches)
+ the full 50-step DDIM reverse loop x_T -> latent (ControlNet + UNet every step), inputs resident in HBM.
N > 1: one rank per GPU, batch-sharded replicas (weak scaling, B per GPU fixed), a single RCCL all-gather of the finished
images per step.  Either launched by torch.distributed.run (RANK / WORLD_SIZE in the environment) or by itself: a plain
`python bench.py --gpus N` starts its N ranks as fresh child processes before touching the GPU.  With fewer than N devices
visible the N ranks share card 0 over gloo (a REHEARSAL of the N > 1 code path, flagged in the JSON, not a measurement).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from makeupdiffuse_amd import dist as mdist  # noqa: E402
from makeupdiffuse_amd.engine import MkdEngine, NetConfig, VaeConfig  # noqa: E402
from makeupdiffuse_amd.schedule import DDIMSchedule  # noqa: E402

def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def usable_cores() -> int:
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
MFMA_PREFIXES = ('gemm_', 'attention', 'tfm_tail')


def _csrc_sha16():
    import hashlib
    src = b''.join(open(os.path.join(ROOT, 'makeupdiffuse_amd', 'csrc', f), 'rb').read()
                   for f in ('engine.hip', 'kernels_gemm.hip', 'kernels_conv.hip', 'kernels_norm.hip', 'kernels_attn.hip', 'kernels_tfm.hip', 'kernels_misc.hip', 'gemm_tuned.inc'))
    return hashlib.sha256(src).hexdigest()[:16]


def live_pmc_passes(args):
    """The HBM-side counters of THIS build on THIS box: two rocprofv3 child runs (one --pmc pass per counter, kernel trace only beside
    it, as MI355X_MICROARCH.md §HBM prescribes) of a 2-step eager run of the same workload, reduced to
    {kernel function: {'FETCH_SIZE': avg KB per launch, 'WRITE_SIZE': ..., 'dispatches': n}}.  Returns (dict | None, note).  Runs after
    the timed region, rank 0, one GPU; every failure (no rocprofv3, a refused counter, a timeout) falls back to the committed summaries."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if not exe:
        return None, 'rocprofv3 not on PATH'
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'MKD_DIST_BACKEND')}
    env['TMPDIR'] = '/tmp'
    work = ['--steps', '1', '--warmup', '0', '--ddim-steps', '2', '--no-cpu-baseline', '--graph', '0', '--decode', '0', '--live-pmc', '0',
            '--batch', str(args.batch), '--res', str(args.res)] + (['--cfg'] if args.cfg else []) + (['--interp', str(args.interp)] if args.interp else [])
    out = {}
    base = tempfile.mkdtemp(prefix='mkd_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(base, counter)
            cmd = [exe, '--kernel-trace', '--output-format', 'csv', '--pmc', counter, '-d', d, '-o', 'live', '--',
                   sys.executable, os.path.abspath(__file__)] + work
            t0 = time.perf_counter()
            p = subprocess.Popen(cmd, env=env, cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                _, err = p.communicate(timeout=float(os.environ.get('MKD_LIVE_PMC_TIMEOUT', '90')))
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)          # exactly the session started above
                p.communicate()
                return None, f'rocprofv3 --pmc {counter}: timed out'
            found = glob.glob(os.path.join(d, '**', 'live_counter_collection.csv'), recursive=True)
            if p.returncode != 0 or not found:
                return None, f'rocprofv3 --pmc {counter}: rc {p.returncode}, {len(found)} counter file(s): {(err or "")[-300:]}'
            tot = {}; ids = {}
            for r in csv.DictReader(open(found[0])):
                if r['Counter_Name'] != counter:
                    continue
                k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
                tot[k] = tot.get(k, 0.0) + float(r['Counter_Value'])
                ids.setdefault(k, set()).add(r['Dispatch_Id'])
            for k in tot:
                e = out.setdefault(k, {})
                e[counter] = tot[k] / len(ids[k]); e['dispatches'] = len(ids[k])
            log(f'live PMC pass {counter}: {len(tot)} kernels, {time.perf_counter() - t0:.0f} s')
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return out, 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py ' + ' '.join(work)


def config_tag(args):
    """Name of the workload in the committed profile files (tools/collect_profiles.sh TAG=...)."""
    if args.interp:
        return 'interp' if (args.batch == 4 and args.interp == 11 and args.res == 256) else None
    if args.cfg:
        return 'cfg' if (args.batch == 8 and args.res == 256) else None
    if args.res == 512:
        return 'r512' if args.batch == 8 else None
    return 'b8_256' if (args.batch == 8 and args.res == 256) else None


def pmc_traffic_bytes_per_launch(kernel_fn, live=None, live_note=None, tag='b8_256'):
    """HBM-side bytes per launch of `kernel_fn`: FETCH_SIZE and WRITE_SIZE (KB -> bytes, FETCH doubled for wide streaming reads on gfx950,
    MI355X_MICROARCH.md §HBM).  `live` (live_pmc_passes): measured by this run on this box.  Otherwise from the COMMITTED rocprofv3
    summaries (profiles/), and the second return value says which files / commit they come from and whether the kernel sources are
    still the ones that were profiled."""
    import csv
    if live:
        n = 0; b = 0.0
        for k, e in live.items():
            if k.startswith(kernel_fn) and 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
                n += e['dispatches']; b += e['dispatches'] * (2.0 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024.0
        if n:
            return b / n, {'collected_live': True, 'dispatches': n, 'command': live_note,
                           'formula': '2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, averaged over the launches of the kernel function'}
    cands = [(f'r4_pmc_fetch_size_kb_{tag}.csv', f'r4_pmc_write_size_kb_{tag}.csv', f'r4_provenance_{tag}.json')] if tag else []
    if tag == 'b8_256':
        cands += [('r3_pmc_fetch_size_kb.csv', 'r3_pmc_write_size_kb.csv', 'r3_provenance.json')]
    for f_fetch, f_write, f_prov in cands:
        tot = {}
        for name, mult in ((f_fetch, 2.0), (f_write, 1.0)):
            path = os.path.join(ROOT, 'profiles', name)
            if not os.path.exists(path):
                tot = None
                break
            n = 0; b = 0.0
            for r in csv.DictReader(open(path)):
                if r['kernel'].startswith(kernel_fn):
                    col = [c for c in r if c.startswith('avg_') and c != 'avg_us'][0]
                    n += int(r['dispatches']); b += int(r['dispatches']) * float(r[col]) * 1024.0 * mult
            if n == 0:
                tot = None
                break
            tot[name] = b / n
        if tot is None:
            continue
        src = {'files': sorted('profiles/' + k for k in tot), 'collected_live': False, 'live_attempt': live_note}
        prov = os.path.join(ROOT, 'profiles', f_prov)
        if os.path.exists(prov):
            pv = json.load(open(prov))
            src.update(commit=pv.get('commit'), utc=pv.get('utc'), same_kernel_sources_as_this_run=pv.get('csrc_sha256_16') == _csrc_sha16())
        return sum(tot.values()), src
    return None, {'collected_live': False, 'files': None, 'live_attempt': live_note,
                  'note': 'no committed rocprofv3 summaries for this workload (tools/collect_profiles.sh TAG=...): traffic not reported'}


def l2_traffic_per_eval(tag, attn_launches_per_eval):
    """L2 -> CU traffic of ONE eps evaluation from the committed rocprofv3 summaries of this workload (tag): per kernel, (TCC_HIT_sum +
    TCC_MISS_sum) requests per launch (profiles/r4_pmc_l2_hit_<tag>.csv, 128 B each) x its launches per evaluation.  The number of
    evaluations inside the profiled run is DERIVED: calls of the attention kernels in the run / attention launches of one
    evaluation in this run's plan - the profiled bench also runs eps_profile passes beside the loop.  None when the summaries are
    absent."""
    import csv
    if not tag or not attn_launches_per_eval:
        return None
    try:
        f_stats = os.path.join(ROOT, 'profiles', f'r4_kernel_stats_bench_{tag}.csv'); f_l2 = os.path.join(ROOT, 'profiles', f'r4_pmc_l2_hit_{tag}.csv')
        stats = {}
        for r in csv.DictReader(open(f_stats)):
            n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            stats[n] = stats.get(n, 0) + int(r['Calls'])
        attn_calls = sum(v for k, v in stats.items() if k.startswith('attention_'))      # (attention_kernel<...>, attention_dma40_kernel<...>)
        evals = attn_calls / float(attn_launches_per_eval)
        if evals < 1:
            return None
        req = hit = 0.0
        for r in csv.DictReader(open(f_l2)):
            k = r['kernel']
            if k in ('merge_ff_out_kernel', 'pack_conv_weight_kernel', 'fold_layernorm_kernel', 'tfm_pack_units_kernel') or not (k.split('<')[0] in stats or k in stats):
                continue                                    # (load-time kernels; torch's own kernels)
            per_eval = stats.get(k, 0) / evals
            req += (float(r['avg_TCC_HIT_sum']) + float(r['avg_TCC_MISS_sum'])) * per_eval
            hit += float(r['avg_TCC_HIT_sum']) * per_eval
        out = {'gb_per_eval': req * 128 / 1e9, 'hit_fraction': hit / req if req else None, 'evaluations_in_profiled_run': round(evals, 2),
               'source': f'profiles/r4_pmc_l2_hit_{tag}.csv x profiles/r4_kernel_stats_bench_{tag}.csv'}
        prov = os.path.join(ROOT, 'profiles', f'r4_provenance_{tag}.json')
        if os.path.exists(prov):
            out['same_kernel_sources_as_this_run'] = json.load(open(prov)).get('csrc_sha256_16') == _csrc_sha16()
        return out
    except Exception:
        return None


def synth_inputs(lo, hi, res, ctx_dim, device):
    """SURVEY.md §8d synthetic inputs, generated PER SAMPLE so shards do not depend on the world size."""
    h = res // 8
    xs, hints, ctxs, uctxs = [], [], [], []
    for i in range(lo, hi):
        g = torch.Generator().manual_seed(1234 + i)
        xs.append(torch.randn(1, 4, h, h, generator=g))
        g = torch.Generator().manual_seed(5678 + i)
        hints.append(torch.rand(1, 6, res, res, generator=g))
        g = torch.Generator().manual_seed(91011 + i)
        ctxs.append(torch.randn(1, 77, ctx_dim, generator=g))
        uctxs.append(torch.randn(1, 77, ctx_dim, generator=g))
    cat = lambda v: torch.cat(v).to(device)
    return cat(xs), cat(hints), cat(ctxs), cat(uctxs)


def gen_weights(eng, seed, keep_cpu):
    """Seeded N(0,1/fan_in) weights generated on the device, optionally mirrored to the host
    for the CPU baseline so both legs use the same numbers."""
    g = torch.Generator(device=eng.device)
    g.manual_seed(seed)
    cpu = {} if keep_cpu else None
    for name, shape in eng.expected_params().items():
        is_norm = ('.norm' in name or 'in_layers.0' in name or 'out_layers.0' in name
                   or name.endswith('out.0.weight') or name.endswith('out.0.bias'))
        if len(shape) == 1:
            if is_norm:
                t = (torch.ones if name.endswith('weight') else torch.zeros)(shape, device=eng.device)
            else:
                t = 0.02 * torch.randn(shape, generator=g, device=eng.device)
        else:
            t = torch.randn(shape, generator=g, device=eng.device) / float(np.prod(shape[1:])) ** 0.5
        eng.load_weight(name, t)
        if keep_cpu:
            cpu[name] = t.cpu()
    eng.finalize()
    if eng.vae_cfg is not None:
        eng.finalize_vae()
    return cpu


def cpu_baseline(sd_cpu, res, x1, hint1, ctx1, gpu_eps1, gpu_lat, cpu_steps, ddim_steps):
    """The reference's CPU path cannot run here (ldm/cldm absent), so this times the fp32 torch RESTATEMENT (oracle/, kind "port")
    on the host cores: BASELINE config 1 END TO END - B=1, fp32, cpu_steps (20) DDIM steps from x_T, eta 0, no CFG - and scales
    the time to the metric's ddim_steps.  One extra evaluation first checks the GPU's eps against the CPU's."""
    from oracle import nets, sampler
    threads = usable_cores()
    torch.set_num_threads(threads)
    cfg = nets.FULL
    cond = {'c_crossattn': [ctx1], 'c_concat': [hint1]}
    ref = sampler.apply_model(sd_cpu, cfg, x1, torch.tensor([981]), cond)
    rel = float(((gpu_eps1.cpu() - ref).norm() / ref.norm()).item())
    cos = float(torch.nn.functional.cosine_similarity(gpu_eps1.cpu().flatten(), ref.flatten(), dim=0).item())
    t0 = time.perf_counter()
    lat = sampler.sample(sampler.make_eps_fn(sd_cpu, cfg), sampler.Schedule(), x1, cond, cpu_steps)
    s_run = time.perf_counter() - t0
    log(f'cpu baseline: {cpu_steps}-step DDIM loop of B=1 in {s_run:.1f} s on {threads} threads')
    s_eval = s_run / cpu_steps
    lat_cos = float(torch.nn.functional.cosine_similarity(gpu_lat.cpu().flatten(), lat.flatten(), dim=0).item())
    return {'value': 1.0 / (s_eval * ddim_steps), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'sample': f'BASELINE config 1 end to end: B=1 {res}x{res} fp32, {cpu_steps} DDIM steps from x_T, eta 0, no CFG (oracle/ restatement, '
                      f'same weights) in {s_run:.1f} s = {1.0 / s_run:.4f} images/s at {cpu_steps} steps; scaled x{ddim_steps}/{cpu_steps} to the metric',
            's_per_eval': s_eval, 'config1_seconds': s_run, 'gpu_vs_cpu_eps_rel_l2': rel, 'gpu_vs_cpu_eps_cos': cos,
            'gpu_vs_cpu_latent_cos_after_config1': lat_cos}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=8, help='samples per GPU')
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--ddim-steps', type=int, default=50)
    ap.add_argument('--cfg', action='store_true', help='classifier-free guidance 9.0 (2 evals / step)')
    ap.add_argument('--interp', type=int, default=0, metavar='N_ALPHA',
                    help='BASELINE config 5: makeup interpolation sweep, --batch SOURCES per GPU x N_ALPHA alpha points (two references)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=20, help='DDIM steps of the CPU baseline run (BASELINE config 1: 20; ~1.2 s each on 16 cores)')
    ap.add_argument('--graph', type=int, default=1, help='replay the DDIM step as a hipGraph (0 = eager launches)')
    ap.add_argument('--decode', type=int, default=1, help='1: VAE-decode the latents to images inside the timed step (images out)')
    ap.add_argument('--live-pmc', type=int, default=-1, help='1 (one GPU): after the timed region, two rocprofv3 --pmc child runs of a 2-step '
                    'eager run measure FETCH_SIZE / WRITE_SIZE for roofline.traffic on this box; 0 / failure: the committed summaries; '
                    '-1 (default): on for the full run (CPU baseline on, not itself under a profiler), off for --no-cpu-baseline experiment runs')
    ap.add_argument('--ops-csv', default=None, help='write per-launch-group device times of one eps evaluation')
    args = ap.parse_args()
    if args.live_pmc < 0:
        profiled = any(k.startswith('ROCPROF') for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')
        args.live_pmc = int(not args.no_cpu_baseline and not profiled)

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # self-launch: nothing in this process has touched the GPU (device_count() does not initialise HIP on this image)
        extra = {}
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            if args.gpus > 6:
                raise SystemExit(f'--gpus {args.gpus} with {ndev} device(s) visible: a rehearsal on one card is limited to 6 ranks')
            log(f'{ndev} device(s) visible for --gpus {args.gpus}: REHEARSAL on card 0 over gloo (code path only, not a measurement)')
            extra = {'MKD_BENCH_SINGLE_DEVICE': '1', 'MKD_DIST_BACKEND': 'gloo'}
        raise SystemExit(mdist.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, extra))
    rank, world, local = mdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} '
                         f'or without WORLD_SIZE set (bench.py then starts its own ranks)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU path for the product)')
    if os.environ.get('MKD_BENCH_SINGLE_DEVICE') == '1':      # rehearsal of the N>1 path on a 1-GPU box (with MKD_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device(f'cuda:{local}')

    cfg = NetConfig()
    eng = MkdEngine(cfg, dev)
    if args.decode:
        eng.configure_vae(VaeConfig())
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline      # (the CPU baseline is timed at N = 1 only)
    log(f'rank {rank}/{world}: generating 1.22 G synthetic weights on {dev}')
    sd_cpu = gen_weights(eng, seed=0, keep_cpu=want_cpu)
    log('weights loaded')

    B = args.batch
    n_total = B * world
    lo, hi = mdist.shard_range(n_total, rank, world)
    x_T, hint, ctx, uctx = synth_inputs(lo, hi, args.res, cfg.context_dim, dev)
    hint2 = alpha = None
    if args.interp:
        # every source is sampled at every alpha from the same start noise; E = (1 - a) E(src||ref1) + a E(src||ref2)
        k = args.interp
        _, other, _, _ = synth_inputs(lo + 100000, hi + 100000, args.res, cfg.context_dim, dev)
        hint2 = torch.cat([hint[:, :3], other[:, 3:]], 1)                       # same source, second reference
        rep = lambda t: t.repeat_interleave(k, 0)
        x_T, hint, hint2, ctx = rep(x_T), rep(hint), rep(hint2), rep(ctx)
        alpha = torch.linspace(0.0, 1.0, k, device=dev).repeat(B)
        B = B * k; n_total = n_total * k
    sch = DDIMSchedule().make_ddim(args.ddim_steps)
    if args.interp and args.cfg:
        raise SystemExit('--interp and --cfg are separate workloads')
    cfg_scale = 9.0 if args.cfg else 1.0

    def one_step():
        if args.cfg:       # uncond first (cddim.py:25-31), same hint (diffusion_makeup.py:401)
            eng.prepare(torch.cat([hint, hint]), torch.cat([uctx, ctx]))
        elif args.interp:
            eng.prepare(hint, ctx, hint2=hint2, alpha=alpha)
        else:
            eng.prepare(hint, ctx)
        lat = eng.sample(x_T, sch.ddim_timesteps, sch.ddim_alphas, sch.ddim_alphas_prev, sch.ddim_sqrt_one_minus_alphas,
                         cfg_scale=cfg_scale, use_graph=bool(args.graph))
        out = eng.decode(lat) if args.decode else lat          # decode_first_stage (diffusion_makeup.py:396)
        return mdist.gather_shards(out, n_total)

    for i in range(args.warmup):
        t_w = time.perf_counter()
        out = one_step()
        torch.cuda.synchronize()
        log(f'warmup step {i + 1}/{args.warmup}: {time.perf_counter() - t_w:.2f} s')
    mdist.barrier(); torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = one_step()
    ev1.record()
    mdist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = mdist.max_over_ranks(dt, dev)
    assert torch.isfinite(out).all() or os.environ.get('MKD_BENCH_ALLOW_NONFINITE') == '1', 'non-finite latents'      # (timing experiments with stubbed kernels only)
    log(f'timed {args.steps} steps in {dt:.2f} s')

    # which device every rank ran on (gathered over the process group: shows N distinct GPUs, or one in a rehearsal)
    me = f'{torch.cuda.get_device_name(local)} #{local}' + (f' pci {torch.cuda.get_device_properties(local).pci_bus_id}'
                                                               if hasattr(torch.cuda.get_device_properties(local), 'pci_bus_id') else '')
    dev_names = [me]
    if world > 1:
        dev_names = [None] * world
        torch.distributed.all_gather_object(dev_names, me)
    if rank == 0:
        evals_per_step = args.ddim_steps
        eps_flops = eng.eps_flops()                      # executed FLOPs of one eval at the prepared batch
        loop_ms = ev0.elapsed_time(ev1) / args.steps
        # per-kernel-class device time of ONE eps evaluation, HIP events around every launch on the same stream
        tt = torch.full((eng.batch,), int(sch.ddim_timesteps[-1]), dtype=torch.int64, device=dev)
        x_in = torch.cat([x_T, x_T]) if args.cfg else x_T
        eng.eps_profile(x_in, tt)
        prof = eng.eps_profile(x_in, tt, csv_path=args.ops_csv)
        tot_ms = sum(v['ms'] for v in prof.values())
        # kernel FUNCTIONS behind the classes: the implicit-GEMM gather kernel (all tile configs), the LDS-staged
        # 3x3 conv kernel, the attention kernel.  The dominant one (by device time) carries the roofline.
        fam = {'gemm_kernel': lambda k: k.startswith('gemm_') and 'patch' not in k,
               'conv3x3_patch_kernel': lambda k: k.startswith('gemm_') and 'patch' in k,
               'attention_kernel': lambda k: k == 'attention',
               'tfm_tail_kernel': lambda k: k == 'tfm_tail'}
        agg = {f: {'ms': sum(v['ms'] for k, v in prof.items() if sel(k)), 'flops': sum(v['flops'] for k, v in prof.items() if sel(k)),
                   'launches': sum(v['launches'] for k, v in prof.items() if sel(k))} for f, sel in fam.items()}
        dom = max(agg, key=lambda f: agg[f]['ms'])
        d = agg[dom]
        ach = d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
        live, live_note = live_pmc_passes(args) if (args.live_pmc and world == 1) else (None, 'not attempted (--live-pmc 0 or more than one rank)')
        if live is None:
            log(f'live PMC: {live_note}; roofline.traffic from the committed summaries')
        traffic, traffic_src = pmc_traffic_bytes_per_launch(dom, live, live_note, config_tag(args))
        b2b_ms = sum(v['ms_b2b'] for k, v in prof.items() if fam[dom](k))
        roofline = {'bound': 'mfma', 'kernel': dom + ' (all tile configurations, one eps evaluation)', 'achieved': ach,
                    'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                    'traffic': traffic, 'traffic_source': traffic_src,
                    'launches_per_eval': d['launches'], 'avg_launch_us': 1e3 * d['ms'] / max(1, d['launches']),
                    'flops_per_launch': d['flops'] / max(1, d['launches']),
                    'share_of_eval_time': d['ms'] / tot_ms if tot_ms else None,
                    # the same launches replayed back to back between ONE event pair: no 3-5 us of event overhead per launch
                    'avg_launch_us_back_to_back': 1e3 * b2b_ms / max(1, d['launches']),
                    'achieved_back_to_back': d['flops'] / (b2b_ms * 1e-3) / 1e12 if b2b_ms > 0 else None,
                    'other_kernels_tflops': {f: (v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] else 0.0) for f, v in agg.items() if f != dom}}
        # the dominant HBM-bound class (north_star: "HBM GB/s AND MFMA utilisation against peak"): algorithmic bytes from the plan
        hbm_fn = {'groupnorm': 'gn_fused_kernel / gn_slab_kernel', 'layernorm': 'layernorm_kernel'}
        hb = {k: v for k, v in prof.items() if k in hbm_fn and v['launches'] > 0 and v['ms'] > 0}
        roofline_hbm = None
        if hb:
            hk = max(hb, key=lambda k: hb[k]['ms'])
            hv = hb[hk]
            gbs = hv['bytes'] / (hv['ms'] * 1e-3) / 1e9
            gbs_b2b = hv['bytes'] / (hv['ms_b2b'] * 1e-3) / 1e9 if hv['ms_b2b'] > 0 else None
            h_traffic, h_src = pmc_traffic_bytes_per_launch(tuple(hbm_fn[hk].split(' / ')), live, live_note, config_tag(args))      # (every kernel function of the class)
            roofline_hbm = {'bound': 'hbm', 'kernel': f'{hbm_fn[hk]} ({hk} class, one eps evaluation)', 'achieved': gbs, 'peak': PEAK_HBM_GBS,
                            'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS, 'traffic': h_traffic, 'traffic_source': h_src,
                            'launches_per_eval': hv['launches'], 'avg_launch_us': 1e3 * hv['ms'] / hv['launches'],
                            'bytes_per_launch': hv['bytes'] / hv['launches'], 'share_of_eval_time': hv['ms'] / tot_ms if tot_ms else None,
                            'avg_launch_us_back_to_back': 1e3 * hv['ms_b2b'] / hv['launches'],
                            'achieved_back_to_back': gbs_b2b, 'frac_back_to_back': gbs_b2b / PEAK_HBM_GBS if gbs_b2b else None,
                            'other_classes_gbs': {k: v['bytes'] / (v['ms'] * 1e-3) / 1e9 for k, v in hb.items() if k != hk}}
        loop_tflops = eps_flops * evals_per_step / (loop_ms * 1e-3) / 1e12
        result = {
            'metric': 'makeup-transfer images/sec @256x256, 50 DDIM steps' if args.res == 256 and args.ddim_steps == 50
                      else f'makeup-transfer images/sec @{args.res}x{args.res}, {args.ddim_steps} DDIM steps',
            'value': n_total * args.steps / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'batch={B}/GPU {args.res}x{args.res}, {args.ddim_steps} DDIM steps, eta 0, '
                                   + (f'makeup interpolation sweep: {args.batch} sources x {args.interp} alpha, ' if args.interp else '')
                                   + ('CFG 9.0 (2 evals/step)' if args.cfg else 'no CFG (1 eval/step)')
                                   + ', ControlNet+UNet every step, random-init SD-1.5 ControlNet weights, '
                                   + ('VAE-decoded images out' if args.decode else 'latents out'),
                       'global_batch': n_total, 'parallelism': f'batch-shard x{world}, 1 all-gather/step'},
            # evidence of the N > 1 run: what the process group itself reports
            'ranks_seen': torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            'backend': torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
            'rehearsal_single_device': os.environ.get('MKD_BENCH_SINGLE_DEVICE') == '1',
            'devices': sorted(set(dev_names)),
            'roofline': roofline,
            'roofline_hbm': roofline_hbm,
            'loop': {'ms_per_eval': loop_ms / evals_per_step, 'executed_tflop_per_eval': eps_flops / 1e12,
                     'mfma_tflops_whole_loop': loop_tflops, 'mfma_frac_whole_loop': loop_tflops / PEAK_BF16_TFLOPS,
                     'launches_per_eval': eng.step_launches(use_graph=bool(args.graph), cfg=bool(args.cfg)) / (2 if args.cfg else 1), 'launches_per_step': eng.step_launches(use_graph=bool(args.graph), cfg=bool(args.cfg)),
                     'plan_options': {k: eng.get_option(k) for k in ('tfm_tail', 'tfm_tail_min_rows', 'tfm_head', 'skip_fold', 'dec_lanes', 'ln_fly', 'gn_2k_min_hw', 'gn_slab_min_channels', 'xcd_auto_ratio', 'graph_steps')},
                     'launches_per_standalone_eps': eng.eps_launches(),
                     'device_gb': eng.device_bytes() / 1e9,
                     'hipgraph': bool(args.graph), 'vae_decode': bool(args.decode),
                     'vae_decode_tflop_per_batch': eng.decode_flops() / 1e12 if args.decode else None,
                     # what the cache hierarchy moves per evaluation (committed PMC summaries of the default workload)
                     'l2_traffic': (lambda t: None if t is None else dict(t, tb_per_s_over_the_loop=t['gb_per_eval'] / (loop_ms / evals_per_step)))(
                         l2_traffic_per_eval(config_tag(args), prof.get('attention', {}).get('launches', 0)))},
            'kernel_classes_ms_per_eval': {k: round(v['ms'], 4) for k, v in prof.items() if v['ms'] > 0},
            # the same classes with their launches replayed back to back between one event pair (no per-launch event overhead)
            'kernel_classes_ms_per_eval_back_to_back': {k: round(v['ms_b2b'], 4) for k, v in prof.items() if v['ms_b2b'] > 0},
        }
        if want_cpu:
            eng.prepare(hint[:1], ctx[:1])
            g1 = eng.eps(x_T[:1], torch.tensor([981], device=dev))
            sch1 = DDIMSchedule().make_ddim(args.cpu_steps)
            l1 = eng.sample(x_T[:1], sch1.ddim_timesteps, sch1.ddim_alphas, sch1.ddim_alphas_prev, sch1.ddim_sqrt_one_minus_alphas,
                            use_graph=bool(args.graph))
            result['cpu_baseline'] = cpu_baseline(sd_cpu, args.res, x_T[:1].cpu(), hint[:1].cpu(), ctx[:1].cpu(), g1, l1,
                                                  args.cpu_steps, args.ddim_steps)
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result), flush=True)
    mdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
