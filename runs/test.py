#!/usr/bin/env python
"""runs/test.py — inference entry point with the reference's shape (reference runs/test.py:27,59-64):
create_model(yaml) -> load_state_dict(ckpt) -> for batch in data: model.test_step(batch, i).

The reference hard-codes its constants in a "modify" block, needs the MT-Dataset, a trained checkpoint, CLIP and
PyTorch-Lightning; none exist offline, so every input is an argument here and the synthetic mode draws the batch
dict (src_img / ref_img / txt_emb) of SURVEY.md §8d.  One process per GPU; under torchrun the pair list is sharded.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from makeupdiffuse_amd import dist as mdist  # noqa: E402
from makeupdiffuse_amd.config import create_model, load_state_dict  # noqa: E402


def synthetic_batch(lo, hi, res, ctx_dim):
    src, ref, txt = [], [], []
    for i in range(lo, hi):
        g = torch.Generator().manual_seed(5678 + i)
        src.append(torch.rand(1, 3, res, res, generator=g)); ref.append(torch.rand(1, 3, res, res, generator=g))
        g = torch.Generator().manual_seed(91011 + i)
        txt.append(torch.randn(1, 77, ctx_dim, generator=g))
    return {'src_img': torch.cat(src), 'ref_img': torch.cat(ref), 'txt_emb': torch.cat(txt),
            'txt': ['makeup transfer'] * (hi - lo), 'img_name': [f'{i:04d}&{i:04d}' for i in range(lo, hi)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default=os.path.join(os.path.dirname(__file__), '..', 'diffmodels', 'test_diffusion_makeup.yaml'))
    ap.add_argument('--ckpt', default=None, help='upstream-named state_dict (.safetensors / tensor-only .ckpt); default: seeded random init')
    ap.add_argument('--pairs', type=int, default=2)
    ap.add_argument('--batch-size', type=int, default=1)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--ddim-steps', type=int, default=None)
    ap.add_argument('--only-mid-control', action='store_true')
    ap.add_argument('--out', default='./results')
    ap.add_argument('--data-root', default=None, help='folder with images/ and a pairs file (reference TestFixed_Dataset layout)')
    ap.add_argument('--pairs-file', default='test_0412.txt')
    ap.add_argument('--tokenizer', default=None, help='local CLIP tokenizer directory (vocab.json + merges.txt): prompts then go through the text encoder')
    ap.add_argument('--seed', type=int, default=None, help='start noise x_T drawn per PAIR from seed + pair index (results independent of '
                    'batch size / sharding); default: torch.randn like the reference')
    ap.add_argument('--txt-emb', default=None, help='.pt/.safetensors with a [1,77,768] tensor: the CLIP embedding of the prompt (offline stand-in)')
    args = ap.parse_args()

    rank, world, local = mdist.init_from_env()
    model = create_model(args.config).cpu()
    if args.ddim_steps is not None:
        model.ddim_steps = args.ddim_steps
    if args.ckpt:
        model.load_state_dict(load_state_dict(args.ckpt, location='cpu'))
    torch.cuda.set_device(local)
    model.cuda(local)
    if not args.ckpt:
        model.engine.init_random(seed=0)
    model.only_mid_control = args.only_mid_control
    if args.tokenizer and model.cond_stage_model is not None:
        from makeupdiffuse_amd.clip import load_tokenizer
        model.cond_stage_model.tokenizer = load_tokenizer(args.tokenizer)
    use_clip = model.cond_stage_model is not None and model.cond_stage_model.tokenizer is not None
    if not use_clip:
        model.uncond_embedding = torch.zeros(1, 77, model.net_config.context_dim)   # stands for CLIP("") without a tokenizer
    model.eval()

    model.saved_dir = args.out
    model.test_pairs_file = os.path.join(args.out, f'test_pairs_rank{rank}.txt')
    dataset = None
    if args.data_root:
        from makeupdiffuse_amd.imageio import PairFolderDataset, collate
        dataset = PairFolderDataset(args.data_root, args.pairs_file, (args.res, args.res))
        args.pairs = len(dataset)
    txt_emb = None
    if args.txt_emb:
        t = load_state_dict(args.txt_emb)
        txt_emb = (next(iter(t.values())) if isinstance(t, dict) else t).float().reshape(1, 77, -1)
    lo, hi = mdist.shard_range(args.pairs, rank, world)
    os.makedirs(args.out, exist_ok=True)
    model.on_test_epoch_start()
    for b0 in range(lo, hi, args.batch_size):
        b1 = min(hi, b0 + args.batch_size)
        if dataset is not None:
            batch = collate([dataset[i] for i in range(b0, b1)])
            if not use_clip:
                g = torch.Generator().manual_seed(91011)
                e = txt_emb if txt_emb is not None else torch.randn(1, 77, model.net_config.context_dim, generator=g)
                batch['txt_emb'] = e.expand(b1 - b0, -1, -1).contiguous()
        else:
            batch = synthetic_batch(b0, b1, args.res, model.net_config.context_dim)
            if use_clip:
                del batch['txt_emb']          # 'txt' -> tokenizer -> mkd_clip_encode
        x_T = None
        if args.seed is not None:
            h8 = args.res // 8
            x_T = torch.cat([torch.randn(1, model.channels, h8, h8, generator=torch.Generator().manual_seed(args.seed + i))
                             for i in range(b0, b1)]).cuda(local)
        out = model.test_step(batch, b0, x_T=x_T)
        model.on_test_batch_end(out, batch, b0)
        torch.save({k: v for k, v in out.items() if isinstance(v, torch.Tensor)},
                   os.path.join(args.out, f'latents_{b0:04d}.pt'))
        print(f'[rank {rank}] pairs {b0}..{min(hi, b0 + args.batch_size) - 1}: ' +
              ', '.join(f'{k} {tuple(v.shape)}' for k, v in out.items() if isinstance(v, torch.Tensor)), flush=True)
    mdist.barrier()


if __name__ == '__main__':
    main()
