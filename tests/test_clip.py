"""Text conditioning stage: the oracle is pinned against transformers.CLIPTextModel itself
(importable offline; random weights — the pretrained ones are not on disk), libmkd is checked against the oracle."""
import ctypes as C
import json
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import clip as oc  # noqa: E402


def _hf_model(cfg):
    from transformers import CLIPTextConfig, CLIPTextModel
    return CLIPTextModel(CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.width, intermediate_size=cfg.intermediate,
                                        num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                                        max_position_embeddings=cfg.max_positions, hidden_act='quick_gelu', layer_norm_eps=cfg.ln_eps,
                                        bos_token_id=0, eos_token_id=2, pad_token_id=1)).eval()


def test_oracle_equals_transformers_cliptextmodel():
    cfg = oc.SMALL
    sd = oc.init_state_dict(cfg, 3)
    hf = _hf_model(cfg)
    pre = 'text_model.' if any(k.startswith('text_model.') for k in hf.state_dict()) else ''
    res = hf.load_state_dict({pre + k[len(oc.PREFIX):]: v for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all('position_ids' in k for k in res.missing_keys)
    tok = torch.randint(0, cfg.vocab_size, (3, 77), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf(input_ids=tok).last_hidden_state
    out = oc.encode_tokens(sd, cfg, tok)
    assert (out - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    short = oc.encode_tokens(sd, cfg, tok[:, :20])            # causal: a prefix is encoded identically
    assert torch.allclose(short, out[:, :20], atol=1e-5)


def test_full_config_matches_the_published_text_tower_size():
    n = sum(int(torch.tensor(s).prod()) for s in oc.param_spec(oc.FULL).values())
    assert n == 123_060_480                                    # openai/clip-vit-large-patch14 text model (no projection)


def test_tokenizer_wrapper_uses_local_files_only(tmp_path):
    """FrozenCLIPEmbedder tokenisation contract (max_length 77 padding, BOS/EOS) on a synthetic byte-pair vocabulary."""
    from makeupdiffuse_amd.clip import load_tokenizer
    words = ['<|startoftext|>', '<|endoftext|>', 'm</w>', 'a</w>', 'k</w>', 'e</w>', 'u</w>', 'p</w>', 'm', 'a', 'k', 'e', 'u', 'p', 'ma', 'ke</w>', 'up</w>',
             'make</w>', 't', 'r', 'n', 's', 'f', 't</w>', 'r</w>', 'n</w>', 's</w>', 'f</w>']
    vocab = {w: i for i, w in enumerate(words)}
    (tmp_path / 'vocab.json').write_text(json.dumps(vocab))
    (tmp_path / 'merges.txt').write_text('#version: 0.2\nm a\nk e</w>\nu p</w>\nma ke</w>\n')
    with pytest.raises(FileNotFoundError):
        load_tokenizer(str(tmp_path / 'missing'))
    tok = load_tokenizer(str(tmp_path))
    ids = tok(['make up', ''], truncation=True, max_length=77, padding='max_length', return_tensors='pt')['input_ids']
    assert tuple(ids.shape) == (2, 77)
    assert ids[0, :4].tolist() == [0, vocab['make</w>'], vocab['up</w>'], 1]
    assert ids[1, :2].tolist() == [0, 1] and set(ids[0, 4:].tolist()) == {1}


# ---- device ---------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize('B,T,heads,dh', [(2, 77, 12, 64), (1, 77, 2, 64), (3, 20, 4, 32), (1, 130, 2, 64)])
def test_causal_attention_kernel(B, T, heads, dh):
    from gpu_util import DEV, L, P, assert_close_bf16, bf, sync
    lib = L()
    g = torch.Generator().manual_seed(T + dh)
    d = heads * dh
    qkv = bf(torch.randn(B * T, 3 * d, generator=g))
    o = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    base = qkv.data_ptr()
    rc = lib.mkd_attention_causal(C.c_void_p(base), 3 * d, C.c_void_p(base + 2 * d), 3 * d, C.c_void_p(base + 4 * d), 3 * d, P(o), d,
                                  B, T, T, heads, dh, dh ** -0.5, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    q, k, v = (qkv[:, i * d:(i + 1) * d].float().view(B, T, heads, dh).transpose(1, 2) for i in range(3))
    mask = torch.full((T, T), float('-inf'), device=DEV).triu(1)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5 + mask, -1) @ v).transpose(1, 2).reshape(B * T, d)
    assert_close_bf16(o, ref, rel=8e-3, what='causal attention')


@gpu
def test_gemm_quick_gelu_epilogue():
    from gpu_util import DEV, L, P, assert_close_bf16, bf, sync
    lib = L()
    g = torch.Generator().manual_seed(9)
    M, N, K = 154, 256, 128
    a = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = (0.1 * torch.randn(N, generator=g)).to(DEV)
    c = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    rc = lib.mkd_gemm_bf16(P(a), K, P(w), K, P(bias), None, 0, 1, None, 0, 1.0, 3, P(c), N, 0, M, N, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None)
    assert rc == 0, lib.mkd_last_error()
    sync()
    assert_close_bf16(c, oc.quick_gelu(a.float() @ w.float().t() + bias), what='gemm + quick_gelu')


def _engine_with_clip(cfg, sd):
    from makeupdiffuse_amd.engine import ClipConfig, MkdEngine, NetConfig
    small_net = dict(model_channels=64, channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2, context_dim=64,
                     hint_widths=(16, 16, 32, 32, 32, 32, 64))
    eng = MkdEngine(NetConfig(**small_net), torch.device('cuda:0'))
    eng.configure_clip(ClipConfig(**cfg.__dict__))
    for k, v in sd.items():
        eng.load_weight(k, v)
    eng.finalize_clip()
    return eng


@gpu
@pytest.mark.parametrize('which', ['small', 'full'])
def test_clip_encode_vs_oracle(which):
    """bf16 residual stream through 2 / 12 pre-LN blocks vs the fp32 oracle: rel-L2 <= 1e-2 (small) / 2e-2 (full, the one-eval
    budget of SURVEY.md §8c), cosine >= 0.9995; padded batch rows and a T < 77 call included."""
    from gpu_util import rel_l2
    cfg = oc.SMALL if which == 'small' else oc.FULL
    sd = oc.init_state_dict(cfg, 5)
    eng = _engine_with_clip(cfg, sd)
    g = torch.Generator().manual_seed(2)
    tok = torch.randint(0, cfg.vocab_size, (3 if which == 'small' else 2, 77), generator=g)
    tok[0, 5:] = 1                                               # a short prompt: pad ids after EOS, still attended causally
    out = eng.encode_tokens(tok)
    torch.cuda.synchronize()
    ref = oc.encode_tokens(sd, cfg, tok)
    r = rel_l2(out, ref)
    cos = torch.nn.functional.cosine_similarity(out.cpu().flatten(), ref.flatten(), dim=0).item()
    assert torch.isfinite(out).all() and r <= (1e-2 if which == 'small' else 2e-2) and cos >= 0.9995, (r, cos)
    out20 = eng.encode_tokens(tok[:, :20])
    assert rel_l2(out20, ref[:, :20]) <= (1e-2 if which == 'small' else 2e-2)
    again = eng.encode_tokens(tok)                               # re-planned back to T = 77: bit-identical
    assert torch.equal(again, out)
    with pytest.raises(ValueError):
        eng.encode_tokens(torch.full((1, 77), cfg.vocab_size))
    eng.close()


@gpu
def test_clip_missing_weight_is_loud():
    from makeupdiffuse_amd.engine import ClipConfig, MkdEngine, NetConfig
    from makeupdiffuse_amd.lib import MkdError
    eng = MkdEngine(NetConfig(), torch.device('cuda:0'))
    with pytest.raises(MkdError):
        eng.encode_tokens(torch.zeros(1, 77, dtype=torch.long))   # not configured
    eng.configure_clip(ClipConfig(**oc.SMALL.__dict__))
    with pytest.raises(MkdError, match='weight not loaded'):
        eng.encode_tokens(torch.zeros(1, 77, dtype=torch.long))
    eng.close()
