"""ctypes binding of libmkd.so (include/mkd.h).  There is NO fallback: a missing library or a
missing GPU is an error, never a silent CPU path."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MKD_LIB_PATH') or os.path.join(HERE, 'libmkd.so')      # override: A/B experiments only

ABI_VERSION = 1


class MkdError(RuntimeError):
    pass


class VaeConfigC(C.Structure):
    _fields_ = [('z_channels', C.c_int32), ('embed_dim', C.c_int32), ('ch', C.c_int32), ('n_levels', C.c_int32),
                ('ch_mult', C.c_int32 * 8), ('num_res_blocks', C.c_int32), ('out_ch', C.c_int32)]


class ClipConfigC(C.Structure):
    _fields_ = [('vocab_size', C.c_int32), ('max_positions', C.c_int32), ('width', C.c_int32), ('layers', C.c_int32),
                ('heads', C.c_int32), ('intermediate', C.c_int32), ('ln_eps', C.c_float)]


class NetConfigC(C.Structure):
    _fields_ = [
        ('in_channels', C.c_int32), ('out_channels', C.c_int32), ('hint_channels', C.c_int32),
        ('model_channels', C.c_int32), ('num_res_blocks', C.c_int32), ('n_levels', C.c_int32),
        ('channel_mult', C.c_int32 * 8), ('n_attention_resolutions', C.c_int32),
        ('attention_resolutions', C.c_int32 * 8), ('num_heads', C.c_int32),
        ('transformer_depth', C.c_int32), ('context_dim', C.c_int32), ('hint_widths', C.c_int32 * 7),
    ]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_int64

# name -> (restype, argtypes); every symbol include/mkd.h declares
SIGNATURES = {
    'mkd_last_error': (C.c_char_p, []),
    'mkd_abi_version': (_I, []),
    'mkd_grouped_launches_available': (_I, []),
    'mkd_ctx_create': (_I, [C.POINTER(NetConfigC), C.POINTER(_P)]),
    'mkd_ctx_destroy': (None, [_P]),
    'mkd_load_weight': (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_L)]),
    'mkd_weights_finalize': (_I, [_P]),
    'mkd_param_count': (_L, [_P, _I]),
    'mkd_param_total': (_I, [_P]),
    'mkd_param_name': (C.c_char_p, [_P, _I]),
    'mkd_param_shape': (_I, [_P, _I, C.POINTER(_L)]),
    'mkd_ctx_set_option': (_I, [_P, C.c_char_p, C.c_double]),
    'mkd_ctx_get_option': (_I, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    'mkd_debug_tfm_trace': (_I, [_P]),
    'mkd_debug_attn_trace': (_I, [_P]),
    'mkd_live_contexts': (_I, []),
    'mkd_prepare': (_I, [_P, _I, _I, _I, _P, _P, C.POINTER(_F), _I, _P]),
    'mkd_prepare_interp': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, C.POINTER(_F), _I, _P]),
    'mkd_eps': (_I, [_P, _P, _P, _P, _P]),
    'mkd_ddim_step': (_I, [_P, _P, _P, _F, _F, _F, _F, _F, _P, _F, _P, _P, _L, _P]),
    'mkd_sample': (_I, [_P, _P, _I, _I, C.POINTER(_L), C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _F, _P, _I, _P]),
    'mkd_sample_eta': (_I, [_P, _P, _I, _I, C.POINTER(_L), C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _P, _F, _F, _P, _I, _P]),
    'mkd_vae_configure': (_I, [_P, C.POINTER(VaeConfigC)]),
    'mkd_vae_finalize': (_I, [_P]),
    'mkd_decode': (_I, [_P, _P, _I, _I, _I, _F, _P, _P]),
    'mkd_decode_flops': (C.c_double, [_P]),
    'mkd_clip_configure': (_I, [_P, C.POINTER(ClipConfigC)]),
    'mkd_clip_finalize': (_I, [_P]),
    'mkd_clip_encode': (_I, [_P, _P, _I, _I, _P, _P]),
    'mkd_kind_count': (_I, []),
    'mkd_kind_name': (C.c_char_p, [_I]),
    'mkd_eps_profile': (_I, [_P, _P, _P, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I), C.c_char_p]),
    'mkd_eps_profile2': (_I, [_P, _P, _P, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I), C.POINTER(C.c_double),
                              C.POINTER(C.c_double), C.c_char_p]),
    'mkd_eps_flops': (C.c_double, [_P]),
    'mkd_eps_launches': (_I, [_P]),
    'mkd_step_launches': (_I, [_P]),
    'mkd_step_launches_ex': (_I, [_P, _I, _I]),
    'mkd_device_bytes': (_L, [_P]),
    'mkd_gemm_bf16': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _F, _I, _P, _I, _I, _I, _I, _I,
                           _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'mkd_conv3x3_fold_bf16': (_I, [_P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'mkd_gemm_gnstat_bf16': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _F, _I, _P, _I, _I, _I, _I, _I,
                                  _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    'mkd_gemm_groupnorm_bf16': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _F, _P, _I, _I, _I, _I, _I,
                                     _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _F, _I, _P, _I, _P]),
    'mkd_gn_colstats': (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'mkd_gn_apply_stats': (_I, [_P, _I, _P, _P, _F, _I, _P, _I, _I, _I, _I, _P, _P]),
    'mkd_gemm_force_tile': (_I, [_I]),
    'mkd_gemm_set_xcd_mode': (_I, [_I]),
    'mkd_debug_poison': (_I, [_P]),
    'mkd_gemm_set_override': (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    'mkd_gemm_cfg_supported': (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    'mkd_fold_layernorm': (_I, [_P, _P, _P, _P, _I, _I, _P, _I, _I, _P, _P, _P]),
    'mkd_gemm_ln_bf16': (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _F, _I, _P, _I, _I, _I, _I, _P]),
    'mkd_gemm_rowstats_bf16': (_I, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P, _I, C.POINTER(_I), _P]),
    'mkd_groupnorm': (_I, [_P, _I, _P, _P, _F, _I, _P, _I, _I, _I, _I, _I, _P]),
    'mkd_tfm_tail_create': (_I, [_I] + [_P] * 15 + [C.POINTER(_P)]),
    'mkd_tfm_tail_destroy': (None, [_P]),
    'mkd_tfm_tail_set_context': (_I, [_P, _P, _I, _I, _I, _P]),
    'mkd_tfm_tail_run': (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P]),
    'mkd_tfm_head_create': (_I, [_I] + [_P] * 9 + [C.POINTER(_P)]),
    'mkd_tfm_head_destroy': (None, [_P]),
    'mkd_tfm_head_run': (_I, [_P, _P, _I, _F, _P, _P, _I, _I, _P]),
    'mkd_layernorm': (_I, [_P, _P, _P, _F, _P, _I, _I, _P]),
    'mkd_layernorm_ld': (_I, [_P, _I, _P, _P, _F, _P, _I, _I, _P]),
    'mkd_attention': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    'mkd_attention_causal': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    'mkd_geglu': (_I, [_P, _P, _I, _I, _P]),
    'mkd_conv3x3_direct': (_I, [_P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mkd_pack_conv_weight': (_I, [_P, _P, _I, _I, _I, _I, _P]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen libmkd.so and bind every declared symbol; raises MkdError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MkdError(f'{LIB_PATH} is missing: build it with `python -m makeupdiffuse_amd.build` '
                       '(there is no CPU fallback for the hot path)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mkd_abi_version() != ABI_VERSION:
        raise MkdError(f'libmkd.so ABI {lib.mkd_abi_version()} != binding ABI {ABI_VERSION}: rebuild')
    _lib = lib
    return lib


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        msg = load().mkd_last_error()
        raise MkdError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')
