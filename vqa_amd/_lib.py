"""ctypes binding of libvqafuse.so (the C ABI declared in include/vqaf.h).

The library is the product path: there is no CPU or PyTorch fallback.  Loading fails loudly
if the shared object has not been built (``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C vqa_amd/csrc``), and every call checks the status code the C ABI returns.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvqafuse.so")

F32 = 0
BF16 = 1
NOISE_NONE, NOISE_GAUSSIAN, NOISE_UNIFORM = 0, 1, 2

_P = c_void_p


class PoolMod(ctypes.Structure):
    """vqaf_pool_mod of include/vqaf.h."""
    _fields_ = [("seq", c_void_p), ("idx", c_void_p), ("noise", c_void_p), ("rng_state", c_void_p),
                ("noise_scale", c_float), ("alpha", c_float), ("noise_kind", c_int), ("N", c_int),
                ("K", c_int), ("col0", c_int)]


class FoldDesc(ctypes.Structure):
    """vqaf_fold_desc of include/vqaf.h."""
    _fields_ = [("q", c_void_p), ("in_proj_weight", c_void_p), ("in_proj_bias", c_void_p),
                ("qproj_ws", c_void_p), ("u", c_void_p)]


class ScoresDesc(ctypes.Structure):
    """vqaf_scores_desc of include/vqaf.h."""
    _fields_ = [("seq", c_void_p), ("u", c_void_p), ("s", c_void_p), ("N", c_int)]


class SelectDesc(ctypes.Structure):
    """vqaf_select_desc of include/vqaf.h."""
    _fields_ = [("s", c_void_p), ("w_in", c_void_p), ("pos", c_void_p), ("src", c_void_p), ("w_out", c_void_p),
                ("top", c_void_p), ("bottom", c_void_p), ("pos_out", c_void_p), ("src_out", c_void_p),
                ("salt", c_uint64), ("N", c_int), ("K", c_int), ("adv_mode", c_int), ("k_adv", c_int)]


class FusedMod(ctypes.Structure):
    """vqaf_fused_mod of include/vqaf.h."""
    _fields_ = [("seq", c_void_p), ("u", c_void_p), ("pos", c_void_p), ("src", c_void_p), ("noise", c_void_p),
                ("noise_rng_state", c_void_p), ("idx_out", c_void_p), ("w_out", c_void_p), ("salt", c_uint64),
                ("alpha", c_float), ("noise_scale", c_float), ("noise_kind", c_int), ("N", c_int), ("K", c_int),
                ("adv_mode", c_int), ("k_adv", c_int), ("col0", c_int), ("noise_counter_offset", c_int)]


class GemmDesc(ctypes.Structure):
    """vqaf_gemm_desc of include/vqaf.h."""
    _fields_ = [("A", c_void_p), ("Bm", c_void_p), ("bias", c_void_p), ("C", c_void_p),
                ("a_sg", c_long), ("a_sm", c_long), ("a_sk", c_long), ("b_sg", c_long), ("b_sk", c_long),
                ("b_sn", c_long), ("bias_sg", c_long), ("c_sg", c_long), ("c_sm", c_long),
                ("a_dtype", c_int), ("b_dtype", c_int), ("bias_dtype", c_int),
                ("G", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("alpha", c_float), ("store", c_int), ("c_dtype", c_int)]


class DenseFwdMod(ctypes.Structure):
    """vqaf_dense_fwd_mod of include/vqaf.h."""
    _fields_ = [("seq", c_void_p), ("s", c_void_p), ("p", c_void_p), ("z", c_void_p), ("N", c_int)]


class DenseBwdMod(ctypes.Structure):
    """vqaf_dense_bwd_mod of include/vqaf.h."""
    _fields_ = [("seq", c_void_p), ("p", c_void_p), ("dp", c_void_p), ("u", c_void_p), ("t", c_void_p),
                ("du", c_void_p), ("grad_seq", c_void_p), ("N", c_int)]


class PoolBwdMod(ctypes.Structure):
    """vqaf_pool_bwd_mod of include/vqaf.h."""
    _fields_ = [("idx", c_void_p), ("grad_seq", c_void_p), ("alpha", c_float), ("N", c_int), ("K", c_int),
                ("col0", c_int)]


# name -> (restype, argtypes); mirrors include/vqaf.h one to one.
SIGNATURES = {
    "vqaf_version": (c_int, []),
    "vqaf_status_string": (c_char_p, [c_int]),
    "vqaf_fold_query": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "vqaf_fold_query_multi": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vqaf_fold_query_multi_advance": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_uint64, _P]),
    "vqaf_head_scores": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vqaf_head_scores_multi": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vqaf_attn_weights": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "vqaf_score_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vqaf_score": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t,
                           _P, _P]),
    "vqaf_score_noalloc": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vqaf_topk": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "vqaf_adv_swap": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vqaf_adv_swap_random": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "vqaf_select": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_uint64, _P, _P, _P,
                            _P, _P, _P]),
    "vqaf_select_multi": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "vqaf_rng_advance": (c_int, [_P, c_uint64, _P]),
    "vqaf_gather_mix": (c_int, [_P, _P, _P, c_int, c_float, _P, c_float, c_int, c_int, c_int,
                                c_int, c_int, c_int, _P, _P]),
    "vqaf_gather_mix_bwd": (c_int, [_P, _P, c_float, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                    _P]),
    "vqaf_mean_pool": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "vqaf_mean_pool_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vqaf_gather_mix_pool": (c_int, [_P, _P, _P, c_int, c_float, _P, c_float, c_int, c_int, c_int,
                                     c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "vqaf_gather_mix_pool_bwd": (c_int, [_P, c_int, c_int, c_int, _P, c_float, c_int, c_int, c_int,
                                         c_int, c_int, _P, _P]),
    "vqaf_softmax_heads": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "vqaf_softmax_heads_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "vqaf_weighted_token_sum": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vqaf_token_outer": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vqaf_gather_mix_pool_multi": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "vqaf_gather_mix_pool_bwd_multi": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "vqaf_prune_mix_supported": (c_int, [c_int, c_int, c_int]),
    "vqaf_prune_mix": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_uint64, _P, c_int,
                               c_float, _P, c_int, c_float, c_int, _P, _P, _P]),
    "vqaf_fused_prune_pool_multi": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P]),
    "vqaf_fused_prune_pool_onepass_multi": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int,
                                                    c_int, _P]),
    "vqaf_fused_prune_pool_onepass_supported": (c_int, [c_int, _P, c_int, c_int]),
    "vqaf_gemm_f32": (c_int, [_P, c_int, c_long, c_long, c_long, _P, c_int, c_long, c_long, c_long,
                              _P, c_int, c_long, _P, c_long, c_long, c_int, c_int, c_int, c_int,
                              c_float, c_int, _P]),
    "vqaf_gemm_f32_atomic": (c_int, [_P, c_int, c_long, c_long, c_long, _P, c_int, c_long, c_long, c_long,
                                     _P, c_int, c_long, _P, c_long, c_long, c_int, c_int, c_int, c_int,
                                     c_float, c_int, _P]),
    "vqaf_gemm_f32_atomic_sets": (c_int, [c_int, _P, c_int, c_long, c_long, c_long, _P, c_int, c_long, c_long, c_long,
                                          _P, c_int, c_long, _P, c_long, c_long, c_int, c_int, c_int, c_int,
                                          c_float, c_int, _P]),
    "vqaf_gemm_f32_group": (c_int, [c_int, _P, _P]),
    "vqaf_softmax_wsum_multi": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "vqaf_token_bwd_multi": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vqaf_gemm_bf16_supported": (c_int, [c_long, c_int, c_long, c_int, c_int, c_int, c_int]),
    "vqaf_clipped_sum_adamw_group": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, c_long, _P, c_float, c_float,
                                             _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, _P, _P]),
    "vqaf_dp_adamw_ranges": (c_int, [c_int, _P, _P, _P, _P, c_long, c_float, c_float, _P, _P, _P, _P, _P, c_float,
                                     c_float, c_float, c_float, c_float, _P, _P]),
    "vqaf_linear_groupnorm_supported": (c_int, [c_long, c_long, c_int, c_int, c_int, c_int]),
    "vqaf_linear_dgrad_groupnorm_supported": (c_int, [c_long, c_long, c_int, c_int, c_int, c_int]),
    "vqaf_linear_dgrad_groupnorm_bwd": (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int,
                                                c_int, c_float, _P, _P, _P]),
    "vqaf_linear_groupnorm_fwd": (c_int, [_P, c_long, _P, c_long, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int,
                                          c_float, c_int, c_float, c_uint64, _P, _P, _P, _P, _P]),
    "vqaf_gemm_bf16_wgrad_group": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, _P, _P]),
    "vqaf_gemm_bf16": (c_int, [_P, c_long, c_int, _P, c_long, c_int, c_int, c_int, c_int, _P, c_int, _P, c_long,
                               c_int, c_int, _P, _P]),
    "vqaf_ghost_sqnorm_linear": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vqaf_ghost_sqnorm_affine": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "vqaf_clip_coef": (c_int, [_P, c_int, c_float, c_float, _P, _P, _P]),
    "vqaf_ghost_clip": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_int, _P, _P, _P,
                                _P]),
    "vqaf_ghost_clip_zero": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_int, _P, _P, _P,
                                     _P, _P, c_int, _P]),
    "vqaf_ghost_clip_zero_bump": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_int, _P, _P,
                                          _P, _P, _P, c_int, _P, _P]),
    "vqaf_clipped_sum_linear": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "vqaf_clipped_sum_affine": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "vqaf_clipped_sum_affine_multi": (c_int, [c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, _P]),
    "vqaf_noise_scale": (c_int, [_P, _P, _P, c_float, c_float, c_size_t, _P, _P]),
    "vqaf_dp_adamw": (c_int, [_P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, c_size_t, c_float, c_float,
                              c_float, c_float, c_float, _P, c_int, _P]),
    "vqaf_zero_multi": (c_int, [_P, _P, c_int, _P]),
    "vqaf_groupnorm_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_int, _P, _P,
                                   _P, c_float, ctypes.c_uint64, _P, _P]),
    "vqaf_groupnorm_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, _P,
                                   _P]),
    "vqaf_ce_conf": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, _P]),
    "vqaf_ce_mean": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_long, c_int, c_float, _P, _P, _P]),
    "vqaf_scale_cast": (c_int, [_P, _P, c_size_t, c_int, _P, _P]),
    "vqaf_attn_fwd": (c_int, [_P, _P, _P, c_long, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, c_long,
                              _P, _P]),
    "vqaf_attn_bwd": (c_int, [_P, _P, _P, c_long, _P, _P, c_long, _P, c_int, c_int, c_int, c_int, c_float, c_float,
                              _P, _P, _P, _P, c_long, _P, _P]),
    "vqaf_layernorm_fwd": (c_int, [_P, _P, _P, _P, c_int, c_long, c_int, c_float, c_int, _P, _P, _P, _P, _P]),
    "vqaf_layernorm_bwd_workspace_bytes": (c_size_t, [c_long, c_int]),
    "vqaf_layernorm_bwd": (c_int, [_P, _P, _P, c_int, _P, _P, _P, c_long, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "vqaf_gelu_fwd": (c_int, [_P, c_size_t, c_int, _P, _P]),
    "vqaf_gelu_bwd": (c_int, [_P, _P, c_size_t, c_int, _P, _P]),
    "vqaf_embed_fwd": (c_int, [_P, _P, c_long, _P, _P, c_long, _P, _P, c_long, c_int, c_int, c_int, _P, c_int, _P]),
    "vqaf_embed_bwd": (c_int, [_P, c_int, _P, c_int, c_int, c_long, c_long, _P, c_int, _P]),
    "vqaf_linear_nt_supported": (c_int, [c_long, c_long, c_int, c_long, c_int, c_int, c_int]),
    "vqaf_linear_nt": (c_int, [_P, c_long, _P, c_long, c_int, _P, c_int, c_int, c_int, c_int, _P, c_long, c_int, _P,
                               c_long, _P]),
    "vqaf_wgrad_tn_supported": (c_int, [c_long, c_long, c_long, c_int, c_int, c_int]),
    "vqaf_wgrad_tn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vqaf_wgrad_tn": (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, _P, c_long, c_int, _P, c_int, _P, c_size_t,
                              _P]),
}

_lib = None


class VqafError(RuntimeError):
    """A libvqafuse entry point returned a negative status."""


def load(path: str | None = None) -> ctypes.CDLL:
    """Loads the shared library once and types every symbol of SIGNATURES."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            f"{p} is missing: build the HIP extension first (make -C vqa_amd/csrc, or "
            "__graft_entry__.build()). There is no CPU fallback for the fusion kernels.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().vqaf_status_string(status).decode()
        raise VqafError(f"{what} failed with status {status}: {msg}")


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"libvqafuse supports float32 and bfloat16 tensors, got {t.dtype}")


def torch_dtype(code: int) -> torch.dtype:
    return torch.bfloat16 if code == BF16 else torch.float32


def ptr(t: torch.Tensor | None) -> int | None:
    """Device address of a contiguous CUDA/HIP tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(
            "libvqafuse kernels run on the GPU only; got a tensor on "
            f"{t.device}. Move the module and its inputs to a HIP device.")
    if not t.is_contiguous():
        raise RuntimeError("libvqafuse needs contiguous tensors")
    return t.data_ptr()


def stream_ptr(device: torch.device | None = None) -> int:
    return torch.cuda.current_stream(device).cuda_stream
