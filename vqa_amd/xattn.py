"""Dense single-query cross-attention: the attention OUTPUT of
nn.MultiheadAttention(query=q[:,None], key=seq, value=seq) used by the non-pruning branch of
CoAttention / Attention (utils/fusion.py:257-268, 173-179), forward and backward.

One query per sample lets both projections fold out of the token loop (see include/vqaf.h), so
all work proportional to the number of tokens is four bandwidth-bound HIP kernels; what is left
is a handful of (B, E)-sized matrix products, done with torch so that autograd delivers the
gradients of in_proj / out_proj exactly as the reference's module does.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import check, dtype_code, load, ptr, stream_ptr


def _head_scores(seq: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """s (B,N,H) f32 = seq . u."""
    B, N, D = seq.shape
    H = u.shape[1]
    lib, st = load(), stream_ptr(seq.device)
    s = torch.empty(B, N, H, dtype=torch.float32, device=seq.device)
    check(lib.vqaf_head_scores(ptr(seq), ptr(u), B, N, D, H, dtype_code(seq), ptr(s), st), "vqaf_head_scores")
    return s


class _SingleQueryTokens(torch.autograd.Function):
    """z[b,h,:] = sum_n softmax_n(seq[b,n,:] . u[b,h,:]) seq[b,n,:]  and its backward."""

    @staticmethod
    def forward(ctx, seq, u):
        B, N, D = seq.shape
        H = u.shape[1]
        seq_c, u_c = seq.contiguous(), u.contiguous().float()
        lib, st = load(), stream_ptr(seq.device)
        s = _head_scores(seq_c, u_c)
        p = torch.empty(B, H, N, dtype=torch.float32, device=seq.device)
        check(lib.vqaf_softmax_heads(ptr(s), B, N, H, ptr(p), st), "vqaf_softmax_heads")
        z = torch.empty(B, H, D, dtype=torch.float32, device=seq.device)
        check(lib.vqaf_weighted_token_sum(ptr(seq_c), ptr(p), B, N, D, H, dtype_code(seq_c), ptr(z), st),
              "vqaf_weighted_token_sum")
        ctx.save_for_backward(seq_c, u_c, p)
        return z

    @staticmethod
    def backward(ctx, dz):
        seq_c, u_c, p = ctx.saved_tensors
        B, N, D = seq_c.shape
        H = u_c.shape[1]
        lib, st = load(), stream_ptr(seq_c.device)
        t = dz.contiguous().float()
        dp = _head_scores(seq_c, t)
        ds = torch.empty_like(p)
        check(lib.vqaf_softmax_heads_bwd(ptr(p), ptr(dp), B, N, H, ptr(ds), st), "vqaf_softmax_heads_bwd")
        du = None
        if ctx.needs_input_grad[1]:
            du = torch.empty_like(u_c)
            check(lib.vqaf_weighted_token_sum(ptr(seq_c), ptr(ds), B, N, D, H, dtype_code(seq_c), ptr(du), st),
                  "vqaf_weighted_token_sum")
        gseq = None
        if ctx.needs_input_grad[0]:
            gseq = torch.empty_like(seq_c)
            check(lib.vqaf_token_outer(ptr(ds), ptr(u_c), ptr(p), ptr(t), B, N, D, H, dtype_code(gseq),
                                       ptr(gseq), st), "vqaf_token_outer")
        return gseq, du


def single_query_attention(mha: nn.MultiheadAttention, query: torch.Tensor, seq: torch.Tensor) -> torch.Tensor:
    """(B,E) attention output for query (B,E) over seq (B,N,E); differentiable w.r.t. query, seq and
    every parameter of `mha` (the key bias receives an exactly zero gradient: it shifts all scores
    of a head equally)."""
    B, N, E = seq.shape
    H = mha.num_heads
    dh = E // H
    w_q, w_k, w_v = mha.in_proj_weight.float().chunk(3)
    b_q, _b_k, b_v = mha.in_proj_bias.float().chunk(3)
    qp = F.linear(query.reshape(B, E).float(), w_q, b_q)                                     # (B,E)
    u = torch.einsum("bhj,hjd->bhd", qp.view(B, H, dh), w_k.view(H, dh, E)) / math.sqrt(dh)  # (B,H,E)
    z = _SingleQueryTokens.apply(seq, u)                                                     # (B,H,E)
    attn = torch.einsum("bhd,hjd->bhj", z, w_v.view(H, dh, E)).reshape(B, E) + b_v
    out = F.linear(attn, mha.out_proj.weight.float(), mha.out_proj.bias.float())
    return out.to(seq.dtype)
