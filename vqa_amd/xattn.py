"""Dense single-query cross-attention: the attention OUTPUT of
nn.MultiheadAttention(query=q[:,None], key=seq, value=seq) used by the non-pruning branch of
CoAttention / Attention (utils/fusion.py:257-268, 173-179), forward and backward.

One query per sample lets both projections fold out of the token loop (see include/vqaf.h), so
all work proportional to the number of tokens is four bandwidth-bound HIP kernels; what is left
is a handful of (B, E)-sized matrix products, each one strided launch of the library's f32 GEMM,
with a hand-derived backward that delivers the gradients of query, tokens, in_proj and out_proj
as the reference's module does.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import check, dtype_code, load, ptr, stream_ptr


def _head_scores(seq: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """s (B,N,H) f32 = seq . u."""
    B, N, D = seq.shape
    H = u.shape[1]
    lib, st = load(), stream_ptr(seq.device)
    s = torch.empty(B, N, H, dtype=torch.float32, device=seq.device)
    check(lib.vqaf_head_scores(ptr(seq), ptr(u), B, N, D, H, dtype_code(seq), ptr(s), st), "vqaf_head_scores")
    return s


def _tokens_forward(seq_c: torch.Tensor, u_c: torch.Tensor):
    """z[b,h,:] = sum_n softmax_n(seq[b,n,:] . u[b,h,:]) seq[b,n,:]; returns (z, p)."""
    B, N, D = seq_c.shape
    H = u_c.shape[1]
    lib, st = load(), stream_ptr(seq_c.device)
    s = _head_scores(seq_c, u_c)
    p = torch.empty(B, H, N, dtype=torch.float32, device=seq_c.device)
    check(lib.vqaf_softmax_heads(ptr(s), B, N, H, ptr(p), st), "vqaf_softmax_heads")
    z = torch.empty(B, H, D, dtype=torch.float32, device=seq_c.device)
    check(lib.vqaf_weighted_token_sum(ptr(seq_c), ptr(p), B, N, D, H, dtype_code(seq_c), ptr(z), st),
          "vqaf_weighted_token_sum")
    return z, p


def _tokens_backward(seq_c, u_c, p, t, need_seq: bool, need_u: bool):
    """Backward of _tokens_forward for dz = t (B,H,D) f32: (grad_seq or None, du or None)."""
    B, N, D = seq_c.shape
    H = u_c.shape[1]
    lib, st = load(), stream_ptr(seq_c.device)
    dp = _head_scores(seq_c, t)
    ds = torch.empty_like(p)
    check(lib.vqaf_softmax_heads_bwd(ptr(p), ptr(dp), B, N, H, ptr(ds), st), "vqaf_softmax_heads_bwd")
    du = None
    if need_u:
        du = torch.empty_like(u_c)
        check(lib.vqaf_weighted_token_sum(ptr(seq_c), ptr(ds), B, N, D, H, dtype_code(seq_c), ptr(du), st),
              "vqaf_weighted_token_sum")
    gseq = None
    if need_seq:
        gseq = torch.empty_like(seq_c)
        check(lib.vqaf_token_outer(ptr(ds), ptr(u_c), ptr(p), ptr(t), B, N, D, H, dtype_code(gseq),
                                   ptr(gseq), st), "vqaf_token_outer")
    return gseq, du


class _SingleQueryTokens(torch.autograd.Function):
    """z[b,h,:] = sum_n softmax_n(seq[b,n,:] . u[b,h,:]) seq[b,n,:]  and its backward."""

    @staticmethod
    def forward(ctx, seq, u):
        seq_c, u_c = seq.contiguous(), u.contiguous().float()
        z, p = _tokens_forward(seq_c, u_c)
        ctx.save_for_backward(seq_c, u_c, p)
        return z

    @staticmethod
    def backward(ctx, dz):
        seq_c, u_c, p = ctx.saved_tensors
        return _tokens_backward(seq_c, u_c, p, dz.contiguous().float(), ctx.needs_input_grad[0],
                                ctx.needs_input_grad[1])


def _zeroed(dev, *shapes):
    """f32 tensors of the given shapes carved out of ONE zero-filled allocation (one fill launch): outputs of
    _gemm(..., into_zeros=True), which then adds into them instead of clearing each one with its own memset."""
    sizes = [(math.prod(sh) + 3) // 4 * 4 for sh in shapes]  # 16-byte aligned pieces
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    out, o = [], 0
    for sh, n in zip(shapes, sizes):
        out.append(flat[o:o + math.prod(sh)].view(sh))
        o += n
    return out


def _gemm(A, a_s, Bm, b_s, C, c_s, G, M, N, K, alpha=1.0, bias=None, bias_sg=0, into_zeros=False):
    """C[g][m][n] = alpha sum_k A[g][m][k] B[g][k][n] (+ bias[g][n]) through vqaf_gemm_f32_atomic.  A, Bm, C and
    bias are (tensor, element offset) pairs addressed by element strides a_s = (s_g, s_m, s_k),
    b_s = (s_g, s_k, s_n), c_s = (s_g, s_m); transposes and head slices cost nothing."""
    def addr(pair):
        t, off = pair
        return t.data_ptr() + off * t.element_size()

    def code(t):
        return _lib.BF16 if t.dtype == torch.bfloat16 else _lib.F32

    (at, _), (bt, _), (ct, _) = A, Bm, C
    assert ct.dtype == torch.float32 and at.dtype in (torch.float32, torch.bfloat16) and \
        bt.dtype in (torch.float32, torch.bfloat16)
    check(load().vqaf_gemm_f32_atomic(addr(A), code(at), a_s[0], a_s[1], a_s[2], addr(Bm), code(bt), b_s[0], b_s[1], b_s[2],
                               addr(bias) if bias is not None else None,
                               code(bias[0]) if bias is not None else _lib.F32, bias_sg, addr(C), c_s[0], c_s[1],
                                      G, M, N, K, float(alpha), int(into_zeros), stream_ptr(ct.device)),
          "vqaf_gemm_f32_atomic")


class _DenseSingleQuery(torch.autograd.Function):
    """The whole attention output of MHA(query[:,None], seq, seq) -- query and key folds, token passes,
    value and output projections -- with a hand-derived backward: every (B,E)-sized product is one
    strided (per-head batched) launch of the library's f32 GEMM, four forward and nine backward,
    instead of the several dozen small launches autograd spends on the same algebra."""

    @staticmethod
    def forward(ctx, query, seq, w_in, b_in, w_out, b_out, H):
        B, N, E = seq.shape
        dh = E // H
        dev = seq.device
        q_c, seq_c = query.reshape(B, E).contiguous(), seq.contiguous()
        w_in_c, b_in_c, w_out_c, b_out_c = (t.detach().contiguous() for t in (w_in, b_in, w_out, b_out))
        if q_c.dtype not in (torch.float32, torch.bfloat16):
            q_c = q_c.float()
        qp, u, attn, out = _zeroed(dev, (B, E), (B, H, E), (B, E), (B, E))
        if q_c.dtype == torch.bfloat16 and w_in_c.dtype == torch.bfloat16 and b_in_c.dtype == torch.bfloat16 \
                and E == 512 and dh == 64:
            # both products in one launch on the bf16 matrix cores (f32-accurate: see vqaf_fold_query)
            check(load().vqaf_fold_query(ptr(q_c), ptr(w_in_c), ptr(b_in_c), B, E, H, _lib.BF16, _lib.BF16, ptr(qp),
                                         ptr(u), stream_ptr(dev)), "vqaf_fold_query")
        else:
            # qp = query Wq^T + bq
            _gemm((q_c, 0), (0, E, 1), (w_in_c, 0), (0, 1, E), (qp, 0), (0, E), 1, B, E, E, bias=(b_in_c, 0),
                  into_zeros=True)
            # u[:,h,:] = qp[:, head h] Wk[head h, :] / sqrt(dh)
            _gemm((qp, 0), (dh, E, 1), (w_in_c, E * E), (dh * E, E, 1), (u, 0), (E, H * E), H, B, E, dh,
                  alpha=1.0 / math.sqrt(dh), into_zeros=True)
        z, p = _tokens_forward(seq_c, u)
        # attn[:, head h] = z[:,h,:] Wv[head h, :]^T + bv[head h]
        _gemm((z, 0), (E, H * E, 1), (w_in_c, 2 * E * E), (dh * E, 1, E), (attn, 0), (dh, E), H, B, dh, E,
              bias=(b_in_c, 2 * E), bias_sg=dh, into_zeros=True)
        _gemm((attn, 0), (0, E, 1), (w_out_c, 0), (0, 1, E), (out, 0), (0, E), 1, B, E, E, bias=(b_out_c, 0),
              into_zeros=True)
        ctx.save_for_backward(q_c, seq_c, w_in_c, w_out_c, qp, u, p, z, attn)
        ctx.H = H
        ctx.dtypes = (query.dtype, w_in.dtype, b_in.dtype, w_out.dtype, b_out.dtype)
        ctx.q_shape = query.shape
        return out

    @staticmethod
    def backward(ctx, g_out):
        q_c, seq_c, w_in_c, w_out_c, qp, u, p, z, attn = ctx.saved_tensors
        B, N, E = seq_c.shape
        H = ctx.H
        dh = E // H
        dev = seq_c.device
        f32 = dict(dtype=torch.float32, device=dev)
        need_q, need_seq, need_win, need_bin, need_wout, need_bout = ctx.needs_input_grad[:6]
        g = g_out.contiguous().float()
        scale = 1.0 / math.sqrt(dh)
        # every f32 output of the products below comes out of one zero-filled allocation (one fill launch
        # instead of a memset per split product)
        d_attn, d_wout, t, d_win, d_qp, d_q = _zeroed(dev, (B, E), (E, E), (B, H, E), (3 * E, E), (B, E), (B, E))
        # through the output projection
        _gemm((g, 0), (0, E, 1), (w_out_c, 0), (0, E, 1), (d_attn, 0), (0, E), 1, B, E, E, into_zeros=True)
        d_bout = None
        if need_wout:
            _gemm((g, 0), (0, 1, E), (attn, 0), (0, E, 1), (d_wout, 0), (0, E), 1, E, E, B, into_zeros=True)
        else:
            d_wout = None
        if need_bout:
            d_bout = g.sum(0)
        # through the value projection: t[:,h,:] = d_attn[:, head h] Wv[head h, :]
        _gemm((d_attn, 0), (dh, E, 1), (w_in_c, 2 * E * E), (dh * E, E, 1), (t, 0), (E, H * E), H, B, E, dh,
              into_zeros=True)
        if not need_win:
            d_win = None
        d_bin = torch.zeros(3 * E, **f32) if need_bin else None  # the key bias shifts every score of a head alike
        if need_win:  # dWv[head h rows] = d_attn[:, head h]^T z[:,h,:]
            _gemm((d_attn, 0), (dh, 1, E), (z, 0), (E, H * E, 1), (d_win, 2 * E * E), (dh * E, E), H, dh, E, B,
                  into_zeros=True)
        if need_bin:
            torch.sum(d_attn, 0, out=d_bin[2 * E:])
        # through the token passes
        need_u = need_q or need_win or need_bin
        gseq, du = _tokens_backward(seq_c, u, p, t, need_seq, need_u)
        if need_u:
            # d_qp[:, head h] = du[:,h,:] Wk[head h, :]^T / sqrt(dh)
            _gemm((du, 0), (E, H * E, 1), (w_in_c, E * E), (dh * E, 1, E), (d_qp, 0), (dh, E), H, B, dh, E,
                  alpha=scale, into_zeros=True)
            if need_win:
                # dWk[head h rows] = qp[:, head h]^T du[:,h,:] / sqrt(dh);  dWq = d_qp^T query
                _gemm((qp, 0), (dh, 1, E), (du, 0), (E, H * E, 1), (d_win, E * E), (dh * E, E), H, dh, E, B,
                      alpha=scale, into_zeros=True)
                _gemm((d_qp, 0), (0, 1, E), (q_c, 0), (0, E, 1), (d_win, 0), (0, E), 1, E, E, B, into_zeros=True)
            if need_bin:
                torch.sum(d_qp, 0, out=d_bin[:E])
            if need_q:
                _gemm((d_qp, 0), (0, E, 1), (w_in_c, 0), (0, E, 1), (d_q, 0), (0, E), 1, B, E, E, into_zeros=True)
        if not (need_u and need_q):
            d_q = None
        qd, wid, bid, wod, bod = ctx.dtypes
        cast = lambda x, d: None if x is None else (x if x.dtype == d else x.to(d))  # noqa: E731
        return (cast(d_q.view(ctx.q_shape), qd) if d_q is not None else None, gseq, cast(d_win, wid),
                cast(d_bin, bid), cast(d_wout, wod), cast(d_bout, bod), None)


def single_query_attention(mha: nn.MultiheadAttention, query: torch.Tensor, seq: torch.Tensor) -> torch.Tensor:
    """(B,E) attention output for query (B,E) over seq (B,N,E); differentiable w.r.t. query, seq and
    every parameter of `mha` (the key bias receives an exactly zero gradient: it shifts all scores
    of a head equally)."""
    out = _DenseSingleQuery.apply(query, seq, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                  mha.out_proj.bias, mha.num_heads)
    return out.to(seq.dtype)
