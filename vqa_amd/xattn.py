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


def _zeroed(dev, *shapes, with_flat=False, zero_first=None):
    """f32 tensors of the given shapes carved out of ONE zero-filled allocation (one fill launch): outputs of
    _gemm(..., into_zeros=True), which then adds into them instead of clearing each one with its own memset.
    with_flat: also returns the allocation itself and the pieces' element offsets in it.  zero_first = n: only
    the first n pieces are cleared (the others are outputs that a product STORES)."""
    sizes = [(math.prod(sh) + 3) // 4 * 4 for sh in shapes]  # 16-byte aligned pieces
    if zero_first is None or zero_first >= len(shapes):
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    else:
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        flat[:sum(sizes[:zero_first])].zero_()
    out, offs, o = [], [], 0
    for sh, n in zip(shapes, sizes):
        out.append(flat[o:o + math.prod(sh)].view(sh))
        offs.append(o)
        o += n
    return (out, flat, offs) if with_flat else out


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


def _gemm_sets(A, a_s, Bm, b_s, C, c_s, G, M, N, K, alpha=1.0, bias=None, bias_sg=0):
    """_gemm for several problems of ONE shape, stride set and dtype set in one launch: A, Bm, C (and bias)
    are lists of (tensor, element offset) pairs, one per problem.  The outputs are ADDED into C, which the
    caller provides zero-filled (long contractions are split over workgroups with f32 atomics)."""
    import ctypes

    def addrs(pairs):
        return (ctypes.c_void_p * len(pairs))(*[t.data_ptr() + off * t.element_size() for t, off in pairs])

    def code(t):
        return _lib.BF16 if t.dtype == torch.bfloat16 else _lib.F32

    n = len(A)
    at, bt, ct = A[0][0], Bm[0][0], C[0][0]
    assert ct.dtype == torch.float32 and all(a.dtype == at.dtype for a, _ in A) and all(b.dtype == bt.dtype for b, _ in Bm)
    assert at.dtype in (torch.float32, torch.bfloat16) and bt.dtype in (torch.float32, torch.bfloat16)
    check(load().vqaf_gemm_f32_atomic_sets(
        n, addrs(A), code(at), a_s[0], a_s[1], a_s[2], addrs(Bm), code(bt), b_s[0], b_s[1], b_s[2],
        addrs(bias) if bias is not None else None, code(bias[0][0]) if bias is not None else _lib.F32, bias_sg,
        addrs(C), c_s[0], c_s[1], G, M, N, K, float(alpha), 1, stream_ptr(ct.device)), "vqaf_gemm_f32_atomic_sets")


_ONES: dict = {}


def _ones(dev):
    """A few f32 ones on `dev`: with zero strides they are the all-ones operand that turns a column sum into
    one more problem of a grouped product launch."""
    t = _ONES.get(dev)
    if t is None:
        t = torch.ones(4, dtype=torch.float32, device=dev)
        if not torch.cuda.is_current_stream_capturing():  # (a tensor made during a capture lives in that graph's pool)
            _ONES[dev] = t
    return t


def _prob(A, a_s, Bm, b_s, C, c_s, G, M, N, K, alpha=1.0, bias=None, bias_sg=0, store=False):
    """One problem of _gemm_group: the arguments of _gemm.  store: C = result (plain stores, contraction not
    split) instead of C += result."""
    return (A, a_s, Bm, b_s, C, c_s, G, M, N, K, alpha, bias, bias_sg, store)


def _gemm_group(probs):
    """Up to 12 different products (see _gemm for the arguments of each) in ONE launch, each added into its
    zero-filled C (or stored, see _prob): products that do not depend on each other share the launch instead of occupying the
    stream one after the other."""
    def addr(pair):
        t, off = pair
        return t.data_ptr() + off * t.element_size()

    def code(t):
        return _lib.BF16 if t.dtype == torch.bfloat16 else _lib.F32

    arr = (_lib.GemmDesc * len(probs))()
    for i, (A, a_s, Bm, b_s, C, c_s, G, M, N, K, alpha, bias, bias_sg, store) in enumerate(probs):
        assert C[0].dtype == torch.float32 or (store and C[0].dtype == torch.bfloat16)
        arr[i] = _lib.GemmDesc(addr(A), addr(Bm), addr(bias) if bias is not None else None, addr(C),
                               a_s[0], a_s[1], a_s[2], b_s[0], b_s[1], b_s[2], bias_sg, c_s[0], c_s[1],
                               code(A[0]), code(Bm[0]), code(bias[0]) if bias is not None else _lib.F32,
                               G, M, N, K, float(alpha), int(store), code(C[0]))
    import ctypes
    check(load().vqaf_gemm_f32_group(len(probs), ctypes.cast(arr, ctypes.c_void_p), stream_ptr(probs[0][4][0].device)),
          "vqaf_gemm_f32_group")


class _DenseCoAttention(torch.autograd.Function):
    """The attention outputs of SEVERAL single-query MHA calls of one batch size and width -- the two branches
    of CoAttention without pruning (utils/fusion.py:257-268) -- concatenated along the feature axis, with every
    step of every branch in the same launch as its counterpart: one query fold, one scoring launch, one
    softmax + weighted-token-sum launch, one launch per (B,E)-sized projection; the backward likewise (the
    softmax backward, the folded-query gradient and the token gradient share one pass over the tokens).
    Same arithmetic as _DenseSingleQuery per branch."""

    @staticmethod
    def _bwd_zero_words(nb, B, E):
        """f32 words of the block that the backward's split products are ADDED into: d_attn and d_qp (nb, B, E each)
        and the in-projection bias gradients (nb, 3E; the key third is an exact zero that nothing writes)."""
        return 2 * nb * B * E + nb * 3 * E

    @staticmethod
    def forward(ctx, H, out_dtype, *flat):
        import ctypes
        nb = len(flat) // 6
        br = [flat[6 * i:6 * i + 6] for i in range(nb)]
        B, _, E = br[0][1].shape
        dh = E // H
        dev = br[0][1].device
        lib, st = load(), stream_ptr(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        q_c = [b[0].reshape(B, E).contiguous() for b in br]
        if any(q.dtype not in (torch.float32, torch.bfloat16) or q.dtype != q_c[0].dtype for q in q_c):
            q_c = [q.float() for q in q_c]
        seq_c = [b[1].contiguous() for b in br]
        w_in = [b[2].detach().contiguous() for b in br]
        b_in = [b[3].detach().to(w.dtype).contiguous() for b, w in zip(br, w_in)]
        w_out = [b[4].detach().contiguous() for b in br]
        b_out = [b[5].detach().to(w.dtype).contiguous() for b, w in zip(br, w_out)]
        Ns = [s.shape[1] for s in seq_c]
        qp, u, z = torch.empty(nb, B, E, **f32), torch.empty(nb, B, H, E, **f32), torch.empty(nb, B, H, E, **f32)
        folds = (_lib.FoldDesc * nb)(*[_lib.FoldDesc(ptr(q_c[i]), ptr(w_in[i]), ptr(b_in[i]), ptr(qp[i]), ptr(u[i]))
                                       for i in range(nb)])
        check(lib.vqaf_fold_query_multi(ctypes.cast(folds, ctypes.c_void_p), nb, B, E, H, dtype_code(q_c[0]),
                                        dtype_code(w_in[0]), st), "vqaf_fold_query_multi")
        s = [torch.empty(B, n, H, **f32) for n in Ns]
        sd = (_lib.ScoresDesc * nb)(*[_lib.ScoresDesc(ptr(seq_c[i]), ptr(u[i]), ptr(s[i]), Ns[i]) for i in range(nb)])
        check(lib.vqaf_head_scores_multi(ctypes.cast(sd, ctypes.c_void_p), nb, B, E, H, dtype_code(seq_c[0]), st),
              "vqaf_head_scores_multi")
        p = [torch.empty(B, H, n, **f32) for n in Ns]
        fm = (_lib.DenseFwdMod * nb)(*[_lib.DenseFwdMod(ptr(seq_c[i]), ptr(s[i]), ptr(p[i]), ptr(z[i]), Ns[i])
                                       for i in range(nb)])
        # (the same launch clears attn and out, which the two projections below build with atomic adds)
        # ... and, when a backward will follow, the block its products are added into: no fill launch there
        wants_grad = any(ctx.needs_input_grad)
        zbuf = torch.empty(2 * nb * B * E + (_DenseCoAttention._bwd_zero_words(nb, B, E) if wants_grad else 0), **f32)
        attn, out = zbuf[:nb * B * E].view(nb, B, E), zbuf[nb * B * E:2 * nb * B * E].view(B, nb * E)
        ctx.bwd_zero = zbuf[2 * nb * B * E:] if wants_grad else None
        check(lib.vqaf_softmax_wsum_multi(ctypes.cast(fm, ctypes.c_void_p), nb, B, E, H, dtype_code(seq_c[0]),
                                          ptr(zbuf), zbuf.numel(), st), "vqaf_softmax_wsum_multi")
        # attn_i[:, head h] = z_i[:,h,:] Wv_i[head h, :]^T + bv_i[head h]
        _gemm_sets([(z[i], 0) for i in range(nb)], (E, H * E, 1), [(w_in[i], 2 * E * E) for i in range(nb)],
                   (dh * E, 1, E), [(attn[i], 0) for i in range(nb)], (dh, E), H, B, dh, E,
                   bias=[(b_in[i], 2 * E) for i in range(nb)], bias_sg=dh)
        # out[:, i*E:(i+1)*E] = attn_i Wout_i^T + bout_i
        _gemm_sets([(attn[i], 0) for i in range(nb)], (0, E, 1), [(w_out[i], 0) for i in range(nb)], (0, 1, E),
                   [(out, i * E) for i in range(nb)], (0, nb * E), 1, B, E, E, bias=[(b_out[i], 0) for i in range(nb)])
        ctx.save_for_backward(qp, u, z, attn, *q_c, *seq_c, *w_in, *w_out, *p)
        ctx.H, ctx.nb = H, nb
        ctx.dtypes = [tuple(t.dtype for t in (b[0], b[2], b[3], b[4], b[5])) for b in br]
        ctx.q_shapes = [b[0].shape for b in br]
        return out if out.dtype == out_dtype else out.to(out_dtype)

    @staticmethod
    def backward(ctx, g_out):
        import ctypes
        nb, H = ctx.nb, ctx.H
        saved = ctx.saved_tensors
        qp, u, z, attn = saved[:4]
        q_c, seq_c, w_in, w_out, p = (saved[4 + k * nb:4 + (k + 1) * nb] for k in range(5))
        B, _, E = seq_c[0].shape
        dh = E // H
        dev = seq_c[0].device
        lib, st = load(), stream_ptr(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        Ns = [s.shape[1] for s in seq_c]
        need = [ctx.needs_input_grad[2 + 6 * i:8 + 6 * i] for i in range(nb)]  # (q, seq, w_in, b_in, w_out, b_out)
        need_q = any(n[0] for n in need)
        need_seq = any(n[1] for n in need)
        need_par = any(any(n[2:]) for n in need)
        need_u = need_q or need_par
        g = g_out.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        scale = 1.0 / math.sqrt(dh)
        rng = range(nb)
        # Everything handed back is STORED by the product that computes it, in the type its consumer wants (the
        # inputs' common dtype; f32 when they differ, cast per tensor at the end): the weight gradients and the bias
        # gradients contract over the batch only (a few fetch stages), the query gradient over the width (16 stages,
        # beside the others in its launch).  Nothing of it needs clearing, no atomics, no cast launch.  What is built
        # with atomic adds over split contractions (d_attn, d_qp) and the key-bias zeros sit in one small block that
        # the forward's clearing launch already cleared; t, the gradient at the per-head token sums, is stored too.
        want = {d for dts, n in zip(ctx.dtypes, need) for d, nd in zip(dts, (n[0], n[2], n[3], n[4], n[5])) if nd}
        rd = next(iter(want)) if len(want) == 1 and next(iter(want)) in (torch.float32, torch.bfloat16) else torch.float32
        zb = ctx.bwd_zero
        ctx.bwd_zero = None  # (a second backward through the same graph clears a block of its own)
        if zb is None:
            zb = torch.zeros(_DenseCoAttention._bwd_zero_words(nb, B, E), **f32)
        n1 = nb * B * E
        d_attn, d_qp = zb[:n1].view(nb, B, E), zb[n1:2 * n1].view(nb, B, E)
        d_bin = zb[2 * n1:].view(rd)[:nb * 3 * E].view(nb, 3 * E)
        ret = dict(dtype=rd, device=dev)
        d_q, d_bout = torch.empty(nb, B, E, **ret), torch.empty(nb, E, **ret)
        d_win, d_wout = torch.empty(nb, 3 * E, E, **ret), torch.empty(nb, E, E, **ret)
        t = torch.empty(nb, B, H, E, **f32)
        # through the output projection: d_attn_i = g[:, i*E:(i+1)*E] Wout_i
        _gemm_sets([(g, i * E) for i in rng], (0, nb * E, 1), [(w_out[i], 0) for i in rng], (0, E, 1),
                   [(d_attn[i], 0) for i in rng], (0, E), 1, B, E, E)
        # Everything that follows from d_attn alone shares ONE launch: launches on a stream do not overlap and each
        # of these products is only a few microseconds of load -> MFMA -> atomic add.
        ones = (_ones(dev), 0)
        group = []
        for i in rng:
            # through the value projection: t_i[:,h,:] = d_attn_i[:, head h] Wv_i[head h, :]
            group.append(_prob((d_attn[i], 0), (dh, E, 1), (w_in[i], 2 * E * E), (dh * E, E, 1), (t[i], 0), (E, H * E),
                               H, B, E, dh, store=True))
            if need_par:
                # dWout_i = g_i^T attn_i;  dWv_i[head h rows] = d_attn_i[:, head h]^T z_i[:,h,:]
                group.append(_prob((g, i * E), (0, 1, nb * E), (attn[i], 0), (0, E, 1), (d_wout[i], 0), (0, E), 1, E, E, B,
                                   store=True))
                group.append(_prob((d_attn[i], 0), (dh, 1, E), (z[i], 0), (E, H * E, 1), (d_win[i], 2 * E * E),
                                   (dh * E, E), H, dh, E, B, store=True))
                # bias gradients = column sums = products with a row of ones: of g (output bias) and of d_attn (value
                # bias); the key bias shifts every score of a head alike and keeps its exact zero
                group.append(_prob(ones, (0, 0, 0), (g, i * E), (0, nb * E, 1), (d_bout[i], 0), (0, E), 1, 1, E, B,
                                   store=True))
                group.append(_prob(ones, (0, 0, 0), (d_attn[i], 0), (0, E, 1), (d_bin[i], 2 * E), (0, E), 1, 1, E, B,
                                   store=True))
        _gemm_group(group)
        # through the token passes
        dp = [torch.empty(B, n, H, **f32) for n in Ns]
        sd = (_lib.ScoresDesc * nb)(*[_lib.ScoresDesc(ptr(seq_c[i]), ptr(t[i]), ptr(dp[i]), Ns[i]) for i in rng])
        check(lib.vqaf_head_scores_multi(ctypes.cast(sd, ctypes.c_void_p), nb, B, E, H, dtype_code(seq_c[0]), st),
              "vqaf_head_scores_multi")
        du = torch.empty(nb, B, H, E, **f32) if need_u else None
        gseq = [torch.empty_like(s) for s in seq_c] if need_seq else [None] * nb
        if need_u or need_seq:
            bm = (_lib.DenseBwdMod * nb)(*[_lib.DenseBwdMod(ptr(seq_c[i]), ptr(p[i]), ptr(dp[i]), ptr(u[i]), ptr(t[i]),
                                                          ptr(du[i]) if need_u else None, ptr(gseq[i]), Ns[i])
                                           for i in rng])
            check(lib.vqaf_token_bwd_multi(ctypes.cast(bm, ctypes.c_void_p), nb, B, E, H, dtype_code(seq_c[0]), st),
                  "vqaf_token_bwd_multi")
        if need_u:
            group = []
            for i in rng:
                # d_qp_i[:, head h] = du_i[:,h,:] Wk_i[head h, :]^T / sqrt(dh)
                group.append(_prob((du[i], 0), (E, H * E, 1), (w_in[i], E * E), (dh * E, 1, E), (d_qp[i], 0), (dh, E),
                                   H, B, dh, E, alpha=scale))
                if need_par:
                    # dWk_i[head h rows] = qp_i[:, head h]^T du_i[:,h,:] / sqrt(dh)
                    group.append(_prob((qp[i], 0), (dh, 1, E), (du[i], 0), (E, H * E, 1), (d_win[i], E * E), (dh * E, E),
                                       H, dh, E, B, alpha=scale, store=True))
            _gemm_group(group)
            group = []
            for i in rng:
                if need_par:
                    # dWq_i = d_qp_i^T query_i; the query bias gradient = column sums of d_qp
                    group.append(_prob((d_qp[i], 0), (0, 1, E), (q_c[i], 0), (0, E, 1), (d_win[i], 0), (0, E), 1, E, E, B,
                                       store=True))
                    group.append(_prob(ones, (0, 0, 0), (d_qp[i], 0), (0, E, 1), (d_bin[i], 0), (0, E), 1, 1, E, B,
                                       store=True))
                if need_q:
                    group.append(_prob((d_qp[i], 0), (0, E, 1), (w_in[i], 0), (0, E, 1), (d_q[i], 0), (0, E), 1, B, E, E,
                                       store=True))
            if group:
                _gemm_group(group)
        grads = [None, None]
        for i in rng:
            qd, wid, bid, wod, bod = ctx.dtypes[i]
            n = need[i]
            cast = lambda x, d: x if x.dtype == d else x.to(d)  # noqa: E731
            grads += [cast(d_q[i].view(ctx.q_shapes[i]), qd) if n[0] else None,
                      gseq[i] if n[1] else None,
                      cast(d_win[i], wid) if n[2] else None,
                      cast(d_bin[i], bid) if n[3] else None,
                      cast(d_wout[i], wod) if n[4] else None,
                      cast(d_bout[i], bod) if n[5] else None]
        return tuple(grads)


def dense_coattention_supported(mhas, queries, seqs) -> bool:
    """True when dense_coattention() covers these branches: one batch size, width, head count and set of
    dtypes, at most four branches, at most 8 heads, per-sample coefficients within the kernels' LDS."""
    if not (1 <= len(mhas) <= 4) or any(s.dim() != 3 for s in seqs):
        return False
    B, _, E = seqs[0].shape
    H = mhas[0].num_heads
    if H > 8 or E % H != 0 or E % 8 != 0 or B > 65535:
        return False
    for m, q, s in zip(mhas, queries, seqs):
        if s.shape[0] != B or s.shape[2] != E or m.num_heads != H or q.numel() != B * E or \
                s.dtype != seqs[0].dtype or s.dtype not in (torch.float32, torch.bfloat16) or \
                q.dtype != queries[0].dtype or m.in_proj_weight is None or \
                m.in_proj_weight.dtype != mhas[0].in_proj_weight.dtype or \
                m.out_proj.weight.dtype != mhas[0].out_proj.weight.dtype or \
                m.in_proj_weight.dtype not in (torch.float32, torch.bfloat16) or m.in_proj_bias is None or \
                m.out_proj.bias is None:
            return False
        if ((2 * 8 + 4) * s.shape[1] + 2048 + 8) * 4 > 128 * 1024:
            return False
    return True


def dense_coattention(mhas, queries, seqs) -> torch.Tensor:
    """cat_i MHA_i(query_i[:,None], seq_i, seq_i) along the feature axis, (B, len(mhas) * E) in the tokens'
    dtype, every branch in the same launches (see _DenseCoAttention)."""
    flat = []
    for m, q, s in zip(mhas, queries, seqs):
        flat += [q, s, m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias]
    return _DenseCoAttention.apply(mhas[0].num_heads, seqs[0].dtype, *flat)


def single_query_attention(mha: nn.MultiheadAttention, query: torch.Tensor, seq: torch.Tensor) -> torch.Tensor:
    """(B,E) attention output for query (B,E) over seq (B,N,E); differentiable w.r.t. query, seq and
    every parameter of `mha` (the key bias receives an exactly zero gradient: it shifts all scores
    of a head equally)."""
    out = _DenseSingleQuery.apply(query, seq, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                  mha.out_proj.bias, mha.num_heads)
    return out.to(seq.dtype)
