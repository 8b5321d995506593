"""Tensor-level wrappers over the libvqafuse C ABI.

Every function here takes and returns torch tensors that live on a HIP device, enqueues the
kernels on torch's current stream and never synchronises with the host.  torch is used for
device memory, streams and autograd bookkeeping only; the arithmetic is in the kernels.
"""
from __future__ import annotations

from typing import Optional, Tuple

import weakref

import torch

from . import _lib
from ._lib import check, dtype_code, load, ptr, stream_ptr


# --------------------------------------------------------------------------------------------
# Device-side counter-based RNG state {seed, counter}
# --------------------------------------------------------------------------------------------
class PhiloxState:
    """Four 64-bit words in device memory: {seed, counter, ticket, reserved}.

    Kernels that draw random numbers read {seed, counter} and address each draw by
    (seed, counter, element); ``advance()`` bumps the counter on the stream, so a captured HIP
    graph draws fresh numbers on every replay.  Kernels that bump the counter themselves (their last
    workgroup does it) count arrivals in `ticket`, which is zero between launches.
    """

    def __init__(self, device: torch.device, seed: Optional[int] = None, stream_salt: int = 0):
        if seed is None:
            seed = torch.initial_seed()
        self.seed = (int(seed) + 0x9E3779B97F4A7C15 * int(stream_salt)) & 0xFFFFFFFFFFFFFFFF
        signed = self.seed - (1 << 64) if self.seed >= (1 << 63) else self.seed
        self.state = torch.tensor([signed, 0, 0, 0], dtype=torch.int64, device=device)

    def advance(self, delta: int = 1) -> None:
        check(load().vqaf_rng_advance(ptr(self.state), delta, stream_ptr(self.state.device)),
              "vqaf_rng_advance")


_default_states: dict = {}


def default_rng(device: torch.device) -> PhiloxState:
    """Per-device state seeded from torch's global seed (re-made after torch.manual_seed)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    seed = torch.initial_seed()
    st = _default_states.get(key)
    if st is None or st[0] != seed:
        st = (seed, PhiloxState(device, seed))
        _default_states[key] = st
    return st[1]


ADVANCE_RNG_IN_FOLD = True  # the query-fold launch of fused_prune_pool advances the generator state on the side
IN_KERNEL_RNG_BUMP = False  # let vqaf_fused_prune_pool_multi advance its generator state itself
FUSED_ONE_PASS = False  # True: the one-launch pruning branch takes the one-pass kernel where it applies (D = 512, N <= 208)

# Optional measurement probe: bench.py installs a callable (name, phase) -> None that records
# stream events around a named kernel launch; None in normal use.
PROBE = None


def _probe(name: str, phase: str) -> None:
    if PROBE is not None:
        PROBE(name, phase)


def require_gpu(*tensors: torch.Tensor) -> None:
    """Raises the library's "GPU only" error for CPU tensors (there is no CPU fallback)."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            ptr(t)


def _noise_kind(noise_type: Optional[str]) -> int:
    if noise_type is None:
        return _lib.NOISE_NONE
    if noise_type == "gaussian":
        return _lib.NOISE_GAUSSIAN
    if noise_type == "uniform":
        return _lib.NOISE_UNIFORM
    # same message as utils/fusion.py:111 of the reference
    raise ValueError(f"Unsupported noise type: {noise_type}")


# --------------------------------------------------------------------------------------------
# Scoring and selection (no gradients flow through these: the weights only feed indices)
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def attention_scores(q: torch.Tensor, seq: torch.Tensor, in_proj_weight: torch.Tensor,
                     in_proj_bias: torch.Tensor, num_heads: int) -> torch.Tensor:
    """Head-averaged single-query attention weights (B,N) f32; see vqaf_score."""
    B, N, D = seq.shape
    seq = seq.contiguous()
    q = q.reshape(B, D).to(seq.dtype).contiguous()
    w_ = in_proj_weight.detach().contiguous()
    b_ = in_proj_bias.detach().to(w_.dtype).contiguous()
    lib = load()
    nbytes = lib.vqaf_score_workspace_bytes(B, N, D, num_heads)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=seq.device)
    out = torch.empty(B, N, dtype=torch.float32, device=seq.device)
    check(lib.vqaf_score(ptr(q), ptr(seq), ptr(w_), ptr(b_), B, N, D, num_heads, dtype_code(seq),
                         dtype_code(w_), ptr(ws), nbytes, ptr(out), stream_ptr(seq.device)),
          "vqaf_score")
    return out


@torch.no_grad()
def head_score_matrix(q: torch.Tensor, seq: torch.Tensor, in_proj_weight: torch.Tensor,
                      in_proj_bias: torch.Tensor, num_heads: int) -> torch.Tensor:
    """Raw scaled per-head scores s (B,N,H) f32 (query fold + one pass over the tokens); the
    softmax / head mean is left to `select_indices`, which fuses it with the ranking."""
    B, N, D = seq.shape
    H = num_heads
    seq = seq.contiguous()
    ptr(seq)  # fails loudly on CPU tensors before any device call
    q = q.reshape(B, D).to(seq.dtype).contiguous()
    w_ = in_proj_weight.detach().contiguous()
    b_ = in_proj_bias.detach().to(w_.dtype).contiguous()
    lib, st, dev = load(), stream_ptr(seq.device), seq.device
    qp = torch.empty(B, D, dtype=torch.float32, device=dev)
    u = torch.empty(B, H, D, dtype=torch.float32, device=dev)
    check(lib.vqaf_fold_query(ptr(q), ptr(w_), ptr(b_), B, D, H, dtype_code(seq), dtype_code(w_), ptr(qp),
                              ptr(u), st), "vqaf_fold_query")
    s = torch.empty(B, N, H, dtype=torch.float32, device=dev)
    check(lib.vqaf_head_scores(ptr(seq), ptr(u), B, N, D, H, dtype_code(seq), ptr(s), st),
          "vqaf_head_scores")
    return s


@torch.no_grad()
def select_indices(k: int, *, scores: Optional[torch.Tensor] = None, weights: Optional[torch.Tensor] = None,
                   k_adv: int = 0, pos: Optional[torch.Tensor] = None, src: Optional[torch.Tensor] = None,
                   rng: Optional["PhiloxState"] = None, want_weights: bool = False,
                   want_bottom: bool = False, record: bool = False, salt: int = 0,
                   advance: bool = True) -> dict:
    """Keep indices in one launch (vqaf_select): softmax/head-mean of `scores` (B,N,H) or given
    `weights` (B,N), ranking, and the adversarial swap -- injected (`pos`, `src`), drawn in the
    kernel (`rng`), or none (k_adv == 0)."""
    if (scores is None) == (weights is None):
        raise ValueError("pass exactly one of scores / weights")
    if scores is not None:
        scores = scores.contiguous()
        B, N, H = scores.shape
        dev = scores.device
    else:
        if weights.dim() == 3:
            weights = weights.squeeze(1)
        weights = weights.to(torch.float32).contiguous()
        (B, N), H, dev = weights.shape, 0, weights.device
    mode = 0
    if k_adv > 0:
        mode = 1 if pos is not None else 2
        if mode == 1:
            pos, src = pos.to(torch.int64).contiguous(), src.to(torch.int64).contiguous()
        elif rng is None:
            rng = default_rng(dev)
    out = {"top": torch.empty(B, k, dtype=torch.int64, device=dev), "bottom": None, "weights": None,
           "pos": None, "src": None}
    if want_bottom and N > k:
        out["bottom"] = torch.empty(B, N - k, dtype=torch.int64, device=dev)
    if want_weights:
        out["weights"] = torch.empty(B, N, dtype=torch.float32, device=dev)
    if record and mode == 2:
        out["pos"] = torch.empty(B, k_adv, dtype=torch.int64, device=dev)
        out["src"] = torch.empty(B, k_adv, dtype=torch.int64, device=dev)
    check(load().vqaf_select(ptr(scores), ptr(weights), B, N, H, k, mode, k_adv,
                             ptr(pos) if mode == 1 else None, ptr(src) if mode == 1 else None,
                             ptr(rng.state) if mode == 2 else None, int(salt), ptr(out["weights"]),
                             ptr(out["top"]), ptr(out["bottom"]), ptr(out["pos"]), ptr(out["src"]),
                             stream_ptr(dev)), "vqaf_select")
    out["drew"] = mode == 2
    if mode == 2 and advance:
        rng.advance()
    return out


@torch.no_grad()
def score_select_multi(specs, rng: Optional["PhiloxState"] = None):
    """Scoring + keep-index selection for several modalities that share batch, width, head count
    and dtypes (CoAttention's two branches) in FOUR launches in total: two for all query folds, one
    for all token scoring passes, one for all selections.

    specs: list of dicts(q, seq, in_proj_weight, in_proj_bias, num_heads, k, k_adv, pos, src, salt).
    Returns (list of keep-index tensors (B,k_i) int64, drew) where `drew` tells whether random
    numbers were consumed (the caller advances `rng` once)."""
    import ctypes
    n = len(specs)
    seq0 = specs[0]["seq"]
    B, _, D = seq0.shape
    H = specs[0]["num_heads"]
    dev = seq0.device
    ptr(seq0.contiguous())  # fails loudly on CPU tensors before any device call
    lib, st = load(), stream_ptr(dev)
    w_dt = specs[0]["in_proj_weight"].dtype
    keep = []  # keeps every temporary alive until the launches are enqueued
    folds = (_lib.FoldDesc * n)()
    scores = (_lib.ScoresDesc * n)()
    sels = (_lib.SelectDesc * n)()
    tops, drew = [], False
    for i, sp in enumerate(specs):
        seq = sp["seq"].contiguous()
        N = seq.shape[1]
        q = sp["q"].reshape(B, D).to(seq.dtype).contiguous()
        w_ = sp["in_proj_weight"].detach().contiguous()
        b_ = sp["in_proj_bias"].detach().to(w_.dtype).contiguous()
        qp = torch.empty(B, D, dtype=torch.float32, device=dev)
        u = torch.empty(B, H, D, dtype=torch.float32, device=dev)
        sc = torch.empty(B, N, H, dtype=torch.float32, device=dev)
        top = torch.empty(B, sp["k"], dtype=torch.int64, device=dev)
        k_adv = int(sp.get("k_adv", 0))
        pos, src = sp.get("pos"), sp.get("src")
        mode = 0
        if k_adv > 0:
            mode = 1 if pos is not None else 2
            if mode == 1:
                pos, src = pos.to(torch.int64).contiguous(), src.to(torch.int64).contiguous()
            else:
                drew = True
        keep.append((seq, q, w_, b_, qp, u, sc, pos, src))
        folds[i] = _lib.FoldDesc(ptr(q), ptr(w_), ptr(b_), ptr(qp), ptr(u))
        scores[i] = _lib.ScoresDesc(ptr(seq), ptr(u), ptr(sc), N)
        sels[i] = _lib.SelectDesc(ptr(sc), None, ptr(pos) if mode == 1 else None, ptr(src) if mode == 1 else None,
                                  None, ptr(top), None, None, None, int(sp.get("salt", i)), N, sp["k"], mode, k_adv)
        tops.append(top)
    if drew and rng is None:
        rng = default_rng(dev)
    check(lib.vqaf_fold_query_multi(ctypes.cast(folds, ctypes.c_void_p), n, B, D, H, dtype_code(seq0),
                                    _lib.BF16 if w_dt == torch.bfloat16 else _lib.F32, st), "vqaf_fold_query_multi")
    check(lib.vqaf_head_scores_multi(ctypes.cast(scores, ctypes.c_void_p), n, B, D, H, dtype_code(seq0), st),
          "vqaf_head_scores_multi")
    check(lib.vqaf_select_multi(ctypes.cast(sels, ctypes.c_void_p), n, B, H, ptr(rng.state) if drew else None, st),
          "vqaf_select_multi")
    return tops, drew


@torch.no_grad()
def topk_indices(w: torch.Tensor, k: int, need_bottom: bool = True
                 ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(top (B,k) int64 weight-descending, bottom (B,N-k) int64 weight-ascending)."""
    if w.dim() == 3:
        w = w.squeeze(1)
    w = w.to(torch.float32).contiguous()
    B, N = w.shape
    top = torch.empty(B, k, dtype=torch.int64, device=w.device)
    bottom = torch.empty(B, N - k, dtype=torch.int64, device=w.device) if (need_bottom and N > k) else None
    check(load().vqaf_topk(ptr(w), B, N, k, ptr(top), ptr(bottom), stream_ptr(w.device)), "vqaf_topk")
    return top, bottom


@torch.no_grad()
def adv_swap_(top: torch.Tensor, bottom: torch.Tensor, pos: torch.Tensor, src: torch.Tensor) -> None:
    B, K = top.shape
    R = bottom.shape[1]
    k_adv = pos.shape[1]
    check(load().vqaf_adv_swap(ptr(top), ptr(bottom), ptr(pos.contiguous()), ptr(src.contiguous()),
                               B, K, R, k_adv, stream_ptr(top.device)), "vqaf_adv_swap")


@torch.no_grad()
def adv_swap_random_(top: torch.Tensor, bottom: torch.Tensor, k_adv: int, rng: PhiloxState,
                     record: bool = False):
    B, K = top.shape
    R = bottom.shape[1]
    pos = src = None
    if record:
        pos = torch.empty(B, k_adv, dtype=torch.int64, device=top.device)
        src = torch.empty(B, k_adv, dtype=torch.int64, device=top.device)
    check(load().vqaf_adv_swap_random(ptr(top), ptr(bottom), B, K, R, k_adv, ptr(rng.state),
                                      ptr(pos), ptr(src), stream_ptr(top.device)),
          "vqaf_adv_swap_random")
    rng.advance()
    return pos, src


# --------------------------------------------------------------------------------------------
# Gather + noise + mix token (differentiable w.r.t. seq)
# --------------------------------------------------------------------------------------------
class _GatherMix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seq, idx, noise, noise_kind, noise_scale, rng_state, alpha, out_dtype):
        B, N, D = seq.shape
        K = idx.shape[1]
        seq_c = seq.contiguous()
        idx_c = idx.contiguous()
        kout = K + (1 if alpha > 0 else 0)
        out = torch.empty(B, kout, D, dtype=out_dtype, device=seq.device)
        noise_c = noise.to(torch.float32).contiguous() if noise is not None else None
        check(load().vqaf_gather_mix(ptr(seq_c), ptr(idx_c), ptr(noise_c), noise_kind,
                                     float(noise_scale), ptr(rng_state), float(alpha), B, N, K, D,
                                     dtype_code(seq_c), dtype_code(out), ptr(out),
                                     stream_ptr(seq.device)), "vqaf_gather_mix")
        ctx.save_for_backward(idx_c)
        ctx.meta = (B, N, K, D, float(alpha), seq.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx_c,) = ctx.saved_tensors
        B, N, K, D, alpha, sdt = ctx.meta
        g = grad_out.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        grad_seq = torch.empty(B, N, D, dtype=sdt, device=g.device)
        check(load().vqaf_gather_mix_bwd(ptr(g), ptr(idx_c), alpha, B, N, K, D, dtype_code(g),
                                         dtype_code(grad_seq), ptr(grad_seq), stream_ptr(g.device)),
              "vqaf_gather_mix_bwd")
        return grad_seq, None, None, None, None, None, None, None


def gather_mix(seq: torch.Tensor, idx: torch.Tensor, alpha: float, noise: Optional[torch.Tensor] = None,
               noise_type: Optional[str] = None, noise_scale: float = 0.0,
               rng: Optional[PhiloxState] = None, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """out (B,K+1,D) (or (B,K,D) if alpha <= 0); see vqaf_gather_mix."""
    kind = _noise_kind(noise_type) if noise is None else _lib.NOISE_NONE
    state = None
    if kind != _lib.NOISE_NONE:
        rng = rng or default_rng(seq.device)
        state = rng.state
    out = _GatherMix.apply(seq, idx, noise, kind, noise_scale, state, alpha, out_dtype or seq.dtype)
    if kind != _lib.NOISE_NONE:
        rng.advance()
    return out


def prune_mix_supported(n_tokens: int, width: int, dtype: torch.dtype) -> bool:
    """Whether vqaf_prune_mix takes a (.., n_tokens, width) sequence of this dtype."""
    if dtype not in (torch.float32, torch.bfloat16):
        return False
    return bool(load().vqaf_prune_mix_supported(int(n_tokens), int(width),
                                                _lib.BF16 if dtype == torch.bfloat16 else _lib.F32))


class _PruneMix(torch.autograd.Function):
    """Selection + gather + noise + mix token of one sequence in one launch (vqaf_prune_mix)."""

    @staticmethod
    def forward(ctx, seq, w, K, mode, k_adv, pos, src, rng_state, salt, noise, noise_kind, noise_scale,
                noise_off, alpha, out_dtype):
        B, N, D = seq.shape
        seq_c = seq.contiguous()
        kout = K + (1 if alpha > 0 else 0)
        out = torch.empty(B, kout, D, dtype=out_dtype, device=seq.device)
        idx = torch.empty(B, K, dtype=torch.int64, device=seq.device)
        noise_c = noise.to(torch.float32).contiguous() if noise is not None else None
        need_noise_state = noise_c is None and noise_kind != _lib.NOISE_NONE
        check(load().vqaf_prune_mix(ptr(seq_c), ptr(w), B, N, K, D, dtype_code(seq_c), mode, k_adv,
                                    ptr(pos) if mode == 1 else None, ptr(src) if mode == 1 else None,
                                    ptr(rng_state) if mode == 2 else None, int(salt), ptr(noise_c), noise_kind,
                                    float(noise_scale), ptr(rng_state) if need_noise_state else None,
                                    int(noise_off), float(alpha), dtype_code(out), ptr(out), ptr(idx),
                                    stream_ptr(seq.device)), "vqaf_prune_mix")
        ctx.save_for_backward(idx)
        ctx.meta = (B, N, K, D, float(alpha), seq.dtype)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return out, idx

    @staticmethod
    def backward(ctx, grad_out, _grad_idx):
        if grad_out is None:
            return (None,) * 15
        (idx,) = ctx.saved_tensors
        B, N, K, D, alpha, sdt = ctx.meta
        g = grad_out.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        grad_seq = torch.empty(B, N, D, dtype=sdt, device=g.device)
        check(load().vqaf_gather_mix_bwd(ptr(g), ptr(idx), alpha, B, N, K, D, dtype_code(g),
                                         dtype_code(grad_seq), ptr(grad_seq), stream_ptr(g.device)),
              "vqaf_gather_mix_bwd")
        return (grad_seq,) + (None,) * 14


def prune_mix(seq: torch.Tensor, weights: torch.Tensor, k: int, alpha: float, k_adv: int = 0,
              pos: Optional[torch.Tensor] = None, src: Optional[torch.Tensor] = None,
              noise: Optional[torch.Tensor] = None, noise_type: Optional[str] = None, noise_scale: float = 0.0,
              rng: Optional[PhiloxState] = None, out_dtype: Optional[torch.dtype] = None, salt: int = 0):
    """(out (B,K+1,D) or (B,K,D) if alpha <= 0, keep indices (B,K) int64) from attention weights in
    ONE launch.  Same results and the same random streams as select_indices() followed by
    gather_mix(): the adversarial draws come from the state's current counter, the noise from the
    next one when both are drawn, and the state then advances once per stream used."""
    if weights.dim() == 3:
        weights = weights.squeeze(1)
    w = weights.detach().to(torch.float32).contiguous()
    mode = 0
    if k_adv > 0:
        mode = 1 if pos is not None else 2
        if mode == 1:
            pos, src = pos.to(torch.int64).contiguous(), src.to(torch.int64).contiguous()
    kind = _noise_kind(noise_type) if noise is None else _lib.NOISE_NONE
    draws_noise = kind != _lib.NOISE_NONE
    state = None
    if mode == 2 or draws_noise:
        rng = rng or default_rng(seq.device)
        state = rng.state
    noise_off = 1 if (mode == 2 and draws_noise) else 0
    out, idx = _PruneMix.apply(seq, w, int(k), mode, int(k_adv), pos, src, state, salt, noise, kind,
                               noise_scale, noise_off, alpha, out_dtype or seq.dtype)
    if state is not None:
        rng.advance((1 if mode == 2 else 0) + (1 if draws_noise else 0))
    return out, idx


class _MeanPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, M, D = x.shape
        xc = x.contiguous()
        out = torch.empty(B, D, dtype=x.dtype, device=x.device)
        check(load().vqaf_mean_pool(ptr(xc), B, M, D, dtype_code(xc), ptr(out), D, 0, dtype_code(out),
                                    stream_ptr(x.device)), "vqaf_mean_pool")
        ctx.meta = (B, M, D, x.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        B, M, D, dt = ctx.meta
        g = grad_out.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        gx = torch.empty(B, M, D, dtype=dt, device=g.device)
        check(load().vqaf_mean_pool_bwd(ptr(g), D, 0, dtype_code(g), B, M, D, dtype_code(gx), ptr(gx),
                                        stream_ptr(g.device)), "vqaf_mean_pool_bwd")
        return gx


def mean_pool(x: torch.Tensor) -> torch.Tensor:
    """x.mean(dim=1) for (B,M,D) tokens."""
    return _MeanPool.apply(x)


class _PrunedPool(torch.autograd.Function):
    """fused[:, off:off+D] = mean_j prune(seq_m, idx_m)[:, j, :] for each modality m, without
    writing the pruned token tensors.  Inputs: a list of (seq, idx, alpha, noise, kind, scale,
    rng_state) flattened; output (B, sum D_m).  Modalities that share dtype and width (the vision
    and text halves of CoAttention) go through ONE launch forward and ONE backward."""

    @staticmethod
    def forward(ctx, n_mod, out_dtype, *flat):
        import ctypes
        mods = [flat[i * 7:(i + 1) * 7] for i in range(n_mod)]
        B = mods[0][0].shape[0]
        dev = mods[0][0].device
        widths = [m[0].shape[2] for m in mods]
        ld = sum(widths)
        out = torch.empty(B, ld, dtype=out_dtype, device=dev)
        lib, st = load(), stream_ptr(dev)
        col0 = 0
        saved, metas, keep = [], [], []
        for (seq, idx, alpha, noise, kind, scale, state), D in zip(mods, widths):
            seq_c, idx_c = seq.contiguous(), idx.contiguous()
            noise_c = noise.to(torch.float32).contiguous() if noise is not None else None
            keep.append((seq_c, idx_c, noise_c, state, float(alpha), int(kind), float(scale)))
            saved.append(idx_c)
            metas.append((seq.shape[1], idx.shape[1], D, float(alpha), seq.dtype, col0))
            col0 += D
        uniform = n_mod > 1 and n_mod <= 4 and len(set(widths)) == 1 and len({m[0].dtype for m in mods}) == 1
        if uniform:
            arr = (_lib.PoolMod * n_mod)()
            for i, ((seq_c, idx_c, noise_c, state, alpha, kind, scale), meta) in enumerate(zip(keep, metas)):
                arr[i] = _lib.PoolMod(ptr(seq_c), ptr(idx_c), ptr(noise_c), ptr(state), scale, alpha, kind,
                                      meta[0], meta[1], meta[5])
            _probe("gather_mix_pool_multi", "begin")
            check(lib.vqaf_gather_mix_pool_multi(ctypes.cast(arr, ctypes.c_void_p), n_mod, B, widths[0],
                                                 dtype_code(keep[0][0]), ptr(out), ld, dtype_code(out), st),
                  "vqaf_gather_mix_pool_multi")
            _probe("gather_mix_pool_multi", "end")
        else:
            for i, ((seq_c, idx_c, noise_c, state, alpha, kind, scale), (N, K, D, _a, _dt, c0)) in \
                    enumerate(zip(keep, metas)):
                _probe(f"gather_mix_pool:{i}", "begin")
                check(lib.vqaf_gather_mix_pool(ptr(seq_c), ptr(idx_c), ptr(noise_c), kind, scale, ptr(state),
                                               alpha, B, N, K, D, dtype_code(seq_c), ptr(out), ld, c0,
                                               dtype_code(out), st), "vqaf_gather_mix_pool")
                _probe(f"gather_mix_pool:{i}", "end")
        for (_s, _i, _n, state, _al, kind, _sc) in keep:
            if state is not None and kind != _lib.NOISE_NONE:  # next consumer gets a fresh stream
                check(lib.vqaf_rng_advance(ptr(state), 1, st), "vqaf_rng_advance")
        ctx.save_for_backward(*saved)
        ctx.metas = metas
        ctx.B, ctx.ld, ctx.n_mod, ctx.uniform = B, ld, n_mod, uniform
        return out

    @staticmethod
    def backward(ctx, grad_out):
        import ctypes
        g = grad_out.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        lib, st = load(), stream_ptr(g.device)
        grads = [None, None]
        outs = []
        for i, (N, K, D, alpha, sdt, col0) in enumerate(ctx.metas):
            need = ctx.needs_input_grad[2 + 7 * i]
            outs.append(torch.empty(ctx.B, N, D, dtype=sdt, device=g.device) if need else None)
        if ctx.uniform and any(o is not None for o in outs):
            arr = (_lib.PoolBwdMod * ctx.n_mod)()
            for i, (idx_c, (N, K, D, alpha, sdt, col0)) in enumerate(zip(ctx.saved_tensors, ctx.metas)):
                arr[i] = _lib.PoolBwdMod(ptr(idx_c), ptr(outs[i]), alpha, N, K, col0)
            sdt0 = ctx.metas[0][4]
            check(lib.vqaf_gather_mix_pool_bwd_multi(ptr(g), ctx.ld, dtype_code(g), ctypes.cast(arr, ctypes.c_void_p),
                                                     ctx.n_mod, ctx.B, ctx.metas[0][2],
                                                     _lib.BF16 if sdt0 == torch.bfloat16 else _lib.F32, st),
                  "vqaf_gather_mix_pool_bwd_multi")
        else:
            for i, (idx_c, (N, K, D, alpha, sdt, col0)) in enumerate(zip(ctx.saved_tensors, ctx.metas)):
                if outs[i] is not None:
                    check(lib.vqaf_gather_mix_pool_bwd(ptr(g), ctx.ld, col0, dtype_code(g), ptr(idx_c), alpha,
                                                       ctx.B, N, K, D, dtype_code(outs[i]), ptr(outs[i]), st),
                          "vqaf_gather_mix_pool_bwd")
        for o in outs:
            grads.extend([o, None, None, None, None, None, None])
        return tuple(grads)


FUSED_WIDTHS = (64, 128, 256, 512, 1024)


def fused_prune_pool_supported(specs) -> bool:
    """True when vqaf_fused_prune_pool_multi covers these modalities (bf16 tokens of one width and
    batch, at most 8 heads, per-sample state within 64 KiB of LDS)."""
    s0 = specs[0]["seq"]
    B, _, D = s0.shape
    H = specs[0]["num_heads"]
    if not (1 <= len(specs) <= 4) or s0.dtype != torch.bfloat16 or D not in FUSED_WIDTHS or H > 8:
        return False
    for sp in specs:
        seq = sp["seq"]
        if seq.dtype != torch.bfloat16 or seq.shape[0] != B or seq.shape[2] != D or sp["num_heads"] != H:
            return False
        N = seq.shape[1]
        NP = 64
        while NP < N:
            NP *= 2
        lds = max(32 * (D + 8), 16384) + 8 * NP + N * H * 4 + 12 * N + 8 * max(1, sp.get("k_adv", 0)) + 512
        if N > 4096 or lds > 64 * 1024:
            return False
    return True


class _FusedPrunePool(torch.autograd.Function):
    """Scoring + selection + pooled prune/mix of several modalities: two launches for the query
    folds, ONE for everything that touches tokens.  Returns (fused (B, n*D), idx_0, idx_1, ...).
    The backward is the pooled prune/mix backward (scores and indices carry no gradient)."""

    @staticmethod
    def forward(ctx, n_mod, out_dtype, rng_state, rng_advance, sel_draws, *flat):
        import ctypes
        per = 14
        mods = [flat[i * per:(i + 1) * per] for i in range(n_mod)]
        seq0 = mods[0][0]
        B, _, D = seq0.shape
        dev = seq0.device
        H = mods[0][4]
        lib, st = load(), stream_ptr(dev)
        w_dt = mods[0][2].dtype
        ld = n_mod * D
        out = torch.empty(B, ld, dtype=out_dtype, device=dev)
        folds = (_lib.FoldDesc * n_mod)()
        fm = (_lib.FusedMod * n_mod)()
        keep, idxs, metas = [], [], []
        for i, (seq, q, w, bia, _h, k, k_adv, pos, src, salt, alpha, noise, kind, scale_state) in enumerate(mods):
            noise_scale, noise_state = scale_state
            seq_c = seq.contiguous()
            N = seq_c.shape[1]
            q_c = q.reshape(B, D).to(seq_c.dtype).contiguous()
            w_c = w.detach().contiguous()
            b_c = bia.detach().to(w_c.dtype).contiguous()
            qp = torch.empty(B, D, dtype=torch.float32, device=dev)
            u = torch.empty(B, H, D, dtype=torch.float32, device=dev)
            idx = torch.empty(B, k, dtype=torch.int64, device=dev)
            mode = 0
            if k_adv > 0:
                mode = 1 if pos is not None else 2
                if mode == 1:
                    pos, src = pos.to(torch.int64).contiguous(), src.to(torch.int64).contiguous()
            noise_c = noise.to(torch.float32).contiguous() if noise is not None else None
            keep.append((seq_c, q_c, w_c, b_c, qp, u, pos, src, noise_c, noise_state))
            folds[i] = _lib.FoldDesc(ptr(q_c), ptr(w_c), ptr(b_c), ptr(qp), ptr(u))
            fm[i] = _lib.FusedMod(ptr(seq_c), ptr(u), ptr(pos) if mode == 1 else None, ptr(src) if mode == 1 else None,
                                  ptr(noise_c), ptr(noise_state), ptr(idx), None, int(salt), float(alpha),
                                  float(noise_scale), int(kind), N, int(k), mode, int(k_adv), i * D,
                                  int(sel_draws and rng_state is not None and noise_state is not None
                                      and noise_state.data_ptr() == rng_state.data_ptr()))
            idxs.append(idx)
            metas.append((N, int(k), D, float(alpha), seq.dtype, i * D))
        if rng_advance < 0:  # the fold launch advances the generator state for the launch behind it
            check(lib.vqaf_fold_query_multi_advance(ctypes.cast(folds, ctypes.c_void_p), n_mod, B, D, H,
                                                    dtype_code(seq0), _lib.BF16 if w_dt == torch.bfloat16 else _lib.F32,
                                                    ptr(rng_state), -int(rng_advance), st),
                  "vqaf_fold_query_multi_advance")
        else:
            check(lib.vqaf_fold_query_multi(ctypes.cast(folds, ctypes.c_void_p), n_mod, B, D, H, dtype_code(seq0),
                                            _lib.BF16 if w_dt == torch.bfloat16 else _lib.F32, st),
                  "vqaf_fold_query_multi")
        _probe("fused_prune_pool", "begin")
        entry = lib.vqaf_fused_prune_pool_multi
        if FUSED_ONE_PASS:
            ns = (ctypes.c_int * n_mod)(*[m[0].shape[1] for m in mods])
            if lib.vqaf_fused_prune_pool_onepass_supported(n_mod, ctypes.cast(ns, ctypes.c_void_p), D, H):
                entry = lib.vqaf_fused_prune_pool_onepass_multi
        check(entry(ctypes.cast(fm, ctypes.c_void_p), n_mod, B, D, H, dtype_code(seq0),
                    ptr(rng_state), int(rng_advance), ptr(out), ld, dtype_code(out), st),
              "vqaf_fused_prune_pool_multi")
        _probe("fused_prune_pool", "end")
        if not rng_advance:  # the kernel did not bump the counter itself: one launch per noise stream
            for (_s, _q, _w, _b, _qp, _u, _p, _sr, noise_c, noise_state), m in zip(keep, mods):
                if noise_state is not None and m[12] != _lib.NOISE_NONE and noise_c is None:
                    check(lib.vqaf_rng_advance(ptr(noise_state), 1, st), "vqaf_rng_advance")
        ctx.save_for_backward(*idxs)
        ctx.metas = metas
        ctx.B, ctx.ld, ctx.n_mod, ctx.per = B, ld, n_mod, per
        ctx.mark_non_differentiable(*idxs)
        ctx.set_materialize_grads(False)  # no zero-filled int64 "gradients" for the index outputs
        return (out, *idxs)

    @staticmethod
    def backward(ctx, grad_out, *_unused):
        import ctypes
        if grad_out is None:
            return (None,) * (5 + ctx.per * ctx.n_mod)
        g = grad_out.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        lib, st = load(), stream_ptr(g.device)
        outs = []
        for i, (N, K, D, alpha, sdt, col0) in enumerate(ctx.metas):
            need = ctx.needs_input_grad[5 + ctx.per * i]
            outs.append(torch.empty(ctx.B, N, D, dtype=sdt, device=g.device) if need else None)
        if any(o is not None for o in outs):
            arr = (_lib.PoolBwdMod * ctx.n_mod)()
            for i, (idx_c, (N, K, D, alpha, sdt, col0)) in enumerate(zip(ctx.saved_tensors, ctx.metas)):
                arr[i] = _lib.PoolBwdMod(ptr(idx_c), ptr(outs[i]), alpha, N, K, col0)
            check(lib.vqaf_gather_mix_pool_bwd_multi(ptr(g), ctx.ld, dtype_code(g), ctypes.cast(arr, ctypes.c_void_p),
                                                     ctx.n_mod, ctx.B, ctx.metas[0][2], _lib.BF16, st),
                  "vqaf_gather_mix_pool_bwd_multi")
        grads = [None, None, None, None, None]
        for o in outs:
            grads.extend([o] + [None] * (ctx.per - 1))
        return tuple(grads)


def fused_prune_pool(specs, out_dtype: torch.dtype, rng: Optional["PhiloxState"] = None):
    """specs: list of dicts(q, seq, in_proj_weight, in_proj_bias, num_heads, k, k_adv, pos, src, salt,
    alpha, noise, noise_type, noise_scale, noise_rng).  Returns (fused, [keep indices], drew)."""
    flat, drew, noise_states = [], False, []
    dev = specs[0]["seq"].device
    for i, sp in enumerate(specs):
        kind = _noise_kind(sp.get("noise_type")) if sp.get("noise") is None else _lib.NOISE_NONE
        state = None
        if kind != _lib.NOISE_NONE:
            state = (sp.get("noise_rng") or default_rng(dev)).state
            noise_states.append(state)
        k_adv = int(sp.get("k_adv", 0))
        if k_adv > 0 and sp.get("pos") is None:
            drew = True
        flat.extend([sp["seq"], sp["q"], sp["in_proj_weight"], sp["in_proj_bias"], sp["num_heads"], sp["k"], k_adv,
                     sp.get("pos"), sp.get("src"), sp.get("salt", i), sp["alpha"], sp.get("noise"), kind,
                     (sp.get("noise_scale", 0.0), state)])
    if drew and rng is None:
        rng = default_rng(dev)
    # Every stream this launch draws from: the selection's and the in-kernel noise streams.  When
    # they are one and the same state the kernel's last workgroup bumps its counter (once per
    # stream position consumed) and no counter launch follows; the caller then has nothing to advance.
    streams = list(noise_states) + ([rng.state] if drew else [])
    # (off by default: the bump costs every workgroup a returning device-scope atomic at its end, which
    # at thousands of workgroups per launch is slower than the one-thread launch it saves)
    one_state = bool(streams) and all(t.data_ptr() == streams[0].data_ptr() for t in streams)
    in_kernel = IN_KERNEL_RNG_BUMP and one_state and streams[0].numel() >= 4
    # (the default) the query-fold launch in front of the token launch advances the counter on the side, and
    # the token launch draws from the positions in front of the advance: same values, no counter launch
    ahead = ADVANCE_RNG_IN_FOLD and one_state and not in_kernel
    res = _FusedPrunePool.apply(len(specs), out_dtype,
                                streams[0] if (in_kernel or ahead) else (rng.state if drew else None),
                                len(streams) if in_kernel else (-len(streams) if ahead else 0), drew, *flat)
    return res[0], list(res[1:]), drew and not (in_kernel or ahead)


def pruned_pool(mods, out_dtype: torch.dtype) -> torch.Tensor:
    """mods: list of dicts(seq, idx, alpha, noise, noise_type, noise_scale, rng)."""
    flat = []
    for m in mods:
        kind = _noise_kind(m.get("noise_type")) if m.get("noise") is None else _lib.NOISE_NONE
        state = None
        if kind != _lib.NOISE_NONE:
            state = (m.get("rng") or default_rng(m["seq"].device)).state
        flat.extend([m["seq"], m["idx"], m["alpha"], m.get("noise"), kind, m.get("noise_scale", 0.0), state])
    return _PrunedPool.apply(len(mods), out_dtype, *flat)


# --------------------------------------------------------------------------------------------
# Small helpers used by the DP-SGD engine and the MIA scorer
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def ce_conf(logits: torch.Tensor, targets: Optional[torch.Tensor] = None, want_loss: bool = True,
            want_conf: bool = True, want_pred: bool = False, grad_scale: Optional[float] = None,
            grad_dtype: torch.dtype = torch.float32):
    """Per-sample cross-entropy, max-softmax confidence, argmax and (optionally) the gradient of
    grad_scale * sum_i CE_i w.r.t. the logits (f32, or bf16 on request).  Returns (loss, conf, pred, grad)."""
    B, C = logits.shape
    lg = logits.contiguous()
    dev = lg.device
    loss = torch.empty(B, dtype=torch.float32, device=dev) if (want_loss and targets is not None) else None
    conf = torch.empty(B, dtype=torch.float32, device=dev) if want_conf else None
    pred = torch.empty(B, dtype=torch.int64, device=dev) if want_pred else None
    grad = torch.empty(B, C, dtype=grad_dtype, device=dev) if grad_scale is not None else None
    tg = targets.to(torch.int64).contiguous() if targets is not None else None
    check(load().vqaf_ce_conf(ptr(lg), ptr(tg), B, C, dtype_code(lg), ptr(loss), ptr(conf), ptr(pred),
                              ptr(grad), dtype_code(grad) if grad is not None else 0, float(grad_scale or 0.0),
                              stream_ptr(dev)), "vqaf_ce_conf")
    return loss, conf, pred, grad


_ce_tickets: dict = {}
_zero_padded: dict = {}  # id(buffer) -> weak reference: (B, ld) buffers whose columns past the logical width are zero


def zero_padded_pitch(t: torch.Tensor) -> Optional[int]:
    """Row pitch of `t` if it is the leading-columns view of one of the zero-padded gradient buffers the
    mean cross-entropy hands out (the library's bf16 GEMMs may then read whole 16-byte chunks past its
    logical width), else None."""
    base = t._base
    if base is None or t.dim() != 2 or t.stride(1) != 1 or t.storage_offset() != 0:
        return None
    ref = _zero_padded.get(id(base))
    if ref is None or ref() is not base or t.stride(0) != base.shape[1] or t.shape[0] != base.shape[0]:
        return None
    return int(t.stride(0))


class _MeanCE(torch.autograd.Function):
    """Mean cross-entropy (nn.CrossEntropyLoss() as the training loop uses it, train.py:104,
    utils/src.py:96) with its logits gradient produced by the same kernel launch."""

    @staticmethod
    def forward(ctx, logits, targets):
        B = logits.shape[0]
        # the gradient is written in the logits' dtype: with the usual incoming gradient of 1 it is handed on as is
        gdt = logits.dtype if logits.dtype in (torch.float32, torch.bfloat16) else torch.float32
        lg = logits.contiguous()
        dev = lg.device
        C = lg.shape[1]
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        # bf16 gradient rows are padded with zeros to a multiple of 8 elements (16 bytes): a 430-class head's
        # backprop is then a legal operand of the library's own bf16 GEMMs (input gradient, clipped sum)
        ldg = (C + 7) // 8 * 8 if gdt == torch.bfloat16 else C
        buf = torch.empty(B, ldg, dtype=gdt, device=dev)
        if ldg != C:
            _zero_padded[id(buf)] = weakref.ref(buf, lambda _r, k=id(buf): _zero_padded.pop(k, None))
        grad = buf[:, :C]
        mean = torch.empty((), dtype=torch.float32, device=dev)
        # one self-resetting ticket word per device, made before any graph capture can be under way (a
        # tensor allocated during a capture would live in that graph's memory pool); mean losses computed
        # on different streams of one device must not overlap
        ticket = _ce_tickets.get(dev)
        if ticket is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("cross_entropy_mean: run one step eagerly before capturing a HIP graph")
            ticket = _ce_tickets[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
        check(load().vqaf_ce_mean(ptr(lg), ptr(targets.to(torch.int64).contiguous()), B, C, dtype_code(lg), ptr(loss),
                                  ptr(buf), ldg, dtype_code(buf), 1.0 / B, ptr(mean), ptr(ticket), stream_ptr(dev)),
              "vqaf_ce_mean")
        ctx.buf = buf  # the unit-gradient backward hands on a view of it (its base stays this object)
        ctx.save_for_backward(grad)
        ctx.in_dtype = logits.dtype
        # kept for the rare backward with a non-unit incoming gradient (see backward); no copy is made
        ctx.logits, ctx.targets = (logits.detach(), targets) if gdt != torch.float32 else (None, None)
        return mean

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        if grad_out is _unit_grads.get((grad_out.device, grad_out.dtype)) and grad.dtype == ctx.in_dtype:
            # loss.backward(gradient=unit_gradient(loss)): the factor is exactly 1
            return ctx.buf[:, :grad.shape[1]], None
        if grad.dtype == torch.float32 and ctx.in_dtype in (torch.float32, torch.bfloat16) and grad_out.numel() == 1:
            # one launch: the product is rounded straight into the logits' dtype
            out = torch.empty(grad.shape, dtype=ctx.in_dtype, device=grad.device)
            go = grad_out.reshape(1).to(torch.float32)  # a view for the usual f32 scalar
            check(load().vqaf_scale_cast(ptr(grad), ptr(go), grad.numel(), dtype_code(out), ptr(out),
                                         stream_ptr(grad.device)), "vqaf_scale_cast")
            return out, None
        # bf16 logits and another incoming gradient (a weighted or scaled loss): the saved gradient was
        # rounded to bf16 for the unit-gradient hand-over, so scaling it would round twice; form it again
        # from the logits in f32 and round the product once
        if grad.dtype != torch.float32 and ctx.logits is not None:
            _, _, _, g32 = ce_conf(ctx.logits, ctx.targets, want_loss=False, want_conf=False,
                                   grad_scale=1.0 / grad.shape[0], grad_dtype=torch.float32)
            return (g32 * grad_out.float()).to(ctx.in_dtype), None
        return (grad.float() * grad_out).to(ctx.in_dtype), None


def cross_entropy_mean(logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    return _MeanCE.apply(logits, targets)


_unit_grads: dict = {}


def unit_gradient(loss: torch.Tensor) -> torch.Tensor:
    """A cached scalar 1 of the loss's dtype and device for `loss.backward(gradient=...)`: what
    `loss.backward()` uses, without the fill launch autograd spends on making it every step."""
    key = (loss.device, loss.dtype)
    one = _unit_grads.get(key)
    if one is None:
        one = torch.ones((), dtype=loss.dtype, device=loss.device)
        _unit_grads[key] = one
    return one


def gemm_bf16_supported(lda: int, a_kmajor: bool, ldb: int, b_kmajor: bool, M: int, N: int, K: int) -> bool:
    return bool(load().vqaf_gemm_bf16_supported(lda, int(a_kmajor), ldb, int(b_kmajor), M, N, K))


@torch.no_grad()
def gemm_bf16(A: torch.Tensor, a_kmajor: bool, B: torch.Tensor, b_kmajor: bool, M: int, N: int, K: int,
              out: torch.Tensor, bias: Optional[torch.Tensor] = None, accumulate: bool = False,
              colsum: Optional[torch.Tensor] = None, name: Optional[str] = None) -> torch.Tensor:
    """out (M,N) = op(A) op(B) (+ bias) on bf16 MFMA (vqaf_gemm_bf16); A, B are 2-D bf16 tensors whose
    second stride is 1 (row pitch = first stride), out f32 or bf16."""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.stride(1) == 1 and B.stride(1) == 1
    assert out.stride(1) == 1
    if name is not None:
        _probe(name, "begin")
    check(load().vqaf_gemm_bf16(A.data_ptr(), A.stride(0), int(a_kmajor), B.data_ptr(), B.stride(0), int(b_kmajor),
                                M, N, K, ptr(bias), dtype_code(bias) if bias is not None else 0, out.data_ptr(),
                                out.stride(0), dtype_code(out), int(accumulate), ptr(colsum),
                                stream_ptr(A.device)), "vqaf_gemm_bf16")
    if name is not None:
        _probe(name, "end")
    return out


@torch.no_grad()
def gemm_bf16_wgrad_group(items, K: int, accumulate: bool = False, name: Optional[str] = None) -> bool:
    """items: list of (A (K,M) bf16, B (K,N) bf16, C (M,N) f32 view, colsum (M) f32 or None).  Launches all
    C_i = A_i^T B_i in one kernel (vqaf_gemm_bf16_wgrad_group); returns False, having launched nothing, when
    the products cannot share a launch (the caller then issues them one by one)."""
    import ctypes
    n = len(items)
    if not 1 <= n <= 4:
        return False
    for A, B, Cm, cs in items:
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and Cm.dtype == torch.float32
        assert A.stride(1) == 1 and B.stride(1) == 1 and Cm.stride(1) == 1 and A.shape[0] == K and B.shape[0] == K
    vp, lg, it = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    has_cs = any(cs is not None for _, _, _, cs in items)
    if name is not None:
        _probe(name, "begin")
    status = load().vqaf_gemm_bf16_wgrad_group(
        n, vp(*[A.data_ptr() for A, _, _, _ in items]), lg(*[A.stride(0) for A, _, _, _ in items]),
        vp(*[B.data_ptr() for _, B, _, _ in items]), lg(*[B.stride(0) for _, B, _, _ in items]),
        it(*[A.shape[1] for A, _, _, _ in items]), it(*[B.shape[1] for _, B, _, _ in items]), K,
        vp(*[Cm.data_ptr() for _, _, Cm, _ in items]), lg(*[Cm.stride(0) for _, _, Cm, _ in items]), int(accumulate),
        vp(*[None if cs is None else cs.data_ptr() for _, _, _, cs in items]) if has_cs else None,
        stream_ptr(items[0][0].device))
    if name is not None:
        _probe(name, "end")
    if status == -1:  # VQAF_E_SHAPE: not groupable
        return False
    check(status, "vqaf_gemm_bf16_wgrad_group")
    return True


def linear_nt_supported(x2: torch.Tensor, w: torch.Tensor, w_kmajor: bool = False) -> bool:
    """Whether y = x2 op(w) can run on vqaf_linear_nt: x2 (M,K); w (N,K), or (K,N) with w_kmajor; both bf16 with
    unit inner stride."""
    for t in (x2, w):
        if t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1 or t.data_ptr() % 16:
            return False
    M, K = x2.shape
    N = w.shape[1] if w_kmajor else w.shape[0]
    if (w.shape[0] if w_kmajor else w.shape[1]) != K:
        return False
    return bool(load().vqaf_linear_nt_supported(x2.stride(0), w.stride(0), int(w_kmajor), N, M, N, K))


@torch.no_grad()
def linear_nt(x2: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, w_kmajor: bool = False,
              epilogue: int = 0, aux: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y (M,N) bf16 = x2 (M,K) op(w) (+ bias) on the library's large-tile bf16 MFMA kernel (vqaf_linear_nt).
    epilogue 1: aux (M,N), allocated by the caller, receives gelu(y); epilogue 2: y = (x2 op(w)) * gelu'(aux)."""
    M, K = x2.shape
    N = w.shape[1] if w_kmajor else w.shape[0]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    check(load().vqaf_linear_nt(x2.data_ptr(), x2.stride(0), w.data_ptr(), w.stride(0), int(w_kmajor), ptr(bias),
                                dtype_code(bias) if bias is not None else 0, M, N, K, y.data_ptr(), N, epilogue,
                                ptr(aux), aux.stride(0) if aux is not None else 0, stream_ptr(x2.device)),
          "vqaf_linear_nt")
    return y


def wgrad_tn_supported(gy2: torch.Tensor, x2: torch.Tensor) -> bool:
    if gy2.dtype != torch.bfloat16 or x2.dtype != torch.bfloat16 or gy2.stride(1) != 1 or x2.stride(1) != 1:
        return False
    if gy2.data_ptr() % 16 or x2.data_ptr() % 16 or gy2.shape[0] != x2.shape[0]:
        return False
    K, M = gy2.shape
    N = x2.shape[1]
    return bool(load().vqaf_wgrad_tn_supported(gy2.stride(0), x2.stride(0), N, M, N, K))


@torch.no_grad()
def wgrad_tn(gy2: torch.Tensor, x2: torch.Tensor, w_dtype: torch.dtype = torch.bfloat16,
             bias_dtype: Optional[torch.dtype] = torch.bfloat16):
    """(gy2^T x2 (M,N) `w_dtype`, column sums of gy2 (M) `bias_dtype` or None) for gy2 (K,M), x2 (K,N) bf16:
    weight and bias gradient of a Linear whose batch has K tokens (vqaf_wgrad_tn)."""
    K, M = gy2.shape
    N = x2.shape[1]
    lib = load()
    gw = torch.empty(M, N, dtype=w_dtype, device=gy2.device)
    gb = torch.empty(M, dtype=bias_dtype, device=gy2.device) if bias_dtype is not None else None
    ws = torch.empty(lib.vqaf_wgrad_tn_workspace_bytes(M, N, K), dtype=torch.uint8, device=gy2.device)
    check(lib.vqaf_wgrad_tn(gy2.data_ptr(), gy2.stride(0), x2.data_ptr(), x2.stride(0), M, N, K, gw.data_ptr(), N,
                            dtype_code(gw), ptr(gb), dtype_code(gb) if gb is not None else 0, ws.data_ptr(),
                            ws.numel(), stream_ptr(gy2.device)), "vqaf_wgrad_tn")
    return gw, gb


def keep_count(n_tokens: int, prune_ratio: float) -> int:
    """k = max(1, int(N * (1 - prune_ratio))) in Python double arithmetic (utils/fusion.py:41)."""
    return max(1, int(n_tokens * (1 - prune_ratio)))


def adv_count(k: int, adv_ratio: float) -> int:
    """k_adv = max(1, int(k * adv_ratio)) (utils/fusion.py:54)."""
    return max(1, int(k * adv_ratio))
