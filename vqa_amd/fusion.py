"""Drop-in fusion modules: same classes, constructor arguments, attribute and parameter names
as utils/fusion.py of the reference, running on the libvqafuse HIP kernels.

    PrivacyAwareTokenPruning  utils/fusion.py:5-116
    Concat                    utils/fusion.py:118-126
    Attention                 utils/fusion.py:129-183
    CoAttention               utils/fusion.py:186-275

Differences a user can observe, all documented where they occur:
  * tie order of equal attention weights is defined (weight desc, index asc); torch.topk leaves
    it unspecified;
  * random draws (adversarial positions, noise) come from a counter-based generator seeded by
    torch's global seed, so the stream differs from torch.randperm / torch.normal (as the
    reference's own CPU and GPU streams differ); pre-drawn tensors can be injected;
  * tensors keep their storage dtype (a bf16 `seq` gives a bf16 result; the reference's
    f32 mask/noise temporaries would silently promote it to f32); sums accumulate in f32.
"""
from __future__ import annotations

from typing import Optional

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class PrivacyAwareTokenPruning(nn.Module):
    """Attention-score token pruning with adversarial index mixing, noise and a mix token.

    Same hyper-parameters and defaults as utils/fusion.py:14-24.
    """

    def __init__(self, prune_ratio=0.5, mixup_alpha=0.05, adversarial_mix=True, adv_ratio=0.2,
                 noise_injection=True, noise_scale=0.01, noise_type='gaussian'):
        super().__init__()
        self.prune_ratio = prune_ratio
        self.mixup_alpha = mixup_alpha
        self.adversarial_mix = adversarial_mix
        self.adv_ratio = adv_ratio
        self.noise_injection = noise_injection
        self.noise_scale = noise_scale
        self.noise_type = noise_type
        self.rng: Optional[ops.PhiloxState] = None  # None -> per-device default stream
        self.last_indices: Optional[torch.Tensor] = None
        self._drew = False

    # -- index selection: utils/fusion.py:35-65 ------------------------------------------------
    @torch.no_grad()
    def select(self, attn_weights: Optional[torch.Tensor], n_tokens: int,
               replace_positions: Optional[torch.Tensor] = None,
               random_bottom: Optional[torch.Tensor] = None,
               head_scores: Optional[torch.Tensor] = None, salt: int = 0,
               advance: bool = True) -> torch.Tensor:
        """Keep indices (B,K) int64, one kernel launch.  Give either `attn_weights` ((B,1,N) or
        (B,N), the module-boundary form) or raw `head_scores` (B,N,H) whose softmax / head mean is
        then fused into the same launch.  `replace_positions` / `random_bottom` (B,k_adv) inject
        the draws of utils/fusion.py:59,62 (positions into the keep list / into the bottom list)."""
        k = ops.keep_count(n_tokens, self.prune_ratio)
        remaining_k = n_tokens - k
        k_adv = 0
        if self.adversarial_mix:
            k_adv = ops.adv_count(k, self.adv_ratio)
            # utils/fusion.py:57 -- the swap is skipped when fewer than k_adv tokens were dropped
            if remaining_k < k_adv:
                k_adv = 0
        if (replace_positions is None) != (random_bottom is None):
            raise ValueError("inject both replace_positions and random_bottom, or neither")
        res = ops.select_indices(k, scores=head_scores, weights=attn_weights if head_scores is None else None,
                                 k_adv=k_adv, pos=replace_positions if k_adv else None,
                                 src=random_bottom if k_adv else None, rng=self.rng, salt=salt,
                                 advance=advance)
        self._drew = res["drew"]  # True if the kernel consumed random numbers (caller may batch the advance)
        return res["top"]

    def plan(self, n_tokens: int):
        """(k, k_adv) for a sequence of n_tokens: tokens kept (utils/fusion.py:41) and keep slots
        swapped adversarially (:54; 0 when mixing is off or skipped by the guard at :57)."""
        k = ops.keep_count(n_tokens, self.prune_ratio)
        k_adv = 0
        if self.adversarial_mix:
            k_adv = ops.adv_count(k, self.adv_ratio)
            if n_tokens - k < k_adv:
                k_adv = 0
        return k, k_adv

    def _noise_args(self, noise: Optional[torch.Tensor]):
        if not self.noise_injection:
            return None, None
        if self.noise_type not in ('gaussian', 'uniform'):
            raise ValueError(f"Unsupported noise type: {self.noise_type}")  # utils/fusion.py:111
        if noise is not None:
            return noise, None
        return None, self.noise_type

    def forward(self, seq, attn_weights, replace_positions=None, random_bottom=None, noise=None):
        """seq (B,N,D), attn_weights (B,1,N) or (B,N) -> (B,K+1,D) ((B,K,D) if mixup_alpha <= 0)."""
        B, N, D = seq.shape
        noise_t, noise_type = self._noise_args(noise)
        if (replace_positions is None) != (random_bottom is None):
            raise ValueError("inject both replace_positions and random_bottom, or neither")
        if attn_weights is not None and ops.prune_mix_supported(N, D, seq.dtype):
            # one launch: the row loads do not wait for the ranking
            k, k_adv = self.plan(N)
            out, idx = ops.prune_mix(seq, attn_weights, k, float(self.mixup_alpha), k_adv=k_adv,
                                     pos=replace_positions if k_adv else None,
                                     src=random_bottom if k_adv else None, noise=noise_t,
                                     noise_type=noise_type, noise_scale=float(self.noise_scale), rng=self.rng)
            self.last_indices = idx
            self._drew = k_adv > 0 and replace_positions is None
            return out
        idx = self.select(attn_weights, N, replace_positions, random_bottom)
        self.last_indices = idx
        return ops.gather_mix(seq, idx, float(self.mixup_alpha), noise=noise_t, noise_type=noise_type,
                              noise_scale=float(self.noise_scale), rng=self.rng)

    def pool_spec(self, seq, attn_weights, replace_positions=None, random_bottom=None, noise=None,
                  head_scores=None, salt=0, advance=True):
        """Arguments for ops.pruned_pool: the same pruning, consumed as a token mean."""
        B, N, D = seq.shape
        idx = self.select(attn_weights, N, replace_positions, random_bottom, head_scores=head_scores,
                          salt=salt, advance=advance)
        self.last_indices = idx
        noise_t, noise_type = self._noise_args(noise)
        return dict(seq=seq, idx=idx, alpha=float(self.mixup_alpha), noise=noise_t,
                    noise_type=noise_type, noise_scale=float(self.noise_scale), rng=self.rng)


class Concat(nn.Module):
    """cat([v_global, q_global]) -- utils/fusion.py:118-126 (no kernel needed)."""

    def __init__(self, v_dim=512, q_dim=512):
        super().__init__()
        self.output_dim = v_dim + q_dim

    def forward(self, v_seq, v_global, q_seq, q_global):
        return torch.cat([v_global, q_global], dim=1)


def _scores(mha: nn.MultiheadAttention, query: torch.Tensor, seq: torch.Tensor) -> torch.Tensor:
    """Head-averaged attention weights (B,N): what the reference's MHA call returns."""
    return ops.attention_scores(query, seq, mha.in_proj_weight, mha.in_proj_bias, mha.num_heads)


@torch.no_grad()
def _scores_reference_order(mha: nn.MultiheadAttention, query: torch.Tensor, seq: torch.Tensor) -> torch.Tensor:
    """The same weights computed in the operation order of nn.MultiheadAttention itself
    (torch.nn.functional.multi_head_attention_forward as called at utils/fusion.py:234-238): every token is
    projected to its key first (one (2E)-wide in-projection GEMM for key and value), the query is scaled by
    sqrt(1/dh), scores by bmm, softmax, mean over heads.  The kernels fold the query into the key projection
    instead, which is equal in exact arithmetic and differs in the last bits; this form spends the
    2*N*E*2E flops per sample of the projection to rank tokens exactly as the reference formulation does
    (`score_order = "reference"` on the fusion modules).  f32 library GEMMs on the GPU; forward only."""
    B, N, E = seq.shape
    H = mha.num_heads
    dh = E // H
    w_q, w_kv = mha.in_proj_weight.split([E, 2 * E])
    b_q, b_kv = mha.in_proj_bias.split([E, 2 * E])
    q = F.linear(query.float().reshape(B, 1, E).transpose(0, 1), w_q.float(), b_q.float())
    kv = F.linear(seq.float().transpose(0, 1), w_kv.float(), b_kv.float())
    k = kv.unflatten(-1, (2, E))[..., 0, :].contiguous()
    q = q.reshape(1, B * H, dh).transpose(0, 1)
    k = k.reshape(N, B * H, dh).transpose(0, 1)
    scores = torch.bmm(q * math.sqrt(1.0 / float(dh)), k.transpose(-2, -1))
    return torch.softmax(scores, dim=-1).view(B, H, 1, N).mean(dim=1).reshape(B, N)


def _head_scores(mha: nn.MultiheadAttention, query: torch.Tensor, seq: torch.Tensor) -> torch.Tensor:
    """Raw per-head scores (B,N,H); their softmax / head mean is fused into the selection kernel."""
    return ops.head_score_matrix(query, seq, mha.in_proj_weight, mha.in_proj_bias, mha.num_heads)


class Attention(nn.Module):
    """text -> image attention with optional token pruning (utils/fusion.py:129-183)."""

    def __init__(self, embed_dim=512, num_heads=8, use_pruning=False, prune_ratio=0.5,
                 mixup_alpha=0.05, adversarial_mix=True, adv_ratio=0.2):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads, batch_first=True)
        self.output_dim = embed_dim * 2
        self.use_pruning = use_pruning
        self.fuse_pool = True  # skip materialising the pruned tokens (same result)
        self.fuse_select = True  # scoring + selection + pooling of a sample in one workgroup (bf16 tokens)
        self.score_order = "fold"  # "reference": rank tokens with the reference's own operation order
        self.batch_launches = True  # dense (non-pruning) mode: fused token passes and shared launches
        if use_pruning:
            # utils/fusion.py:141-146 passes no noise arguments, so the class defaults apply
            # (noise_injection=True, noise_scale=0.01, gaussian).
            self.pruning = PrivacyAwareTokenPruning(prune_ratio=prune_ratio, mixup_alpha=mixup_alpha,
                                                    adversarial_mix=adversarial_mix, adv_ratio=adv_ratio)

    def forward(self, v_seq, v_global, q_seq, q_global, **inject):
        ops.require_gpu(v_seq)  # no CPU fallback: fail before anything else is touched
        if self.use_pruning and self.score_order == "reference":
            w = _scores_reference_order(self.attention, q_global, v_seq)
            attended_v = ops.mean_pool(self.pruning(v_seq, w, **inject))
        elif self.use_pruning:
            if self.score_order != "fold":
                raise ValueError(f"score_order must be 'fold' or 'reference', got {self.score_order!r}")
            spec = None
            if self.fuse_pool and self.fuse_select:
                k, k_adv = self.pruning.plan(v_seq.shape[1])
                noise_t, noise_type = self.pruning._noise_args(inject.get("noise"))
                spec = dict(q=q_global, seq=v_seq, in_proj_weight=self.attention.in_proj_weight,
                            in_proj_bias=self.attention.in_proj_bias, num_heads=self.attention.num_heads, k=k,
                            k_adv=k_adv, pos=inject.get("replace_positions"), src=inject.get("random_bottom"), salt=0,
                            alpha=float(self.pruning.mixup_alpha), noise=noise_t, noise_type=noise_type,
                            noise_scale=float(self.pruning.noise_scale), noise_rng=self.pruning.rng)
                if not ops.fused_prune_pool_supported([spec]):
                    spec = None
            if spec is not None:
                rng = self.pruning.rng or ops.default_rng(v_seq.device)
                attended_v, (idx,), drew = ops.fused_prune_pool([spec], v_seq.dtype, rng)
                if drew:
                    rng.advance()
                self.pruning.last_indices = idx
            elif self.fuse_pool:
                hs = _head_scores(self.attention, q_global, v_seq)
                attended_v = ops.pruned_pool([self.pruning.pool_spec(v_seq, None, head_scores=hs, **inject)],
                                             v_seq.dtype)
            else:
                w = _scores(self.attention, q_global, v_seq)
                attended_v = ops.mean_pool(self.pruning(v_seq, w, **inject))
        else:
            from .xattn import dense_coattention, dense_coattention_supported, single_query_attention
            if self.batch_launches and dense_coattention_supported((self.attention,), (q_global,), (v_seq,)):
                # the launch-sharing form of the dense branch (softmax inside the token passes, independent
                # products of the backward in shared launches), here with one branch
                attended_v = dense_coattention((self.attention,), (q_global,), (v_seq,))
            else:
                attended_v = single_query_attention(self.attention, q_global, v_seq)
        return torch.cat([attended_v, q_global.to(attended_v.dtype)], dim=1)


_SIDE_STREAMS: dict = {}


def _side_stream(device):
    """A second stream of `device` for the dense co-attention's second branch (module level: a module that
    is copied or pickled never meets a stream object); None while a graph is being captured before the
    stream exists (a stream cannot be created then)."""
    if device.type != "cuda":
        return None
    st = _SIDE_STREAMS.get(device.index)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        st = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[device.index] = st
    return st


class CoAttention(nn.Module):
    """Co-attention with privacy-aware token pruning (utils/fusion.py:186-275)."""

    def __init__(self, embed_dim=512, num_heads=8, use_pruning=False, prune_ratio=0.5,
                 mixup_alpha=0.05, adversarial_mix=True, adv_ratio=0.2,
                 noise_injection=True, noise_scale=0.01, noise_type='gaussian'):
        super().__init__()
        self.txt_to_vis_attn = nn.MultiheadAttention(embed_dim, num_heads, batch_first=True)
        self.vis_to_txt_attn = nn.MultiheadAttention(embed_dim, num_heads, batch_first=True)
        self.output_dim = embed_dim * 2
        self.use_pruning = use_pruning
        self.fuse_pool = True
        self.fuse_select = True  # scoring + selection + pooling of a sample in one workgroup (bf16 tokens)
        self.score_order = "fold"  # "reference": rank tokens with the reference's own operation order
        self.batch_branches = True  # dense (non-pruning) mode: the two attention branches share every launch
        self.overlap_branches = True  # ... or, where that form does not apply, run on two streams
        if use_pruning:
            self.v_pruning = PrivacyAwareTokenPruning(
                prune_ratio=prune_ratio, mixup_alpha=mixup_alpha, adversarial_mix=adversarial_mix,
                adv_ratio=adv_ratio, noise_injection=noise_injection, noise_scale=noise_scale,
                noise_type=noise_type)
            # text tokens never get noise (utils/fusion.py:216-224)
            self.q_pruning = PrivacyAwareTokenPruning(
                prune_ratio=prune_ratio, mixup_alpha=mixup_alpha, adversarial_mix=adversarial_mix,
                adv_ratio=adv_ratio, noise_injection=False, noise_scale=0, noise_type=noise_type)

    def forward(self, v_seq, v_global, q_seq, q_global, v_inject=None, q_inject=None):
        """v_inject / q_inject: optional dicts(replace_positions=, random_bottom=, noise=) that
        replay pre-drawn randomness in the vision / text pruning (parity tests)."""
        v_inject = v_inject or {}
        q_inject = q_inject or {}
        ops.require_gpu(v_seq, q_seq)  # no CPU fallback: fail before anything else is touched
        if self.use_pruning:
            if self.score_order == "reference":
                v_w = _scores_reference_order(self.txt_to_vis_attn, q_global, v_seq)
                q_w = _scores_reference_order(self.vis_to_txt_attn, v_global, q_seq)
                attended_v = ops.mean_pool(self.v_pruning(v_seq, v_w, **v_inject))
                attended_q = ops.mean_pool(self.q_pruning(q_seq, q_w, **q_inject))
                return torch.cat([attended_v, attended_q.to(attended_v.dtype)], dim=1)
            if self.score_order != "fold":
                raise ValueError(f"score_order must be 'fold' or 'reference', got {self.score_order!r}")
            if self.fuse_pool:
                # Both branches go through the same four launches (query folds x2, token scoring,
                # selection); they draw from one counter with different salts, advanced once.
                a_v, a_q = self.txt_to_vis_attn, self.vis_to_txt_attn
                uniform = (v_seq.shape[0] == q_seq.shape[0] and v_seq.shape[2] == q_seq.shape[2]
                           and v_seq.dtype == q_seq.dtype and a_v.num_heads == a_q.num_heads
                           and a_v.in_proj_weight.dtype == a_q.in_proj_weight.dtype
                           and (self.v_pruning.rng is self.q_pruning.rng))
                if uniform:
                    specs = []
                    for salt, (mha, query, seq, prune, inj) in enumerate((
                            (a_v, q_global, v_seq, self.v_pruning, v_inject),      # utils/fusion.py:234-240
                            (a_q, v_global, q_seq, self.q_pruning, q_inject))):    # utils/fusion.py:243-249
                        k, k_adv = prune.plan(seq.shape[1])
                        specs.append(dict(q=query, seq=seq, in_proj_weight=mha.in_proj_weight,
                                          in_proj_bias=mha.in_proj_bias, num_heads=mha.num_heads, k=k,
                                          k_adv=k_adv, pos=inj.get("replace_positions"),
                                          src=inj.get("random_bottom"), salt=salt))
                    rng = self.v_pruning.rng or ops.default_rng(v_seq.device)
                    if self.fuse_select and ops.fused_prune_pool_supported(specs):
                        # one launch scores, selects and pools both modalities (the tokens of a
                        # sample are read from HBM once)
                        v_noise, v_ntype = self.v_pruning._noise_args(v_inject.get("noise"))
                        specs[0].update(alpha=float(self.v_pruning.mixup_alpha), noise=v_noise, noise_type=v_ntype,
                                        noise_scale=float(self.v_pruning.noise_scale), noise_rng=self.v_pruning.rng)
                        specs[1].update(alpha=float(self.q_pruning.mixup_alpha), noise=None, noise_type=None,
                                        noise_scale=0.0, noise_rng=self.q_pruning.rng)
                        fused, (v_idx, q_idx), drew = ops.fused_prune_pool(specs, v_seq.dtype, rng)
                        if drew:
                            rng.advance()
                        self.v_pruning.last_indices, self.q_pruning.last_indices = v_idx, q_idx
                        return fused
                    (v_idx, q_idx), drew = ops.score_select_multi(specs, rng)
                    if drew:
                        rng.advance()
                    self.v_pruning.last_indices, self.q_pruning.last_indices = v_idx, q_idx
                    v_noise, v_ntype = self.v_pruning._noise_args(v_inject.get("noise"))
                    v_spec = dict(seq=v_seq, idx=v_idx, alpha=float(self.v_pruning.mixup_alpha), noise=v_noise,
                                  noise_type=v_ntype, noise_scale=float(self.v_pruning.noise_scale),
                                  rng=self.v_pruning.rng)
                    q_spec = dict(seq=q_seq, idx=q_idx, alpha=float(self.q_pruning.mixup_alpha), noise=None,
                                  noise_type=None, noise_scale=0.0, rng=self.q_pruning.rng)
                    return ops.pruned_pool([v_spec, q_spec], v_seq.dtype)          # :253-254, :274
                v_hs = _head_scores(self.txt_to_vis_attn, q_global, v_seq)   # utils/fusion.py:234-238
                v_spec = self.v_pruning.pool_spec(v_seq, None, head_scores=v_hs, salt=0, advance=False,
                                                  **v_inject)
                q_hs = _head_scores(self.vis_to_txt_attn, v_global, q_seq)   # utils/fusion.py:243-247
                q_spec = self.q_pruning.pool_spec(q_seq, None, head_scores=q_hs, salt=1, advance=False,
                                                  **q_inject)
                rngs = {id(r): r for m, r in ((self.v_pruning, self.v_pruning.rng or ops.default_rng(v_seq.device)),
                                              (self.q_pruning, self.q_pruning.rng or ops.default_rng(v_seq.device)))
                        if m._drew}
                for r in rngs.values():
                    r.advance()
                return ops.pruned_pool([v_spec, q_spec], v_seq.dtype)        # :253-254, :274
            v_w = _scores(self.txt_to_vis_attn, q_global, v_seq)
            v_pruned = self.v_pruning(v_seq, v_w, **v_inject)         # :240
            q_w = _scores(self.vis_to_txt_attn, v_global, q_seq)
            q_pruned = self.q_pruning(q_seq, q_w, **q_inject)         # :249
            attended_v = ops.mean_pool(v_pruned)
            attended_q = ops.mean_pool(q_pruned)
        else:
            from .xattn import dense_coattention, dense_coattention_supported, single_query_attention
            mhas, queries, seqs = (self.txt_to_vis_attn, self.vis_to_txt_attn), (q_global, v_global), (v_seq, q_seq)
            if self.batch_branches and dense_coattention_supported(mhas, queries, seqs):
                # both branches in the same launches (:257-261 and :264-268), concatenated by the last one (:274)
                return dense_coattention(mhas, queries, seqs)
            side = _side_stream(v_seq.device) if self.overlap_branches else None
            if side is not None:
                # The two branches are independent chains of small launches (a dozen forward, two dozen
                # backward, each bound by its launch and fill latency at these sizes): on two streams one
                # chain fills the gaps of the other, forward and backward (autograd runs a node's backward
                # on the stream of its forward).
                main = torch.cuda.current_stream(v_seq.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    attended_q = single_query_attention(self.vis_to_txt_attn, v_global, q_seq)  # :264-268
                attended_v = single_query_attention(self.txt_to_vis_attn, q_global, v_seq)  # :257-261
                main.wait_stream(side)
                attended_q.record_stream(main)  # allocated on the side stream, consumed on this one
            else:
                attended_v = single_query_attention(self.txt_to_vis_attn, q_global, v_seq)  # :257-261
                attended_q = single_query_attention(self.vis_to_txt_attn, v_global, q_seq)  # :264-268
        return torch.cat([attended_v, attended_q.to(attended_v.dtype)], dim=1)
