"""CPU restatement of the reference's fusion hot path -- TEST INFRASTRUCTURE ONLY.

This file is the checker the HIP kernels are compared against; nothing in vqa_amd/ may import
it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

It restates, with plain PyTorch CPU ops in the order the reference issues them, what
/root/reference/utils/fusion.py computes (file:line given per function).  It is pinned to the
reference itself: oracle/gen_golden.py imports the reference's own modules in the build
container, asserts that these functions reproduce their outputs bit for bit on the same inputs
(with the same pre-drawn random numbers replayed into torch.randperm / torch.normal), and writes
the compact expected outputs checked by tests/test_oracle_golden.py.

Randomness is always explicit here: callers pass the draws (`replace_positions`,
`random_bottom`, `noise`) that the reference would have taken from torch's global generator.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def keep_count(n_tokens: int, prune_ratio: float) -> int:
    """utils/fusion.py:41."""
    return max(1, int(n_tokens * (1 - prune_ratio)))


def adv_count(k: int, adv_ratio: float) -> int:
    """utils/fusion.py:54."""
    return max(1, int(k * adv_ratio))


def attention_weights(query: torch.Tensor, seq: torch.Tensor, in_proj_weight: torch.Tensor,
                      in_proj_bias: torch.Tensor, num_heads: int) -> torch.Tensor:
    """Head-averaged attention weights (B,1,N) of nn.MultiheadAttention(batch_first=True) called
    as attn(query=query[:,None], key=seq, value=seq) with need_weights=True -- the calls at
    utils/fusion.py:234-238, 243-247, 159-163.  Op order follows
    torch.nn.functional.multi_head_attention_forward: packed in-projection (key is value but
    query is not key, so q is projected alone and k,v by ONE (2E)-wide GEMM), q * sqrt(1/dh),
    bmm, softmax, mean over heads."""
    B, N, E = seq.shape
    dh = E // num_heads
    w_q, w_kv = in_proj_weight.split([E, 2 * E])
    b_q, b_kv = in_proj_bias.split([E, 2 * E])
    q = F.linear(query.reshape(B, 1, E).transpose(0, 1), w_q, b_q)        # (1,B,E)
    kv = F.linear(seq.transpose(0, 1), w_kv, b_kv)                         # (N,B,2E)
    k = kv.unflatten(-1, (2, E))[..., 0, :].contiguous()                   # (N,B,E)
    q = q.reshape(1, B * num_heads, dh).transpose(0, 1)                    # (B*H,1,dh)
    k = k.reshape(N, B * num_heads, dh).transpose(0, 1)                    # (B*H,N,dh)
    q_scaled = q * math.sqrt(1.0 / float(dh))
    scores = torch.bmm(q_scaled, k.transpose(-2, -1))                      # (B*H,1,N)
    probs = torch.softmax(scores, dim=-1)
    return probs.view(B, num_heads, 1, N).mean(dim=1)                      # (B,1,N)


def single_query_attention(query: torch.Tensor, seq: torch.Tensor, in_proj_weight: torch.Tensor,
                           in_proj_bias: torch.Tensor, out_proj_weight: torch.Tensor,
                           out_proj_bias: torch.Tensor, num_heads: int) -> torch.Tensor:
    """Attention OUTPUT (B,E) of the same call (non-pruning branch, utils/fusion.py:257-268,
    173-179): softmax(q k^T / sqrt(dh)) v per head, heads concatenated, out_proj."""
    B, N, E = seq.shape
    dh = E // num_heads
    w_q, w_kv = in_proj_weight.split([E, 2 * E])
    b_q, b_kv = in_proj_bias.split([E, 2 * E])
    q = F.linear(query.reshape(B, 1, E).transpose(0, 1), w_q, b_q)
    kv = F.linear(seq.transpose(0, 1), w_kv, b_kv).unflatten(-1, (2, E))
    k = kv[..., 0, :].contiguous()
    v = kv[..., 1, :].contiguous()
    q = q.reshape(1, B * num_heads, dh).transpose(0, 1)
    k = k.reshape(N, B * num_heads, dh).transpose(0, 1)
    v = v.reshape(N, B * num_heads, dh).transpose(0, 1)
    probs = torch.softmax(torch.bmm(q * math.sqrt(1.0 / float(dh)), k.transpose(-2, -1)), dim=-1)
    ctx = torch.bmm(probs, v)                                              # (B*H,1,dh)
    ctx = ctx.transpose(0, 1).reshape(B, E)
    return F.linear(ctx, out_proj_weight, out_proj_bias)


def select_indices(attn_weights: torch.Tensor, n_tokens: int, prune_ratio: float,
                   adversarial_mix: bool, adv_ratio: float,
                   replace_positions: Optional[torch.Tensor] = None,
                   random_bottom: Optional[torch.Tensor] = None
                   ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Keep indices (B,K) after adversarial mixing -- utils/fusion.py:37-65.
    replace_positions[b] plays randperm(k)[:k_adv] (:59); random_bottom[b] plays
    randperm(N-k)[:k_adv] (:62), i.e. positions INTO the bottom list.
    Returns (top_indices, bottom_indices)."""
    if attn_weights.dim() == 3:
        attn_weights = attn_weights.squeeze(1)
    B = attn_weights.shape[0]
    k = keep_count(n_tokens, prune_ratio)
    top = torch.topk(attn_weights, k=k, dim=-1).indices.clone()
    bottom = None
    if adversarial_mix:
        remaining_k = n_tokens - k
        bottom = torch.topk(attn_weights, k=remaining_k, dim=-1, largest=False).indices
        k_adv = adv_count(k, adv_ratio)
        if remaining_k >= k_adv:
            if replace_positions is None or random_bottom is None:
                raise ValueError("the oracle needs the adversarial draws to be passed in")
            for b in range(B):
                top[b, replace_positions[b]] = bottom[b][random_bottom[b]]
    return top, bottom


def stable_order(attn_weights: torch.Tensor) -> torch.Tensor:
    """The library's documented total order (weight desc, index asc) as a full ranking (B,N);
    equals torch.topk's order whenever a row has no exact ties."""
    if attn_weights.dim() == 3:
        attn_weights = attn_weights.squeeze(1)
    return torch.sort(attn_weights, dim=-1, descending=True, stable=True).indices


def gather_mix(seq: torch.Tensor, idx: torch.Tensor, mixup_alpha: float,
               noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """utils/fusion.py:67-92 on given keep indices: gather, optional additive noise on the kept
    tokens (:73-74, 114), then the alpha-scaled mean of the dropped tokens appended as one more
    token (:77-92)."""
    B, N, D = seq.shape
    sel = torch.gather(seq, 1, idx.unsqueeze(-1).expand(-1, -1, D))
    if noise is not None:
        sel = sel + noise
    if mixup_alpha > 0:
        mask = torch.ones(B, N)
        mask.scatter_(1, idx, 0.0)
        rem_sum = (seq * mask.unsqueeze(-1).expand(-1, -1, D)).sum(dim=1, keepdim=True)
        rem_cnt = mask.sum(dim=1, keepdim=True).unsqueeze(-1) + 1e-10
        rem = mixup_alpha * (rem_sum / rem_cnt)
        sel = torch.cat([sel, rem], dim=1)
    return sel


def prune_tokens(seq, attn_weights, prune_ratio=0.5, mixup_alpha=0.05, adversarial_mix=True,
                 adv_ratio=0.2, replace_positions=None, random_bottom=None, noise=None):
    """PrivacyAwareTokenPruning.forward (utils/fusion.py:26-94) with explicit draws.
    Returns (pruned tokens, keep indices)."""
    idx, _ = select_indices(attn_weights, seq.shape[1], prune_ratio, adversarial_mix, adv_ratio,
                            replace_positions, random_bottom)
    return gather_mix(seq, idx, mixup_alpha, noise), idx


def co_attention_pruned(v_seq, v_global, q_seq, q_global, t2v, v2t, num_heads=8, prune_ratio=0.5,
                        mixup_alpha=0.05, adversarial_mix=True, adv_ratio=0.2, v_draws=None,
                        q_draws=None, v_noise=None):
    """CoAttention.forward, pruning branch (utils/fusion.py:231-254, 274).  t2v / v2t are
    (in_proj_weight, in_proj_bias) of txt_to_vis_attn / vis_to_txt_attn; *_draws are
    (replace_positions, random_bottom) or None.  Returns (fused (B,2E), v_idx, q_idx)."""
    v_draws = v_draws or (None, None)
    q_draws = q_draws or (None, None)
    v_w = attention_weights(q_global, v_seq, t2v[0], t2v[1], num_heads)
    v_pruned, v_idx = prune_tokens(v_seq, v_w, prune_ratio, mixup_alpha, adversarial_mix, adv_ratio,
                                   v_draws[0], v_draws[1], v_noise)
    q_w = attention_weights(v_global, q_seq, v2t[0], v2t[1], num_heads)
    q_pruned, q_idx = prune_tokens(q_seq, q_w, prune_ratio, mixup_alpha, adversarial_mix, adv_ratio,
                                   q_draws[0], q_draws[1], None)
    fused = torch.cat([v_pruned.mean(dim=1), q_pruned.mean(dim=1)], dim=1)
    return fused, v_idx, q_idx


def attention_pruned(v_seq, q_global, attn, num_heads=8, prune_ratio=0.5, mixup_alpha=0.05,
                     adversarial_mix=True, adv_ratio=0.2, draws=None, noise=None):
    """Attention.forward, pruning branch (utils/fusion.py:157-170, 182)."""
    draws = draws or (None, None)
    w = attention_weights(q_global, v_seq, attn[0], attn[1], num_heads)
    pruned, idx = prune_tokens(v_seq, w, prune_ratio, mixup_alpha, adversarial_mix, adv_ratio,
                               draws[0], draws[1], noise)
    return torch.cat([pruned.mean(dim=1), q_global], dim=1), idx


def co_attention_dense(v_seq, v_global, q_seq, q_global, t2v, v2t, num_heads=8):
    """CoAttention.forward, non-pruning branch (utils/fusion.py:255-274).  t2v / v2t are
    (in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias)."""
    a_v = single_query_attention(q_global, v_seq, *t2v, num_heads)
    a_q = single_query_attention(v_global, q_seq, *v2t, num_heads)
    return torch.cat([a_v, a_q], dim=1)


def per_sample_ce(logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """nn.CrossEntropyLoss(reduction='none') -- attack/metric_src.py:313-314, 335, 354."""
    return F.cross_entropy(logits, targets, reduction='none')


def max_softmax(logits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """softmax -> max -- attack/metric_src.py:376-377."""
    return torch.max(torch.softmax(logits, dim=1), dim=1)
