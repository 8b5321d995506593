"""Generates tests/golden/*.npz from the REFERENCE's own modules and pins the oracle to them.

Run in the build container only (the reference tree is mounted read-only at /root/reference and
never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

For every case it (1) builds seeded inputs (oracle/inputs.py), (2) runs the reference's
utils/fusion.py / model/model.py classes on them -- with torch.randperm / torch.normal patched
to replay pre-drawn numbers where the reference is stochastic --, (3) asserts that the oracle's
restatement reproduces the reference output exactly, and (4) stores the seeds, hyper-parameters
and COMPACT expected outputs (weights, recovered keep indices, mix token, pooled features,
logits, gradient projections).  Fixtures are data only; no reference source text is stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from utils.fusion import Attention as RefAttention  # noqa: E402  (reference)
from utils.fusion import CoAttention as RefCoAttention  # noqa: E402
from utils.fusion import PrivacyAwareTokenPruning as RefPruning  # noqa: E402
from model.model import VQAModel as RefVQAModel  # noqa: E402

from oracle import fusion_oracle as fo  # noqa: E402
from oracle import inputs as gi  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)


class Replay:
    """Patches torch.randperm / torch.normal so the reference consumes recorded draws."""

    def __init__(self, perms=None, normals=None):
        self.perms = list(perms or [])
        self.normals = list(normals or [])

    def __enter__(self):
        self._rp, self._nm = torch.randperm, torch.normal

        def randperm(n, *a, **k):
            p = self.perms.pop(0)
            assert p.numel() == n, (p.numel(), n)
            return p.clone()

        def normal(mean, std, size=None, **k):
            z = self.normals.pop(0)
            assert tuple(z.shape) == tuple(size)
            return mean + std * z

        torch.randperm, torch.normal = randperm, normal
        return self

    def __exit__(self, *exc):
        torch.randperm, torch.normal = self._rp, self._nm
        assert not self.perms and not self.normals, "unused recorded draws"


def full_perms(seed, B, k, remaining_k):
    """Full permutations, in the order the reference's loop draws them (utils/fusion.py:59,62)."""
    rng = np.random.default_rng(seed)
    pk = [torch.from_numpy(rng.permutation(k).astype(np.int64)) for _ in range(B)]
    pr = [torch.from_numpy(rng.permutation(remaining_k).astype(np.int64)) for _ in range(B)]
    seq = []
    for b in range(B):
        seq += [pk[b], pr[b]]
    return pk, pr, seq


def load_mha(mha: nn.MultiheadAttention, params):
    mha.load_state_dict(params)


def recover_indices(out_coded: torch.Tensor, k: int) -> torch.Tensor:
    return out_coded[:, :k, 0].round().long()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


def case_scores():
    """(1) attention weights of the reference's MHA call for N in {30, 49, 197, 577}."""
    for N, B in ((30, 4), (49, 4), (197, 4), (577, 2)):
        seed = 1000 + N
        seq, _ = gi.tokens(seed, B, N)
        q = gi.normal(seed + 1, B, 512)
        p = gi.mha_params(seed + 2)
        ref = RefCoAttention(use_pruning=True)
        load_mha(ref.txt_to_vis_attn, p)
        _, w_ref = ref.txt_to_vis_attn(query=q.unsqueeze(1), key=seq, value=seq)
        # the module's own Parameters are passed: on CPU the GEMM path (and so the last bits)
        # depends on whether the weight tensor requires grad
        m = ref.txt_to_vis_attn
        w_or = fo.attention_weights(q, seq, m.in_proj_weight, m.in_proj_bias, 8)
        assert torch.equal(w_ref, w_or), (w_ref - w_or).abs().max()
        save(f"scores_N{N}", seed=seed, B=B, N=N, w=w_ref.squeeze(1))


def case_topk_and_mix():
    """(2)+(3) keep indices and pruned tokens of PrivacyAwareTokenPruning without randomness.
    Indices are recovered from the reference's OUTPUT by pruning index-coded tokens
    (seq[b,n,:] = n), since the reference does not return them."""
    for N in (30, 197, 577):
        for p_ratio in (0.1, 0.25, 0.5, 0.75):
            for alpha in (0.0, 0.05):
                seed = 2000 + N
                B, D = 3, 512
                seq, _ = gi.tokens(seed, B, N, D)
                w = torch.softmax(gi.normal(seed + 1, B, N, scale=2.0), dim=-1).unsqueeze(1)
                ref = RefPruning(prune_ratio=p_ratio, mixup_alpha=alpha, adversarial_mix=False,
                                 noise_injection=False)
                out_ref = ref(seq, w)
                coded = torch.arange(N, dtype=torch.float32).view(1, N, 1).expand(B, N, 8).contiguous()
                k = fo.keep_count(N, p_ratio)
                idx_ref = recover_indices(ref(coded, w), k)
                out_or, idx_or = fo.prune_tokens(seq, w, p_ratio, alpha, False, 0.2)
                assert torch.equal(idx_ref, idx_or)
                assert torch.equal(out_ref, out_or)
                # bottom list: the reference only uses it with adversarial_mix; pin torch.topk's
                # smallest-first order next to the library's documented order
                bottom = torch.topk(w.squeeze(1), k=N - k, dim=-1, largest=False).indices
                assert torch.equal(fo.stable_order(w)[:, :k], idx_ref)
                assert torch.equal(fo.stable_order(w).flip(-1)[:, :N - k], bottom)
                tag = f"prune_N{N}_p{int(p_ratio * 100)}_a{int(alpha * 100)}"
                arrays = dict(seed=seed, B=B, N=N, D=D, prune_ratio=p_ratio, alpha=alpha, w=w.squeeze(1),
                              idx=idx_ref, bottom=bottom, pooled=out_ref.mean(dim=1))
                if alpha > 0:
                    arrays["mix_token"] = out_ref[:, k, :]
                save(tag, **arrays)


def case_injected():
    """(4) adversarial mixing + gaussian noise with replayed draws."""
    for N, p_ratio, adv_ratio in ((197, 0.5, 0.2), (577, 0.5, 0.2), (577, 0.1, 0.2), (30, 0.5, 0.2)):
        seed = 3000 + N + int(p_ratio * 100)
        B, D, alpha, sigma = 3, 512, 0.05, 0.01
        seq, _ = gi.tokens(seed, B, N, D)
        w = torch.softmax(gi.normal(seed + 1, B, N, scale=2.0), dim=-1).unsqueeze(1)
        k = fo.keep_count(N, p_ratio)
        rk = N - k
        k_adv = fo.adv_count(k, adv_ratio)
        swap_runs = rk >= k_adv
        pk, pr, order = full_perms(seed + 2, B, k, rk)
        z = gi.normal(seed + 3, B, k, D)
        ref = RefPruning(prune_ratio=p_ratio, mixup_alpha=alpha, adversarial_mix=True,
                         adv_ratio=adv_ratio, noise_injection=True, noise_scale=sigma)
        with Replay(perms=order if swap_runs else [], normals=[z]):
            out_ref = ref(seq, w)
        coded = torch.arange(N, dtype=torch.float32).view(1, N, 1).expand(B, N, 8).contiguous()
        ref_nonoise = RefPruning(prune_ratio=p_ratio, mixup_alpha=alpha, adversarial_mix=True,
                                 adv_ratio=adv_ratio, noise_injection=False)
        with Replay(perms=order if swap_runs else []):
            idx_ref = recover_indices(ref_nonoise(coded, w), k)
        pos = torch.stack([p[:k_adv] for p in pk])
        src = torch.stack([p[:k_adv] for p in pr])
        out_or, idx_or = fo.prune_tokens(seq, w, p_ratio, alpha, True, adv_ratio, pos, src,
                                         noise=0.0 + sigma * z)
        assert torch.equal(idx_ref, idx_or)
        assert torch.equal(out_ref, out_or), (out_ref - out_or).abs().max()
        save(f"inject_N{N}_p{int(p_ratio * 100)}", seed=seed, B=B, N=N, D=D, prune_ratio=p_ratio,
             alpha=alpha, adv_ratio=adv_ratio, noise_scale=sigma, swap_runs=swap_runs, w=w.squeeze(1),
             pos=pos, src=src, idx=idx_ref, mix_token=out_ref[:, k, :], pooled=out_ref.mean(dim=1),
             noisy_row0=out_ref[:, 0, :])


def _fusion_inputs(seed, B, N, L=30):
    v_seq, v_global = gi.tokens(seed, B, N)
    q_seq, q_global = gi.tokens(seed + 1, B, L)
    return v_seq, v_global, q_seq, q_global


def case_coattention():
    """(5)+(7) CoAttention.forward, pruning and dense branches, with input gradients."""
    for N, B in ((49, 4), (197, 4), (577, 2)):
        seed = 4000 + N
        v_seq, v_global, q_seq, q_global = _fusion_inputs(seed, B, N)
        t2v, v2t = gi.mha_params(seed + 2), gi.mha_params(seed + 6)
        cot = gi.normal(seed + 10, B, 1024)
        # pruning branch, deterministic configuration
        ref = RefCoAttention(use_pruning=True, adversarial_mix=False, noise_injection=False)
        load_mha(ref.txt_to_vis_attn, t2v)
        load_mha(ref.vis_to_txt_attn, v2t)
        with torch.enable_grad():
            vs, qs = v_seq.clone().requires_grad_(), q_seq.clone().requires_grad_()
            fused_ref = ref(vs, v_global, qs, q_global)
            (fused_ref * cot).sum().backward()
        fused_or, v_idx, q_idx = fo.co_attention_pruned(
            v_seq, v_global, q_seq, q_global,
            (ref.txt_to_vis_attn.in_proj_weight, ref.txt_to_vis_attn.in_proj_bias),
            (ref.vis_to_txt_attn.in_proj_weight, ref.vis_to_txt_attn.in_proj_bias),
            adversarial_mix=False)
        assert torch.equal(fused_ref.detach(), fused_or)
        assert all(p.grad is None for p in ref.parameters())  # scoring is forward-only
        # dense branch
        refd = RefCoAttention(use_pruning=False)
        load_mha(refd.txt_to_vis_attn, t2v)
        load_mha(refd.vis_to_txt_attn, v2t)
        with torch.enable_grad():
            vd, qd = v_seq.clone().requires_grad_(), q_seq.clone().requires_grad_()
            vg, qg = v_global.clone().requires_grad_(), q_global.clone().requires_grad_()
            dense_ref = refd(vd, vg, qd, qg)
            (dense_ref * cot).sum().backward()
        def _p(m):
            return (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias)
        t2v_t, v2t_t = _p(refd.txt_to_vis_attn), _p(refd.vis_to_txt_attn)
        dense_or = fo.co_attention_dense(v_seq, v_global, q_seq, q_global, t2v_t, v2t_t)
        assert torch.allclose(dense_ref.detach(), dense_or, rtol=0, atol=0), (dense_ref - dense_or).abs().max()
        pg = {n: p.grad for n, p in refd.named_parameters()}
        save(f"coattn_N{N}", seed=seed, B=B, N=N, L=30, fused=fused_ref.detach(), v_idx=v_idx, q_idx=q_idx,
             gv_rowsum=vs.grad.sum(dim=2), gv_colsum=vs.grad.sum(dim=1),
             gq_rowsum=qs.grad.sum(dim=2), gq_colsum=qs.grad.sum(dim=1),
             dense=dense_ref.detach(),
             dense_gv_rowsum=vd.grad.sum(dim=2), dense_gv_colsum=vd.grad.sum(dim=1),
             dense_gq_rowsum=qd.grad.sum(dim=2), dense_gq_colsum=qd.grad.sum(dim=1),
             dense_gvglobal=vg.grad, dense_gqglobal=qg.grad,
             dense_g_t2v_in_w_rowsum=pg["txt_to_vis_attn.in_proj_weight"].sum(dim=1),
             dense_g_t2v_in_b=pg["txt_to_vis_attn.in_proj_bias"],
             dense_g_t2v_out_w_rowsum=pg["txt_to_vis_attn.out_proj.weight"].sum(dim=1),
             dense_g_t2v_out_b=pg["txt_to_vis_attn.out_proj.bias"],
             dense_g_v2t_in_w_rowsum=pg["vis_to_txt_attn.in_proj_weight"].sum(dim=1),
             dense_g_v2t_in_b=pg["vis_to_txt_attn.in_proj_bias"])


def case_attention():
    """Attention.forward pruning branch: its pruning module keeps the class-default noise
    (utils/fusion.py:141-146), so the noise draw is replayed."""
    N, B = 197, 3
    seed = 4500
    v_seq, v_global, q_seq, q_global = _fusion_inputs(seed, B, N)
    p = gi.mha_params(seed + 2)
    k = fo.keep_count(N, 0.5)
    z = gi.normal(seed + 3, B, k, 512)
    ref = RefAttention(use_pruning=True, adversarial_mix=False)
    load_mha(ref.attention, p)
    with Replay(normals=[z]):
        out_ref = ref(v_seq, v_global, q_seq, q_global)
    out_or, idx = fo.attention_pruned(v_seq, q_global,
                                      (ref.attention.in_proj_weight, ref.attention.in_proj_bias),
                                      adversarial_mix=False, noise=0.0 + 0.01 * z)
    assert torch.equal(out_ref, out_or)
    save("attention_N197", seed=seed, B=B, N=N, L=30, fused=out_ref, idx=idx)


class _StubEncoder(nn.Module):
    """Encoder stand-in obeying the (seq, global) contract of model/model.py:16-17,66-67."""
    seq = None
    glob = None

    def __init__(self, out_features=512, hidden_dim=512):
        super().__init__()
        self.output_dim = 512

    def forward(self, *args):
        return type(self).seq, type(self).glob


class _StubVision(_StubEncoder):
    pass


class _StubText(_StubEncoder):
    pass


def case_model():
    """(6)+(9) VQAModel.forward with stub encoders (eval mode) -> logits, per-sample CE and
    max-softmax confidence as the MIA loops compute them (attack/metric_src.py:313-314,376-377)."""
    seed, B, N, C = 5000, 4, 197, 430
    v_seq, v_global, q_seq, q_global = _fusion_inputs(seed, B, N)
    _StubVision.seq, _StubVision.glob = v_seq, v_global
    _StubText.seq, _StubText.glob = q_seq, q_global
    torch.manual_seed(42)
    ref = RefVQAModel(vision=_StubVision, text=_StubText, fusion_type="co_attention", hidden_dim=1024,
                      num_classes=C, use_token_pruning=True, adversarial_mix=False,
                      noise_injection=False).eval()
    t2v, v2t = gi.mha_params(seed + 2), gi.mha_params(seed + 6)
    load_mha(ref.fusion_module.txt_to_vis_attn, t2v)
    load_mha(ref.fusion_module.vis_to_txt_attn, v2t)
    cls = gi.classifier_params(seed + 20, 1024, 1024, C)
    sd = ref.classifier.state_dict()
    for kname, val in cls.items():
        sd[kname] = val
    ref.classifier.load_state_dict(sd)
    logits = ref(images=torch.zeros(B, 3, 8, 8), input_ids=torch.zeros(B, 30, dtype=torch.long),
                 attention_mask=torch.ones(B, 30, dtype=torch.long))
    y = gi.labels(seed + 30, B, C)
    loss = nn.CrossEntropyLoss(reduction='none')(logits, y)
    conf, pred = torch.max(torch.softmax(logits, dim=1), dim=1)
    assert torch.equal(loss, fo.per_sample_ce(logits, y))
    assert torch.equal(conf, fo.max_softmax(logits)[0])
    save("model_N197", seed=seed, B=B, N=N, C=C, logits=logits, targets=y, loss=loss, conf=conf, pred=pred,
         state_keys=np.array(sorted(ref.state_dict().keys())))


if __name__ == "__main__":
    case_scores()
    case_topk_and_mix()
    case_injected()
    case_coattention()
    case_attention()
    case_model()
    print("golden vectors written; oracle == reference on every case")
