"""CPU: the oracle (oracle/fusion_oracle.py) against the golden vectors that
oracle/gen_golden.py captured from the reference's own modules."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import fusion_oracle as fo
from oracle import inputs as gi



@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def _names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def test_golden_files_present():
    assert len(_names("scores_")) == 4
    assert len(_names("prune_")) == 24
    assert len(_names("inject_")) == 4
    assert len(_names("coattn_")) == 3


@pytest.mark.parametrize("name", _names("scores_"))
def test_scores(name):
    g = load_golden(name)
    seed, B, N = int(g["seed"]), int(g["B"]), int(g["N"])
    seq, _ = gi.tokens(seed, B, N)
    q = gi.normal(seed + 1, B, 512)
    p = gi.mha_params(seed + 2)
    w = fo.attention_weights(q, seq, p["in_proj_weight"], p["in_proj_bias"], 8).squeeze(1)
    # last-bit differences only: the CPU GEMM path depends on whether weights require grad
    np.testing.assert_allclose(w.numpy(), g["w"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(w.sum(1).numpy(), np.ones(B), rtol=1e-5)


@pytest.mark.parametrize("name", _names("prune_"))
def test_prune(name):
    g = load_golden(name)
    seed, B, N, D = int(g["seed"]), int(g["B"]), int(g["N"]), int(g["D"])
    p_ratio, alpha = float(g["prune_ratio"]), float(g["alpha"])
    seq, _ = gi.tokens(seed, B, N, D)
    w = torch.from_numpy(g["w"])
    out, idx = fo.prune_tokens(seq, w, p_ratio, alpha, False, 0.2)
    k = fo.keep_count(N, p_ratio)
    assert np.array_equal(idx.numpy(), g["idx"])
    order = fo.stable_order(w)
    assert np.array_equal(order[:, :k].numpy(), g["idx"])
    assert np.array_equal(order.flip(-1)[:, :N - k].numpy(), g["bottom"])
    assert out.shape == (B, k + (1 if alpha > 0 else 0), D)
    assert np.array_equal(out.mean(dim=1).numpy(), g["pooled"])
    if alpha > 0:
        assert np.array_equal(out[:, k].numpy(), g["mix_token"])
    # gathered rows are copies
    assert torch.equal(out[:, :k], torch.stack([seq[b, idx[b]] for b in range(B)]))


@pytest.mark.parametrize("name", _names("inject_"))
def test_injected(name):
    g = load_golden(name)
    seed, B, N, D = int(g["seed"]), int(g["B"]), int(g["N"]), int(g["D"])
    p_ratio, alpha, adv = float(g["prune_ratio"]), float(g["alpha"]), float(g["adv_ratio"])
    sigma = float(g["noise_scale"])
    seq, _ = gi.tokens(seed, B, N, D)
    k = fo.keep_count(N, p_ratio)
    z = gi.normal(seed + 3, B, k, D)
    out, idx = fo.prune_tokens(seq, torch.from_numpy(g["w"]), p_ratio, alpha, True, adv,
                               torch.from_numpy(g["pos"]), torch.from_numpy(g["src"]),
                               noise=0.0 + sigma * z)
    assert np.array_equal(idx.numpy(), g["idx"])
    for b in range(B):
        assert len(set(idx[b].tolist())) == k  # the swap keeps the set duplicate-free
    assert np.array_equal(out[:, k].numpy(), g["mix_token"])
    assert np.array_equal(out[:, 0].numpy(), g["noisy_row0"])
    assert np.array_equal(out.mean(dim=1).numpy(), g["pooled"])
    assert bool(g["swap_runs"]) == (N - k >= fo.adv_count(k, adv))


@pytest.mark.parametrize("name", _names("coattn_"))
def test_coattention(name):
    g = load_golden(name)
    seed, B, N = int(g["seed"]), int(g["B"]), int(g["N"])
    v_seq, v_global = gi.tokens(seed, B, N)
    q_seq, q_global = gi.tokens(seed + 1, B, 30)
    t2v, v2t = gi.mha_params(seed + 2), gi.mha_params(seed + 6)
    fused, v_idx, q_idx = fo.co_attention_pruned(
        v_seq, v_global, q_seq, q_global, (t2v["in_proj_weight"], t2v["in_proj_bias"]),
        (v2t["in_proj_weight"], v2t["in_proj_bias"]), adversarial_mix=False)
    assert np.array_equal(v_idx.numpy(), g["v_idx"])
    assert np.array_equal(q_idx.numpy(), g["q_idx"])
    np.testing.assert_allclose(fused.numpy(), g["fused"], rtol=1e-6, atol=1e-7)
    keys = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")
    dense = fo.co_attention_dense(v_seq, v_global, q_seq, q_global, tuple(t2v[k] for k in keys),
                                  tuple(v2t[k] for k in keys))
    np.testing.assert_allclose(dense.numpy(), g["dense"], rtol=1e-4, atol=2e-6)


def test_coattention_backward_projections():
    """Input gradients of the pruning branch, via autograd on the oracle, against the
    projections captured from the reference's backward."""
    g = load_golden("coattn_N197")
    seed, B, N = int(g["seed"]), int(g["B"]), int(g["N"])
    v_seq, v_global = gi.tokens(seed, B, N)
    q_seq, q_global = gi.tokens(seed + 1, B, 30)
    t2v, v2t = gi.mha_params(seed + 2), gi.mha_params(seed + 6)
    cot = gi.normal(seed + 10, B, 1024)
    with torch.enable_grad():
        vs, qs = v_seq.clone().requires_grad_(), q_seq.clone().requires_grad_()
        fused, _, _ = fo.co_attention_pruned(
            vs, v_global, qs, q_global, (t2v["in_proj_weight"], t2v["in_proj_bias"]),
            (v2t["in_proj_weight"], v2t["in_proj_bias"]), adversarial_mix=False)
        (fused * cot).sum().backward()
    np.testing.assert_allclose(vs.grad.sum(2).numpy(), g["gv_rowsum"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(vs.grad.sum(1).numpy(), g["gv_colsum"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(qs.grad.sum(2).numpy(), g["gq_rowsum"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(qs.grad.sum(1).numpy(), g["gq_colsum"], rtol=1e-5, atol=1e-7)


def test_attention_module_golden():
    g = load_golden("attention_N197")
    seed, B, N = int(g["seed"]), int(g["B"]), int(g["N"])
    v_seq, _ = gi.tokens(seed, B, N)
    _, q_global = gi.tokens(seed + 1, B, 30)
    p = gi.mha_params(seed + 2)
    z = gi.normal(seed + 3, B, fo.keep_count(N, 0.5), 512)
    out, idx = fo.attention_pruned(v_seq, q_global, (p["in_proj_weight"], p["in_proj_bias"]),
                                   adversarial_mix=False, noise=0.0 + 0.01 * z)
    assert np.array_equal(idx.numpy(), g["idx"])
    np.testing.assert_allclose(out.numpy(), g["fused"], rtol=1e-6, atol=1e-7)


def test_mia_scores_golden():
    g = load_golden("model_N197")
    logits, y = torch.from_numpy(g["logits"]), torch.from_numpy(g["targets"])
    assert np.array_equal(fo.per_sample_ce(logits, y).numpy(), g["loss"])
    conf, pred = fo.max_softmax(logits)
    assert np.array_equal(conf.numpy(), g["conf"])
    assert np.array_equal(pred.numpy(), g["pred"])


def test_keep_count_matches_reference_arithmetic():
    # utils/fusion.py:41 / :54 in Python double arithmetic
    assert fo.keep_count(577, 0.5) == 288 and fo.keep_count(197, 0.5) == 98
    assert fo.keep_count(30, 0.5) == 15 and fo.keep_count(577, 0.75) == 144
    assert fo.keep_count(577, 0.1) == 519 and fo.keep_count(10, 0.99) == 1
    assert fo.adv_count(288, 0.2) == 57 and fo.adv_count(519, 0.2) == 103 and fo.adv_count(3, 0.2) == 1
