"""GPU: the one-launch pruning branch (scoring -> selection -> pooled prune/mix in one workgroup per
sample, vqaf_fused_prune_pool_multi) against the separate launches it replaces and against the
oracle.  Keep indices must be identical (same scoring arithmetic, same ranking code, same draws);
pooled features agree to f32 summation order (rounded to bf16: at most one bf16 step)."""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import inputs as gi

pytestmark = pytest.mark.gpu
BF16_STEP = 2.0 ** -7


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _module(dev, **kw):
    from vqa_amd import fusion
    m = fusion.CoAttention(use_pruning=True, **kw)
    m.txt_to_vis_attn.load_state_dict(gi.mha_params(3))
    m.vis_to_txt_attn.load_state_dict(gi.mha_params(7))
    return m.to(dev)


def _run(m, fuse, v, vg, q, qg, cot, seed, **inject):
    from vqa_amd import ops
    m.fuse_select = fuse
    rng = ops.PhiloxState(v.device, seed=seed)
    m.v_pruning.rng = m.q_pruning.rng = rng
    vs, qs = v.clone().requires_grad_(), q.clone().requires_grad_()
    out = m(vs, vg, qs, qg, **inject)
    (out.float() * cot).sum().backward()
    torch.cuda.synchronize()
    return (out.detach(), m.v_pruning.last_indices.clone(), m.q_pruning.last_indices.clone(), vs.grad, qs.grad,
            int(rng.state[1].item()))


@pytest.mark.parametrize("B,Nv,Nq", [(5, 197, 30), (3, 577, 30), (2, 49, 77), (70, 197, 12)])
@pytest.mark.parametrize("noise", [False, True])
def test_one_launch_branch_equals_separate_launches(dev, B, Nv, Nq, noise):
    v, vg = gi.tokens(1, B, Nv)
    q, qg = gi.tokens(2, B, Nq)
    v, vg, q, qg = (t.to(dev).bfloat16() for t in (v, vg, q, qg))
    cot = gi.normal(11, B, 1024).to(dev)
    m = _module(dev, noise_injection=noise, noise_scale=0.05).bfloat16()
    # in-kernel draws: the same rng state gives the same adversarial draws and the same noise
    a = _run(m, True, v, vg, q, qg, cot, 21)
    b = _run(m, False, v, vg, q, qg, cot, 21)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    np.testing.assert_allclose(a[0].float().cpu().numpy(), b[0].float().cpu().numpy(), rtol=BF16_STEP, atol=1e-4)
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])  # same indices -> same backward
    assert a[5] == b[5]  # both paths consumed the same number of stream positions
    # injected draws (and injected noise)
    kv, kq = fo.keep_count(Nv, 0.5), fo.keep_count(Nq, 0.5)
    pv, sv = gi.adversarial_draws(31, B, kv, Nv - kv, fo.adv_count(kv, 0.2))
    pq, sq = gi.adversarial_draws(32, B, kq, Nq - kq, fo.adv_count(kq, 0.2))
    inj = dict(v_inject=dict(replace_positions=pv.to(dev), random_bottom=sv.to(dev)),
               q_inject=dict(replace_positions=pq.to(dev), random_bottom=sq.to(dev)))
    if noise:
        inj["v_inject"]["noise"] = gi.normal(33, B, kv, 512, scale=0.05).to(dev)
    a = _run(m, True, v, vg, q, qg, cot, 22, **inj)
    b = _run(m, False, v, vg, q, qg, cot, 22, **inj)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    np.testing.assert_allclose(a[0].float().cpu().numpy(), b[0].float().cpu().numpy(), rtol=BF16_STEP, atol=1e-4)
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


@pytest.mark.parametrize("noise", [False, True])
def test_counter_advanced_by_the_fold_launch_equals_the_separate_counter_launch(dev, noise):
    """The query-fold launch advancing the generator state on the side (the default) and the separate
    counter launches give the same draws, the same results bit for bit and the same final state."""
    from vqa_amd import ops
    B, Nv, Nq = 9, 197, 30
    v, vg = gi.tokens(1, B, Nv)
    q, qg = gi.tokens(2, B, Nq)
    v, vg, q, qg = (t.to(dev).bfloat16() for t in (v, vg, q, qg))
    cot = gi.normal(11, B, 1024).to(dev)
    m = _module(dev, noise_injection=noise, noise_scale=0.05).bfloat16()
    saved = ops.ADVANCE_RNG_IN_FOLD
    try:
        ops.ADVANCE_RNG_IN_FOLD = True
        a = _run(m, True, v, vg, q, qg, cot, 77)
        ops.ADVANCE_RNG_IN_FOLD = False
        b = _run(m, True, v, vg, q, qg, cot, 77)
    finally:
        ops.ADVANCE_RNG_IN_FOLD = saved
    assert all(torch.equal(x, y) for x, y in zip(a[:5], b[:5])) and a[5] == b[5]
    assert a[5] == (2 if noise else 1)


def test_one_launch_branch_vs_oracle(dev):
    """Against the reference's op sequence on the same bf16-rounded tokens (f32 arithmetic): indices
    equal, fused features and input gradients within the bf16 bar."""
    B, Nv, Nq = 6, 197, 30
    v, vg = gi.tokens(1, B, Nv)
    q, qg = gi.tokens(2, B, Nq)
    vb, vgb, qb, qgb = (t.bfloat16() for t in (v, vg, q, qg))
    t2v, v2t = gi.mha_params(3), gi.mha_params(7)
    kv, kq = fo.keep_count(Nv, 0.5), fo.keep_count(Nq, 0.5)
    vd = gi.adversarial_draws(41, B, kv, Nv - kv, fo.adv_count(kv, 0.2))
    qd = gi.adversarial_draws(42, B, kq, Nq - kq, fo.adv_count(kq, 0.2))
    cot = gi.normal(43, B, 1024)
    vr, qr = vb.float().requires_grad_(), qb.float().requires_grad_()
    want, v_idx, q_idx = fo.co_attention_pruned(vr, vgb.float(), qr, qgb.float(),
                                                (t2v["in_proj_weight"], t2v["in_proj_bias"]),
                                                (v2t["in_proj_weight"], v2t["in_proj_bias"]), v_draws=vd, q_draws=qd)
    (want * cot).sum().backward()
    m = _module(dev, noise_injection=False)  # f32 parameters, bf16 tokens
    m.fuse_select = True
    vs, qs = vb.to(dev).requires_grad_(), qb.to(dev).requires_grad_()
    out = m(vs, vgb.to(dev), qs, qgb.to(dev),
            v_inject=dict(replace_positions=vd[0].to(dev), random_bottom=vd[1].to(dev)),
            q_inject=dict(replace_positions=qd[0].to(dev), random_bottom=qd[1].to(dev)))
    (out.float() * cot.to(dev)).sum().backward()
    assert torch.equal(m.v_pruning.last_indices.cpu(), v_idx) and torch.equal(m.q_pruning.last_indices.cpu(), q_idx)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), want.detach().numpy(), rtol=1e-2, atol=1e-3)
    np.testing.assert_allclose(vs.grad.float().cpu().numpy(), vr.grad.numpy(), rtol=1e-2, atol=1e-5)
    np.testing.assert_allclose(qs.grad.float().cpu().numpy(), qr.grad.numpy(), rtol=1e-2, atol=1e-5)


def test_fused_entry_point_rejects_what_it_does_not_cover(dev):
    from vqa_amd import ops
    seq32 = torch.zeros(2, 30, 512, device=dev)
    spec = dict(seq=seq32, num_heads=8, k=15, k_adv=3)
    assert not ops.fused_prune_pool_supported([spec])                       # f32 tokens
    spec = dict(seq=seq32.bfloat16()[:, :, :96].contiguous(), num_heads=8, k=15, k_adv=3)
    assert not ops.fused_prune_pool_supported([spec])                       # width without a kernel
    spec = dict(seq=torch.zeros(1, 2000, 512, device=dev, dtype=torch.bfloat16), num_heads=8, k=1000, k_adv=0)
    assert not ops.fused_prune_pool_supported([spec])                       # per-sample state beyond LDS
    assert ops.fused_prune_pool_supported([dict(seq=seq32.bfloat16(), num_heads=8, k=15, k_adv=3)])
