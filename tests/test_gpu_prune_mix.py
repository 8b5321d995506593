"""GPU: PrivacyAwareTokenPruning.forward as ONE launch (vqaf_prune_mix) against the two launches it
replaces (vqaf_select, vqaf_gather_mix) and against the oracle (utils/fusion.py:26-94).

Bars: keep indices identical; kept rows bit-identical (raw copies without noise; the same
element-addressed draws with noise); the mix token within f32 summation order of the oracle
(rounded once more for bf16 outputs); gradients identical (same backward kernel, same indices)."""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import inputs as gi

pytestmark = pytest.mark.gpu
F32_RTOL, F32_ATOL = 2e-5, 2e-6
BF16_RTOL, BF16_ATOL = 2.0 ** -7, 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _weights(seed, B, N):
    return torch.softmax(gi.normal(seed, B, N, scale=2.0), dim=-1)


def _two_launch(seq, w, k, alpha, k_adv, pos, src, noise, noise_type, sigma, seed, out_dtype=None):
    from vqa_amd import ops
    rng = ops.PhiloxState(seq.device, seed=seed)
    sel = ops.select_indices(k, weights=w, k_adv=k_adv, pos=pos, src=src, rng=rng)
    out = ops.gather_mix(seq, sel["top"], alpha, noise=noise, noise_type=noise_type, noise_scale=sigma, rng=rng,
                         out_dtype=out_dtype)
    return out, sel["top"], int(rng.state[1].item())


def _one_launch(seq, w, k, alpha, k_adv, pos, src, noise, noise_type, sigma, seed, out_dtype=None):
    from vqa_amd import ops
    rng = ops.PhiloxState(seq.device, seed=seed)
    out, idx = ops.prune_mix(seq, w, k, alpha, k_adv=k_adv, pos=pos, src=src, noise=noise, noise_type=noise_type,
                             noise_scale=sigma, rng=rng, out_dtype=out_dtype)
    return out, idx, int(rng.state[1].item())


SHAPES = [
    # B, N, D, dtype
    (4, 577, 512, torch.bfloat16),   # ViT-L/14-336 patches
    (3, 384, 768, torch.bfloat16),   # three 256-column slices
    (5, 197, 512, torch.bfloat16),   # ranking by counting (N <= 256)
    (7, 30, 512, torch.bfloat16),    # text branch
    (2, 100, 320, torch.bfloat16),   # width not a multiple of the 256-column slice
    (3, 197, 512, torch.float32),
    (2, 577, 264, torch.float32),    # f32 at the ViT-L token count, ragged slice
    (1, 1500, 64, torch.bfloat16),   # 2048-key ranking, quarter slice
    (2, 9, 8, torch.float32),        # one partial slice, K tiny
]


@pytest.mark.parametrize("B,N,D,dtype", SHAPES)
@pytest.mark.parametrize("keep", [0.25, 0.5, 0.9])
def test_one_launch_equals_two_launches(dev, B, N, D, dtype, keep):
    from vqa_amd import ops
    assert ops.prune_mix_supported(N, D, dtype)
    seq = gi.normal(N + 1, B, N, D).to(dtype).to(dev)
    w = _weights(N + 2, B, N).to(dev)
    k = fo.keep_count(N, 1.0 - keep)
    k_adv = fo.adv_count(k, 0.2)
    if N - k < k_adv:
        k_adv = 0
    for alpha in (0.05, 0.0):
        for mode in ("none", "drawn", "injected"):
            ka = 0 if mode == "none" else k_adv
            pos = src = None
            if mode == "injected" and ka:
                pos, src = (t.to(dev) for t in gi.adversarial_draws(N, B, k, N - k, ka))
            for noise_type in (None, "gaussian", "uniform"):
                a = _one_launch(seq, w, k, alpha, ka, pos, src, None, noise_type, 0.05, 77)
                b = _two_launch(seq, w, k, alpha, ka, pos, src, None, noise_type, 0.05, 77)
                assert torch.equal(a[1], b[1]), (mode, noise_type)
                assert a[2] == b[2], "the two paths consume the same number of stream positions"
                assert torch.equal(a[0][:, :k], b[0][:, :k]), (mode, noise_type, alpha)
                if alpha > 0:
                    rt, at = (BF16_RTOL, BF16_ATOL) if dtype == torch.bfloat16 else (F32_RTOL, F32_ATOL)
                    np.testing.assert_allclose(a[0][:, k].float().cpu().numpy(), b[0][:, k].float().cpu().numpy(),
                                               rtol=rt, atol=at)


@pytest.mark.parametrize("B,N,D,dtype", SHAPES)
def test_one_launch_vs_oracle_with_injected_draws_and_noise(dev, B, N, D, dtype):
    from vqa_amd import ops
    seq = gi.normal(N + 5, B, N, D).to(dtype)
    w = _weights(N + 6, B, N)
    p_ratio, alpha, adv = 0.5, 0.05, 0.2
    k = fo.keep_count(N, p_ratio)
    k_adv = fo.adv_count(k, adv)
    pos = src = None
    if N - k >= k_adv and k_adv > 0:
        pos, src = gi.adversarial_draws(N + 7, B, k, N - k, k_adv)
    else:
        k_adv = 0
    z = 0.03 * gi.normal(N + 8, B, k, D)
    want, want_idx = fo.prune_tokens(seq.float(), w, p_ratio, alpha, k_adv > 0, adv, pos, src, z)
    out, idx = ops.prune_mix(seq.to(dev), w.to(dev), k, alpha, k_adv=k_adv,
                             pos=pos.to(dev) if k_adv else None, src=src.to(dev) if k_adv else None,
                             noise=z.to(dev), out_dtype=torch.float32)
    assert torch.equal(idx.cpu(), want_idx)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=F32_RTOL, atol=F32_ATOL)
    if dtype == torch.bfloat16:
        out_b, _ = ops.prune_mix(seq.to(dev), w.to(dev), k, alpha, k_adv=k_adv,
                                 pos=pos.to(dev) if k_adv else None, src=src.to(dev) if k_adv else None,
                                 noise=z.to(dev))
        assert out_b.dtype == torch.bfloat16
        np.testing.assert_allclose(out_b.float().cpu().numpy(), want.numpy(), rtol=BF16_RTOL, atol=BF16_ATOL)


def test_duplicate_keep_rows_and_nan_payload(dev):
    """Injected draws may name one dropped token twice (utils/fusion.py:62 draws without
    replacement, a caller-supplied list need not): both keep slots receive the row, the row is not
    part of the mix token.  NaN payloads are copied, not arithmetic'd."""
    from vqa_amd import ops
    B, N, D = 2, 40, 128
    seq = gi.normal(1, B, N, D)
    seq[0, 3, 5] = float("nan")
    w = _weights(2, B, N)
    w[0, 3] = 10.0  # token 3 of sample 0 is certainly kept
    k, k_adv = 20, 4
    pos = torch.tensor([[1, 5, 7, 9], [1, 2, 3, 4]])
    src = torch.tensor([[2, 2, 6, 2], [0, 1, 1, 19]])
    top, _ = fo.select_indices(w, N, 0.5, True, 0.2, pos, src)
    out, idx = ops.prune_mix(seq.to(dev), w.to(dev), k, 0.05, k_adv=k_adv, pos=pos.to(dev), src=src.to(dev))
    assert torch.equal(idx.cpu(), top)
    got = out.cpu()
    rows = seq[torch.arange(B)[:, None], top]
    assert torch.equal(got[:, :k].view(torch.int32), rows.view(torch.int32))
    want = fo.gather_mix(torch.nan_to_num(seq), top, 0.05)
    np.testing.assert_allclose(got[:, k].numpy(), want[:, k].numpy(), rtol=F32_RTOL, atol=F32_ATOL)
    two = ops.gather_mix(seq.to(dev), idx, 0.05).cpu()
    np.testing.assert_allclose(got[:, k].numpy(), two[:, k].numpy(), rtol=F32_RTOL, atol=F32_ATOL)


def test_ties_follow_the_documented_order(dev):
    from vqa_amd import ops
    B, N, D = 3, 64, 128
    seq = gi.normal(4, B, N, D).to(dev)
    w = torch.zeros(B, N)
    w[1, ::3] = 1.0
    w[2] = torch.arange(N).float() % 4
    k = 17
    _, idx = ops.prune_mix(seq, w.to(dev), k, 0.05)
    assert torch.equal(idx.cpu(), fo.stable_order(w)[:, :k])


def test_backward_equals_two_launch_backward(dev):
    from vqa_amd import ops
    for dtype in (torch.float32, torch.bfloat16):
        B, N, D = 4, 197, 512
        seq = gi.normal(9, B, N, D).to(dtype).to(dev)
        w = _weights(10, B, N).to(dev)
        k = 99
        cot = gi.normal(12, B, k + 1, D).to(dev)
        s1 = seq.clone().requires_grad_()
        out, idx = ops.prune_mix(s1, w, k, 0.05, k_adv=19, rng=ops.PhiloxState(dev, seed=3))
        assert not idx.requires_grad
        (out.float() * cot).sum().backward()
        s2 = seq.clone().requires_grad_()
        sel = ops.select_indices(k, weights=w, k_adv=19, rng=ops.PhiloxState(dev, seed=3))
        (ops.gather_mix(s2, sel["top"], 0.05).float() * cot).sum().backward()
        assert torch.equal(idx, sel["top"])
        assert torch.equal(s1.grad, s2.grad)


def test_module_routes_through_the_one_launch_op(dev):
    from vqa_amd import fusion, ops
    torch.manual_seed(0)
    mod = fusion.PrivacyAwareTokenPruning(0.5, 0.05, True, 0.2, True, 0.01, "gaussian")
    for N, dtype, supported in ((577, torch.bfloat16, True), (577, torch.float32, True), (30, torch.float32, True)):
        assert ops.prune_mix_supported(N, 512, dtype) == supported
        seq = gi.normal(N, 2, N, 512).to(dtype).to(dev)
        w = _weights(N + 1, 2, N).to(dev)
        mod.rng = ops.PhiloxState(dev, seed=5)
        out = mod(seq, w.unsqueeze(1))
        k = fo.keep_count(N, 0.5)
        assert out.shape == (2, k + 1, 512) and out.dtype == dtype
        # same stream positions whichever path ran: one for the draws, one for the noise
        assert int(mod.rng.state[1].item()) == 2
        two = _two_launch(seq, w, k, 0.05, fo.adv_count(k, 0.2), None, None, None, "gaussian", 0.01, 5)
        assert torch.equal(mod.last_indices, two[1])
        assert torch.equal(out[:, :k], two[0][:, :k])


def test_rejects_bad_arguments(dev):
    from vqa_amd import _lib, ops
    seq = gi.normal(1, 2, 40, 128).to(dev)
    w = _weights(2, 2, 40).to(dev)
    with pytest.raises(RuntimeError):
        ops.prune_mix(seq, w, 41, 0.05)                      # K > N
    with pytest.raises(RuntimeError):
        ops.prune_mix(seq, w, 30, 0.05, k_adv=11)            # fewer dropped tokens than swaps
    with pytest.raises(RuntimeError):
        ops.prune_mix(gi.normal(1, 2, 40, 132).to(dev), w, 20, 0.05)  # width not a multiple of 8
    assert not ops.prune_mix_supported(4097, 512, torch.bfloat16)
    assert not ops.prune_mix_supported(4096, 1 << 17, torch.float32)  # a sample of 2 GiB
    assert _lib.load().vqaf_prune_mix_supported(40, 128, 99) == 0


@pytest.mark.parametrize("keep", [0.25, 0.5, 0.75, 0.9])
def test_full_size_properties(dev, keep):
    """B=128, N=577, D=512 bf16 (the module-boundary benchmark shape): every kept row is the source
    row its index names, indices are a duplicate-free subset, and keep rows + count * mix / alpha
    reproduce the column sums of the whole sequence (nothing lost, nothing counted twice)."""
    from vqa_amd import ops
    B, N, D = 128, 577, 512
    g = torch.Generator(device="cpu").manual_seed(3)
    seq = torch.randn(B, N, D, generator=g).bfloat16().to(dev)
    w = torch.softmax(2 * torch.randn(B, N, generator=g), -1).to(dev)
    k = fo.keep_count(N, 1.0 - keep)
    k_adv = fo.adv_count(k, 0.2)
    if N - k < k_adv:
        k_adv = 0
    alpha = 0.05
    out, idx = ops.prune_mix(seq, w, k, alpha, k_adv=k_adv, rng=ops.PhiloxState(dev, seed=1))
    assert idx.min() >= 0 and idx.max() < N
    srt = idx.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    rows = torch.gather(seq, 1, idx.unsqueeze(-1).expand(-1, -1, D))
    assert torch.equal(out[:, :k], rows)
    total = seq.float().sum(1)
    recon = rows.float().sum(1) + out[:, k].float() * ((N - k) / alpha)
    # the mix token is rounded to bf16 once: relative 2^-9 of a sum of (N-k) unit normals
    tol = 2.0 ** -8 * (N - k) ** 0.5 * 4 + 1e-2
    assert float((total - recon).abs().max()) < tol


@pytest.mark.parametrize("N,keep,one_launch", [(1600, 0.95, True), (2048, 0.9, False), (4096, 0.5, False)])
def test_long_sequences_take_whichever_path_holds_their_state(dev, N, keep, one_launch):
    """vqaf_prune_mix_supported answers for the worst K / k_adv of a length, so the module never meets a
    shape refusal: long rows go through the one launch while their selection state fits LDS, through
    select + gather_mix beyond, with the same results either way."""
    from vqa_amd import fusion, ops
    assert ops.prune_mix_supported(N, 64, torch.bfloat16) == one_launch
    mod = fusion.PrivacyAwareTokenPruning(1.0 - keep, 0.05, True, 0.2, False, 0.0, "gaussian")
    seq = gi.normal(N, 2, N, 64).bfloat16().to(dev)
    w = _weights(N + 1, 2, N).to(dev)
    mod.rng = ops.PhiloxState(dev, seed=5)
    out = mod(seq, w.unsqueeze(1))
    k = fo.keep_count(N, 1.0 - keep)
    k_adv = fo.adv_count(k, 0.2) if N - k >= fo.adv_count(k, 0.2) else 0
    assert out.shape == (2, k + 1, 64)
    two = _two_launch(seq, w, k, 0.05, k_adv, None, None, None, None, 0.0, 5)
    assert torch.equal(mod.last_indices, two[1])
    assert torch.equal(out[:, :k], two[0][:, :k])
