"""GPU: the one-pass form of the one-launch pruning branch (vqaf_fused_prune_pool_onepass_multi: rows read
once, held in registers / LDS, pooled on the matrix cores) against the two-pass kernel: identical keep
indices and random streams, pooled features equal to f32 summation order."""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import inputs as gi

pytestmark = pytest.mark.gpu


def _specs(dev, B, Nv, Nq, noise, seed):
    from vqa_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    v = torch.randn(B, Nv, 512, generator=g).bfloat16().to(dev)
    q = torch.randn(B, Nq, 512, generator=g).bfloat16().to(dev)
    w = (torch.randn(1536, 512, generator=g) / 512 ** 0.5).to(dev)
    bias = (torch.randn(1536, generator=g) * 0.1).to(dev)
    qg = torch.randn(B, 512, generator=g).bfloat16().to(dev)
    out = []
    for seq, N in ((v, Nv), (q, Nq)):
        k = ops.keep_count(N, 0.5)
        d = dict(q=qg, seq=seq, in_proj_weight=w, in_proj_bias=bias, num_heads=8, k=k, k_adv=ops.adv_count(k, 0.2),
                 alpha=0.05)
        if noise and N == Nv:
            d.update(noise_type=noise, noise_scale=0.05)
        out.append(d)
    return out


@pytest.mark.parametrize("B,Nv,Nq", [(5, 197, 30), (3, 208, 77), (2, 49, 12), (70, 197, 30), (4, 193, 16), (3, 208, 33)])
@pytest.mark.parametrize("noise", [None, "gaussian"])
def test_onepass_equals_twopass(B, Nv, Nq, noise):
    from vqa_amd import ops
    dev = torch.device("cuda:0")
    res = []
    for one in (True, False):
        ops.FUSED_ONE_PASS = one
        rng = ops.PhiloxState(dev, seed=5)
        specs = _specs(dev, B, Nv, Nq, noise, 3)
        for sp in specs:
            sp["noise_rng"] = rng
        with torch.no_grad():
            fused, idxs, _ = ops.fused_prune_pool(specs, torch.float32, rng)
        torch.cuda.synchronize()
        res.append((fused.cpu(), [i.cpu() for i in idxs]))
    ops.FUSED_ONE_PASS = False
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    np.testing.assert_allclose(res[0][0].numpy(), res[1][0].numpy(), rtol=2e-5, atol=2e-6)


def test_onepass_limits():
    import ctypes
    from vqa_amd import _lib
    lib = _lib.load()
    ok = lambda ns, D=512, H=8: lib.vqaf_fused_prune_pool_onepass_supported(  # noqa: E731
        len(ns), ctypes.cast((ctypes.c_int * len(ns))(*ns), ctypes.c_void_p), D, H)
    assert ok([197, 30]) == 1 and ok([208]) == 1
    assert ok([209]) == 0 and ok([197], D=256) == 0 and ok([197], H=12) == 0 and ok([577, 30]) == 0


@pytest.mark.parametrize("one_pass", [False, True])
def test_full_size_properties(one_pass):
    """B = 1024, 197 + 30 tokens (the benchmarked launch, grid 1024 x 2) on either kernel: keep indices are
    duplicate-free subsets of the right size, the pooled features are the mean of the kept rows plus the
    mix token (recomputed from the indices with torch), and a 64-sample slice equals the separate launches."""
    from vqa_amd import ops
    dev = torch.device("cuda:0")
    B, Nv, Nq = 1024, 197, 30
    specs = _specs(dev, B, Nv, Nq, None, 9)
    rng = ops.PhiloxState(dev, seed=11)
    ops.FUSED_ONE_PASS = one_pass
    try:
        with torch.no_grad():
            fused, idxs, _ = ops.fused_prune_pool(specs, torch.float32, rng)
    finally:
        ops.FUSED_ONE_PASS = False
    torch.cuda.synchronize()
    for sp, idx, c0 in zip(specs, idxs, (0, 512)):
        seq, N, k = sp["seq"], sp["seq"].shape[1], sp["k"]
        assert idx.shape == (B, k) and int(idx.min()) >= 0 and int(idx.max()) < N
        srt = idx.sort(dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())
        x = seq.float()
        kept = torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, 512)).sum(1)
        mix = 0.05 * (x.sum(1) - kept) / ((N - k) + 1e-10)
        want = (kept + mix) / (k + 1)
        np.testing.assert_allclose(fused[:, c0:c0 + 512].cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-5)
    # a slice through the separate launches (same rng state: the draws are addressed by sample index)
    sl = [dict(sp, seq=sp["seq"][:64].contiguous(), q=sp["q"][:64].contiguous()) for sp in specs]
    tops, _ = ops.score_select_multi(sl, ops.PhiloxState(dev, seed=11))
    for idx, top in zip(idxs, tops):
        assert torch.equal(idx[:64], top)
