"""GPU: vqaf_gemm_bf16 (LDS-DMA ring + MFMA) against an f64 contraction of the same bf16 operands,
for the three operand layouts the classifier uses, benchmark-sized and ragged shapes, row pitches
larger than the row, bias / accumulate / column-sum epilogues and the split contraction."""
import numpy as np
import pytest
import torch

from oracle import inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _operand(seed, rows, cols, pitch, dev, scale=1.0):
    """(device view (rows, cols) with row pitch `pitch`, f64 host copy); the padding holds junk."""
    full = (gi.normal(seed, rows, pitch) * scale).bfloat16()
    return full.to(dev)[:, :cols], full[:, :cols].double()


CASES = [
    # layout, M, N, K                      (a_kmajor, b_kmajor)
    ("fwd", 1024, 2048, 1024), ("fwd", 1024, 1024, 2048), ("fwd", 1024, 430, 1024),
    ("dgrad", 1024, 1024, 2048), ("dgrad", 1024, 2048, 1024),
    ("wgrad", 2048, 1024, 1024), ("wgrad", 1024, 2048, 1024), ("wgrad", 430, 1024, 1024),
    # ragged: partial tiles in M and N, k tails, one k-step, more k-steps than ring slots
    ("fwd", 37, 70, 64), ("fwd", 130, 9, 200), ("fwd", 1, 1, 8), ("fwd", 300, 129, 520),
    ("dgrad", 70, 100, 72), ("dgrad", 33, 64, 448), ("dgrad", 257, 130, 8),
    ("wgrad", 100, 70, 37), ("wgrad", 13, 66, 9), ("wgrad", 200, 136, 300), ("wgrad", 64, 64, 1),
]


@pytest.mark.parametrize("layout,M,N,K", CASES)
def test_gemm_bf16_layouts_and_edges(dev, layout, M, N, K):
    from vqa_amd import ops
    up8 = lambda v: (v + 7) // 8 * 8  # noqa: E731
    if layout == "fwd":      # A (M,K) k-contiguous, B (N,K) k-contiguous
        A, Ah = _operand(1, M, K, up8(K) + 8, dev)
        B, Bh = _operand(2, N, K, up8(K), dev)
        want = Ah @ Bh.t()
        ak, bk = False, False
    elif layout == "dgrad":  # A (M,K) k-contiguous, B (K,N) k-major
        A, Ah = _operand(3, M, K, up8(K), dev)
        B, Bh = _operand(4, K, N, up8(N) + 16, dev)
        want = Ah @ Bh
        ak, bk = False, True
    else:                    # A (K,M) k-major, B (K,N) k-major
        A, Ah = _operand(5, K, M, up8(M), dev, scale=0.1)
        B, Bh = _operand(6, K, N, up8(N), dev)
        want = Ah.t() @ Bh
        ak, bk = True, True
    assert ops.gemm_bf16_supported(A.stride(0), ak, B.stride(0), bk, M, N, K)
    scale = float(want.abs().max()) + 1e-30
    # f32 output, plain store
    C = torch.full((M, N + 3), float("nan"), device=dev)[:, :N]
    colsum = torch.full((M,), float("nan"), device=dev) if ak else None
    ops.gemm_bf16(A, ak, B, bk, M, N, K, C, colsum=colsum)
    np.testing.assert_allclose(C.cpu().numpy(), want.numpy(), rtol=1e-5, atol=3e-6 * scale)
    if ak:
        cs = Ah.sum(0)
        np.testing.assert_allclose(colsum.cpu().numpy(), cs.numpy(), rtol=1e-5, atol=3e-6 * float(cs.abs().max()))
    # accumulate adds a second product
    ops.gemm_bf16(A, ak, B, bk, M, N, K, C, accumulate=True, colsum=colsum)
    np.testing.assert_allclose(C.cpu().numpy(), 2 * want.numpy(), rtol=1e-5, atol=6e-6 * scale)
    # bf16 output with the f32 column sums beside it (weight + bias gradient of an encoder Linear)
    if ak:
        Cb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        colsum.fill_(float("nan"))
        ops.gemm_bf16(A, ak, B, bk, M, N, K, Cb, colsum=colsum)
        np.testing.assert_allclose(Cb.double().cpu().numpy(), want.numpy(), rtol=2 ** -8, atol=2 ** -8 * scale)
        np.testing.assert_allclose(colsum.cpu().numpy(), cs.numpy(), rtol=1e-5, atol=3e-6 * float(cs.abs().max()))
    # bf16 output with a bias (what the forward uses); f32 bias and bf16 bias
    if not ak:
        bias = gi.normal(7, N)
        for bt in (bias.to(dev), bias.bfloat16().to(dev)):
            Cb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm_bf16(A, ak, B, bk, M, N, K, Cb, bias=bt)
            wb = want + bt.double().cpu()[None, :]
            np.testing.assert_allclose(Cb.double().cpu().numpy(), wb.numpy(), rtol=2 ** -8, atol=2 ** -8 * scale)


def test_gemm_bf16_rejects_uncovered_layouts(dev):
    from vqa_amd import ops
    assert not ops.gemm_bf16_supported(430, True, 1024, True, 430, 1024, 1024)   # row pitch not a multiple of 8
    assert not ops.gemm_bf16_supported(430, False, 1024, True, 64, 1024, 430)    # k-contiguous operand, K % 8 != 0
    assert ops.gemm_bf16_supported(432, True, 1024, True, 430, 1024, 1024)       # padded pitch covers the tail


def test_wgrad_group_equals_single_launches(dev):
    """Two and three clipped-sum-shaped products in one launch against the per-product launches and f64;
    products that cannot share a launch are refused without launching anything."""
    from vqa_amd import ops
    K = 1024
    shapes = [(2048, 1024), (1024, 2048), (2048, 1536)]
    items, singles, wants = [], [], []
    for i, (M, N) in enumerate(shapes):
        A, Ah = _operand(20 + i, K, M, M, dev, scale=0.1)
        B, Bh = _operand(30 + i, K, N, N + 8, dev)
        C = torch.full((M, N), float("nan"), device=dev)
        cs = torch.full((M,), float("nan"), device=dev)
        items.append((A, B, C, cs if i != 1 else None))
        C1 = torch.empty(M, N, device=dev)
        cs1 = torch.empty(M, device=dev)
        ops.gemm_bf16(A, True, B, True, M, N, K, C1, colsum=cs1)
        singles.append((C1, cs1))
        wants.append((Ah.t() @ Bh, Ah.sum(0)))
    for n in (2, 3):
        for _, _, C, cs in items:
            C.fill_(float("nan"))
        assert ops.gemm_bf16_wgrad_group(items[:n], K)
        for (A, B, C, cs), (C1, cs1), (want, wcs) in list(zip(items, singles, wants))[:n]:
            assert torch.equal(C, C1)  # same tiles, same order of accumulation
            np.testing.assert_allclose(C.cpu().numpy(), want.numpy(), rtol=1e-5, atol=3e-6 * float(want.abs().max()))
            if cs is not None:
                assert torch.equal(cs, cs1)
    # accumulate adds a second product
    assert ops.gemm_bf16_wgrad_group(items[:2], K, accumulate=True)
    np.testing.assert_allclose(items[0][2].cpu().numpy(), 2 * wants[0][0].numpy(), rtol=1e-5,
                               atol=6e-6 * float(wants[0][0].abs().max()))
    # a narrow product joins a launch that fills the chip without it (the 430-wide head beside the wide layers:
    # row pitch 432, the rows' padding zero) and equals its own launch up to the split contraction of the latter
    Ap = torch.zeros(K, 432, device=dev, dtype=torch.bfloat16)
    Ap[:, :430] = (torch.randn(K, 430, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.1).bfloat16()
    A = Ap[:, :430]
    B, Bh = _operand(42, K, 1024, 1024, dev)
    C = torch.full((430, 1024), float("nan"), device=dev)
    cs = torch.full((430,), float("nan"), device=dev)
    assert ops.gemm_bf16_wgrad_group([items[0], (A, B, C, cs)], K)
    want = A.double().cpu().t() @ Bh.double()
    np.testing.assert_allclose(C.cpu().numpy(), want.numpy(), rtol=1e-5, atol=3e-6 * float(want.abs().max()))
    np.testing.assert_allclose(cs.cpu().numpy(), A.double().cpu().sum(0).numpy(), rtol=1e-5, atol=1e-5)
    # too few tiles altogether (the products want their contraction split): refused, nothing launched
    A2, _ = _operand(41, K, 256, 256, dev)
    B2, _ = _operand(43, K, 128, 128, dev)
    C2 = torch.zeros(256, 128, device=dev)
    assert ops.gemm_bf16_wgrad_group([(A2, B2, C2, None), (A2, B2, C2, None)], K) is False
    assert float(C2.abs().max()) == 0.0


@pytest.mark.gpu
def test_dgrad_with_a_contraction_that_is_not_a_multiple_of_8(dev):
    """The 430-class head's input gradient gx = gy W: gy rows padded with zeros to 432 elements are read in
    whole 16-byte chunks, W's rows past 430 are never read (they come from the block of zeros)."""
    from vqa_amd import ops
    M, N, K = 200, 1024, 430
    gen = torch.Generator(device=dev).manual_seed(9)
    buf = torch.zeros(M, 432, device=dev, dtype=torch.bfloat16)
    buf[:, :K] = torch.randn(M, K, device=dev, generator=gen).bfloat16()
    gy = buf[:, :K]
    # W is the LAST allocation-sized block here: junk or unmapped memory past its 430 rows must not matter
    W = torch.randn(K, N, device=dev, generator=gen).bfloat16()
    assert ops.gemm_bf16_supported(gy.stride(0), False, N, True, M, N, K)
    assert not ops.gemm_bf16_supported(K, False, N, True, M, N, K)          # unpadded rows: refused
    assert not ops.gemm_bf16_supported(432, False, 432, False, M, N, K)     # both k-contiguous: refused
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm_bf16(gy, False, W, True, M, N, K, out)
    want = gy.double() @ W.double()
    np.testing.assert_allclose(out.double().cpu().numpy(), want.cpu().numpy(), rtol=2.0 ** -7, atol=2e-2)
