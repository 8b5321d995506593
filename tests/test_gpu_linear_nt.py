"""vqaf_linear_nt (large-tile bf16 Linear with bias / GELU epilogues, both weight layouts) and vqaf_wgrad_tn
against f64 references.  Tolerance: the kernel accumulates bf16 products in f32 and rounds the result to bf16 once, so
an output may differ from the rounded f64 value by one bf16 ulp (2^-8 relative) plus the f32 summation error."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _close(got, want64):
    want = want64.to(torch.float64)
    tol = 2.0 ** -7 * want.abs() + 1e-3 * want.abs().max() + 1e-6
    err = (got.to(torch.float64) - want).abs()
    bad = err > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outside tolerance; worst {err.max().item():.4g}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (1, 8, 64), (12608 // 8, 768, 768),
                                   (257, 392, 192), (2048, 3072, 768)])
@pytest.mark.parametrize("bias", ["bf16", "f32", None])
def test_forward_with_bias(M, N, K, bias):
    from vqa_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
    b = None
    if bias is not None:
        b = torch.randn(N, generator=g).to(dev)
        b = b.bfloat16() if bias == "bf16" else b
    assert ops.linear_nt_supported(x, w)
    y = ops.linear_nt(x, w, b)
    want = x.double() @ w.double().t()
    if b is not None:
        want = want + b.double()
    assert y.dtype == torch.bfloat16 and y.shape == (M, N)
    _close(y, want)


def test_pitched_operands_and_unsupported_shapes():
    from vqa_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    xb = torch.randn(200, 256, generator=g).bfloat16().to(dev)
    wb = torch.randn(136, 192, generator=g).bfloat16().to(dev)
    x, w = xb[:, :128], wb[:, :128]          # row pitches 256 and 192, K = 128
    assert ops.linear_nt_supported(x, w)
    _close(ops.linear_nt(x, w), x.double() @ w.double().t())
    assert not ops.linear_nt_supported(xb[:, :96], wb[:, :96])      # K not a multiple of 64
    assert not ops.linear_nt_supported(xb[:, :128], wb[:133, :128])  # N not a multiple of 8
    assert not ops.linear_nt_supported(xb[:, 4:132], wb[:, :128])    # operand not 16-byte aligned
    assert not ops.linear_nt_supported(xb.float()[:, :128], w)


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1576, 3072, 768)])
def test_gelu_epilogues(M, N, K):
    from vqa_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11 + M)
    x = torch.randn(M, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    h = ops.linear_nt(x, w, b, False, 1, y)
    # the pre-activation is the plain product, bit for bit; the activation is GELU of the ROUNDED pre-activation
    assert torch.equal(h, ops.linear_nt(x, w, b))
    want = F.gelu(h.double())
    assert (y.double() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() + 1e-6
    # backward epilogue: (gy W2t^T) * gelu'(h), the product rounded to bf16 before the multiplication
    gy = torch.randn(M, K, generator=g).bfloat16().to(dev)
    wt = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
    gh = ops.linear_nt(gy, wt, None, False, 2, h)
    plain = ops.linear_nt(gy, wt).double()
    hd = h.double()
    grad = 0.5 * (1 + torch.erf(hd / 2 ** 0.5)) + hd * torch.exp(-0.5 * hd * hd) / (2 * torch.pi) ** 0.5
    want = plain * grad
    assert (gh.double() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item() + 1e-6


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (1, 8, 64), (1576, 768, 3072), (257, 392, 192)])
def test_input_gradient_reads_the_weight_as_stored(M, N, K):
    """gx (M,N) = gy (M,K) W (K,N): W is the (out,in) matrix of the layer, K = out its slow index."""
    from vqa_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    gy = torch.randn(M, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(K, N, generator=g) * 0.05).bfloat16().to(dev)
    assert ops.linear_nt_supported(gy, w, True)
    gx = ops.linear_nt(gy, w, None, True)
    _close(gx, gy.double() @ w.double())
    # pitched weight and GELU-backward epilogue on the same layout
    wide = torch.zeros(K, N + 16, dtype=torch.bfloat16, device=dev)
    wide[:, :N] = w
    assert torch.equal(ops.linear_nt(gy, wide[:, :N], None, True), gx)
    h = torch.randn(M, N, generator=g).bfloat16().to(dev)
    gh = ops.linear_nt(gy, w, None, True, 2, h)
    hd = h.double()
    grad = 0.5 * (1 + torch.erf(hd / 2 ** 0.5)) + hd * torch.exp(-0.5 * hd * hd) / (2 * torch.pi) ** 0.5
    want = gx.double() * grad
    assert (gh.double() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item() + 1e-6


@pytest.mark.parametrize("K,M,N", [(64, 128, 128), (12608, 768, 768), (1000, 200, 136), (4103, 3072, 768), (7, 8, 8),
                                   (12608, 768, 3072)])
@pytest.mark.parametrize("out", ["bf16", "f32"])
def test_weight_and_bias_gradient(K, M, N, out):
    from vqa_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(K + M + N)
    gy = torch.randn(K, M, generator=g).bfloat16().to(dev)
    x = torch.randn(K, N, generator=g).bfloat16().to(dev)
    assert ops.wgrad_tn_supported(gy, x)
    dt = torch.bfloat16 if out == "bf16" else torch.float32
    gw, gb = ops.wgrad_tn(gy, x, dt, dt)
    assert gw.dtype == dt and gb.dtype == dt and gw.shape == (M, N) and gb.shape == (M,)
    want_w, want_b = gy.double().t() @ x.double(), gy.double().sum(0)
    rel = 2.0 ** -8 if out == "bf16" else 1e-5
    assert (gw.double() - want_w).abs().max().item() <= rel * want_w.abs().max().item() + 1e-4
    assert (gb.double() - want_b).abs().max().item() <= rel * want_b.abs().max().item() + 1e-4
    # fixed summation order: a second call gives the same bits
    gw2, gb2 = ops.wgrad_tn(gy, x, dt, dt)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    gw3, none = ops.wgrad_tn(gy, x, dt, None)
    assert none is None and torch.equal(gw3, gw)


@pytest.mark.parametrize("T,in_f,out_f", [(6304, 768, 3072), (6304, 3072, 768), (4200, 768, 768)])
def test_encoder_linear_function_on_the_large_tile_kernels(T, in_f, out_f):
    """encoders._LinearFn at token counts where all three products take the large-tile kernels (forward and
    input gradient on vqaf_linear_nt, weight + bias gradient on vqaf_wgrad_tn) against torch autograd in f64."""
    from vqa_amd import encoders
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(T + in_f)
    x = torch.randn(2, T // 2, in_f, generator=g).bfloat16().to(dev).requires_grad_()
    w = (torch.randn(out_f, in_f, generator=g) * 0.03).bfloat16().to(dev).requires_grad_()
    b = torch.randn(out_f, generator=g).bfloat16().to(dev).requires_grad_()
    assert encoders._big_tiles(T, out_f) and encoders._big_tiles(T, in_f) and T >= encoders._WGRAD_TN_MIN_TOKENS
    y = encoders._LinearFn.apply(x, w, b)
    cot = torch.randn(2, T // 2, out_f, generator=g).bfloat16().to(dev)
    y.backward(cot)
    x64, w64, b64 = (t.detach().double().requires_grad_() for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64, b64)
    y64.backward(cot.double())
    assert y.shape == y64.shape
    for got, want in ((y, y64), (x.grad, x64.grad), (w.grad, w64.grad), (b.grad, b64.grad)):
        assert got.dtype == torch.bfloat16
        _close(got, want.detach())
