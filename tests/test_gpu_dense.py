"""GPU parity of the dense (non-pruning) cross-attention branch against the oracle and the
golden vectors captured from the reference's CoAttention(use_pruning=False), forward and backward
(inputs and every MHA parameter)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import fusion_oracle as fo
from oracle import inputs as gi

pytestmark = pytest.mark.gpu
KEYS = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _close(got, want, rtol=1e-4, atol_rel=1e-5):
    want = np.asarray(want)
    np.testing.assert_allclose(np.asarray(got), want, rtol=rtol, atol=atol_rel * max(1e-12, np.abs(want).max()))


@pytest.mark.parametrize("batched", [True, False])
@pytest.mark.parametrize("name", ["coattn_N49", "coattn_N197", "coattn_N577"])
def test_coattention_dense_vs_golden(dev, name, batched):
    """batched: both branches in the same launches (the default); else one chain of launches per branch."""
    from vqa_amd import fusion
    g = load_golden(name)
    seed, B, N = int(g["seed"]), int(g["B"]), int(g["N"])
    v_seq, v_global = gi.tokens(seed, B, N)
    q_seq, q_global = gi.tokens(seed + 1, B, 30)
    cot = gi.normal(seed + 10, B, 1024)
    m = fusion.CoAttention(use_pruning=False)
    m.txt_to_vis_attn.load_state_dict(gi.mha_params(seed + 2))
    m.vis_to_txt_attn.load_state_dict(gi.mha_params(seed + 6))
    m = m.to(dev)
    m.batch_branches = batched
    vs, qs = v_seq.to(dev).requires_grad_(), q_seq.to(dev).requires_grad_()
    vg, qg = v_global.to(dev).requires_grad_(), q_global.to(dev).requires_grad_()
    out = m(vs, vg, qs, qg)
    _close(out.detach().cpu(), g["dense"])
    (out * cot.to(dev)).sum().backward()
    _close(vs.grad.sum(2).cpu(), g["dense_gv_rowsum"])
    _close(vs.grad.sum(1).cpu(), g["dense_gv_colsum"])
    _close(qs.grad.sum(2).cpu(), g["dense_gq_rowsum"])
    _close(qs.grad.sum(1).cpu(), g["dense_gq_colsum"])
    _close(vg.grad.cpu(), g["dense_gvglobal"])
    _close(qg.grad.cpu(), g["dense_gqglobal"])
    pg = {n: p.grad.cpu() for n, p in m.named_parameters()}
    _close(pg["txt_to_vis_attn.in_proj_weight"].sum(1), g["dense_g_t2v_in_w_rowsum"])
    _close(pg["txt_to_vis_attn.in_proj_bias"], g["dense_g_t2v_in_b"])
    _close(pg["txt_to_vis_attn.out_proj.weight"].sum(1), g["dense_g_t2v_out_w_rowsum"])
    _close(pg["txt_to_vis_attn.out_proj.bias"], g["dense_g_t2v_out_b"])
    _close(pg["vis_to_txt_attn.in_proj_weight"].sum(1), g["dense_g_v2t_in_w_rowsum"])
    _close(pg["vis_to_txt_attn.in_proj_bias"], g["dense_g_v2t_in_b"])


@pytest.mark.parametrize("E,H,N,B,dtype", [(512, 8, 197, 3, torch.float32), (256, 4, 50, 5, torch.float32),
                                           (64, 1, 9, 2, torch.float32), (512, 8, 300, 4, torch.bfloat16),
                                           (128, 2, 33, 70, torch.bfloat16)])
@pytest.mark.parametrize("shared", [True, False])
def test_attention_dense_full_gradients_vs_oracle(dev, E, H, N, B, dtype, shared):
    """Attention(use_pruning=False): every gradient element against autograd on the oracle, on the
    launch-sharing form (the default) and on the per-product chain."""
    from vqa_amd import fusion
    v_seq, v_global = gi.tokens(40 + N, B, N, E)
    q_seq, q_global = gi.tokens(41 + N, B, 12, E)
    p = gi.mha_params(42 + N, E)
    cot = gi.normal(43 + N, B, 2 * E)
    if dtype == torch.bfloat16:
        v_seq, q_global = v_seq.bfloat16().float(), q_global.bfloat16().float()
    ref_p = {k: v.clone().requires_grad_() for k, v in p.items()}
    vr, qr = v_seq.clone().requires_grad_(), q_global.clone().requires_grad_()
    want = torch.cat([fo.single_query_attention(qr, vr, *(ref_p[k] for k in KEYS), H), qr], dim=1)
    (want * cot).sum().backward()

    m = fusion.Attention(embed_dim=E, num_heads=H, use_pruning=False)
    m.attention.load_state_dict(p)
    m = m.to(dev)
    m.batch_launches = shared
    vs = v_seq.to(dev).to(dtype).requires_grad_()
    qg = q_global.to(dev).to(dtype).requires_grad_()
    out = m(vs, v_global.to(dev).to(dtype), q_seq.to(dev).to(dtype), qg)
    (out.float() * cot.to(dev)).sum().backward()
    if dtype == torch.float32:
        rt, at = 1e-4, 2e-5
    else:
        rt, at = 1e-2, 1e-2
    _close(out.detach().float().cpu(), want.detach(), rt, at)
    _close(vs.grad.float().cpu(), vr.grad, rt, at)
    _close(qg.grad.float().cpu(), qr.grad, rt, at)
    for k in KEYS:
        _close(getattr_path(m.attention, k).grad.cpu(), ref_p[k].grad, rt, at)


def getattr_path(mod, path):
    for part in path.split("."):
        mod = getattr(mod, part)
    return mod


@pytest.mark.parametrize("B,Nv,Nq,dtype,need_globals", [(5, 197, 30, torch.bfloat16, True), (64, 197, 30, torch.bfloat16, False),
                                                        (3, 577, 12, torch.bfloat16, True), (4, 50, 77, torch.float32, True),
                                                        (2, 900, 30, torch.bfloat16, False)])
def test_batched_branches_equal_the_per_branch_launches(dev, B, Nv, Nq, dtype, need_globals):
    """CoAttention without pruning: both branches in the same launches against one chain of launches per
    branch.  The token kernels share their arithmetic and summation order, so the token gradients are
    bit-identical given the same coefficients; the (B,E)-sized products are split over workgroups with f32
    atomics in both forms, so everything downstream of them agrees to f32 rounding."""
    from vqa_amd import fusion
    torch.manual_seed(5)
    m = fusion.CoAttention(use_pruning=False).to(dev).to(dtype)
    gen = torch.Generator(device=dev).manual_seed(9)
    v = torch.randn(B, Nv, 512, device=dev, generator=gen).to(dtype)
    q = torch.randn(B, Nq, 512, device=dev, generator=gen).to(dtype)
    vg = torch.randn(B, 512, device=dev, generator=gen).to(dtype)
    qg = torch.randn(B, 512, device=dev, generator=gen).to(dtype)
    cot = torch.randn(B, 1024, device=dev, generator=gen)
    res = []
    for batched in (True, False):
        m.batch_branches = batched
        m.zero_grad(set_to_none=True)
        ins = [t.clone().requires_grad_(need_globals or i % 2 == 0) for i, t in enumerate((v, vg, q, qg))]
        out = m(*ins)
        assert out.dtype == dtype and out.shape == (B, 1024)
        (out.float() * cot).sum().backward()
        res.append((out.detach().float(), [None if t.grad is None else t.grad.float() for t in ins],
                    {n: p.grad.float() for n, p in m.named_parameters()}))
    (oa, ia, pa), (ob, ib, pb) = res
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= tol * scale, what

    close(oa, ob, "output")
    for i, (a, b) in enumerate(zip(ia, ib)):
        assert (a is None) == (b is None)
        if a is not None:
            close(a, b, f"input gradient {i}")
    assert pa.keys() == pb.keys()
    for n in pa:
        close(pa[n], pb[n], n)
    for n in ("txt_to_vis_attn.in_proj_bias", "vis_to_txt_attn.in_proj_bias"):  # key bias: exactly zero
        assert float(pa[n][512:1024].abs().max()) == 0.0


def test_batched_backward_with_frozen_parameters_and_a_second_backward(dev):
    """The launch-sharing backward when only some gradients are asked for -- parameters frozen (query and token
    gradients only), tokens frozen (parameter gradients only) -- and when it runs twice through one graph
    (retain_graph): the block it adds into is cleared by the forward for the first backward only, so the second
    must clear its own.  Every variant equals the corresponding gradients of the full run."""
    from vqa_amd.fusion import CoAttention
    torch.manual_seed(4)
    dtype, B, Nv, Nq = torch.bfloat16, 7, 120, 30
    m = CoAttention(embed_dim=512, num_heads=8, use_pruning=False).to(dev).to(dtype)
    gen = torch.Generator(device=dev).manual_seed(8)
    base = [torch.randn(*sh, device=dev, generator=gen).to(dtype) for sh in ((B, Nv, 512), (B, 512), (B, Nq, 512), (B, 512))]
    cot = torch.randn(B, 1024, device=dev, generator=gen)

    def run(req_inputs, req_params, twice=False):
        for p_ in m.parameters():
            p_.requires_grad_(req_params)
        m.zero_grad(set_to_none=True)
        ins = [t.clone().requires_grad_(req_inputs) for t in base]
        loss = (m(*ins).float() * cot).sum()
        if twice:
            loss.backward(retain_graph=True)
            first = ([t.grad.clone() for t in ins], {n: p_.grad.clone() for n, p_ in m.named_parameters()})
            for t in ins:
                t.grad = None
            m.zero_grad(set_to_none=True)
            loss.backward()
            second = ([t.grad for t in ins], {n: p_.grad for n, p_ in m.named_parameters()})
            # (not bit for bit: d_attn and d_qp are sums over split contractions, whose atomic order is not fixed)
            for a, b in zip(first[0], second[0]):
                scale = float(a.float().abs().max()) + 1e-30
                assert float((a.float() - b.float()).abs().max()) <= 1e-2 * scale
            for n in first[1]:
                scale = float(first[1][n].float().abs().max()) + 1e-30
                assert float((first[1][n].float() - second[1][n].float()).abs().max()) <= 1e-2 * scale, n
        else:
            loss.backward()
        return ([None if t.grad is None else t.grad.float() for t in ins],
                {n: None if p_.grad is None else p_.grad.float() for n, p_ in m.named_parameters()})

    full_i, full_p = run(True, True)
    twice_i, twice_p = run(True, True, twice=True)
    frozen_i, frozen_p = run(True, False)
    only_p_i, only_p = run(False, True)
    for p_ in m.parameters():
        p_.requires_grad_(True)

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 1e-2 * scale, what

    for i in range(4):
        close(twice_i[i], full_i[i], f"second backward, input {i}")
        close(frozen_i[i], full_i[i], f"frozen parameters, input {i}")
        assert only_p_i[i] is None
    for n in full_p:
        close(twice_p[n], full_p[n], f"second backward, {n}")
        close(only_p[n], full_p[n], f"frozen inputs, {n}")
        assert frozen_p[n] is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("H", [8, 4, 3])
def test_fused_token_kernels_are_bit_identical_to_the_separate_launches(dev, dtype, H):
    """vqaf_softmax_wsum_multi / vqaf_token_bwd_multi / vqaf_gemm_f32_atomic_sets / vqaf_gemm_f32_group through
    the C ABI against the launches they replace, two problems of different token counts per launch."""
    import ctypes
    from vqa_amd import _lib
    from vqa_amd._lib import check, dtype_code, load, ptr, stream_ptr
    lib, st = load(), stream_ptr(dev)
    B, D, Ns = 6, 256, (77, 301)
    gen = torch.Generator(device=dev).manual_seed(17)
    f32 = dict(dtype=torch.float32, device=dev)
    seqs = [torch.randn(B, n, D, device=dev, generator=gen).to(dtype) for n in Ns]
    ss = [torch.randn(B, n, H, device=dev, generator=gen) * 3 for n in Ns]
    us = [torch.randn(B, H, D, device=dev, generator=gen) for _ in Ns]
    ts = [torch.randn(B, H, D, device=dev, generator=gen) for _ in Ns]
    dps = [torch.randn(B, n, H, device=dev, generator=gen) for n in Ns]
    # forward
    p1 = [torch.empty(B, H, n, **f32) for n in Ns]
    z1 = [torch.empty(B, H, D, **f32) for _ in Ns]
    fm = (_lib.DenseFwdMod * 2)(*[_lib.DenseFwdMod(ptr(seqs[i]), ptr(ss[i]), ptr(p1[i]), ptr(z1[i]), Ns[i]) for i in range(2)])
    side = torch.ones(5000, **f32)  # a range cleared on the side: exactly the elements asked for
    check(lib.vqaf_softmax_wsum_multi(ctypes.cast(fm, ctypes.c_void_p), 2, B, D, H, dtype_code(seqs[0]),
                                      ptr(side[4:]), 4001, st), "fwd")
    assert float(side[:4].sum()) == 4.0 and float(side[4:4005].abs().sum()) == 0.0 and float(side[4005:].sum()) == 995.0
    for i, n in enumerate(Ns):
        p0, z0 = torch.empty(B, H, n, **f32), torch.empty(B, H, D, **f32)
        check(lib.vqaf_softmax_heads(ptr(ss[i]), B, n, H, ptr(p0), st), "softmax")
        check(lib.vqaf_weighted_token_sum(ptr(seqs[i]), ptr(p0), B, n, D, H, dtype_code(seqs[i]), ptr(z0), st), "wsum")
        assert torch.equal(p0, p1[i]) and torch.equal(z0, z1[i])
    # backward, every combination of requested outputs
    for need_u, need_seq in ((True, True), (False, True), (True, False)):
        du1 = [torch.empty(B, H, D, **f32) if need_u else None for _ in Ns]
        gs1 = [torch.empty_like(s) if need_seq else None for s in seqs]
        bm = (_lib.DenseBwdMod * 2)(*[_lib.DenseBwdMod(ptr(seqs[i]), ptr(p1[i]), ptr(dps[i]), ptr(us[i]), ptr(ts[i]),
                                                       ptr(du1[i]), ptr(gs1[i]), Ns[i]) for i in range(2)])
        check(lib.vqaf_token_bwd_multi(ctypes.cast(bm, ctypes.c_void_p), 2, B, D, H, dtype_code(seqs[0]), st), "bwd")
        for i, n in enumerate(Ns):
            ds0 = torch.empty(B, H, n, **f32)
            check(lib.vqaf_softmax_heads_bwd(ptr(p1[i]), ptr(dps[i]), B, n, H, ptr(ds0), st), "softmax_bwd")
            if need_u:
                du0 = torch.empty(B, H, D, **f32)
                check(lib.vqaf_weighted_token_sum(ptr(seqs[i]), ptr(ds0), B, n, D, H, dtype_code(seqs[i]), ptr(du0), st), "du")
                assert torch.equal(du0, du1[i])
            if need_seq:
                g0 = torch.empty_like(seqs[i])
                check(lib.vqaf_token_outer(ptr(ds0), ptr(us[i]), ptr(p1[i]), ptr(ts[i]), B, n, D, H, dtype_code(g0),
                                           ptr(g0), st), "outer")
                assert torch.equal(g0, gs1[i])
    # mixed requests across problems are refused, as is a launch with nothing to write
    bm = (_lib.DenseBwdMod * 2)(*[_lib.DenseBwdMod(ptr(seqs[i]), ptr(p1[i]), ptr(dps[i]), ptr(us[i]), ptr(ts[i]),
                                                   ptr(z1[i]) if i == 0 else None, None, Ns[i]) for i in range(2)])
    assert lib.vqaf_token_bwd_multi(ctypes.cast(bm, ctypes.c_void_p), 2, B, D, H, dtype_code(seqs[0]), st) != 0
    # two equal-shaped products in one launch against one launch each
    M, N, K = 40, 96, 512
    As = [torch.randn(M, K, device=dev, generator=gen) for _ in range(2)]
    Bs = [torch.randn(K, N, device=dev, generator=gen) for _ in range(2)]
    C = torch.zeros(2, M, N, **f32)
    vp = ctypes.c_void_p * 2
    check(lib.vqaf_gemm_f32_atomic_sets(2, vp(*[ptr(x) for x in As]), _lib.F32, 0, K, 1, vp(*[ptr(x) for x in Bs]), _lib.F32,
                                        0, N, 1, None, _lib.F32, 0, vp(ptr(C[0]), ptr(C[1])), 0, N, 1, M, N, K, 0.5, 1, st),
          "gemm sets")
    for i in range(2):
        want = 0.5 * As[i].double() @ Bs[i].double()
        assert float((C[i].double() - want).abs().max()) < 1e-3
    # different products in one launch: mixed shapes, operand types, strides, a bias, a batched problem and a
    # column sum written as a product with a row of ones (all strides zero)
    a1 = torch.randn(70, 300, device=dev, generator=gen).to(dtype)          # (M,K) k-contiguous
    b1 = torch.randn(300, 130, device=dev, generator=gen)                   # (K,N) n-contiguous
    bias1 = torch.randn(130, device=dev, generator=gen)
    a2 = torch.randn(256, 48, device=dev, generator=gen)                    # read as A^T: (K,M) storage
    b2 = torch.randn(256, 96, device=dev, generator=gen).to(dtype)
    a3 = torch.randn(4, 33, 64, device=dev, generator=gen)                  # batched, G = 4
    b3 = torch.randn(4, 64, 40, device=dev, generator=gen)
    ones = torch.ones(4, **f32)
    x4 = torch.randn(700, 200, device=dev, generator=gen).to(dtype)
    c1, c2, c3, c4 = (torch.zeros(*sh, **f32) for sh in ((70, 130), (48, 96), (4, 33, 40), (200,)))
    c2.fill_(7.0)  # problems 2 and 4 are STORED: what their outputs held before does not matter
    c4.fill_(-3.0)
    descs = (_lib.GemmDesc * 4)(
        _lib.GemmDesc(ptr(a1), ptr(b1), ptr(bias1), ptr(c1), 0, 300, 1, 0, 130, 1, 0, 0, 130,
                      dtype_code(a1), _lib.F32, _lib.F32, 1, 70, 130, 300, 0.5, 0, _lib.F32),
        _lib.GemmDesc(ptr(a2), ptr(b2), None, ptr(c2), 0, 1, 48, 0, 96, 1, 0, 0, 96,
                      _lib.F32, dtype_code(b2), _lib.F32, 1, 48, 96, 256, 1.0, 1, _lib.F32),
        _lib.GemmDesc(ptr(a3), ptr(b3), None, ptr(c3), 33 * 64, 64, 1, 64 * 40, 40, 1, 0, 33 * 40, 40,
                      _lib.F32, _lib.F32, _lib.F32, 4, 33, 40, 64, 2.0, 0, _lib.F32),
        _lib.GemmDesc(ptr(ones), ptr(x4), None, ptr(c4), 0, 0, 0, 0, 200, 1, 0, 0, 200,
                      _lib.F32, dtype_code(x4), _lib.F32, 1, 1, 200, 700, 1.0, 1, _lib.F32))
    check(lib.vqaf_gemm_f32_group(4, ctypes.cast(descs, ctypes.c_void_p), st), "gemm group")
    want = (0.5 * a1.double() @ b1.double() + bias1.double(), a2.double().t() @ b2.double(),
            2.0 * torch.bmm(a3.double(), b3.double()), x4.double().sum(0))
    for got, w in zip((c1, c2, c3, c4), want):
        assert float((got.double() - w).abs().max()) < 2e-4 * max(1.0, float(w.abs().max()))
    assert lib.vqaf_gemm_f32_group(13, ctypes.cast(descs, ctypes.c_void_p), st) != 0
    # a stored product may leave as bf16 (one rounding of the f32 result); a product added into its output may not
    c5 = torch.full((4, 33, 40), 9.0, dtype=torch.bfloat16, device=dev)
    d5 = (_lib.GemmDesc * 1)(_lib.GemmDesc(ptr(a3), ptr(b3), None, ptr(c5), 33 * 64, 64, 1, 64 * 40, 40, 1, 0, 33 * 40, 40,
                                          _lib.F32, _lib.F32, _lib.F32, 4, 33, 40, 64, 2.0, 1, _lib.BF16))
    check(lib.vqaf_gemm_f32_group(1, ctypes.cast(d5, ctypes.c_void_p), st), "gemm group bf16")
    w5 = 2.0 * torch.bmm(a3.double(), b3.double())
    assert float((c5.double() - w5).abs().max()) < 2.0 ** -8 * float(w5.abs().max())
    d5[0].store = 0
    assert lib.vqaf_gemm_f32_group(1, ctypes.cast(d5, ctypes.c_void_p), st) != 0
