"""GPU parity of the DP-SGD step kernels against the per-sample-gradient oracle
(oracle/dp_oracle.py; parity unpinned by the reference, see its header), and of the MIA scoring
kernel against the golden vectors captured from the reference model.

Tolerances: f32 rtol 1e-5 on per-sample norms, clip factors and clipped sums (atol scaled to the
magnitude of the sums); bf16 rtol 1e-2.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from oracle import dp_oracle as do
from oracle import fusion_oracle as fo
from oracle import inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _classifier(seed, in_dim, hidden, C):
    m = do.make_classifier(in_dim, hidden, C, dp_fixed=True)
    m.load_state_dict(gi.classifier_params(seed, in_dim, hidden, C))
    return m.eval()  # dropout off: the oracle's vmap cannot replay a dropout stream


def _flat_to_named(opt, module, flat):
    out = {}
    names = {id(p): n for n, p in module.named_parameters()}
    for p in opt.params:
        o = opt._offsets[id(p)]
        out[names[id(p)]] = flat[o:o + p.numel()].view_as(p).cpu()
    return out


@pytest.mark.parametrize("in_dim,hidden,C,B,max_norm", [
    (1024, 1024, 430, 6, 1.0),     # BASELINE classifier, every sample clipped
    (1024, 1024, 430, 5, 1e4),     # nothing clipped
    (64, 64, 13, 33, 1.0),         # small groups (gs = 4), ragged tile edges
    (128, 256, 100, 70, 0.5),
])
def test_dp_step_f32_vs_oracle(dev, in_dim, hidden, C, B, max_norm):
    from vqa_amd import dp
    cpu_m = _classifier(100 + hidden, in_dim, hidden, C)
    x = gi.normal(7 + B, B, in_dim)
    y = gi.labels(8 + B, B, C)
    sigma, E = 1.0, 64
    # the same standard-normal draw on both sides, laid out parameter by parameter
    z = {n: gi.normal(1000 + i, *p.shape) for i, (n, p) in enumerate(cpu_m.named_parameters())}
    norms, clip, sums, grads = do.dp_step(cpu_m, x, y, max_norm, sigma, E, noise=z)

    gpu_m = _classifier(100 + hidden, in_dim, hidden, C).to(dev)
    gsm = dp.GradSampleModule(gpu_m)
    inner = torch.optim.SGD(gpu_m.parameters(), lr=0.1)
    opt = dp.DPOptimizer(inner, gsm, noise_multiplier=sigma, max_grad_norm=max_norm, expected_batch_size=E)
    xg = x.to(dev).requires_grad_()
    logits = gsm(xg)
    loss = nn.CrossEntropyLoss()(logits, y.to(dev))
    loss.backward()
    assert all(p.grad is None for p in gpu_m.parameters())  # no unclipped gradient is ever formed
    S = opt.clip_and_accumulate().clone()
    # groups of 2-4 channels make xhat ill-conditioned (variance of two numbers): 1e-4 there
    rtol_n = 1e-5 if hidden >= 256 else 1e-4
    np.testing.assert_allclose(opt.last_norms.cpu().numpy(), norms.numpy(), rtol=rtol_n)
    np.testing.assert_allclose(opt.last_clip.cpu().numpy(), clip.numpy(), rtol=rtol_n)
    got = _flat_to_named(opt, gsm, S)
    for n, want in sums.items():
        scale = max(1e-6, float(want.abs().max()))
        np.testing.assert_allclose(got["_module." + n].numpy(), want.numpy(), rtol=rtol_n, atol=(2e-6 if hidden >= 256 else 1e-4) * scale,
                                   err_msg=n)
    # noise + scale with the injected draw
    flat_z = torch.zeros(opt._numel)
    for (o, numel, shape), (n, p) in zip(opt.flat_noise_layout(), cpu_m.named_parameters()):
        flat_z[o:o + numel] = z[n].reshape(-1)
    opt.add_noise_and_scale(noise=flat_z.to(dev))
    for n, p in gsm.named_parameters():
        want = grads[n.replace("_module.", "")]
        scale = max(1e-6, float(want.abs().max()))
        np.testing.assert_allclose(p.grad.cpu().numpy(), want.numpy(), rtol=rtol_n, atol=(2e-6 if hidden >= 256 else 1e-4) * scale, err_msg=n)
    # input gradient (flows on to the non-private fusion / encoders): plain autograd value
    cx = x.clone().requires_grad_()
    nn.CrossEntropyLoss()(cpu_m(cx), y).backward()
    gscale = float(cx.grad.abs().max())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), cx.grad.numpy(), rtol=1e-4,
                               atol=(1e-5 if hidden >= 256 else 2e-4) * gscale)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), cpu_m(x).detach().numpy(), rtol=1e-4, atol=1e-5)


def test_dp_kernels_bf16_same_inputs(dev):
    """bf16 storage, kernel by kernel on identical bf16 operands: ghost norms, clipped sums and
    GroupNorm against f32 math on the same rounded values (rtol 1e-2 is the stated bar; the f32
    accumulation actually lands far inside it)."""
    from vqa_amd._lib import BF16, check, load, ptr, stream_ptr
    lib, st = load(), stream_ptr(dev)
    B, In, Out = 37, 1024, 430
    a = gi.normal(1, B, In).bfloat16()
    g = (gi.normal(2, B, Out) * 1e-2).bfloat16()
    coef = torch.rand(B, generator=torch.Generator().manual_seed(3)) + 0.5  # own generator: order-independent
    ad, gd, cd = a.to(dev), g.to(dev), coef.to(dev)
    sq = torch.empty(B, device=dev)
    check(lib.vqaf_ghost_sqnorm_linear(ptr(ad), ptr(gd), B, In, Out, BF16, 1, 0, ptr(sq), st), "sq")
    af, gf = a.double(), g.double()
    want = af.pow(2).sum(1) * gf.pow(2).sum(1) + gf.pow(2).sum(1)
    np.testing.assert_allclose(sq.cpu().numpy(), want.numpy(), rtol=1e-5)
    SW = torch.empty(Out, In, device=dev)
    Sb = torch.empty(Out, device=dev)
    check(lib.vqaf_clipped_sum_linear(ptr(ad), ptr(gd), ptr(cd), B, In, Out, BF16, 0, ptr(SW), ptr(Sb), st), "cs")
    wantW = torch.einsum("b,bo,bi->oi", coef.double(), gf, af)
    # coef * g is re-rounded to bf16 before the MFMA (2^-9 per element), so entries that are
    # sums with cancellation need an absolute term tied to the size of the matrix
    wscale = float(wantW.abs().max())
    np.testing.assert_allclose(SW.cpu().numpy(), wantW.numpy(), rtol=1e-2, atol=3e-3 * wscale)
    assert float((SW.cpu().double() - wantW).norm() / wantW.norm()) < 3e-3
    np.testing.assert_allclose(Sb.cpu().numpy(), (coef.double()[:, None] * gf).sum(0).numpy(), rtol=1e-5, atol=1e-7)
    # accumulate flag adds a second micro-batch
    check(lib.vqaf_clipped_sum_linear(ptr(ad), ptr(gd), ptr(cd), B, In, Out, BF16, 1, ptr(SW), ptr(Sb), st), "cs")
    np.testing.assert_allclose(SW.cpu().numpy(), 2 * wantW.numpy(), rtol=1e-2, atol=4e-3 * wscale)
    # odd row pitches take the narrower load paths
    for (B2, In2, Out2) in ((70, 200, 430), (9, 66, 13), (33, 65, 7)):
        a2, g2 = gi.normal(11, B2, In2).bfloat16(), gi.normal(12, B2, Out2).bfloat16()
        c2 = torch.rand(B2, generator=torch.Generator().manual_seed(4 + B2)) + 0.5
        a2d, g2d, c2d = a2.to(dev), g2.to(dev), c2.to(dev)
        SW2 = torch.empty(Out2, In2, device=dev)
        Sb2 = torch.empty(Out2, device=dev)
        check(lib.vqaf_clipped_sum_linear(ptr(a2d), ptr(g2d), ptr(c2d), B2, In2, Out2, BF16, 0, ptr(SW2),
                                          ptr(Sb2), st), "cs")
        w2 = torch.einsum("b,bo,bi->oi", c2.double(), g2.double(), a2.double())
        np.testing.assert_allclose(SW2.cpu().numpy(), w2.numpy(), rtol=1e-2, atol=4e-3 * float(w2.abs().max()))
        sb_want = (c2.double()[:, None] * g2.double()).sum(0)
        np.testing.assert_allclose(Sb2.cpu().numpy(), sb_want.numpy(), rtol=1e-5, atol=2e-6 * float(sb_want.abs().max()))
    # GroupNorm + ReLU forward / backward in bf16 storage
    C, groups = 2048, 32
    x = gi.normal(3, B, C).bfloat16()
    gam, bet = (1 + 0.1 * gi.normal(4, C)).bfloat16(), (0.1 * gi.normal(5, C)).bfloat16()
    y = torch.empty(B, C, dtype=torch.bfloat16, device=dev)
    xh = torch.empty_like(y)
    rstd = torch.empty(B, groups, device=dev)
    xd, gamd, betd = x.to(dev), gam.to(dev), bet.to(dev)  # keep the device copies alive
    check(lib.vqaf_groupnorm_fwd(ptr(xd), ptr(gamd), ptr(betd), B, C, groups, 1e-5, 1,
                                 BF16, ptr(y), ptr(xh), ptr(rstd), 0.0, 0, None, st), "gn")
    want_y = torch.relu(torch.nn.functional.group_norm(x.float(), groups, gam.float(), bet.float(), 1e-5))
    np.testing.assert_allclose(y.float().cpu().numpy(), want_y.numpy(), rtol=1e-2, atol=1e-2)


def test_dp_step_bf16_end_to_end(dev):
    """Whole classifier in bf16 storage against the f32 oracle.  bf16 rounding of the
    pre-activations flips a few ReLU gates, so element-wise bars do not apply; direction and
    size of every clipped sum must still agree."""
    from vqa_amd import dp
    in_dim, hidden, C, B = 1024, 1024, 430, 8
    cpu_m = _classifier(300, in_dim, hidden, C)
    x = gi.normal(9, B, in_dim).bfloat16().float()
    y = gi.labels(10, B, C)
    norms, clip, sums, _ = do.dp_step(cpu_m, x, y, 1.0, 0.0, B)
    gpu_m = _classifier(300, in_dim, hidden, C).to(dev)
    gsm = dp.GradSampleModule(gpu_m)
    opt = dp.DPOptimizer(torch.optim.SGD(gpu_m.parameters(), lr=0.1), gsm, noise_multiplier=0.0,
                         max_grad_norm=1.0, expected_batch_size=B)
    logits = gsm(x.to(dev).bfloat16())
    assert logits.dtype == torch.bfloat16
    np.testing.assert_allclose(logits.detach().float().cpu().numpy(), cpu_m(x).detach().numpy(), rtol=1e-2, atol=3e-2)
    nn.CrossEntropyLoss()(logits.float(), y.to(dev)).backward()
    S = opt.clip_and_accumulate()
    np.testing.assert_allclose(opt.last_norms.cpu().numpy(), norms.numpy(), rtol=2e-2)
    got = _flat_to_named(opt, gsm, S)
    for n, want in sums.items():
        g = got["_module." + n].double().reshape(-1)
        w = want.double().reshape(-1)
        cos = float((g @ w) / (g.norm() * w.norm()))
        assert cos > 0.99 and abs(float(g.norm() / w.norm()) - 1) < 3e-2, (n, cos)


def test_noise_scale_in_kernel_draws(dev):
    """The in-kernel Gaussian: statistics, reproducibility from (seed, counter), and independence
    of the launch geometry (element i always uses the same draw)."""
    from vqa_amd import ops
    from vqa_amd._lib import check, load, ptr, stream_ptr
    n = 4_644_270  # parameter count of the BASELINE classifier
    S = torch.zeros(n, device=dev)
    out = torch.empty(n, device=dev)
    rng = ops.PhiloxState(dev, seed=99)
    check(load().vqaf_noise_scale(ptr(S), None, ptr(rng.state), 2.0, 0.5, n, ptr(out), stream_ptr(dev)), "ns")
    z = out.cpu().double()  # = N(0,1) since 2.0 * 0.5 = 1
    assert abs(z.mean().item()) < 5 / n ** 0.5
    assert abs(z.std().item() - 1.0) < 3e-3
    assert abs((z ** 4).mean().item() - 3.0) < 2e-2
    assert 4.5 < z.abs().max().item() < 7.0
    # a shorter call sees the same leading draws
    m = 1001
    out2 = torch.empty(m, device=dev)
    check(load().vqaf_noise_scale(ptr(S), None, ptr(rng.state), 2.0, 0.5, m, ptr(out2), stream_ptr(dev)), "ns")
    assert torch.equal(out2.cpu(), out[:m].cpu())
    rng.advance()
    check(load().vqaf_noise_scale(ptr(S), None, ptr(rng.state), 2.0, 0.5, m, ptr(out2), stream_ptr(dev)), "ns")
    assert not torch.equal(out2.cpu(), out[:m].cpu())
    # std 0 -> pure scaling
    S.fill_(3.0)
    check(load().vqaf_noise_scale(ptr(S), None, None, 0.0, 0.25, n, ptr(out), stream_ptr(dev)), "ns")
    assert torch.equal(out.cpu(), torch.full((n,), 0.75))


def test_privacy_engine_surface_and_training_step(dev):
    """The call sequence of train.py:124-164 + utils/src.py:66-106 on a tiny problem."""
    from vqa_amd import dp
    torch.manual_seed(0)
    cls = nn.Sequential(nn.Linear(64, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.3),
                        nn.Linear(128, 64), nn.BatchNorm1d(64), nn.ReLU(), nn.Linear(64, 13))
    cls = dp.ModuleValidator.fix(cls).to(dev)
    assert isinstance(cls[1], nn.GroupNorm) and cls[1].num_groups == 32 and isinstance(cls[5], nn.GroupNorm)
    ds = torch.utils.data.TensorDataset(torch.randn(256, 64), torch.randint(0, 13, (256,)))
    loader = torch.utils.data.DataLoader(ds, batch_size=32, shuffle=True)
    engine = dp.PrivacyEngine(secure_mode=False)
    opt = torch.optim.AdamW(cls.parameters(), lr=1e-2, weight_decay=0.01)
    module, opt, loader = engine.make_private(module=cls, optimizer=opt, data_loader=loader,
                                              noise_multiplier=1.0, max_grad_norm=1.0, poisson_sampling=True)
    assert sorted(module.state_dict())[0].startswith("_module.")
    assert opt.expected_batch_size == 32
    before = [p.detach().clone() for p in module.parameters()]
    steps = 0
    module.train()
    for batch in loader:
        if batch is None:
            continue
        xb, yb = batch
        opt.zero_grad()
        loss = nn.CrossEntropyLoss()(module(xb.to(dev)), yb.to(dev))
        loss.backward()
        opt.step()
        steps += 1
        assert bool((opt.last_norms * opt.last_clip <= 1.0 + 1e-4).all())  # every sample clipped to C
    assert steps >= 6
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, module.parameters()))
    eps = engine.get_epsilon(1e-5)
    assert 0.1 < eps < 50
    engine2 = dp.PrivacyEngine()
    cls2 = dp.ModuleValidator.fix(nn.Sequential(nn.Linear(64, 64), nn.BatchNorm1d(64), nn.ReLU(),
                                                nn.Linear(64, 13))).to(dev)
    _, opt2, _ = engine2.make_private_with_epsilon(
        module=cls2, optimizer=torch.optim.AdamW(cls2.parameters()), target_epsilon=3.0, target_delta=1e-5,
        data_loader=torch.utils.data.DataLoader(ds, batch_size=32), epochs=5, max_grad_norm=10.0)
    assert 0.3 < opt2.noise_multiplier < 5.0


# ------------------------------------------------------------------ MIA scoring
def test_ce_conf_vs_golden(dev):
    from vqa_amd import mia
    g = load_golden("model_N197")
    logits, y = torch.from_numpy(g["logits"]).to(dev), torch.from_numpy(g["targets"]).to(dev)
    loss, conf, pred = mia.score_logits(logits, y)
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(conf.cpu().numpy(), g["conf"], rtol=1e-5, atol=1e-7)
    assert np.array_equal(pred.cpu().numpy(), g["pred"])


def test_ce_conf_shapes_dtypes_and_gradient(dev):
    from vqa_amd import ops
    for B, C in ((1, 1), (7, 13), (130, 430), (33, 1000)):
        lg = gi.normal(B + C, B, C, scale=3.0)
        y = gi.labels(B, B, C)
        loss, conf, pred, grad = ops.ce_conf(lg.to(dev), y.to(dev), want_pred=True, grad_scale=1.0 / B)
        np.testing.assert_allclose(loss.cpu().numpy(), fo.per_sample_ce(lg, y).numpy(), rtol=1e-5, atol=1e-6)
        c_ref, p_ref = fo.max_softmax(lg)
        np.testing.assert_allclose(conf.cpu().numpy(), c_ref.numpy(), rtol=1e-5, atol=1e-7)
        assert torch.equal(pred.cpu(), p_ref)
        lr = lg.clone().requires_grad_()
        nn.CrossEntropyLoss()(lr, y).backward()
        np.testing.assert_allclose(grad.cpu().numpy(), lr.grad.numpy(), rtol=1e-5, atol=1e-8)
        lb = lg.bfloat16()
        loss_b, conf_b, _, _ = ops.ce_conf(lb.to(dev), y.to(dev))
        np.testing.assert_allclose(loss_b.cpu().numpy(), fo.per_sample_ce(lb.float(), y).numpy(), rtol=1e-5, atol=1e-6)


def test_fused_adamw_step_matches_torch_adamw(dev):
    """DPOptimizer's fused noise+scale+AdamW kernel against the unfused path
    (vqaf_noise_scale followed by torch.optim.AdamW.step) over several steps, same injected noise."""
    from vqa_amd import dp
    in_dim, hidden, C, B = 128, 256, 100, 32
    x, y = gi.normal(1, B, in_dim).to(dev), gi.labels(2, B, C).to(dev)

    def make(fuse):
        m = _classifier(77, in_dim, hidden, C).to(dev)
        gsm = dp.GradSampleModule(m)
        inner = torch.optim.AdamW(m.parameters(), lr=3e-3, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-7)
        return gsm, dp.DPOptimizer(inner, gsm, noise_multiplier=0.7, max_grad_norm=1.0,
                                   expected_batch_size=B, fuse_adamw=fuse)

    gsm_f, opt_f = make(True)
    gsm_u, opt_u = make(False)
    assert opt_f._fusable() and not opt_u._fusable()
    for step in range(4):
        z = gi.normal(100 + step, opt_f._numel).to(dev)
        for gsm, opt in ((gsm_f, opt_f), (gsm_u, opt_u)):
            nn.CrossEntropyLoss()(gsm(x), y).backward()
            opt.clip_and_accumulate()
        opt_f._fused_step(noise=z)
        opt_u.add_noise_and_scale(noise=z)
        opt_u.original_optimizer.step()
        for (n, pf), (_, pu) in zip(gsm_f.named_parameters(), gsm_u.named_parameters()):
            np.testing.assert_allclose(pf.detach().cpu().numpy(), pu.detach().cpu().numpy(), rtol=2e-6, atol=1e-7,
                                       err_msg=f"step {step} {n}")
            gsc = float(pu.grad.abs().max())
            np.testing.assert_allclose(pf.grad.cpu().numpy(), pu.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * gsc)
    # the bf16 copy the next forward reads is the rounded parameter bucket
    assert torch.equal(opt_f._shadow, opt_f._P.bfloat16())
    assert gsm_f.shadow is not None and int(opt_f._step[1].item()) == 4
    sd = opt_f.state_dict()
    assert len(sd["state"]) == len(opt_f.params) and float(sd["state"][0]["step"]) == 4.0
    ref_sd = opt_u.state_dict()
    np.testing.assert_allclose(sd["state"][0]["exp_avg"].cpu().numpy(), ref_sd["state"][0]["exp_avg"].cpu().numpy(),
                               rtol=1e-5, atol=1e-9)
    # public step(): in-kernel noise, bf16 activations pick up the shadow weights
    gsm_f.train()
    nn.CrossEntropyLoss()(gsm_f(x.bfloat16()).float(), y).backward()
    before = opt_f._P.clone()
    opt_f.step()
    assert not torch.equal(before, opt_f._P) and int(opt_f._step[1].item()) == 5


def test_ghost_clip_equals_separate_kernels(dev):
    """vqaf_ghost_clip (all layers + clip coefficient + in-place scaling in one launch) against
    vqaf_ghost_sqnorm_linear/_affine + vqaf_clip_coef, f32 and bf16, vector and element paths."""
    import ctypes
    from vqa_amd._lib import check, dtype_code, load, ptr, stream_ptr
    lib, st = load(), stream_ptr(dev)
    B = 37
    for dt in (torch.float32, torch.bfloat16):
        layers = [("lin1", 1024, 2048), ("aff", 2048, 2048), ("lin0", 2048, 430), ("lin1", 66, 13)]
        a = [gi.normal(10 + i, B, L[1]).to(dt).to(dev) for i, L in enumerate(layers)]
        g = [(gi.normal(20 + i, B, L[2]) * 1e-2).to(dt).to(dev) for i, L in enumerate(layers)]
        kinds = [1, 2, 0, 1]
        sq = torch.empty(B, device=dev)
        for i, (L, k) in enumerate(zip(layers, kinds)):
            if k == 2:
                check(lib.vqaf_ghost_sqnorm_affine(ptr(a[i]), ptr(g[i]), B, L[2], dtype_code(g[i]), int(i > 0),
                                                   ptr(sq), st), "aff")
            else:
                check(lib.vqaf_ghost_sqnorm_linear(ptr(a[i]), ptr(g[i]), B, L[1], L[2], dtype_code(g[i]), k,
                                                   int(i > 0), ptr(sq), st), "lin")
        coef_ref, norm_ref = torch.empty(B, device=dev), torch.empty(B, device=dev)
        check(lib.vqaf_clip_coef(ptr(sq), B, 1.0, float(B), ptr(coef_ref), ptr(norm_ref), st), "clip")
        n = len(layers)
        g2 = [t.clone() for t in g]
        pa = (ctypes.c_void_p * n)(*[ptr(t) for t in a])
        pg = (ctypes.c_void_p * n)(*[ptr(t) for t in g2])
        din = (ctypes.c_int * n)(*[L[1] for L in layers])
        dout = (ctypes.c_int * n)(*[L[2] for L in layers])
        kd = (ctypes.c_int * n)(*kinds)
        coef, norms, sq2 = (torch.empty(B, device=dev) for _ in range(3))
        check(lib.vqaf_ghost_clip(n, pa, pg, din, dout, kd, B, dtype_code(g[0]), 1.0, float(B), 1, ptr(coef),
                                  ptr(norms), ptr(sq2), st), "ghost_clip")
        np.testing.assert_allclose(sq2.cpu().numpy(), sq.cpu().numpy(), rtol=2e-6)
        np.testing.assert_allclose(norms.cpu().numpy(), norm_ref.cpu().numpy(), rtol=2e-6)
        np.testing.assert_allclose(coef.cpu().numpy(), coef_ref.cpu().numpy(), rtol=2e-6)
        for i in range(n):  # rows scaled in place by their sample's coefficient
            want = (g[i].float() * coef[:, None]).to(dt)
            assert torch.equal(g2[i], want), i


# ------------------------------------------------------------------ benchmark-sized batches
CLASSIFIER_LINEARS = [(1024, 2048), (2048, 1024), (1024, 430)]


@pytest.mark.parametrize("B", [128, 256, 512, 1024])
@pytest.mark.parametrize("In,Out", CLASSIFIER_LINEARS)
def test_clipped_sum_bf16_at_benchmark_batches(dev, B, In, Out):
    """The bf16 MFMA clipped sum at the batch sizes the DP-SGD step of the benchmark runs
    (2..16 pipeline stages of 64 rows; the 430-row head splits the batch over blockIdx.z and adds
    partial tiles with f32 atomics from B = 256 on) against an f64 contraction of the same bf16
    operands, with the coefficient applied in the kernel and with pre-scaled rows (coef == NULL)."""
    from vqa_amd._lib import BF16, check, load, ptr, stream_ptr
    lib, st = load(), stream_ptr(dev)
    a = gi.normal(31, B, In).bfloat16()
    g = (gi.normal(32, B, Out) * 1e-2).bfloat16()
    coef = torch.from_numpy(np.random.default_rng(33).uniform(0.5, 1.5, B).astype(np.float32))
    ad, gd, cd = a.to(dev), g.to(dev), coef.to(dev)
    SW = torch.full((Out, In), float("nan"), device=dev)
    Sb = torch.full((Out,), float("nan"), device=dev)
    # pre-scaled rows: exact products of bf16 operands, f32 accumulation
    check(lib.vqaf_clipped_sum_linear(ptr(ad), ptr(gd), None, B, In, Out, BF16, 0, ptr(SW), ptr(Sb), st), "cs")
    want = g.double().t() @ a.double()
    wscale = float(want.abs().max())
    np.testing.assert_allclose(SW.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-6 * wscale)
    sb_want = g.double().sum(0)
    np.testing.assert_allclose(Sb.cpu().numpy(), sb_want.numpy(), rtol=1e-5, atol=2e-6 * float(sb_want.abs().max()))
    # micro-batches of 64 rows through the accumulate flag take the one-stage, unsplit path:
    # both decompositions of the same sum agree to f32 summation-order noise
    SW2 = torch.zeros(Out, In, device=dev)
    Sb2 = torch.zeros(Out, device=dev)
    for lo in range(0, B, 64):
        check(lib.vqaf_clipped_sum_linear(ptr(ad[lo:lo + 64]), ptr(gd[lo:lo + 64]), None, 64, In, Out, BF16, 1,
                                          ptr(SW2), ptr(Sb2), st), "cs")
    np.testing.assert_allclose(SW.cpu().numpy(), SW2.cpu().numpy(), rtol=1e-5, atol=2e-6 * wscale)
    np.testing.assert_allclose(Sb.cpu().numpy(), Sb2.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # coefficient applied while staging (coef * g re-rounded to bf16: 2^-9 per element)
    check(lib.vqaf_clipped_sum_linear(ptr(ad), ptr(gd), ptr(cd), B, In, Out, BF16, 0, ptr(SW), ptr(Sb), st), "cs")
    wantc = (coef.double()[:, None] * g.double()).t() @ a.double()
    cscale = float(wantc.abs().max())
    np.testing.assert_allclose(SW.cpu().numpy(), wantc.numpy(), rtol=1e-2, atol=2e-3 * cscale)
    assert float((SW.cpu().double() - wantc).norm() / wantc.norm()) < 3e-3
    np.testing.assert_allclose(Sb.cpu().numpy(), (coef.double()[:, None] * g.double()).sum(0).numpy(),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ghost_clip_at_benchmark_batch(dev, dt):
    """vqaf_ghost_clip over the five wrapped layers of the classifier at B = 1024 against f64
    arithmetic on the same stored values, and against the separate norm / coefficient kernels."""
    import ctypes
    from vqa_amd._lib import check, dtype_code, load, ptr, stream_ptr
    lib, st = load(), stream_ptr(dev)
    B, C_clip = 1024, 1.0
    layers = [(1, 1024, 2048), (2, 2048, 2048), (1, 2048, 1024), (2, 1024, 1024), (1, 1024, 430)]
    a = [gi.normal(40 + i, B, L[1]).to(dt) for i, L in enumerate(layers)]
    # backprops of a mean-reduced loss are O(1/B); a per-sample spread makes some samples clip and some not
    spread = torch.from_numpy(np.random.default_rng(5).uniform(0.02, 3.0, B).astype(np.float32))[:, None]
    g = [(gi.normal(50 + i, B, L[2]) * spread * (2e-2 / B / (L[1] ** 0.5))).to(dt) for i, L in enumerate(layers)]
    sq = torch.zeros(B, dtype=torch.float64)
    for (k, In, Out), ai, gi_ in zip(layers, a, g):
        ad_, gd_ = ai.double(), gi_.double()
        if k == 2:
            sq += (ad_ * gd_).pow(2).sum(1) + gd_.pow(2).sum(1)
        else:
            sq += ad_.pow(2).sum(1) * gd_.pow(2).sum(1) + gd_.pow(2).sum(1)
    norm_want = B * sq.sqrt()
    coef_want = B * (C_clip / (norm_want + 1e-6)).clamp(max=1.0)
    assert 0.2 < float((coef_want < B).double().mean()) < 0.8  # both branches of the clip are exercised
    ad, gd = [t.to(dev) for t in a], [t.to(dev) for t in g]
    n = len(layers)
    pa = (ctypes.c_void_p * n)(*[ptr(t) for t in ad])
    pg = (ctypes.c_void_p * n)(*[ptr(t) for t in gd])
    din = (ctypes.c_int * n)(*[L[1] for L in layers])
    dout = (ctypes.c_int * n)(*[L[2] for L in layers])
    kd = (ctypes.c_int * n)(*[L[0] for L in layers])
    coef, norms, sq2 = (torch.empty(B, device=dev) for _ in range(3))
    check(lib.vqaf_ghost_clip(n, pa, pg, din, dout, kd, B, dtype_code(gd[0]), C_clip, float(B), 1, ptr(coef),
                              ptr(norms), ptr(sq2), st), "ghost_clip")
    np.testing.assert_allclose(sq2.cpu().numpy(), sq.numpy(), rtol=1e-5)
    np.testing.assert_allclose(norms.cpu().numpy(), norm_want.numpy(), rtol=1e-5)
    np.testing.assert_allclose(coef.cpu().numpy(), coef_want.numpy(), rtol=1e-5)
    for i in range(n):  # rows scaled in place by their sample's coefficient
        assert torch.equal(gd[i].cpu(), (g[i].float() * coef.cpu()[:, None]).to(dt)), i
    # the separate kernels on the unscaled rows
    gd0 = [t.to(dev) for t in g]
    sq3 = torch.empty(B, device=dev)
    for i, (k, In, Out) in enumerate(layers):
        if k == 2:
            check(lib.vqaf_ghost_sqnorm_affine(ptr(ad[i]), ptr(gd0[i]), B, Out, dtype_code(gd0[i]), int(i > 0),
                                               ptr(sq3), st), "aff")
        else:
            check(lib.vqaf_ghost_sqnorm_linear(ptr(ad[i]), ptr(gd0[i]), B, In, Out, dtype_code(gd0[i]), k,
                                               int(i > 0), ptr(sq3), st), "lin")
    np.testing.assert_allclose(sq3.cpu().numpy(), sq2.cpu().numpy(), rtol=2e-6)


def _unambiguous_rows(cpu_m, x, tol=2e-5):
    """Rows of x whose ReLU inputs all stay clear of zero: a pre-activation within rounding noise
    of zero may open the gate on one side and not on the other, which changes that sample's
    gradient by a whole unit's worth -- a property of ReLU, not of either implementation."""
    keep = torch.ones(x.shape[0], dtype=torch.bool)
    h = x
    with torch.no_grad():
        for layer in cpu_m:
            if isinstance(layer, nn.ReLU):
                keep &= h.abs().min(dim=1).values > tol
            h = layer(h)
    return keep


@pytest.mark.parametrize("library_sums", [True, False])
def test_dp_step_f32_batch_256_vs_chunked_oracle(dev, library_sums, monkeypatch):
    """One whole DP step of the BASELINE classifier at B = 256 (4 pipeline stages, split head)
    against the per-sample-gradient oracle evaluated in chunks of 32 samples; the clip norm is the
    median per-sample norm, so both branches of the clip are taken.  The Linear layers' weight sums as
    library GEMMs + one grouped bias-sum launch, and all of them in the grouped launch."""
    from vqa_amd import dp
    monkeypatch.setattr(dp, "F32_SUMS_IN_LIBRARY", library_sums)
    in_dim, hidden, C, B, sigma, E = 1024, 1024, 430, 256, 1.0, 1024
    cpu_m = _classifier(500, in_dim, hidden, C)
    x, y = gi.normal(61, B + 64, in_dim), gi.labels(62, B + 64, C)
    keep = _unambiguous_rows(cpu_m, x)
    assert int(keep.sum()) >= B
    x, y = x[keep][:B].contiguous(), y[keep][:B].contiguous()
    norms0, _, _, _ = do.dp_step(cpu_m, x[:32], y[:32], 1.0, 0.0, E)
    max_norm = float(norms0.median())
    norms, clips, sums = [], [], None
    for lo in range(0, B, 32):
        n_, c_, s_, _ = do.dp_step(cpu_m, x[lo:lo + 32], y[lo:lo + 32], max_norm, 0.0, E)
        norms.append(n_)
        clips.append(c_)
        sums = s_ if sums is None else {k: sums[k].double() + s_[k].double() for k in s_}
    norms, clips = torch.cat(norms), torch.cat(clips)
    assert 0.2 < float((clips < 1).double().mean()) < 0.8
    z = {n: gi.normal(2000 + i, *p.shape) for i, (n, p) in enumerate(cpu_m.named_parameters())}

    gpu_m = _classifier(500, in_dim, hidden, C).to(dev)
    gsm = dp.GradSampleModule(gpu_m)
    opt = dp.DPOptimizer(torch.optim.SGD(gpu_m.parameters(), lr=0.1), gsm, noise_multiplier=sigma,
                         max_grad_norm=max_norm, expected_batch_size=E)
    nn.CrossEntropyLoss()(gsm(x.to(dev)), y.to(dev)).backward()
    S = opt.clip_and_accumulate().clone()
    np.testing.assert_allclose(opt.last_norms.cpu().numpy(), norms.numpy(), rtol=1e-5)
    np.testing.assert_allclose(opt.last_clip.cpu().numpy(), clips.numpy(), rtol=1e-5)
    got = _flat_to_named(opt, gsm, S)
    for n, want in sums.items():
        scale = max(1e-6, float(want.abs().max()))
        np.testing.assert_allclose(got["_module." + n].numpy(), want.float().numpy(), rtol=1e-5, atol=4e-6 * scale,
                                   err_msg=n)
    flat_z = torch.zeros(opt._numel)
    for (o, numel, shape), (n, p) in zip(opt.flat_noise_layout(), cpu_m.named_parameters()):
        flat_z[o:o + numel] = z[n].reshape(-1)
    opt.add_noise_and_scale(noise=flat_z.to(dev))
    for n, p in gsm.named_parameters():
        k = n.replace("_module.", "")
        want = ((sums[k].double() + sigma * max_norm * z[k].double()) / E).float()
        np.testing.assert_allclose(p.grad.cpu().numpy(), want.numpy(), rtol=1e-5, atol=4e-6 * float(want.abs().max()),
                                   err_msg=n)


def test_dp_step_bf16_batch_1024_bounds_and_direction(dev):
    """bf16 tapes at the benchmark's batch: every sample's clipped gradient, as the clipped-sum
    kernels consume it (rows re-rounded to bf16 after scaling), has norm <= C; the sums agree in
    direction and size with the f64 contraction of the recorded tapes."""
    from vqa_amd import dp
    in_dim, hidden, C, B = 1024, 1024, 430, 1024
    gpu_m = _classifier(600, in_dim, hidden, C).to(dev)
    gsm = dp.GradSampleModule(gpu_m)
    opt = dp.DPOptimizer(torch.optim.SGD(gpu_m.parameters(), lr=0.1), gsm, noise_multiplier=0.0,
                         max_grad_norm=1.0, expected_batch_size=B)
    x, y = gi.normal(71, B, in_dim), gi.labels(73, B, C)
    logits = gsm(x.to(dev).bfloat16())
    nn.CrossEntropyLoss()(logits.float(), y.to(dev)).backward()
    tapes = [(t.kind, t.module, t.a.detach().clone(), t.g.detach().clone()) for t in gsm.tapes]
    keep_g = [t.g for t in gsm.tapes]  # scaled in place by the step

    def sample_norms(rows):  # per-sample gradient norm of the tapes with the given backprop rows, f64
        sq = torch.zeros(B, dtype=torch.float64)
        for (kind, m, a, _g0), g in zip(tapes, rows):
            a_, g_ = a.double().cpu(), g.double().cpu()
            sq += (a_.pow(2).sum(1) * g_.pow(2).sum(1) if kind == "linear" else (a_ * g_).pow(2).sum(1)) \
                + g_.pow(2).sum(1)
        return sq.sqrt()

    true_norms = B * sample_norms([t[3] for t in tapes])
    max_norm = float(true_norms.median())  # half of the samples clip
    opt.max_grad_norm = max_norm
    S = opt.clip_and_accumulate()
    np.testing.assert_allclose(opt.last_norms.cpu().numpy(), true_norms.numpy(), rtol=1e-5)
    coef = (opt.last_clip * B).double().cpu()
    names = {id(p): n for n, p in gsm.named_parameters()}
    for (kind, m, a, g0), gs in zip(tapes, keep_g):
        assert gs.dtype == torch.bfloat16
        a_, g_ = a.double().cpu(), gs.double().cpu()  # what the clipped sums contract
        want_w = g_.t() @ a_ if kind == "linear" else (a_ * g_).sum(0)
        o = opt._offsets[id(m.weight)]
        got_w = S[o:o + m.weight.numel()].view_as(m.weight).double().cpu()
        np.testing.assert_allclose(got_w.numpy(), want_w.numpy(), rtol=1e-4, atol=1e-5 * float(want_w.abs().max()),
                                   err_msg=names[id(m.weight)])
        # against the unrounded product coef * g (bf16 bar)
        exact = (coef[:, None] * g0.double().cpu())
        ref_w = exact.t() @ a_ if kind == "linear" else (a_ * exact).sum(0)
        rel = float((got_w - ref_w).norm() / ref_w.norm())
        assert rel < 1e-2, (names[id(m.weight)], rel)
    clipped_norm = sample_norms(keep_g)
    assert float(clipped_norm.max()) <= max_norm, (float(clipped_norm.max()), max_norm)
    assert float(clipped_norm.max()) > 0.99 * max_norm  # the margin is small
    frac_clipped = float((opt.last_clip < 0.999).double().mean())
    assert 0.3 < frac_clipped < 0.7, frac_clipped


# ------------------------------------------------------------------ fused small launches
def test_groupnorm_relu_dropout_in_one_launch(dev):
    """GroupNorm -> ReLU -> Dropout(p) as the classifier holds them (model/model.py:56-58) in the
    GroupNorm launch: the output equals relu(group_norm(x)) scaled by 1/(1-p) where kept and is
    exactly zero where dropped, the keep rate is p-consistent, the mask changes from call to call
    and repeats for the same generator state; the backward equals autograd's through the same mask."""
    from vqa_amd import dp, ops
    B, C, p = 256, 2048, 0.3
    torch.manual_seed(3)
    seq = nn.Sequential(nn.Linear(64, C), nn.GroupNorm(32, C), nn.ReLU(), nn.Dropout(p), nn.Linear(C, 8)).to(dev)
    gsm = dp.GradSampleModule(seq)
    gsm.train()
    x = gi.normal(5, B, 64).to(dev)
    h = seq[0](x).detach().requires_grad_()
    tape_mod = dp.GradSampleModule(nn.Sequential(seq[1], seq[2], seq[3]))
    tape_mod.train()
    y1 = tape_mod(h)
    with torch.no_grad():
        ref = torch.relu(torch.nn.functional.group_norm(h.detach(), 32, seq[1].weight, seq[1].bias, seq[1].eps))
    kept = y1.detach() != 0
    live = ref > 0
    assert bool((kept <= live).all())
    np.testing.assert_allclose(y1[kept].detach().cpu().numpy(), (ref[kept] / (1 - p)).cpu().numpy(), rtol=1e-5, atol=1e-6)
    rate = float(kept.sum()) / float(live.sum())
    n_live = float(live.sum())
    assert abs(rate - (1 - p)) < 5 * (p * (1 - p) / n_live) ** 0.5 + 2e-5   # 16-bit threshold granularity
    assert int(tape_mod.dropout_rng.state[1].item()) == 1 and int(tape_mod.dropout_rng.state[2].item()) == 0
    # backward through the same mask
    cot = gi.normal(7, B, C).to(dev)
    (y1 * cot).sum().backward()
    h2 = h.detach().clone().requires_grad_()
    mask = kept.float() / (1 - p)
    (torch.relu(torch.nn.functional.group_norm(h2, 32, seq[1].weight.detach(), seq[1].bias.detach(), seq[1].eps))
     * mask * cot).sum().backward()
    np.testing.assert_allclose(h.grad.cpu().numpy(), h2.grad.cpu().numpy(), rtol=2e-4, atol=2e-5)
    # a second call draws another mask; the same state draws the same one
    y2 = tape_mod(h.detach()).detach()
    assert not torch.equal(y2 != 0, kept)
    tape_mod.dropout_rng.state[1] = 0
    tape_mod._drop_calls = 0
    y3 = tape_mod(h.detach()).detach()
    assert torch.equal(y3, y1.detach())
    # keyed on an optimizer's step count instead (no counter launch of its own): the same (call site,
    # step) gives the same mask, another step or another call site another one
    own = tape_mod.dropout_rng.state.clone()
    tape_mod.step_counter = torch.tensor([0, 5, 0, 0], dtype=torch.int64, device=dev)
    tape_mod._drop_calls = 0
    a = tape_mod(h.detach()).detach()
    b = tape_mod(h.detach()).detach()          # second call site of the same step
    tape_mod._drop_calls = 0
    c = tape_mod(h.detach()).detach()
    tape_mod.step_counter[1] = 6
    tape_mod._drop_calls = 0
    d = tape_mod(h.detach()).detach()
    assert torch.equal(a, c) and not torch.equal(a != 0, b != 0) and not torch.equal(a != 0, d != 0)
    assert torch.equal(tape_mod.dropout_rng.state, own)
    tape_mod.step_counter = None
    # eval mode: no dropout, no draw
    tape_mod.eval()
    before = int(tape_mod.dropout_rng.state[1].item())
    np.testing.assert_allclose(tape_mod(h.detach()).detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert int(tape_mod.dropout_rng.state[1].item()) == before


def test_in_kernel_counter_bumps_and_zero_multi(dev):
    """The generator counters that kernels bump themselves (last workgroup out) end where the
    separate vqaf_rng_advance launches would have left them, and the ticket word is back at zero;
    vqaf_zero_multi clears exactly the ranges it is given."""
    import ctypes
    from vqa_amd import ops
    from vqa_amd._lib import check, load, ptr, stream_ptr
    lib, st = load(), stream_ptr(dev)
    # fused AdamW counting its own steps: many workgroups, several launches
    n = 1 << 20
    S = torch.randn(n, device=dev)
    P, M, V = torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = ops.PhiloxState(dev, seed=5).state
    for k in range(3):
        check(lib.vqaf_dp_adamw(ptr(S), None, ptr(step), 1.0, 1.0 / 64, ptr(P), ptr(M), ptr(V), None, None, n,
                                1e-3, 0.9, 0.999, 1e-8, 0.01, ptr(step), 1, st), "adamw")
        assert step[1:3].tolist() == [k + 1, 0]
    # the same three updates with the caller advancing: identical parameters (same draws, same step counts)
    P2, M2, V2 = torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    torch.manual_seed(0)
    P2.copy_(P * 0 + 1.0)
    P3, M3, V3 = P2.clone(), M2.clone(), V2.clone()
    sa, sb = ops.PhiloxState(dev, seed=9).state, ops.PhiloxState(dev, seed=9).state
    for _ in range(3):
        check(lib.vqaf_dp_adamw(ptr(S), None, ptr(sa), 1.0, 1.0 / 64, ptr(P2), ptr(M2), ptr(V2), None, None, n,
                                1e-3, 0.9, 0.999, 1e-8, 0.01, ptr(sa), 1, st), "adamw")
        check(lib.vqaf_dp_adamw(ptr(S), None, ptr(sb), 1.0, 1.0 / 64, ptr(P3), ptr(M3), ptr(V3), None, None, n,
                                1e-3, 0.9, 0.999, 1e-8, 0.01, ptr(sb), 0, st), "adamw")
        check(lib.vqaf_rng_advance(ptr(sb), 1, st), "adv")
    assert torch.equal(P2, P3) and torch.equal(M2, M3) and torch.equal(V2, V3) and torch.equal(sa, sb)
    # ... and with the count advanced BEFORE the update (step_advance = -1: what the clipping launch of a
    # step does on the side): the same draws and bias corrections again
    P4, M4, V4 = torch.ones(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    sc = ops.PhiloxState(dev, seed=9).state
    for _ in range(3):
        check(lib.vqaf_rng_advance(ptr(sc), 1, st), "adv")
        check(lib.vqaf_dp_adamw(ptr(S), None, ptr(sc), 1.0, 1.0 / 64, ptr(P4), ptr(M4), ptr(V4), None, None, n,
                                1e-3, 0.9, 0.999, 1e-8, 0.01, ptr(sc), -1, st), "adamw")
    assert torch.equal(P4, P3) and torch.equal(M4, M3) and torch.equal(V4, V3) and torch.equal(sc, sb)
    # the clipping launch advancing a count on the side: same coefficients, count + 1, nothing else touched
    Bq = 37
    a = torch.randn(Bq, 64, device=dev).bfloat16()
    g = (torch.randn(Bq, 48, device=dev) * 0.1).bfloat16()
    outs = []
    for count in (None, ops.PhiloxState(dev, seed=3).state):
        gg = g.clone()
        coef, norms = torch.empty(Bq, device=dev), torch.empty(Bq, device=dev)
        before = count.clone() if count is not None else None
        pa, pg = (ctypes.c_void_p * 1)(ptr(a)), (ctypes.c_void_p * 1)(ptr(gg))
        din, dout, kind = (ctypes.c_int * 1)(64), (ctypes.c_int * 1)(48), (ctypes.c_int * 1)(1)
        check(lib.vqaf_ghost_clip_zero_bump(1, pa, pg, din, dout, kind, Bq, 1, 1.0, 1.0, 1, ptr(coef), ptr(norms),
                                            None, None, None, 0, ptr(count), st), "ghost_clip_zero_bump")
        outs.append((coef.clone(), norms.clone(), gg))
        if count is not None:
            assert count[1].item() == before[1].item() + 1 and count[0].item() == before[0].item()
            assert count[2:].tolist() == before[2:].tolist()
    assert all(torch.equal(x, y) for x, y in zip(*outs))
    # zero_multi
    buf = torch.ones(10000, device=dev)
    ranges = [buf[3:10], buf[100:1124], buf[5000:5001], buf[6001:9000]]
    zp = (ctypes.c_void_p * 4)(*[ptr(r) for r in ranges])
    zn = (ctypes.c_size_t * 4)(*[r.numel() for r in ranges])
    check(lib.vqaf_zero_multi(zp, zn, 4, st), "zero")
    want = torch.ones(10000)
    for lo, hi in ((3, 10), (100, 1124), (5000, 5001), (6001, 9000)):
        want[lo:hi] = 0
    assert torch.equal(buf.cpu(), want)
    assert lib.vqaf_zero_multi(zp, zn, 9, st) != 0


def test_affine_sums_of_several_layers_in_one_launch(dev):
    """vqaf_clipped_sum_affine_multi against the per-layer entry point (layers of different widths, with and
    without the clip factors, both storage types)."""
    import ctypes
    from vqa_amd._lib import check, dtype_code, load, ptr, stream_ptr
    lib = load()
    B = 200
    for dtype in (torch.float32, torch.bfloat16):
        widths = [256, 1024, 72]
        xs = [gi.normal(300 + i, B, C).to(dtype).to(dev) for i, C in enumerate(widths)]
        gs = [gi.normal(310 + i, B, C).to(dtype).to(dev) for i, C in enumerate(widths)]
        coef = torch.rand(B, generator=torch.Generator().manual_seed(5)).to(dev)
        for cf in (coef, None):
            want = []
            for x, g, C in zip(xs, gs, widths):
                sg, sb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
                check(lib.vqaf_clipped_sum_affine(ptr(x), ptr(g), ptr(cf), B, C, dtype_code(g), 1, ptr(sg), ptr(sb),
                                                  stream_ptr(dev)), "single")
                want.append((sg, sb))
            got = [(torch.zeros(C, device=dev), torch.zeros(C, device=dev)) for C in widths]
            n = len(widths)
            vp = ctypes.c_void_p * n
            check(lib.vqaf_clipped_sum_affine_multi(n, vp(*[ptr(x) for x in xs]), vp(*[ptr(g) for g in gs]), ptr(cf), B,
                                                    (ctypes.c_int * n)(*widths), dtype_code(gs[0]),
                                                    vp(*[ptr(a) for a, _ in got]), vp(*[ptr(b) for _, b in got]),
                                                    stream_ptr(dev)), "multi")
            for (a, b), (wa, wb) in zip(got, want):
                np.testing.assert_allclose(a.cpu().numpy(), wa.cpu().numpy(), rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(b.cpu().numpy(), wb.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_mean_ce_backward_with_a_scaled_incoming_gradient(dev):
    """loss.backward() with a gradient other than the cached unit (a weighted or scaled loss): for bf16
    logits the logits gradient is formed in f32 and rounded ONCE (the hand-over copy that the unit case
    uses is already rounded to bf16 and must not be scaled and rounded again)."""
    from vqa_amd import ops
    B, C = 64, 430
    lg = gi.normal(71, B, C).to(dev)
    y = gi.labels(72, B, C).to(dev)
    want = torch.softmax(lg.bfloat16().double(), -1)
    want[torch.arange(B), y] -= 1.0
    want = want * (2.0 / B)
    for scale in (2.0, torch.tensor(2.0, device=dev)):
        x = lg.bfloat16().requires_grad_()
        loss = ops.cross_entropy_mean(x, y)
        if isinstance(scale, float):
            (loss * scale).backward()
        else:
            loss.backward(gradient=scale)
        once = want.to(torch.bfloat16)  # the f64 gradient rounded once
        diff = (x.grad.double() - once.double()).abs()
        # identical to single rounding except where f32 vs f64 arithmetic lands on the other side of a tie
        assert float((diff > 0).double().mean()) < 0.02
        assert float((diff / (once.double().abs() + 1e-12)).max()) <= 2.0 ** -7
    x = lg.bfloat16().requires_grad_()
    loss = ops.cross_entropy_mean(x, y)
    loss.backward(gradient=ops.unit_gradient(loss))
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), (want / 2).float().cpu().numpy(), rtol=2.0 ** -7, atol=1e-9)


def test_mean_ce_value_in_one_launch(dev):
    """The mean of the per-row losses comes out of the same launch (last workgroup, fixed order): equal to
    the float64 mean within f32 rounding, identical on repeated calls (the ticket word resets itself),
    for batches that fill the last workgroup or not, rows cached in registers (C <= 512) or not."""
    from vqa_amd import ops
    for B, C in ((1, 3), (5, 430), (64, 430), (1023, 430), (130, 1000)):
        lg = gi.normal(B + C, B, C, scale=3.0)
        y = gi.labels(B, B, C)
        want = float(fo.per_sample_ce(lg, y).double().mean())
        got = [float(ops.cross_entropy_mean(lg.to(dev), y.to(dev))) for _ in range(3)]
        assert got[0] == got[1] == got[2]
        assert abs(got[0] - want) <= 2e-6 * max(1.0, abs(want))
        lb = lg.bfloat16()
        wb = float(fo.per_sample_ce(lb.float(), y).double().mean())
        assert abs(float(ops.cross_entropy_mean(lb.to(dev), y.to(dev))) - wb) <= 2e-6 * max(1.0, abs(wb))


def test_linear_groupnorm_in_one_launch_equals_the_two_launches(dev):
    """GroupNorm (+ReLU +Dropout) as the epilogue of the Linear's GEMM, and its backward as the epilogue of
    the next Linear's input-gradient GEMM: outputs, every tape and the input gradient equal the separate
    launches bit for bit -- whole and ragged row tiles, group sizes 64 and 32 (the classifier's) and 8, with
    and without dropout, a head whose width is not a multiple of 8 (the padded loss gradient), and a
    stand-alone norm between two blocks (no fusion across it)."""
    from vqa_amd import dp, ops

    def nets(kind, In, C, groups, p):
        if kind == "two":
            return [nn.Linear(In, C), nn.GroupNorm(groups, C), nn.ReLU(), nn.Dropout(p), nn.Linear(C, 16)]
        if kind == "classifier":
            return [nn.Linear(In, C), nn.GroupNorm(groups, C), nn.ReLU(), nn.Dropout(p),
                    nn.Linear(C, C // 2), nn.GroupNorm(groups, C // 2), nn.ReLU(), nn.Dropout(p), nn.Linear(C // 2, 430)]
        return [nn.Linear(In, C), nn.GroupNorm(groups, C), nn.ReLU(), nn.GroupNorm(groups, C), nn.Linear(C, 64),
                nn.GroupNorm(8, 64), nn.Linear(64, 16)]

    cases = (("two", 256, 1024, 2048, 32, 0.3), ("two", 100, 2048, 1024, 32, 0.3), ("two", 37, 128, 64, 8, 0.0),
             ("two", 64, 64, 128, 8, 0.5), ("classifier", 130, 256, 512, 32, 0.3), ("mixed", 70, 128, 256, 32, 0.0))
    for kind, B, In, C, groups, p in cases:
        outs = []
        for fuse in (False, True):
            torch.manual_seed(5)
            m = nn.Sequential(*nets(kind, In, C, groups, p)).to(dev)
            gsm = dp.GradSampleModule(m)
            gsm.fuse_linear_groupnorm = fuse
            gsm.train()
            x = gi.normal(3, B, In).to(dev).bfloat16().requires_grad_()
            torch.manual_seed(11)  # the dropout stream is seeded from torch's global seed on first use
            y = gsm(x)
            if kind == "classifier":
                tgt = gi.labels(7, B, 430).to(dev)
                loss = ops.cross_entropy_mean(y, tgt)
                loss.backward(gradient=ops.unit_gradient(loss))
            else:
                y.float().square().sum().backward()
            tapes = [(t.a.clone(), t.g.clone()) for t in gsm.tapes]
            outs.append((y.detach().clone(), x.grad.clone(), tapes))
        (y0, gx0, t0), (y1, gx1, t1) = outs
        assert len(t0) == len(t1) == {"two": 3, "classifier": 5, "mixed": 6}[kind]
        assert torch.equal(y0, y1) and torch.equal(gx0, gx1), kind
        assert float(y0.float().abs().sum()) > 0 and float(gx0.float().abs().sum()) > 0
        for (a0, g0), (a1, g1) in zip(t0, t1):
            assert a0.shape == a1.shape and g0.shape == g1.shape
            assert torch.equal(a0, a1) and torch.equal(g0, g1), kind


def test_update_fused_into_the_clipped_sums_equals_the_separate_launches(dev):
    """The Linear layers' clipped sums with the noise + AdamW update as their epilogue (and the normalisation
    layers' update as one launch over their ranges) against the clipped-sum launch followed by the
    whole-bucket update: parameters, moments, final gradients and the bf16 copy identical bit for bit over
    several steps with in-kernel noise (the same generator addressing), at the classifier's shapes and at a
    ragged one; the step count moves once per step."""
    from vqa_amd import dp, ops
    for in_dim, hidden, C, B in ((1024, 2048, 430, 64), (256, 512, 100, 40)):
        x = gi.normal(1, B, in_dim).to(dev).bfloat16()
        y = gi.labels(2, B, C).to(dev)
        runs = []
        for fuse in (False, True):
            torch.manual_seed(3)
            m = _classifier(77, in_dim, hidden, C).to(dev).train()
            gsm = dp.GradSampleModule(m)
            inner = torch.optim.AdamW(m.parameters(), lr=3e-3, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-7)
            opt = dp.DPOptimizer(inner, gsm, noise_multiplier=0.7, max_grad_norm=1.0, expected_batch_size=B, seed=11)
            opt.fuse_update_into_sums = fuse
            used = []
            for step in range(3):
                loss = ops.cross_entropy_mean(gsm(x), y)
                loss.backward(gradient=ops.unit_gradient(loss))
                opt.local_step()
                used.append(bool(opt._updated))
                opt.reduce_step()
                opt.apply_step()
            assert all(used) == fuse and any(used) == fuse
            assert int(opt._step[1].item()) == 3
            spans = [(opt._offsets[id(p)], p.numel()) for p in opt.params]
            runs.append((opt._P.clone(), opt._M.clone(), opt._V.clone(), opt._grad.clone(), opt._shadow.clone()))
        # every parameter's slice (the padding between slices belongs to no parameter: the whole-bucket launch
        # updates it too, the fused one does not)
        for a, b in zip(*runs):
            for o, n in spans:
                assert torch.equal(a[o:o + n], b[o:o + n])
        assert float(runs[0][0].abs().sum()) > 0 and float(runs[0][3].abs().sum()) > 0


def test_f32_to_bf16_rounding_is_torch_s(dev):
    """Every f32 -> bf16 rounding in the library is one hardware conversion (v_cvt_pk_bf16_f32).  Through
    vqaf_scale_cast (packed pairs, and single values in its tail) it must equal torch's rounding bit for bit on
    ties, subnormals, infinities, the largest finite values and random data, and keep NaN a NaN."""
    import struct
    from vqa_amd._lib import BF16, check, load, ptr, stream_ptr
    lib, st = load(), stream_ptr(dev)

    def f(bits):
        return struct.unpack("<f", struct.pack("<I", bits))[0]

    special = [0.0, -0.0, 1.0, -1.0, f(0x3F808000), f(0x3F818000), f(0x3F807FFF), f(0x3F808001),  # ties to even
               f(0x00000001), f(0x00008000), f(0x00018000), f(0x007FFFFF), f(0x80010000),          # subnormals
               f(0x7F7FFFFF), f(0x7F7F7FFF), f(0x7F7F8000), float("inf"), float("-inf"), 65504.0, 1e-40, 3.0e38]
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.cat([torch.tensor(special, device=dev),
                   torch.randn(100003, device=dev, generator=gen) * 10.0 ** torch.randint(-30, 30, (100003,), device=dev,
                                                                                       generator=gen).float()])
    one = torch.ones(1, device=dev)
    out = torch.empty(x.numel(), dtype=torch.bfloat16, device=dev)
    check(lib.vqaf_scale_cast(ptr(x), ptr(one), x.numel(), BF16, ptr(out), st), "vqaf_scale_cast")
    want = x.to(torch.bfloat16)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    nan = torch.tensor([float("nan"), -float("nan"), f(0x7F800001), f(0xFFC12345), 1.0, 2.0, 3.0, 4.0, 5.0], device=dev)
    outn = torch.empty(nan.numel(), dtype=torch.bfloat16, device=dev)
    check(lib.vqaf_scale_cast(ptr(nan), ptr(one), nan.numel(), BF16, ptr(outn), st), "vqaf_scale_cast")
    assert bool(torch.isnan(outn[:4].float()).all()) and outn[4:].float().tolist() == [1.0, 2.0, 3.0, 4.0, 5.0]
