"""GPU: encoder kernels (attention, LayerNorm+residual, GELU) and the ViT / BERT stacks against the oracle.

Tolerances: the kernels take bf16 operands and accumulate in f32; probabilities and score gradients are
rounded to bf16 before their second product (as the tensors of an unfused bf16 implementation would be), so
results are compared with f64 math on the SAME bf16 operands at bf16 resolution (rtol 1e-2 of the tensor's
scale).  LayerNorm and GELU in f32 storage are held to rtol 1e-5.
"""
import pytest
import torch

from oracle import encoder_oracle as eo
from vqa_amd import encoders

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b).norm() / b.norm().clamp_min(1e-30)).detach())


def _maxerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).detach())


def _qkv(B, N, H, seed, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, N, 3 * H * 64, generator=g) * spread).to(torch.bfloat16)


def _split(qkv, H):
    B, N, _ = qkv.shape
    q, k, v = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    return q, k, v


@pytest.mark.parametrize("B,H,N", [(2, 3, 5), (3, 2, 30), (2, 2, 32), (1, 3, 64), (2, 2, 65), (2, 12, 197), (1, 2, 200),
                                   (1, 3, 577)])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_forward_and_backward(B, H, N, masked):
    qkv = _qkv(B, N, H, seed=N + 7 * H, spread=1.5)
    bias = None
    if masked:
        keep = torch.ones(B, N)
        for b in range(B):
            keep[b, max(1, N - 1 - 3 * b - N // 3):] = 0
        bias = (1 - keep) * torch.finfo(torch.float32).min
    x = qkv.to(DEV).requires_grad_()
    out = encoders.self_attention(x, None if bias is None else bias.to(DEV), H)
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(B, N, H * 64, generator=g).to(torch.bfloat16)
    out.backward(cot.to(DEV))

    ref_in = qkv.double().requires_grad_()
    q, k, v = _split(ref_in, H)
    ref = eo.attention_reference(q, k, v, bias, 0.125).transpose(1, 2).reshape(B, N, H * 64)
    ref.backward(cot.double())
    assert _rel(out, ref) < 6e-3 and _maxerr(out, ref) < 2e-2
    gq, gk, gv = (t.reshape(B, N, H * 64) for t in x.grad.view(B, N, 3, H * 64).unbind(2))
    rq, rk, rv = (t.reshape(B, N, H * 64) for t in ref_in.grad.view(B, N, 3, H * 64).unbind(2))
    for name, a, r in (("dq", gq, rq), ("dk", gk, rk), ("dv", gv, rv)):
        assert _rel(a, r) < 1.2e-2, (name, _rel(a, r))
        assert _maxerr(a, r) < 3e-2, (name, _maxerr(a, r))
    if masked:  # masked keys receive no gradient
        dead = (keep == 0)
        assert float(gk[dead].abs().max()) == 0.0 and float(gv[dead].abs().max()) == 0.0


def _host_keep_scale(seed, counter, B, H, N, p):
    """mask / (1 - p) of every (b,h,n,m) from the host restatement of the generator."""
    import numpy as np
    import philox_np
    groups = (N + 3) // 4
    calls = np.arange(B * H * N * groups, dtype=np.uint64)
    words = philox_np.philox4x32_10(seed, calls, counter).reshape(B, H, N, groups * 4)[..., :N]
    thr = min(int(p * 4294967296.0), 4294967295)
    return torch.from_numpy((words >= np.uint32(thr)).astype(np.float64) / (1.0 - p))


@pytest.mark.parametrize("B,H,N,p", [(2, 2, 30, 0.1), (1, 3, 64, 0.25), (2, 2, 197, 0.1)])
def test_attention_dropout_mask_values_and_gradients(B, H, N, p):
    """The mask is the documented function of (seed, counter, b, h, n, m): the forward applies it after the
    softmax, the backward regenerates it; the state's counter moves by one per call."""
    from vqa_amd import ops
    rng = ops.PhiloxState(DEV, seed=1234 + N)
    rng.advance(3)
    qkv = _qkv(B, N, H, seed=N, spread=1.2)
    x = qkv.to(DEV).requires_grad_()
    out = encoders.self_attention(x, None, H, p, rng)
    assert rng.state.cpu().tolist()[1] == 4
    cot = torch.randn(B, N, H * 64, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16)
    out.backward(cot.to(DEV))
    keep = _host_keep_scale(rng.seed, 3, B, H, N, p)
    assert abs(float((keep > 0).double().mean()) - (1 - p)) < 0.02
    ref_in = qkv.double().requires_grad_()
    q, k, v = _split(ref_in, H)
    probs = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1) * keep
    ref = (probs @ v).transpose(1, 2).reshape(B, N, H * 64)
    ref.backward(cot.double())
    assert _rel(out, ref) < 6e-3, _rel(out, ref)
    assert _rel(x.grad, ref_in.grad) < 1.2e-2, _rel(x.grad, ref_in.grad)
    # a second call draws a different mask, p = 0 none at all
    out2 = encoders.self_attention(x.detach(), None, H, p, rng)
    assert _rel(out2, ref) > 0.05
    out0 = encoders.self_attention(x.detach(), None, H, 0.0, None)
    clean = (torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1) @ v).transpose(1, 2).reshape(B, N, H * 64)
    assert _rel(out0, clean) < 6e-3


def test_bert_train_mode_applies_both_dropouts_and_eval_does_not():
    torch.manual_seed(4)
    cfg = encoders.BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                              vocab_size=101, max_position_embeddings=40)
    m = encoders.BertModel(cfg).to(DEV)
    ids = torch.randint(1, 101, (3, 12), device=DEV)
    mask = torch.ones(3, 12, dtype=torch.long, device=DEV)
    m.eval()
    with torch.no_grad():
        e1, _ = m(ids, mask)
        e2, _ = m(ids, mask)
    assert torch.equal(e1, e2)
    m.train()
    t1, _ = m(ids, mask)
    t2, pooled = m(ids, mask)
    assert not torch.equal(t1, t2)
    assert m._attn_rng.state.cpu().tolist()[1] == 4  # two layers, two calls
    (t2.float().sum() + pooled.float().sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_attention_fully_masked_row_is_uniform_and_finite():
    B, H, N = 2, 2, 9
    qkv = _qkv(B, N, H, seed=1).to(DEV)
    bias = torch.zeros(B, N, device=DEV)
    bias[1] = torch.finfo(torch.float32).min
    out = encoders.self_attention(qkv, bias, H)
    assert torch.isfinite(out.float()).all()
    v = qkv.view(B, N, 3, H * 64)[1, :, 2].float().mean(0)
    torch.testing.assert_close(out[1, 0].float(), v, rtol=2e-2, atol=2e-2)


def test_attention_rejects_what_it_does_not_support():
    with pytest.raises(TypeError):
        encoders.self_attention(torch.zeros(1, 4, 3 * 64, device=DEV), None, 1)
    with pytest.raises(ValueError):
        encoders.self_attention(torch.zeros(1, 4, 3 * 96, device=DEV, dtype=torch.bfloat16), None, 3)


@pytest.mark.parametrize("rows,D", [(5, 128), (37, 768), (64, 1024), (9, 2048), (130, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("residual", [False, True])
def test_add_layer_norm(rows, D, dtype, residual):
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.3).to(dtype)
    r = (torch.randn(rows, D, generator=g)).to(dtype) if residual else None
    w = torch.randn(D, generator=g)
    b = torch.randn(D, generator=g)
    cy = torch.randn(rows, D, generator=g).to(dtype)
    cs = torch.randn(rows, D, generator=g).to(dtype)

    xd = x.to(DEV).requires_grad_()
    rd = None if r is None else r.to(DEV).requires_grad_()
    wd, bd = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y, s = encoders.add_layer_norm(xd, rd, wd, bd, 1e-12)
    loss_terms = (y.float() * cy.to(DEV).float()).sum()
    if residual:
        loss_terms = loss_terms + (s.float() * cs.to(DEV).float()).sum()
    loss_terms.backward()

    x64 = x.double().requires_grad_()
    r64 = None if r is None else r.double().requires_grad_()
    w64, b64 = w.double().requires_grad_(), b.double().requires_grad_()
    s64 = x64 if r64 is None else x64 + r64
    if residual and dtype == torch.bfloat16:
        # the kernel normalises the rounded sum it hands on; a straight-through rounding keeps the gradient
        s64 = s64 + (s64.detach().to(torch.bfloat16).double() - s64.detach())
    y64 = torch.nn.functional.layer_norm(s64, (D,), w64, b64, 1e-12)
    l64 = (y64 * cy.double()).sum()
    if residual:
        l64 = l64 + (s64 * cs.double()).sum()
    l64.backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert _maxerr(y, y64) < tol
    if residual:
        assert _maxerr(s, s64) < tol
        assert _maxerr(rd.grad, r64.grad) < tol
    assert _maxerr(xd.grad, x64.grad) < tol
    assert _maxerr(wd.grad, w64.grad) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert _maxerr(bd.grad, b64.grad) < (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gelu(dtype):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(33, 3072, generator=g) * 3).to(dtype)
    c = torch.randn(33, 3072, generator=g).to(dtype)
    xd = x.to(DEV).requires_grad_()
    y = encoders.gelu(xd)
    y.backward(c.to(DEV))
    x64 = x.double().requires_grad_()
    y64 = torch.nn.functional.gelu(x64)
    y64.backward(c.double())
    tol = 2e-6 if dtype == torch.float32 else 4e-3
    assert float((y.double().cpu() - y64).abs().max()) < tol * 10
    assert float((xd.grad.double().cpu() - x64.grad).abs().max()) < tol * 10


def _grad_check(model, oracle_out, mine_out, names, sd64):
    cot = [torch.randn(o.shape, generator=torch.Generator().manual_seed(11 + i)).double() for i, o in enumerate(oracle_out)]
    sum((o * c).sum() for o, c in zip(oracle_out, cot)).backward()
    sum((o.double() * c.to(DEV)).sum() for o, c in zip(mine_out, cot)).backward()
    params = dict(model.named_parameters())
    for n in names:
        assert _rel(params[n].grad, sd64[n].grad) < 3e-2, (n, _rel(params[n].grad, sd64[n].grad))


def test_vit_stack_matches_oracle_forward_and_gradients():
    torch.manual_seed(0)
    cfg = encoders.ViTConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                             image_size=80)
    m = encoders.ViTModel(cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.08)
    sd64 = {k: v.detach().double().requires_grad_() for k, v in m.state_dict().items()}
    m.to(DEV)
    x = torch.randn(3, 3, 80, 80)
    h, pooled = m(x.to(DEV))
    rh, rp = eo.vit_forward(sd64, x.double(), heads=2, patch=16)
    assert h.dtype == torch.bfloat16 and h.shape == (3, 26, 128)
    assert _rel(h, rh) < 1e-2 and _rel(pooled, rp) < 1e-2
    _grad_check(m, (rh, rp), (h, pooled),
                ["embeddings.cls_token", "embeddings.patch_embeddings.projection.weight",
                 "encoder.layer.0.attention.attention.query.weight", "encoder.layer.1.attention.attention.value.bias",
                 "encoder.layer.0.layernorm_before.weight", "encoder.layer.2.layernorm_after.bias",
                 "encoder.layer.1.intermediate.dense.weight", "encoder.layer.2.output.dense.weight",
                 "layernorm.weight", "pooler.dense.weight"], sd64)


def test_bert_stack_matches_oracle_forward_and_gradients_with_padding():
    torch.manual_seed(1)
    cfg = encoders.BertConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                              vocab_size=101, max_position_embeddings=40, hidden_dropout_prob=0.0,
                              attention_probs_dropout_prob=0.0)
    m = encoders.BertModel(cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.08)
    sd64 = {k: v.detach().double().requires_grad_() for k, v in m.state_dict().items()}
    m.to(DEV)
    ids = torch.randint(1, 101, (4, 30))
    mask = torch.ones(4, 30, dtype=torch.long)
    mask[0, 12:] = 0
    mask[2, 25:] = 0
    h, pooled = m(ids.to(DEV), mask.to(DEV))
    rh, rp = eo.bert_forward(sd64, ids, mask, heads=2)
    live = mask.bool()
    assert _rel(h[live.to(DEV)], rh[live]) < 1e-2 and _rel(pooled, rp) < 1e-2
    # gradients through the attended positions only (padding rows are arbitrary in both implementations)
    cot = torch.randn(rh.shape, generator=torch.Generator().manual_seed(2)).double() * live[..., None]
    ((rh * cot).sum() + rp.sum()).backward()
    ((h.double() * cot.to(DEV)).sum() + pooled.double().sum()).backward()
    params = dict(m.named_parameters())
    for n in ["embeddings.word_embeddings.weight", "embeddings.LayerNorm.weight",
              "encoder.layer.0.attention.self.key.weight", "encoder.layer.2.attention.self.query.weight",
              "encoder.layer.1.attention.output.LayerNorm.weight", "encoder.layer.1.intermediate.dense.bias",
              "encoder.layer.2.output.dense.weight", "encoder.layer.0.output.LayerNorm.bias", "pooler.dense.weight"]:
        # query/key gradients are differences of nearly equal terms (the rows of ds sum to zero): the bf16
        # rounding of the layer's inputs shows up amplified there
        tol = 8e-2 if (".query." in n or ".key." in n) else 3e-2
        assert _rel(params[n].grad, sd64[n].grad) < tol, (n, _rel(params[n].grad, sd64[n].grad))


def test_full_size_encoders_batch8_against_oracle():
    """ViT-B/16 at 224 px and BERT-base at 30 tokens, batch 8, random weights, against the f32 oracle on the
    host: the reference's CPU-runnable configuration of the benchmark."""
    torch.manual_seed(2)
    v = encoders.ViT(out_features=512).eval()
    t = encoders.Bert(hidden_dim=512).eval()
    sdv = {k: p.detach().clone() for k, p in v.vit.state_dict().items()}
    sdt = {k: p.detach().clone() for k, p in t.bert.state_dict().items()}
    imgs = torch.randn(8, 3, 224, 224)
    ids = torch.randint(1, 30000, (8, 30))
    mask = torch.ones(8, 30, dtype=torch.long)
    mask[3, 17:] = 0
    with torch.no_grad():
        rh, rp = eo.vit_forward(sdv, imgs, heads=12, patch=16)
        rv_seq = torch.nn.functional.linear(rh, v.seq_projection.weight, v.seq_projection.bias)
        rv_glob = torch.nn.functional.linear(rp, v.global_projection.weight, v.global_projection.bias)
        th, tp = eo.bert_forward(sdt, ids, mask, heads=12)
        rq_seq = torch.nn.functional.linear(th, t.seq_projection.weight, t.seq_projection.bias)
        rq_glob = torch.nn.functional.linear(tp, t.global_projection.weight, t.global_projection.bias)
        v.to(DEV)
        t.to(DEV)
        v_seq, v_glob = v(imgs.to(DEV))
        q_seq, q_glob = t(ids.to(DEV), mask.to(DEV))
    assert v_seq.shape == (8, 197, 512) and v_glob.shape == (8, 512)
    assert q_seq.shape == (8, 30, 512) and q_glob.shape == (8, 512)
    assert _rel(v_seq, rv_seq) < 1.5e-2 and _rel(v_glob, rv_glob) < 1.5e-2
    live = mask.bool()
    assert _rel(q_seq[live.to(DEV)], rq_seq[live]) < 1.5e-2 and _rel(q_glob, rq_glob) < 1.5e-2


@pytest.mark.parametrize("pruning", [False, True])
def test_whole_model_with_encoders_matches_the_composed_oracle(pruning):
    """VQAModel(ViT, Bert, co_attention) in eval mode, batch 8, 224 px / 30 tokens, against the host composition
    encoder oracle -> projections -> fusion oracle -> classifier (all f32).  The pruning variant may differ in
    a keep index where two attention weights tie within bf16 noise; the logits are compared, with a looser bar."""
    from oracle import fusion_oracle as fo
    from vqa_amd import model as vm

    def vit_cls(out_features=512):
        return encoders.ViT(out_features=out_features, config=encoders.ViTConfig(
            hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512, image_size=224))

    def bert_cls(hidden_dim=512):
        return encoders.Bert(hidden_dim=hidden_dim, config=encoders.BertConfig(
            hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512, vocab_size=500,
            max_position_embeddings=64))

    torch.manual_seed(7)
    m = vm.VQAModel(vit_cls, bert_cls, fusion_type="co_attention", num_classes=13, use_token_pruning=pruning,
                    prune_ratio=0.5, adversarial_mix=False, noise_injection=False).eval()
    with torch.no_grad():
        for p in m.vision_encoder.vit.parameters():
            p.copy_(torch.randn_like(p) * 0.06)
        for p in m.text_encoder.bert.parameters():
            p.copy_(torch.randn_like(p) * 0.06)
    imgs = torch.randn(8, 3, 224, 224)
    ids = torch.randint(1, 500, (8, 30))
    mask = torch.ones(8, 30, dtype=torch.long)
    F = torch.nn.functional
    with torch.no_grad():
        ve, te, fm = m.vision_encoder, m.text_encoder, m.fusion_module
        h, pooled = eo.vit_forward(ve.vit.state_dict(), imgs, heads=4, patch=16)
        v_seq, v_glob = F.linear(h, ve.seq_projection.weight, ve.seq_projection.bias), \
            F.linear(pooled, ve.global_projection.weight, ve.global_projection.bias)
        h, pooled = eo.bert_forward(te.bert.state_dict(), ids, mask, heads=4)
        q_seq, q_glob = F.linear(h, te.seq_projection.weight, te.seq_projection.bias), \
            F.linear(pooled, te.global_projection.weight, te.global_projection.bias)
        t2v, v2t = fm.txt_to_vis_attn, fm.vis_to_txt_attn
        if pruning:
            fused, _, _ = fo.co_attention_pruned(v_seq, v_glob, q_seq, q_glob, (t2v.in_proj_weight, t2v.in_proj_bias),
                                                 (v2t.in_proj_weight, v2t.in_proj_bias), adversarial_mix=False)
        else:
            fused = fo.co_attention_dense(v_seq, v_glob, q_seq, q_glob,
                                          (t2v.in_proj_weight, t2v.in_proj_bias, t2v.out_proj.weight, t2v.out_proj.bias),
                                          (v2t.in_proj_weight, v2t.in_proj_bias, v2t.out_proj.weight, v2t.out_proj.bias))
        want = m.classifier(fused)
        m.to(DEV)
        got = m(imgs.to(DEV), ids.to(DEV), mask.to(DEV))
    assert got.shape == (8, 13)
    assert _rel(got, want) < (6e-2 if pruning else 3e-2), _rel(got, want)


def test_roberta_stack_matches_oracle_with_padding():
    torch.manual_seed(3)
    cfg = encoders.roberta_config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                  vocab_size=101, max_position_embeddings=40)
    m = encoders.BertModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.08)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ids = torch.randint(2, 101, (3, 20))
    mask = torch.ones(3, 20, dtype=torch.long)
    mask[2, 9:] = 0
    ids[2, 9:] = 1
    with torch.no_grad():
        rh, rp = eo.bert_forward(sd, ids, mask, heads=2, eps=1e-5, roberta_pad=1)
        h, pooled = m.to(DEV)(ids.to(DEV), mask.to(DEV))
    live = mask.bool()
    assert _rel(h[live.to(DEV)], rh[live]) < 1e-2 and _rel(pooled, rp) < 1e-2


@pytest.mark.parametrize("wrapper,targets", [("BertQLoRA", None), ("RoBertaQLoRA", ["query", "key", "value"])])
def test_lora_text_encoders_match_the_oracle_on_merged_weights(wrapper, targets):
    """BertQLoRA / RoBertaQLoRA (model/text_encoder.py:91-169; the peft package is absent, so parity with it is
    unpinned): with dropout off the adapted stack computes what the plain stack computes on W + (alpha / r) B A,
    which the oracle evaluates in float64 as a function of A and B; only adapters and projections train."""
    torch.manual_seed(3)
    roberta = wrapper == "RoBertaQLoRA"
    cfg = (encoders.roberta_config if roberta else encoders.BertConfig)(
        hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=101,
        max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = getattr(encoders, wrapper)(hidden_dim=64, lora_r=8, lora_alpha=16, lora_dropout=0.0, target_modules=targets,
                                   config=cfg)
    with torch.no_grad():
        for p in m.bert.parameters():
            p.copy_(torch.randn_like(p) * 0.08)   # (the adapters' B matrices too: zeros would hide them)
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert all(".lora." in n or n.startswith(("seq_projection", "global_projection")) for n in trainable)
    tset = targets or ["query", "value"]
    assert sum(".lora_A." in n for n in trainable) == 2 * len(tset)
    stack = m.bert
    sd64 = {k: v.detach().double() for k, v in stack.state_dict().items() if ".lora." not in k}
    ab = {}
    for i, lyr in enumerate(stack.encoder.layer):
        for t in lyr.lora.targets:
            pair = lyr.lora.adapters[t]
            a = pair.lora_A.weight.detach().double().requires_grad_()
            b = pair.lora_B.weight.detach().double().requires_grad_()
            ab[(i, t)] = (a, b)
            key = f"encoder.layer.{i}.attention.self.{t}.weight"
            sd64[key] = sd64[key] + lyr.lora.scaling * b @ a
    # the helper that exports merged weights agrees with that formula
    merged = encoders.merged_lora_weights(stack)
    for k, w in merged.items():
        assert _rel(w, sd64[k]) < 1e-6
    m.to(DEV).train()
    ids = torch.randint(2, 101, (4, 30))
    mask = torch.ones(4, 30, dtype=torch.long)
    mask[1, 20:] = 0
    if roberta:
        ids[1, 20:] = 1
    h, pooled = stack(ids.to(DEV), mask.to(DEV))
    rh, rp = eo.bert_forward(sd64, ids, mask, heads=2, eps=cfg.layer_norm_eps, roberta_pad=1 if roberta else None)
    live = mask.bool()
    assert _rel(h[live.to(DEV)], rh[live]) < 1e-2 and _rel(pooled, rp) < 1e-2
    cot = torch.randn(rh.shape, generator=torch.Generator().manual_seed(2)).double() * live[..., None]
    ((rh * cot).sum() + rp.sum()).backward()
    ((h.double() * cot.to(DEV)).sum() + pooled.double().sum()).backward()
    for i, lyr in enumerate(stack.encoder.layer):
        for t in lyr.lora.targets:
            pair = lyr.lora.adapters[t]
            a, b = ab[(i, t)]
            tol = 8e-2 if t in ("query", "key") else 3e-2
            assert _rel(pair.lora_A.weight.grad, a.grad) < tol, (i, t, "A", _rel(pair.lora_A.weight.grad, a.grad))
            assert _rel(pair.lora_B.weight.grad, b.grad) < tol, (i, t, "B", _rel(pair.lora_B.weight.grad, b.grad))
    assert all(p.grad is None for n, p in stack.named_parameters() if ".lora." not in n)
    # the wrapper's contract and a step of dropout-on training (masks drawn, shapes kept)
    q_seq, q_global = m(ids.to(DEV), mask.to(DEV))
    assert q_seq.shape == (4, 30, 64) and q_global.shape == (4, 64)
    for lyr in stack.encoder.layer:
        lyr.lora.p = 0.1
    q2, _ = m(ids.to(DEV), mask.to(DEV))
    assert torch.isfinite(q2.float()).all() and not torch.equal(q2, q_seq)
