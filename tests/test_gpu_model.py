"""GPU: whole-model checks through the reference's own call shapes -- VQAModel with stub encoders
against the golden logits, the BASELINE configs[1] shape (bf16, batch 64, forward + backward), the
training loop of train.py/utils/src.py with the DP engine, and MIA scoring at configs[4] size."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from oracle import fusion_oracle as fo
from oracle import inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


class _Stub(nn.Module):
    """Encoder stand-in obeying model/model.py:16-17,66-67: returns preset (seq, global)."""
    seq = glob = None

    def __init__(self, out_features=512, hidden_dim=512):
        super().__init__()
        self.output_dim = 512

    def forward(self, *args):
        return type(self).seq, type(self).glob


class _StubV(_Stub):
    pass


class _StubT(_Stub):
    pass


def test_vqamodel_logits_vs_golden(dev):
    from vqa_amd import mia
    from vqa_amd.model import VQAModel
    g = load_golden("model_N197")
    seed, B, N, C = int(g["seed"]), int(g["B"]), int(g["N"]), int(g["C"])
    v_seq, v_global = gi.tokens(seed, B, N)
    q_seq, q_global = gi.tokens(seed + 1, B, 30)
    _StubV.seq, _StubV.glob = v_seq.to(dev), v_global.to(dev)
    _StubT.seq, _StubT.glob = q_seq.to(dev), q_global.to(dev)
    m = VQAModel(vision=_StubV, text=_StubT, fusion_type="co_attention", hidden_dim=1024, num_classes=C,
                 use_token_pruning=True, adversarial_mix=False, noise_injection=False)
    assert sorted(m.state_dict().keys()) == sorted(str(k) for k in g["state_keys"])  # checkpoint-compatible
    m.fusion_module.txt_to_vis_attn.load_state_dict(gi.mha_params(seed + 2))
    m.fusion_module.vis_to_txt_attn.load_state_dict(gi.mha_params(seed + 6))
    sd = m.classifier.state_dict()
    sd.update(gi.classifier_params(seed + 20, 1024, 1024, C))
    m.classifier.load_state_dict(sd)
    m = m.to(dev).eval()
    with torch.no_grad():
        logits = m(images=torch.zeros(B, 3, 8, 8, device=dev), input_ids=torch.zeros(B, 30, dtype=torch.long, device=dev),
                   attention_mask=torch.ones(B, 30, dtype=torch.long, device=dev))
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=1e-4, atol=1e-4)
    loss, conf, pred = mia.score_logits(logits, torch.from_numpy(g["targets"]).to(dev))
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(conf.cpu().numpy(), g["conf"], rtol=1e-3, atol=1e-6)
    assert np.array_equal(pred.cpu().numpy(), g["pred"])


def test_config2_shape_bf16_forward_backward(dev):
    """BASELINE configs[1]: batch 64, bf16, pruning keep 0.5, forward + backward (non-private)."""
    from vqa_amd import fusion
    B, N = 64, 197
    v_seq, v_global = gi.tokens(11, B, N)
    q_seq, q_global = gi.tokens(12, B, 30)
    t2v, v2t = gi.mha_params(13), gi.mha_params(17)
    cot = gi.normal(21, B, 1024)
    rb = lambda t: t.bfloat16().float()  # noqa: E731  (the oracle sees the same bf16-rounded values)
    vr, qr = rb(v_seq).requires_grad_(), rb(q_seq).requires_grad_()
    want, v_idx, q_idx = fo.co_attention_pruned(
        vr, rb(v_global), qr, rb(q_global), (rb(t2v["in_proj_weight"]), rb(t2v["in_proj_bias"])),
        (rb(v2t["in_proj_weight"]), rb(v2t["in_proj_bias"])), adversarial_mix=False)
    (want * cot).sum().backward()
    m = fusion.CoAttention(use_pruning=True, adversarial_mix=False, noise_injection=False)
    m.txt_to_vis_attn.load_state_dict(t2v)
    m.vis_to_txt_attn.load_state_dict(v2t)
    m = m.to(dev).to(torch.bfloat16)
    vs, qs = v_seq.to(dev).bfloat16().requires_grad_(), q_seq.to(dev).bfloat16().requires_grad_()
    fused = m(vs, v_global.to(dev).bfloat16(), qs, q_global.to(dev).bfloat16())
    assert fused.dtype == torch.bfloat16
    (fused.float() * cot.to(dev)).sum().backward()
    # keep sets: bf16 scores are f32-accurate here (tokens exact, folded query in two bf16 terms)
    # The kernels fold the query into the key projection (x . (Wk^T q)) where the reference projects
    # every token first ((Wk x) . q): the same number in exact arithmetic, a few ulps apart in f32.
    # A keep set may therefore differ from the oracle's only by tokens whose weight ties with the
    # weight at the keep boundary to within that rounding; anything else is a failure.
    w_or = fo.attention_weights(rb(q_global), rb(v_seq), rb(t2v["in_proj_weight"]), rb(t2v["in_proj_bias"]), 8)
    w_or = w_or.reshape(B, N)
    K = v_idx.shape[1]
    for b_, (a, o) in enumerate(zip(m.v_pruning.last_indices.cpu(), v_idx)):
        diff = set(a.tolist()) ^ set(o.tolist())
        if diff:
            boundary = w_or[b_].sort(descending=True).values[K - 1:K + 1].mean()
            gaps = (w_or[b_][sorted(diff)] - boundary).abs() / boundary
            assert len(diff) == 2 and float(gaps.max()) < 1e-4, (b_, sorted(diff), gaps.tolist())
    np.testing.assert_allclose(fused.float().detach().cpu().numpy(), want.detach().numpy(), rtol=1e-2, atol=1e-2)
    gv = vs.grad.float().cpu()
    ok = (m.v_pruning.last_indices.cpu().sort(1).values == v_idx.sort(1).values).all(dim=1)
    np.testing.assert_allclose(gv[ok].numpy(), vr.grad[ok].numpy(), rtol=1e-2, atol=1e-4)


def test_training_loop_with_dp_engine_and_concat_free_path(dev):
    """train.py:107-164 + utils/src.py:60-121 order on a tiny synthetic dataset, through
    vqa_amd.loops.train with the {'main','classifier'} optimizer dict."""
    from vqa_amd import dp, loops
    from vqa_amd.model import VQAModel

    class TinyV(nn.Module):
        def __init__(self, out_features=512):
            super().__init__()
            self.output_dim = out_features
            self.proj = nn.Linear(48, out_features)

        def forward(self, images):
            seq = self.proj(images.reshape(images.shape[0], 16, 48))
            return seq, seq.mean(1)

    class TinyT(nn.Module):
        def __init__(self, hidden_dim=512):
            super().__init__()
            self.output_dim = hidden_dim
            self.emb = nn.Embedding(100, hidden_dim)

        def forward(self, ids, mask):
            seq = self.emb(ids)
            return seq, seq.mean(1)

    torch.manual_seed(42)
    model = VQAModel(vision=TinyV, text=TinyT, fusion_type="co_attention", hidden_dim=64, num_classes=13,
                     use_token_pruning=True)
    model.classifier = dp.ModuleValidator.fix(model.classifier)
    model = model.to(dev)
    data = [{"image": torch.randn(3, 16, 16), "inputs": {"input_ids": torch.randint(0, 100, (12,)),
                                                       "attention_mask": torch.ones(12, dtype=torch.long)},
             "answer": torch.tensor(i % 13)} for i in range(96)]

    def collate(b):
        return {"image": torch.stack([x["image"] for x in b]),
                "inputs": {"input_ids": torch.stack([x["inputs"]["input_ids"] for x in b]),
                           "attention_mask": torch.stack([x["inputs"]["attention_mask"] for x in b])},
                "answer": torch.stack([x["answer"] for x in b])}

    loader = torch.utils.data.DataLoader(data, batch_size=16, shuffle=True, collate_fn=collate)
    engine = dp.PrivacyEngine(secure_mode=False)
    cls_opt = torch.optim.AdamW(model.classifier.parameters(), lr=1e-3, weight_decay=0.01)
    model.classifier, cls_opt, loader = engine.make_private(
        module=model.classifier, optimizer=cls_opt, data_loader=loader, noise_multiplier=1.0,
        max_grad_norm=10.0, poisson_sampling=True)
    main_params = [p for n, p in model.named_parameters() if not n.startswith('classifier.') and p.requires_grad]
    optimizer = {'main': torch.optim.AdamW(main_params, lr=1e-3), 'classifier': cls_opt}
    w0 = model.vision_encoder.proj.weight.detach().clone()
    c0 = next(model.classifier.parameters()).detach().clone()
    loss, acc = loops.train(model, loader, optimizer, nn.CrossEntropyLoss(), dev, use_dp_sgd=True)
    assert np.isfinite(loss) and 0.0 <= acc <= 100.0
    assert not torch.equal(w0, model.vision_encoder.proj.weight)       # encoder trained non-privately
    assert not torch.equal(c0, next(model.classifier.parameters()))     # classifier trained by the DP step
    assert engine.get_epsilon(1e-5) > 0
    vloss, vacc, extra = loops.validate(model, torch.utils.data.DataLoader(data, batch_size=32, collate_fn=collate),
                                        nn.CrossEntropyLoss(), dev)
    assert np.isfinite(vloss) and extra["preds"].shape == (96,)
    assert any(k.startswith("classifier._module.") for k in model.state_dict())


def test_mia_scoring_at_config5_shard_size(dev):
    """configs[4]: 25 000 samples per GPU of the 200k-sample MIA pass, loss and confidence."""
    from vqa_amd import mia
    B, C = 25000, 430
    gen = torch.Generator(device=dev).manual_seed(3)
    logits = torch.randn(B, C, device=dev, generator=gen) * 3
    y = torch.randint(0, C, (B,), device=dev, generator=gen)
    loss, conf, pred = mia.score_logits(logits, y)
    ref_loss = nn.functional.cross_entropy(logits.double(), y, reduction='none')
    ref_conf, ref_pred = torch.softmax(logits.double(), 1).max(1)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(conf.cpu().numpy(), ref_conf.cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert torch.equal(pred, ref_pred)


@pytest.mark.parametrize("name,tokens", [("CNN", 64), ("ResNet50", 4), ("SwinTransformer", 49)])
def test_vision_wrappers_feed_the_fusion_kernels(dev, name, tokens):
    """The reference's vision encoders (model/vision_encoder.py:8-127; its configurations pair ResNet50 with
    Bert) in front of CoAttention with pruning: forward + backward of the whole VQAModel on the GPU in bf16,
    token count and gradients as expected, the pruning kernels fed by the encoder's token layout as it comes
    (channels-last feature maps are contiguous tokens; the 1x1 projection of ResNet50 is a strided view)."""
    import functools
    from vqa_amd import encoders, vision
    from vqa_amd.model import VQAModel
    torch.manual_seed(11)
    vis = vision.get_vision_registry()[name]
    text = functools.partial(encoders.Bert, config=encoders.BertConfig(
        hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=200))
    model = VQAModel(vision=vis, text=text, fusion_type="co_attention", hidden_dim=64, num_classes=13,
                     use_token_pruning=True, prune_ratio=0.5).to(dev).to(torch.bfloat16)
    B = 4
    side = 224 if name == "SwinTransformer" else 64  # (the default Swin-T configuration takes 224 x 224 pixels)
    images = torch.randn(B, 3, side, side, device=dev, dtype=torch.bfloat16)
    ids = torch.randint(0, 200, (B, 12), device=dev)
    mask = torch.ones(B, 12, dtype=torch.long, device=dev)
    seen = {}

    def note(module, args):
        seen["v_seq"] = args[0].shape

    hook = model.fusion_module.register_forward_pre_hook(note)
    logits = model(images, ids, mask)
    hook.remove()
    assert seen["v_seq"] == (B, tokens, 512)
    assert logits.shape == (B, 13) and torch.isfinite(logits.float()).all()
    logits.float().logsumexp(1).sum().backward()
    first = next(model.vision_encoder.parameters())
    assert first.grad is not None and torch.isfinite(first.grad.float()).all() and float(first.grad.float().abs().sum()) > 0
