"""GPU: the training step replayed as HIP graphs (vqa_amd.loops.GraphedStep, loops.train(capture=True))
against the same epochs launched eagerly: same parameters, same accounted steps; batches of varying size
padded onto one graph."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


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


def _batches(sizes):
    g = torch.Generator().manual_seed(11)
    out = []
    for b in sizes:
        out.append({"image": torch.randn(b, 3, 16, 16, generator=g),
                    "inputs": {"input_ids": torch.randint(0, 100, (b, 12), generator=g),
                               "attention_mask": torch.ones(b, 12, dtype=torch.long)},
                    "answer": torch.randint(0, 13, (b,), generator=g)})
    return out


def _run(dev, sizes, capture, pad_to=0, epochs=2, sgd=False, fusion_type="co_attention"):
    from vqa_amd import dp, loops, ops
    from vqa_amd.model import VQAModel
    torch.manual_seed(42)
    model = VQAModel(vision=TinyV, text=TinyT, fusion_type=fusion_type, hidden_dim=64, num_classes=13,
                     use_token_pruning=True, adversarial_mix=True, noise_injection=False)
    model.classifier = dp.ModuleValidator.fix(model.classifier)
    model = model.to(dev)
    rng = ops.PhiloxState(dev, seed=5)
    if fusion_type == "co_attention":
        model.fusion_module.v_pruning.rng = model.fusion_module.q_pruning.rng = rng
    gsm = dp.GradSampleModule(model.classifier)
    model.classifier = gsm
    steps = []
    inner = torch.optim.SGD(gsm.parameters(), lr=0.01) if sgd else torch.optim.AdamW(gsm.parameters(), lr=1e-3, weight_decay=0.01)
    cls_opt = dp.DPOptimizer(inner, gsm, noise_multiplier=0.0, max_grad_norm=1.0, expected_batch_size=16, seed=3)
    cls_opt.attach_step_hook(lambda _o: steps.append(1))
    main_params = [p for n, p in model.named_parameters() if not n.startswith('classifier.') and p.requires_grad]
    main = torch.optim.SGD(main_params, lr=0.01) if sgd else torch.optim.AdamW(main_params, lr=1e-3, capturable=True)
    optimizer = {'main': main, 'classifier': cls_opt}
    out = []
    for _ in range(epochs):
        out.append(loops.train(model, _batches(sizes), optimizer, nn.CrossEntropyLoss(), dev, use_dp_sgd=True,
                               capture=capture, pad_to=pad_to))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1).float() for p in model.parameters()]).cpu()
    launch = model._vqaf_graphed.launch if capture else "eager"
    n_graphs = sum(1 for e in model._vqaf_graphed._sigs.values() if e["graph"] is not None) if capture else 0
    return params, out, len(steps), launch, n_graphs, int(rng.state[1].item())


def test_graphed_epochs_equal_eager_epochs():
    dev = torch.device("cuda:0")
    sizes = [16] * 6
    eager = _run(dev, sizes, capture=False)
    graphed = _run(dev, sizes, capture=True)
    assert graphed[3] == "hipgraph" and graphed[4] == 1
    assert eager[2] == graphed[2] == 12          # the accountant saw every step, replayed or not
    assert eager[5] == graphed[5]                # the same number of random-stream positions consumed
    # same kernels in the same order; only the few f32 atomics of the affine sums may reorder, and Adam
    # turns a last-bit difference of a near-zero gradient into a fraction of one lr-sized step
    np.testing.assert_allclose(graphed[0].numpy(), eager[0].numpy(), rtol=1e-3, atol=2e-4)
    for (le, ae), (lg, ag) in zip(eager[1], graphed[1]):
        assert abs(le - lg) < 1e-3 and abs(ae - ag) < 1e-6 + 100.0 / 96


@pytest.mark.parametrize("fusion_type,sizes,epochs", [("concat", [16, 11, 16, 7, 13, 16, 9], 1),
                                                       ("co_attention", [16, 11, 7, 13], 1)])
def test_varying_batches_share_one_padded_graph(fusion_type, sizes, epochs):
    """Plain SGD on both sides: the padded step forms the same sums in another order (sum / padded rows,
    times padded / real), and Adam would turn those last bits into lr-sized steps wherever a gradient
    is ~0.  Few steps are compared: training this tiny model amplifies a last-bit difference several times
    per step (measured: 1e-7 after the first padded step, 1e-5 after seven, 4e-2 after fourteen at a
    learning rate of 0.05), and with the pruning fusion it can also flip a near-tie of the ranking, after
    which the two runs train on different tokens."""
    dev = torch.device("cuda:0")
    eager = _run(dev, sizes, capture=False, sgd=True, fusion_type=fusion_type, epochs=epochs)
    graphed = _run(dev, sizes, capture=True, pad_to=16, sgd=True, fusion_type=fusion_type, epochs=epochs)
    assert graphed[3] == "hipgraph" and graphed[4] == 1   # every size rides the one 16-row graph
    assert eager[2] == graphed[2] == len(sizes) * epochs
    np.testing.assert_allclose(graphed[0].numpy(), eager[0].numpy(), rtol=1e-3, atol=2e-5)
    for (le, _), (lg, _) in zip(eager[1], graphed[1]):
        assert abs(le - lg) < 1e-4
