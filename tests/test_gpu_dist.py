"""GPU: the data-parallel DP-SGD step with two ranks.  The box has one GPU, so both ranks share
cuda:0 and exchange through gloo; what is checked is the logic RCCL runs unchanged on 8 GPUs:
shard -> per-rank clip/sum -> noise share -> sum all-reduce -> identical fused AdamW on every rank,
and equality with the single-process step on the whole batch when the noise is off."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import dp_oracle as do
from oracle import inputs as gi

pytestmark = pytest.mark.gpu
IN, HID, C, B = 128, 256, 100, 24


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(dev):
    m = do.make_classifier(IN, HID, C, dp_fixed=True)
    m.load_state_dict(gi.classifier_params(60, IN, HID, C))
    return m.eval().to(dev)


def _step(dev, x, y, sigma, expected, steps=2):
    from vqa_amd import dp
    m = _model(dev)
    gsm = dp.GradSampleModule(m)
    opt = dp.DPOptimizer(torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=0.01), gsm,
                         noise_multiplier=sigma, max_grad_norm=1.0, expected_batch_size=expected, seed=7)
    first_grad = None
    for _ in range(steps):
        opt.zero_grad()
        nn.CrossEntropyLoss()(gsm(x), y).backward()
        opt.step()
        if first_grad is None:
            first_grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu(), first_grad


def _graph_step(dev, x, y, expected):
    """Step 1 eagerly, step 2 as: replayed HIP graph (forward, backward, per-rank DP work) -> eager
    all-reduce -> replayed HIP graph (update) -- the launch shape bench.py uses with several ranks."""
    from vqa_amd import dp
    m = _model(dev)
    gsm = dp.GradSampleModule(m)
    opt = dp.DPOptimizer(torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=0.01), gsm,
                         noise_multiplier=0.0, max_grad_norm=1.0, expected_batch_size=expected, seed=7)

    def local():
        opt.zero_grad()
        nn.CrossEntropyLoss()(gsm(x), y).backward()
        opt.local_step()

    local()
    opt.reduce_step()
    opt.apply_step()
    torch.cuda.synchronize()
    g_local, g_apply = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_local, capture_error_mode="thread_local"):
        local()
    with torch.cuda.graph(g_apply, pool=g_local.pool(), capture_error_mode="thread_local"):
        opt.apply_step()
    g_local.replay()
    # (two processes time-share this one GPU: reduce_step itself drains the launch stream first)
    opt.reduce_step()
    g_apply.replay()
    torch.cuda.synchronize()
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), VQAF_DIST_BACKEND="gloo", VQAF_SINGLE_DEVICE="1")
    from vqa_amd import dist as vdist
    r, local, w = vdist.init_from_env()
    dev = vdist.local_device(local)
    torch.cuda.set_device(dev)
    x, y = gi.normal(1, B, IN).to(dev), gi.labels(2, B, C).to(dev)
    lo, hi = vdist.shard_range(B, rank, world)
    quiet, quiet_grad = _step(dev, x[lo:hi], y[lo:hi], 0.0, B)
    noisy, _ = _step(dev, x[lo:hi], y[lo:hi], 1.0, B)
    graphed = _graph_step(dev, x[lo:hi], y[lo:hi], B)
    torch.save({"quiet": quiet, "noisy": noisy, "quiet_grad": quiet_grad, "graphed": graphed},
               os.path.join(out_dir, f"r{rank}.pt"))
    vdist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_share_one_gpu_over_gloo(tmp_path):
    assert torch.cuda.is_available()
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "r0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(str(tmp_path), "r1.pt"), weights_only=True)
    # replicas stay bit-identical: every rank applies the same reduced gradient
    assert torch.equal(r0["quiet"], r1["quiet"]) and torch.equal(r0["noisy"], r1["noisy"])
    # noise off: two shards == one process on the whole batch (per-sample clipping is rank-local)
    dev = torch.device("cuda:0")
    x, y = gi.normal(1, B, IN).to(dev), gi.labels(2, B, C).to(dev)
    single, single_grad = _step(dev, x, y, 0.0, B)
    gs = float(single_grad.abs().max())
    np.testing.assert_allclose(r0["quiet_grad"].numpy(), single_grad.numpy(), rtol=1e-5, atol=2e-6 * gs)
    # Adam divides by sqrt(v): where a gradient is ~0 the last bits of the sum order decide the
    # update's sign, so parameters agree to a fraction of one lr-sized step, not to f32 rounding
    np.testing.assert_allclose(r0["quiet"].numpy(), single.numpy(), rtol=1e-3, atol=2e-4)
    # graph replay around the eager collective: same replicas on both ranks, same update as the
    # eagerly launched step (the few f32 atomics of the affine sums may reorder: last bits only)
    assert torch.equal(r0["graphed"], r1["graphed"])
    np.testing.assert_allclose(r0["graphed"].numpy(), r0["quiet"].numpy(), rtol=1e-3, atol=2e-4)
    assert float((r0["graphed"] - r0["quiet"]).abs().max()) < 2e-4
    # noise on: the parameters moved away from the noise-free ones by a noise-sized amount
    assert float((r0["noisy"] - r0["quiet"]).abs().max()) > 1e-4


# ---------------------------------------------------------------------------------------------------
# Whole training step in a process group: loops.train with the {'main', 'classifier'} optimizers of
# train.py:156-162 -- non-private gradients averaged over the ranks (GradReducer), clipped sums of the
# classifier all-reduced (DPOptimizer) -- keeps encoder AND classifier replicas identical and equals
# one process on the joint batch.
# ---------------------------------------------------------------------------------------------------
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


NS = 24


def _samples():
    g = torch.Generator().manual_seed(9)
    return (torch.randn(NS, 3, 16, 16, generator=g), torch.randint(0, 100, (NS, 12), generator=g),
            torch.randint(0, 13, (NS,), generator=g))


def _batch_of(lo, hi):
    img, ids, ans = _samples()
    return {"image": img[lo:hi], "inputs": {"input_ids": ids[lo:hi], "attention_mask": torch.ones(hi - lo, 12, dtype=torch.long)},
            "answer": ans[lo:hi]}


def _train_steps(dev, batches):
    from vqa_amd import dp, loops
    from vqa_amd.model import VQAModel
    torch.manual_seed(42)
    model = VQAModel(vision=TinyV, text=TinyT, fusion_type="co_attention", hidden_dim=64, num_classes=13,
                     use_token_pruning=True, adversarial_mix=False, noise_injection=False)
    model.classifier = dp.ModuleValidator.fix(model.classifier)
    model.classifier[3].p = 0.0  # dropout masks depend on a sample's position in its batch
    model = model.to(dev)
    gsm = dp.GradSampleModule(model.classifier)
    model.classifier = gsm
    cls_opt = dp.DPOptimizer(torch.optim.SGD(gsm.parameters(), lr=0.05), gsm, noise_multiplier=0.0, max_grad_norm=1.0,
                             expected_batch_size=NS, seed=3)
    main_params = [p for n, p in model.named_parameters() if not n.startswith('classifier.') and p.requires_grad]
    optimizer = {'main': torch.optim.SGD(main_params, lr=0.05), 'classifier': cls_opt}
    loops.train(model, batches, optimizer, nn.CrossEntropyLoss(), dev, use_dp_sgd=True)
    torch.cuda.synchronize()
    enc = torch.cat([p.detach().reshape(-1) for n, p in model.named_parameters() if not n.startswith('classifier.')]).cpu()
    cls = torch.cat([p.detach().reshape(-1) for p in gsm.parameters()]).cpu()
    return enc, cls


def _train_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), VQAF_DIST_BACKEND="gloo", VQAF_SINGLE_DEVICE="1")
    from vqa_amd import dist as vdist
    _, local, _ = vdist.init_from_env()
    dev = vdist.local_device(local)
    torch.cuda.set_device(dev)
    assert vdist.ranks_share_device(dev)
    # step 1: unequal disjoint shards; step 2: rank 1 draws an empty batch
    batches = [[_batch_of(0, 10), _batch_of(0, NS)], [_batch_of(10, NS), None]][rank]
    enc, cls = _train_steps(dev, batches)
    torch.save({"enc": enc, "cls": cls}, os.path.join(out_dir, f"t{rank}.pt"))
    vdist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_training_step_keeps_every_replica_identical(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_train_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "t0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(str(tmp_path), "t1.pt"), weights_only=True)
    assert torch.equal(r0["enc"], r1["enc"])   # encoders + fusion: the non-private optimizer
    assert torch.equal(r0["cls"], r1["cls"])   # classifier: the DP optimizer
    enc, cls = _train_steps(torch.device("cuda:0"), [_batch_of(0, NS), _batch_of(0, NS)])
    np.testing.assert_allclose(r0["enc"].numpy(), enc.numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(r0["cls"].numpy(), cls.numpy(), rtol=1e-4, atol=2e-6)
