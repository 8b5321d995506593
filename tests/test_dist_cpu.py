"""CPU rehearsal of the data-parallel DP-SGD exchange: two gloo ranks, each computing the
clipped sums of its shard with the oracle, reduced with vqa_amd.dist exactly as DPOptimizer does
(sum all-reduce of the flat bucket, noise split N(0, (sigma C)^2 / world) per rank)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dp_oracle as do
from oracle import inputs as gi
from vqa_amd import dist as vdist

IN, HID, C, B = 64, 64, 13, 12


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    m = do.make_classifier(IN, HID, C, dp_fixed=True)
    m.load_state_dict(gi.classifier_params(50, IN, HID, C))
    return m.eval()


def _flat(d):
    return torch.cat([d[k].reshape(-1) for k in sorted(d)])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    r, _, w = vdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    x, y = gi.normal(1, B, IN), gi.labels(2, B, C)
    lo, hi = vdist.shard_range(B, rank, world)
    sigma, clip = 1.0, 1.0
    _, _, sums, _ = do.dp_step(_model(), x[lo:hi], y[lo:hi], clip, 0.0, B)
    bucket = _flat(sums)
    # every rank adds its own share of the noise before the reduce (rank-seeded stream)
    std = vdist.per_rank_noise_std(sigma, clip, world)
    gen = torch.Generator().manual_seed(1234 + rank)
    noise = torch.randn(bucket.shape, generator=gen) * std
    reduced = vdist.allreduce_sum_((bucket + noise) / B)
    noise_total = vdist.allreduce_sum_(noise.clone())
    vdist.barrier()
    if rank == 0:
        torch.save({"reduced": reduced, "noise_total": noise_total}, os.path.join(out_dir, "r0.pt"))
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (1, 7, 1024, 200000):
        for world in (1, 2, 3, 8):
            parts = [vdist.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    assert vdist.shard_range(1024, 3, 8) == (384, 512)


def test_noise_split_adds_up():
    for world in (1, 2, 4, 8):
        std = vdist.per_rank_noise_std(1.3, 0.7, world)
        assert abs(world * std ** 2 - (1.3 * 0.7) ** 2) < 1e-12


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "r0.pt"), weights_only=True)
    x, y = gi.normal(1, B, IN), gi.labels(2, B, C)
    _, _, sums, _ = do.dp_step(_model(), x, y, 1.0, 0.0, B)
    want = _flat(sums) / B
    # per-sample clipping is rank-local, so shard sums add up to the full-batch sum
    np.testing.assert_allclose((got["reduced"] - got["noise_total"] / B).numpy(), want.numpy(),
                               rtol=1e-5, atol=1e-7)
    # the summed noise has variance (sigma C)^2: 2 shares of variance 1/2 each
    n = got["noise_total"]
    assert abs(float(n.std()) - 1.0) < 5.0 / n.numel() ** 0.5 + 0.05


def _sampler_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    vdist.init_from_env(backend="gloo")
    from vqa_amd import dp
    torch.manual_seed(42)  # identically seeded ranks: the case a rank-unaware sampler gets wrong
    n, bs = 96, 12
    x, y = gi.normal(1, n, IN), gi.labels(2, n, C)
    ds = torch.utils.data.TensorDataset(x, y, torch.arange(n))
    loader = torch.utils.data.DataLoader(ds, batch_size=bs)
    engine = dp.PrivacyEngine()
    model = _model()
    gsm, opt, dl = engine.make_private(module=model, optimizer=torch.optim.SGD(model.parameters(), lr=0.1),
                                       data_loader=loader, noise_multiplier=1.0, max_grad_norm=1.0,
                                       poisson_sampling=True, seed=5)
    assert opt.expected_batch_size == bs  # the logical batch, not bs * world
    owned = dl.batch_sampler.owned().tolist()
    steps, total = [], torch.zeros(sum(p.numel() for p in model.parameters()))
    for batch in dl:
        idx = [] if batch is None else batch[2].tolist()
        steps.append(idx)
        bucket = torch.zeros_like(total)
        if idx:  # this rank's clipped sums of the samples it drew (the oracle does what the kernels do)
            _, _, sums, _ = do.dp_step(model, x[idx], y[idx], 1.0, 0.0, bs)
            bucket = _flat(sums)
        total += vdist.allreduce_sum_(bucket)
    gathered = [None] * world
    dist.all_gather_object(gathered, {"owned": owned, "steps": steps})
    if rank == 0:
        torch.save({"ranks": gathered, "total": total}, os.path.join(out_dir, "sampler.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_poisson_shards_are_disjoint(tmp_path):
    """make_private in a process group: rank shards partition the dataset, so no sample enters
    two ranks' batches in a step and the reduced sum counts every drawn sample once, clipped to C."""
    world, port = 2, _free_port()
    mp.spawn(_sampler_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "sampler.pt"), weights_only=False)
    r0, r1 = got["ranks"]
    assert not set(r0["owned"]) & set(r1["owned"])
    assert sorted(r0["owned"] + r1["owned"]) == list(range(96))
    assert len(r0["steps"]) == len(r1["steps"]) == 8
    x, y = gi.normal(1, 96, IN), gi.labels(2, 96, C)
    want = None
    drawn = 0
    for s0, s1 in zip(r0["steps"], r1["steps"]):
        assert not set(s0) & set(s1)
        assert set(s0) <= set(r0["owned"]) and set(s1) <= set(r1["owned"])
        assert len(set(s0)) == len(s0) and len(set(s1)) == len(s1)
        union = s0 + s1
        drawn += len(union)
        if union:
            norms, clip, sums, _ = do.dp_step(_model(), x[union], y[union], 1.0, 0.0, 12)
            assert float((norms * clip).max()) <= 1.0 + 1e-6  # one sample contributes at most C in total
            want = _flat(sums) if want is None else want + _flat(sums)
    assert 96 * 0.5 < drawn < 96 * 1.6  # 8 steps at rate 1/8 draw about one dataset's worth
    assert r0["steps"] != r1["steps"]
    # sums of ~100 clipped per-sample gradients in f32, grouped differently on the two sides
    np.testing.assert_allclose(got["total"].numpy(), want.numpy(), rtol=1e-5,
                               atol=1e-5 * float(want.abs().max()))


# ---------------------------------------------------------------------------------------------------
# Non-private gradients: batch-weighted mean over the ranks (vqa_amd.dist.GradReducer), the exchange
# in front of optimizer['main'].step() (reference: train.py:156-160, utils/src.py:103)
# ---------------------------------------------------------------------------------------------------
class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Linear(IN, HID)
        self.idle = torch.nn.Linear(IN, 3)  # never used in forward: its gradient is None on every rank
        self.head = torch.nn.Linear(HID, C)

    def forward(self, x):
        return self.head(torch.tanh(self.enc(x)))


def _net():
    torch.manual_seed(77)
    return _Net()


def _reducer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    vdist.init_from_env(backend="gloo")
    net = _net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    red = vdist.GradReducer(net.parameters())
    x, y = gi.normal(1, B, IN), gi.labels(2, B, C)
    # step 1: unequal shards (5 and 7 samples); step 2: rank 1 has an EMPTY batch
    shards = [((0, 5), (5, 12)), ((0, 12), None)]
    grads = []
    for step in shards:
        opt.zero_grad()
        part = step[rank]
        if part is None:
            red.reduce(0)
        else:
            lo, hi = part
            torch.nn.functional.cross_entropy(net(x[lo:hi]), y[lo:hi]).backward()
            red.reduce(hi - lo)
        assert net.idle.weight.grad is None
        grads.append(torch.cat([p.grad.reshape(-1) for p in (net.enc.weight, net.enc.bias, net.head.weight, net.head.bias)]))
        opt.step()
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.save({"grads": grads, "params": params}, os.path.join(out_dir, f"red{rank}.pt"))
    vdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_grad_reducer_keeps_replicas_identical(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_reducer_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "red0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(str(tmp_path), "red1.pt"), weights_only=True)
    # every rank leaves a step with the same gradient and the same parameters, bit for bit
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)
    assert torch.equal(r0["params"], r1["params"])
    # ... and they are those of ONE process on the joint batch (mean loss over all 12 samples)
    net = _net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x, y = gi.normal(1, B, IN), gi.labels(2, B, C)
    for g in r0["grads"]:
        opt.zero_grad()
        torch.nn.functional.cross_entropy(net(x), y).backward()
        want = torch.cat([p.grad.reshape(-1) for p in (net.enc.weight, net.enc.bias, net.head.weight, net.head.bias)])
        np.testing.assert_allclose(g.numpy(), want.numpy(), rtol=2e-5, atol=1e-7)
        opt.step()
    np.testing.assert_allclose(r0["params"].numpy(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy(),
                               rtol=2e-5, atol=1e-7)


def test_grad_reducer_is_a_no_op_without_a_process_group():
    net = _net()
    red = vdist.GradReducer(net.parameters())
    torch.nn.functional.cross_entropy(net(gi.normal(1, 4, IN)), gi.labels(2, 4, C)).backward()
    before = net.enc.weight.grad.clone()
    red.reduce(4)
    assert torch.equal(net.enc.weight.grad, before)
