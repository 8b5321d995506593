"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) for GPUs and gloo for the CPU rehearsals in tests/.

The fusion path shards by sample with no exchange at all.  A training step has two exchanges: the
sum all-reduce of the flat clipped-gradient bucket of the private classifier (DPOptimizer.reduce_step)
and the batch-weighted mean of the NON-private gradients (encoders, fusion: the parameters of the
reference's `optimizer['main']`, train.py:156-160, stepped at utils/src.py:103), GradReducer below.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the default group when
    WORLD_SIZE > 1.  HSA_ENABLE_IPC_MODE_LEGACY=0 must stay exported for RCCL on this driver."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # VQAF_DIST_BACKEND=gloo rehearses the multi-rank path where RCCL cannot run
            # (CPU-only hosts, or several ranks sharing one GPU)
            backend = os.environ.get("VQAF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def local_device(local_rank: int) -> torch.device:
    """cuda:<local_rank>, or cuda:0 for every rank when VQAF_SINGLE_DEVICE=1 (rehearsal of N ranks
    on a one-GPU box; only meaningful with the gloo backend)."""
    if os.environ.get("VQAF_SINGLE_DEVICE") == "1":
        return torch.device("cuda", 0)
    return torch.device("cuda", local_rank)


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) shard of n samples for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def per_rank_noise_std(noise_multiplier: float, max_grad_norm: float, world: int) -> float:
    """Each rank adds N(0, std^2) before the sum all-reduce; world independent shares sum to
    N(0, (noise_multiplier * max_grad_norm)^2): the noise is added exactly once."""
    return noise_multiplier * max_grad_norm / (world ** 0.5)


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


_shared_device: Dict[Tuple[int, int], bool] = {}


def ranks_share_device(device: torch.device, group=None) -> bool:
    """True when two ranks of `group` drive the same GPU (a one-GPU rehearsal of a multi-GPU job).
    Found out once per (group, device) by gathering (host, GPU uuid) from every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return False
    if device.type != "cuda":
        return False
    key = (id(group), device.index if device.index is not None else torch.cuda.current_device())
    if key not in _shared_device:
        import socket
        props = torch.cuda.get_device_properties(device)
        ident = (socket.gethostname(), str(getattr(props, "uuid", "")) or f"{props.name}:{key[1]}",
                 getattr(props, "pci_bus_id", -1), getattr(props, "pci_device_id", -1))
        every = [None] * dist.get_world_size(group)
        dist.all_gather_object(every, ident, group=group)
        _shared_device[key] = len(set(every)) < len(every)
    return _shared_device[key]


def drain_if_device_is_shared(device: torch.device, group=None) -> None:
    """Before a collective: when several ranks time-share one GPU, wait for this rank's launch stream.
    A collective's side stream parked on an event of a launch stream that has just replayed a HIP
    graph was seen never to be released in that set-up (two processes, one device); with one process
    per GPU nothing is waited for."""
    if ranks_share_device(device, group):
        torch.cuda.current_stream(device).synchronize()


class GradReducer:
    """Batch-weighted mean of the gradients of `params` over the process group, so that every replica
    steps its non-private optimizer with the gradient of the mean loss over the ranks' joint batch
    (what the single-process reference computes at utils/src.py:96-103).

    One flat bucket per dtype holds the gradients of every parameter that receives one on ANY rank
    (found out by a one-off exchange at the first call; parameters whose gradient is None everywhere
    -- the attention weights in pruning mode, which only feed indices -- stay out).  A step is:
    gradients into the bucket, times the local batch size, ONE sum all-reduce per bucket plus one of
    the batch sizes, division by the joint batch size on the device (no host synchronisation), and
    `param.grad` becomes a view of the bucket.  A rank with an empty batch calls `reduce(0)`: it
    contributes zeros and leaves with the same gradient as everyone else, so replicas stay identical.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.group = process_group
        self._live: Optional[List[int]] = None
        self._buckets: List[Tuple[torch.Tensor, List[torch.Tensor], List[int]]] = []
        self._count: Optional[torch.Tensor] = None

    def world(self) -> int:
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def reset(self) -> None:
        """Forget which parameters take part (call on every rank after changing what the model trains)."""
        self._live, self._buckets, self._count = None, [], None

    def _setup(self, device: torch.device, grads) -> None:
        mask = torch.tensor([0 if g is None else 1 for g in grads], dtype=torch.int32, device=device)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        self._live = [i for i, m in enumerate(mask.cpu().tolist()) if m]
        by_dtype: Dict[torch.dtype, List[int]] = {}
        for i in self._live:
            by_dtype.setdefault(self.params[i].dtype, []).append(i)
        self._buckets = []
        for dt, idx in by_dtype.items():
            sizes = [(self.params[i].numel() + 7) // 8 * 8 for i in idx]  # every slice 16-byte aligned
            flat = torch.zeros(sum(sizes), dtype=dt, device=device)
            views, o = [], 0
            for i, n in zip(idx, sizes):
                views.append(flat[o:o + self.params[i].numel()].view_as(self.params[i]))
                o += n
            self._buckets.append((flat, views, idx))
        self._count = torch.zeros(1, dtype=torch.float32, device=device)

    @torch.no_grad()
    def reduce(self, local_batch: int, grads: Optional[List[Optional[torch.Tensor]]] = None) -> None:
        """`grads` (one entry per parameter of this reducer, None where there is none) names the tensors
        that hold this step's gradients when they are not `param.grad` -- a replayed HIP graph writes
        them where its capture allocated them, while `param.grad` still shows the previous step's view."""
        if self.world() == 1 or not self.params:
            return
        device = self.params[0].device
        if grads is None:
            grads = [p.grad for p in self.params]
        if self._live is None:
            self._setup(device, grads)
        if not self._live:
            return  # nothing trains through this optimizer (known to every rank alike)
        live = set(self._live)
        for i, g in enumerate(grads):
            if g is not None and i not in live:
                raise RuntimeError("GradReducer: a parameter received a gradient that had none on any rank when the "
                                   "buckets were laid out; call reset() on every rank after changing what is trained")
        self._count.fill_(float(local_batch))
        for flat, views, idx in self._buckets:
            dst, src = [], []
            for v, i in zip(views, idx):
                g = grads[i]
                if g is None:
                    v.zero_()
                elif g.data_ptr() != v.data_ptr():
                    dst.append(v)
                    src.append(g if g.dtype == v.dtype else g.to(v.dtype))
            if dst:
                torch._foreach_copy_(dst, src)
            flat.mul_(float(local_batch))
        drain_if_device_is_shared(device, self.group)
        dist.all_reduce(self._count, op=dist.ReduceOp.SUM, group=self.group)
        for flat, _, _ in self._buckets:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        total = self._count.clamp_min(1.0)
        for flat, views, idx in self._buckets:
            flat.div_(total.to(flat.dtype))
            for v, i in zip(views, idx):
                self.params[i].grad = v
