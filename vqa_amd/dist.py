"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) for GPUs and gloo for the CPU rehearsals in tests/.

The fusion path shards by sample with no exchange at all; the DP-SGD step has one exchange per
step, a sum all-reduce of the flat clipped-gradient bucket (DPOptimizer.add_noise_and_scale).
"""
from __future__ import annotations

import os
from typing import Tuple

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
