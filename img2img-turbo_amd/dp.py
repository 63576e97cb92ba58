"""Data-parallel sharding of independent images over the GPUs of one node (one process per GPU).

The forward has no cross-image dependency (GroupNorm / LayerNorm / attention are per sample), so images
shard as contiguous batch slices with replicated weights and NO collective on the data path.  The only
exchange is the optional gather of finished images to rank 0: with backend "nccl" (= RCCL on ROCm) that is
7 direct xGMI sends landing on 7 distinct links of the root, far below the compute time.
The reference has no multi-GPU inference at all (every entry point is single-process, batch 1).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(total, rank, world):
    """Contiguous, balanced slice [lo, hi) of ``total`` images for ``rank`` (first ``total % world`` ranks get one more)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(t, rank, world):
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def gather_images(local, total, dst=0):
    """Gather per-rank output slices [b_r, 3, H, W] into [total, 3, H, W] on ``dst`` (None elsewhere).

    Equal shards use one ``dist.gather``; ragged shards pad to the largest shard."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    buf = local
    if local.shape[0] != mx:
        buf = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[: local.shape[0]] = local
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf.contiguous(), outs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
