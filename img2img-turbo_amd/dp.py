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


class OutputGather:
    """Gather of the finished images to ``dst`` into buffers allocated ONCE (the timed loop of bench.py / a serving loop
    calls it every step): ``local`` is the plan's static output buffer [b_r, ...]; rank ``dst`` owns one
    [world, max_shard, ...] slab whose per-rank views are the gather list.  Ragged shards are padded to the largest.

    ``dist.gather`` with async_op=False makes the CURRENT stream wait for the collective, so a following hipGraph replay
    on that stream cannot overwrite ``local`` while RCCL still reads it."""

    def __init__(self, local, total, dst=0):
        self.local, self.total, self.dst = local, total, dst
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.sizes = [shard_bounds(total, r, self.world)[1] - shard_bounds(total, r, self.world)[0] for r in range(self.world)]
        mx = max(self.sizes)
        self.send = local
        if local.shape[0] != mx:
            self.send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        self.slab = self.views = None
        if self.rank == dst:
            self.slab = torch.empty((self.world, mx) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            self.views = [self.slab[r] for r in range(self.world)]

    def __call__(self):
        """-> on ``dst``: the slab [world, max_shard, ...] (rank r's images are slab[r, :sizes[r]]); None elsewhere."""
        if self.send is not self.local:
            self.send[: self.local.shape[0]].copy_(self.local)
        dist.gather(self.send, self.views, dst=self.dst)
        return self.slab

    def images(self):
        """[total, ...] on ``dst`` in global batch order (a copy only when shards are ragged)."""
        if self.rank != self.dst:
            return None
        if len(set(self.sizes)) == 1:
            return self.slab.reshape((self.total,) + tuple(self.slab.shape[2:]))
        return torch.cat([self.slab[r, :s] for r, s in enumerate(self.sizes)], dim=0)


def gather_images(local, total, dst=0):
    """One-shot form of OutputGather: per-rank slices [b_r, 3, H, W] -> [total, 3, H, W] on ``dst`` (None elsewhere)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    g = OutputGather(local.contiguous(), total, dst)
    g()
    return g.images()


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value, device):
    """[value of rank 0, ..., value of rank world-1] on every rank (one small all_gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
