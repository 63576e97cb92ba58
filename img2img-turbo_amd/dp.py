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


def shared_weights(build, rank, world, tag="weights"):
    """One host copy of the generator weights for all ranks of a node: rank 0 runs ``build()`` (a checkpoint read, or bench.py's
    synthesis of ~950 M random parameters) and writes the state dicts to a safetensors file in /dev/shm; the other ranks map that
    file (safetensors loads are mmap views: every rank's host tensors are the SAME pages) instead of building 8 identical copies
    concurrently.  Weak-scaling set-up time then does not grow with the rank count.  The file is unlinked once every rank has
    mapped it (the mappings keep the pages alive).  world == 1: just ``build()``."""
    if world <= 1:
        return build()
    import json
    from dataclasses import asdict
    from safetensors import safe_open
    from safetensors.torch import save_file
    from .arch import UNetArch, VAEArch
    from .weights import GeneratorWeights
    base = "/dev/shm" if os.path.isdir("/dev/shm") else (os.environ.get("TMPDIR") or "/tmp")
    path = os.path.join(base, "i2i_%s_%s_%s.safetensors" % (tag, os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run")))
    if rank == 0:
        w = build()
        flat = {}
        for pre, sd in (("unet/", w.unet), ("vae/", w.vae), ("vae_b2a/", w.vae_b2a or {})):
            for k, v in sd.items():
                flat[pre + k] = v.contiguous()
        meta = {"unet_arch": json.dumps(asdict(w.unet_arch)), "vae_arch": json.dumps(asdict(w.vae_arch)),
                "unet_scaling": json.dumps(w.unet_scaling), "vae_scaling": json.dumps(w.vae_scaling),
                "meta": json.dumps(w.meta, default=str), "has_b2a": "1" if w.vae_b2a is not None else "0"}
        save_file(flat, path + ".tmp", metadata=meta)
        os.replace(path + ".tmp", path)
    barrier()
    if rank != 0:
        unet, vae, b2a = {}, {}, {}
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata()
            for k in f.keys():
                pre, _, name = k.partition("/")
                {"unet": unet, "vae": vae, "vae_b2a": b2a}[pre][name] = f.get_tensor(k)
        tup = lambda d: {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}
        w = GeneratorWeights(unet, vae, UNetArch(**tup(json.loads(meta["unet_arch"]))), VAEArch(**tup(json.loads(meta["vae_arch"]))),
                             json.loads(meta["unet_scaling"]), json.loads(meta["vae_scaling"]), b2a if meta["has_b2a"] == "1" else None,
                             json.loads(meta["meta"]))
    barrier()
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
    return w


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
    calls it every step): ``local`` is the plan's static output buffer [b_r, ...]; rank ``dst`` owns [world, max_shard, ...]
    slabs whose per-rank views are the gather lists.  Ragged shards are padded to the largest.

    ``overlap=False``: ``dist.gather`` straight from ``local`` with async_op=False -- the CURRENT stream waits for the
    collective, so a following hipGraph replay on that stream cannot overwrite ``local`` while RCCL still reads it; the gather
    of step i is then serial with the replay of step i+1.

    ``overlap=True`` (double-buffered): step i copies ``local`` into staging buffer i % 2 on the compute stream (a device-to-
    device copy of a few MB), and the collective runs on a side stream behind an event -- the compute stream goes straight on
    to the replay of step i+1 while RCCL moves step i over xGMI.  Buffer i % 2 (and its slab on ``dst``) is reused at step
    i+2, after the compute stream has waited for the gather of step i.  ``images()`` / ``wait()`` join the side stream.
    On CPU tensors (gloo, the tests) there are no streams: same buffers, same alternation, synchronous collectives.

    Ownership in overlap mode: the slabs belong to the gather -- the side stream overwrites slab i % 2 two steps later, ordered
    only against the compute stream.  ``__call__`` therefore hands out no slab (returns None) and ``images()`` returns a COPY
    made on the current stream after joining the side stream; the copy is stream-ordered before the next staging copy, whose
    event the gather of step i+2 waits for, so that gather cannot overtake a reader of the returned tensor on this stream."""

    def __init__(self, local, total, dst=0, overlap=False):
        self.local, self.total, self.dst, self.overlap = local, total, dst, overlap
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.sizes = [shard_bounds(total, r, self.world)[1] - shard_bounds(total, r, self.world)[0] for r in range(self.world)]
        mx = max(self.sizes)
        nbuf = 2 if overlap else 1
        tail = tuple(local.shape[1:])
        if overlap or local.shape[0] != mx:
            self.send = [torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device) for _ in range(nbuf)]
        else:
            self.send = [local]
        self.slabs = self.views = None
        if self.rank == dst:
            self.slabs = [torch.empty((self.world, mx) + tail, dtype=local.dtype, device=local.device) for _ in range(nbuf)]
            self.views = [[sl[r] for r in range(self.world)] for sl in self.slabs]
        self.step = 0
        self.cuda = local.is_cuda
        if overlap and self.cuda:
            self.comm = torch.cuda.Stream(device=local.device)
            self.copied = [torch.cuda.Event() for _ in range(2)]       # staging buffer i filled (compute stream)
            self.gathered = [None, None]                               # gather out of staging buffer i done (side stream)

    @property
    def slab(self):
        """The slab the most recent call gathered into (``dst`` only)."""
        return None if self.slabs is None else self.slabs[(self.step - 1) % len(self.slabs)]

    def __call__(self):
        """overlap=False -> on ``dst``: the slab [world, max_shard, ...] of this step, complete on return (rank r's images are
        slab[r, :sizes[r]]); None elsewhere.  overlap=True -> None everywhere: the slab is still being written by the side stream
        and will be overwritten two steps later -- read the step through ``images()``."""
        i = self.step % len(self.send)
        self.step += 1
        send = self.send[i]
        views = self.views[i] if self.views is not None else None
        if not (self.overlap and self.cuda):
            if send is not self.local:
                send[: self.local.shape[0]].copy_(self.local)
            dist.gather(send, views, dst=self.dst)
            return self.slabs[i] if (self.slabs is not None and not self.overlap) else None
        cur = torch.cuda.current_stream(self.local.device)
        if self.gathered[i] is not None:
            cur.wait_event(self.gathered[i])            # the gather of step i-2 has read this staging buffer
        send[: self.local.shape[0]].copy_(self.local)
        self.copied[i].record(cur)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.copied[i])
            dist.gather(send, views, dst=self.dst)      # async_op=False: the SIDE stream waits for the collective
            ev = torch.cuda.Event()
            ev.record(self.comm)
            self.gathered[i] = ev
        return None

    def wait(self):
        """Join the side stream: everything gathered so far is complete for the current stream."""
        if self.overlap and self.cuda:
            torch.cuda.current_stream(self.local.device).wait_stream(self.comm)

    def images(self):
        """[total, ...] on ``dst`` in global batch order, of the most recent step.  overlap=False: a view of the slab when the
        shards are even (a copy when ragged); overlap=True: always a copy (the slab is reused two steps later)."""
        self.wait()
        if self.rank != self.dst:
            return None
        slab = self.slab
        if len(set(self.sizes)) == 1:
            out = slab.reshape((self.total,) + tuple(slab.shape[2:]))
            return out.clone() if self.overlap else out
        return torch.cat([slab[r, :s] for r, s in enumerate(self.sizes)], dim=0)


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
