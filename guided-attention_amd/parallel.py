"""Seed-parallel execution: one process per GPU, seeds striped over ranks, no per-step exchange.

The reference generates seeds serially on one device (run.py:97-112); images for different
(seed, hyper-parameter state) are independent, so the path shards with no data-path collective:
  start-up : ONE broadcast of the model weights from rank 0 (RCCL over xGMI; flattened per-dtype
             buckets so each of the 7 point-to-point links carries a few large messages),
  per image: nothing,
  end      : ONE gather of the final latents (32 KB each at 512^2 fp16) to rank 0.
Works on any torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """-> (rank, world_size, local_rank); initialises the default group from the torchrun environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_seeds(seeds, rank, world):
    """Seeds striped by rank: rank r takes seeds[r], seeds[r + world], ..."""
    return list(seeds)[rank::world]


@torch.no_grad()
def broadcast_module_(module, src=0, bucket_bytes=512 << 20):
    """Broadcast every parameter and buffer of `module` from `src`, packed into per-dtype flat buckets of
    up to `bucket_bytes` (few, large messages: xGMI is point-to-point, ~153 GB/s per link)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    # the parameters themselves (not `.data`): `copy_` below then bumps their version counters, which is what the
    # weight-derived caches (fused QKV, cached text K/V, batched time projections) are keyed on
    tensors = list(module.parameters()) + list(module.buffers())
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    n_msgs = 0
    for (dtype, device), group in by_dtype.items():
        limit = max(1, bucket_bytes // max(1, group[0].element_size()))
        i = 0
        while i < len(group):
            chunk, count = [], 0
            while i < len(group) and (not chunk or count + group[i].numel() <= limit):
                chunk.append(group[i])
                count += group[i].numel()
                i += 1
            flat = torch.empty(count, dtype=dtype, device=device)
            if dist.get_rank() == src:
                torch.cat([t.reshape(-1) for t in chunk], out=flat)
            dist.broadcast(flat, src=src)
            n_msgs += 1
            if dist.get_rank() != src:
                off = 0
                for t in chunk:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
    return n_msgs


def gather_tensors(local, dst=0):
    """local: list of equally-shaped tensors produced on this rank -> on `dst`, the list of per-rank lists
    (rank-major); None elsewhere.  Ranks may hold different counts (striping remainder)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(local)]
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [torch.zeros(1, dtype=torch.long, device=_dev(local)) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(local)], dtype=torch.long, device=_dev(local)))
    counts = [int(c) for c in counts]
    shape_src = local[0] if local else None
    meta = [None] * world
    dist.all_gather_object(meta, None if shape_src is None else (tuple(shape_src.shape), shape_src.dtype))
    shape, dtype = next(m for m in meta if m is not None)
    pad = max(counts)
    buf = torch.zeros((pad,) + shape, dtype=dtype, device=_dev(local))
    for i, t in enumerate(local):
        buf[i] = t
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [[out[r][i] for i in range(counts[r])] for r in range(world)]


def _dev(local):
    if local:
        return local[0].device
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def unstripe(per_rank):
    """Inverse of shard_seeds on gathered results: [[r0 items], [r1 items], ...] -> original seed order."""
    world = len(per_rank)
    total = sum(len(x) for x in per_rank)
    return [per_rank[i % world][i // world] for i in range(total)]


def rank_world():
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def execute_seeds(generate, seeds, module=None):
    """Run `generate(seed) -> tensor` for this rank's share of `seeds`; weights of `module` are first
    broadcast from rank 0; returns on rank 0 the results in seed order (None on other ranks)."""
    rank, world = rank_world()
    if module is not None:
        broadcast_module_(module)
    mine = [generate(s) for s in shard_seeds(seeds, rank, world)]
    gathered = gather_tensors(mine)
    return unstripe(gathered) if gathered is not None else None
