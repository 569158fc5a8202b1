"""Data-parallel sharding of detection inference: one process per GPU, images partitioned across
ranks, and ONE exchange step -- an all-gather of the fixed-shape per-image detections -- when a
metric over the whole set is needed.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device_type="cuda"):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* if WORLD_SIZE > 1.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # YDET_DIST_BACKEND=gloo: rehearse the multi-process control flow on a box with fewer GPUs than ranks
        if device_type == "cuda" and os.environ.get("YDET_DIST_BACKEND", "nccl") == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(n, rank, world_size):
    """Contiguous [start, stop) of `n` items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_rows(t):
    """Concatenate the rows (dim 0) of `t` from every rank, in rank order; row counts may differ.
    One all_gather of the row counts (read on the host: the result's shape depends on them) + one all_gather of the
    padded tensor."""
    rank, ws = world()
    if ws == 1:
        return t
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    padded = t.new_zeros((m,) + tuple(t.shape[1:]))
    padded[: t.shape[0]] = t
    out = t.new_empty((ws * m,) + tuple(t.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous())
    return torch.cat([out[r * m: r * m + c] for r, c in enumerate(counts)], 0)


def all_gather_detections(det, count, image_id, total=None, sizes=None):
    """det [B, max_det, 6] fp32, count [B] int32, image_id [B] int64 (global image indices) of this rank's shard ->
    the same three tensors for the whole set, sorted by image id.

    ONE collective: every rank packs its rows as [detections | count | image id] (the integers ride as raw bits in
    fp32 lanes), pads them to the largest shard and calls all_gather_into_tensor once.  When every rank knows every
    shard's size -- `sizes` (images per rank, contiguous ranges in rank order), or `total` (images of the whole set,
    sharded by shard_range) -- the pad length follows from arithmetic: no size exchange and no host synchronisation.
    Without either the largest shard is found by one all-reduce(MAX) and the rows are picked by their (non-negative)
    image id."""
    rank, ws = world()
    if ws == 1:
        order = torch.argsort(image_id)
        return det[order], count[order], image_id[order]
    B, max_det = det.shape[0], det.shape[1]
    if sizes is None and total is not None:
        sizes = [shard_range(total, r, ws)[1] - shard_range(total, r, ws)[0] for r in range(ws)]
    if sizes is not None:
        assert len(sizes) == ws and sizes[rank] == B, f"rank {rank} holds {B} images, its shard is listed with {sizes[rank]}"
        m = max(max(sizes), 1)
    else:
        mt = torch.tensor([B], device=det.device, dtype=torch.int64)
        dist.all_reduce(mt, op=dist.ReduceOp.MAX)
        m = max(int(mt.item()), 1)
    width = max_det * 6 + 2
    packed = det.new_zeros((m, width), dtype=torch.float32)
    packed[:B, : max_det * 6] = det.reshape(B, max_det * 6).float()
    ints = torch.full((m, 2), -1, device=det.device, dtype=torch.int32)      # pad rows: image id -1
    ints[:B, 0] = count.to(torch.int32)
    ints[:B, 1] = image_id.to(torch.int32)
    packed[:, max_det * 6:] = ints.view(torch.float32)
    out = packed.new_empty((ws * m, width))
    dist.all_gather_into_tensor(out, packed.contiguous())
    if sizes is not None:      # contiguous shards in rank order are already in image order
        out = torch.cat([out[r * m: r * m + c] for r, c in enumerate(sizes)], 0)
    iv = out[:, max_det * 6:].contiguous().view(torch.int32)
    if sizes is None:
        keep = iv[:, 1] >= 0
        out, iv = out[keep], iv[keep]
        order = torch.argsort(iv[:, 1].long())
        out, iv = out[order], iv[order]
    return out[:, : max_det * 6].reshape(-1, max_det, 6), iv[:, 0].contiguous(), iv[:, 1].long()
