"""Clip-parallel multi-GPU plumbing: one process per GPU, RCCL over xGMI through torch.distributed.

The Stage-2 sampler shards over independent units (clips; for one long video, temporal windows within a DDIM step), so
there is no collective on the data path of a clip: rank 0 broadcasts the packed weights once, every rank samples its own
clips, and the decoded frames are gathered to rank 0.  The reference itself is single-GPU at
inference (scripts/pose2vid.py:116-135 only shards the file list).
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Unit (clip / window) indices owned by `rank`: round-robin, every unit exactly once."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_units, world))


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], spec: Dict[str, tuple], src: int = 0, device="cpu",
                         dtype=torch.float32, bucket_bytes: int = 512 << 20) -> Dict[str, torch.Tensor]:
    """Rank `src` holds `sd`; every rank returns the same tensors.  Tensors travel in flat buckets of ~512 MiB (a few large
    broadcasts: xGMI links are point-to-point, so fewer, larger transfers amortise the per-collective latency)."""
    rank = dist.get_rank()
    names = list(spec)
    out: Dict[str, torch.Tensor] = {}
    i = 0
    esz = torch.empty((), dtype=dtype).element_size()
    al = max(1, 16 // esz)                        # every tensor starts 16-byte aligned inside its bucket (the kernels' vector width)
    slot = lambda n: (_numel(spec[n]) + al - 1) // al * al
    while i < len(names):
        j, nbytes = i, 0
        while j < len(names) and (j == i or nbytes + slot(names[j]) * esz <= bucket_bytes):
            nbytes += slot(names[j]) * esz
            j += 1
        total = sum(slot(n) for n in names[i:j])
        flat = torch.zeros((total,), device=device, dtype=dtype)
        if rank == src:
            off = 0
            for n in names[i:j]:
                k = _numel(spec[n])
                flat[off:off + k] = sd[n].reshape(-1).to(device=device, dtype=dtype)
                off += slot(n)
        dist.broadcast(flat, src=src)
        off = 0
        for n in names[i:j]:
            k = _numel(spec[n])
            out[n] = flat[off:off + k].view(tuple(spec[n]))
            off += slot(n)
        i = j
    return out


def gather_frames(frames: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Every rank contributes its decoded clip(s) (any shape, same dtype); rank `dst` receives the list ordered by rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    shape = torch.tensor(list(frames.shape) + [0] * (8 - frames.dim()), device=frames.device, dtype=torch.int64)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    ndim = frames.dim()
    numels = [int(torch.prod(s[:ndim]).item()) if ndim else 1 for s in shapes]
    mx = max(numels)
    pad = torch.zeros((mx,), device=frames.device, dtype=frames.dtype)
    pad[: frames.numel()] = frames.reshape(-1)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:n].view(tuple(int(x) for x in s[:ndim])) for b, n, s in zip(bufs, numels, shapes)]


def allgather_window_predictions(pred: torch.Tensor, group=None) -> List[torch.Tensor]:
    """Window-parallel long video: ranks own disjoint windows of one DDIM step and exchange their
    predictions; every rank then applies the identical overlap-average + CFG + DDIM update."""
    world = dist.get_world_size(group)
    out = [torch.empty_like(pred) for _ in range(world)]
    dist.all_gather(out, pred.contiguous(), group=group)
    return out


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n
