"""Sharding of an image list over ranks / devices.

Images are independent (the reference keeps no state between calls, seqoia.h:439), so multi-GPU
is a partition of the image list: no collective in the data path.  Uniform batches are cut into
contiguous blocks; mixed-size batches use longest-processing-time-first on the pixel count so that
every rank gets about the same number of pixels.
"""
from __future__ import annotations

import heapq
from typing import List, Sequence


def contiguous(n_items: int, world: int, rank: int) -> range:
    """Block `rank` of `world` near-equal contiguous blocks of range(n_items)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def lpt(weights: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first: item indices per rank, heaviest items placed first on the
    least loaded rank.  Deterministic (ties by index), so every rank computes the same plan."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + int(weights[i]), r))
    for part in out:
        part.sort()
    return out


def gather_lengths(local_idx: Sequence[int], local_len: Sequence[int], n_items: int, group=None) -> List[int]:
    """Host-side gather of per-image stream lengths into global order (the only cross-rank step:
    a few bytes per image, outside the timed data path).  Works on any torch.distributed backend."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        out = [0] * n_items
        for i, l in zip(local_idx, local_len):
            out[i] = int(l)
        return out
    world = dist.get_world_size(group)
    # RCCL ("nccl") moves device tensors only; gloo moves host tensors
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([list(map(int, local_idx)), list(map(int, local_len))], dtype=torch.int64, device=dev).reshape(2, -1)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.shape[1]], dtype=torch.int64, device=dev), group=group)
    m = int(max(s.item() for s in sizes))
    pad = torch.full((2, m), -1, dtype=torch.int64, device=dev)
    pad[:, : mine.shape[1]] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    parts = [p.cpu() for p in parts]
    out = [0] * n_items
    for p in parts:
        for i, l in zip(p[0].tolist(), p[1].tolist()):
            if i >= 0:
                out[i] = l
    return out
