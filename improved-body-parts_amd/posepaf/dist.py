"""Image sharding + the one exchange of the inference path: an all-gather of fixed-size per-image records.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).
Images are independent (evaluate.py:262-267), so image i goes to rank i mod W and nothing else is communicated."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import RECORD_BYTES, RECORD_DTYPE


def shard_indices(n_images: int, rank: int, world: int) -> np.ndarray:
    """Indices of the images rank `rank` processes (round-robin, like SURVEY 8e)."""
    return np.arange(rank, n_images, world)


def padded_shard_size(n_images: int, world: int) -> int:
    return (n_images + world - 1) // world


def gather_records(local: torch.Tensor, n_local_valid: int, group=None) -> np.ndarray | None:
    """local: uint8 tensor (S * RECORD_BYTES) with S = padded shard size, same S on every rank.
    Returns on EVERY rank the records of all ranks re-interleaved into global image order (numpy structured
    array of length sum(valid)); ranks whose shard is shorter pad with records that are dropped here."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    S = local.numel() // RECORD_BYTES
    out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[rank] = n_local_valid
    dist.all_reduce(counts, group=group)
    recs = out.cpu().numpy().view(RECORD_DTYPE).reshape(world, S)
    counts = counts.cpu().numpy()
    total = int(counts.sum())
    merged = np.empty(total, RECORD_DTYPE)
    for r in range(world):
        idx = np.arange(r, total, world)[: counts[r]]
        merged[idx] = recs[r, : counts[r]]
    return merged
