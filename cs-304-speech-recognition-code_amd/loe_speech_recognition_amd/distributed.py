"""Multi-GPU plumbing: one process per GPU, utterances sharded in contiguous ranges, RCCL
(torch.distributed backend "nccl" on ROCm) only to gather decoded label records and to sum
training statistics.  Replaces the reference's host-side ProcessPoolExecutor fan-out
(scripts/project5_test_ndigits_with_sil.py:29-42, hidden_markov_model.py:746-750).
All functions fall back to the identity when torch.distributed is not initialised; with the
"gloo" backend they run on CPU tensors (used by the tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def _dist():
    import torch.distributed as dist

    return dist


def is_active() -> bool:
    dist = _dist()
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world() -> Tuple[int, int]:
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of `rank`; sizes differ by at most one, earlier ranks get the extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_reduce_stats(stats):
    """Sum the packed float64 statistics (count, sum_x, sum_xx, transition counts) over all ranks:
    one all-reduce per training iteration; every rank then runs the same host M-step."""
    if is_active():
        dist = _dist()
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def gather_label_records(labels, n_labels, n_total: int):
    """All ranks' fixed-width label records in global utterance order.

    labels: uint8 [B_local, cap] and n_labels: int32 [B_local] of this rank's contiguous shard
    (`shard_bounds(n_total, rank, world)`).  Returns (uint8 [n_total, cap], int32 [n_total]) on every rank.
    Shards are padded to the largest shard so that a single all-gather suffices."""
    import torch

    if not is_active():
        return labels, n_labels
    dist = _dist()
    rank, world = rank_world()
    cap = labels.shape[1]
    biggest = shard_bounds(n_total, 0, world)[1]
    packed = torch.full((biggest, cap + 4), 0xFF, dtype=torch.uint8, device=labels.device)
    b = labels.shape[0]
    packed[:b, :cap] = labels
    packed[:b, cap:] = n_labels.to(torch.int32).contiguous().view(torch.uint8).reshape(b, 4)
    out = torch.empty((world * biggest, cap + 4), dtype=torch.uint8, device=labels.device)
    dist.all_gather_into_tensor(out, packed)
    keep = torch.cat([torch.arange(r * biggest, r * biggest + (shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0]),
                                   device=labels.device) for r in range(world)])
    out = out[keep]
    return out[:, :cap].contiguous(), out[:, cap:].contiguous().view(torch.int32).reshape(-1)


def records_to_strings(labels: np.ndarray, n_labels: np.ndarray, names: Sequence[str]) -> List[str]:
    cap = labels.shape[1]
    return ["".join(names[w] for w in labels[i, :min(int(n_labels[i]), cap)]) for i in range(labels.shape[0])]
