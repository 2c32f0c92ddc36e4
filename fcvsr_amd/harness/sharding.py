"""Clip-parallel work partitioning for multi-GPU inference (one process per GPU, no data-path collective).

Every output frame depends only on its own 7-frame LR window (no recurrent state, reference CVSR_freq.py:2611-2646), so
centre-frame indices are split into contiguous chunks, one per rank; a rank needs a 3-frame halo of LR frames on each
side of its chunk (host-side reads).  `gather_counts` is the only collective: an all-gather of per-rank frame counts /
timings used for reporting.
"""
from __future__ import annotations

from typing import List, Tuple


def shard(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the contiguous chunk of `n_items` owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sequences(seq_lens: List[int], rank: int, world: int) -> List[Tuple[int, int, int]]:
    """Flatten several sequences into one list of frames and return this rank's share as (seq, first, last+1) ranges."""
    total = sum(seq_lens)
    lo, hi = shard(total, rank, world)
    out, base = [], 0
    for s, n in enumerate(seq_lens):
        a, b = max(lo, base), min(hi, base + n)
        if a < b:
            out.append((s, a - base, b - base))
        base += n
    return out


def throughput(frames_per_rank: List[int], seconds_per_rank: List[float]) -> float:
    """Whole-job frames/s = all frames / slowest rank (the bench.py definition)."""
    return sum(frames_per_rank) / max(seconds_per_rank)
