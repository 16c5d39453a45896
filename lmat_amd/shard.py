"""Multi-GPU layout of the path: reads shard across ranks (independent per read, src/read_label.cpp:1742-1748),
the k-mer table is replicated, and the only exchange is the final merge of the per-taxid tallies
(src/read_label.cpp:1760-1800) done as one all-reduce (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) of this rank; sizes differ by at most one, order preserved across ranks."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_tallies(count, score, nomatch, dist=None):
    """Sums the dense tallies (int64 count[n_ids], float64 score[n_ids], int64 nomatch[3]) over all ranks in
    place.  `dist` = torch.distributed (already initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.all_reduce(count)
    dist.all_reduce(score)
    dist.all_reduce(nomatch)
