"""Multi-GPU execution of a batch of independent (mesh, camera) render jobs.

The path shards embarrassingly: every job reads only its own slice of the packed
arrays and writes only its own output images, so ranks exchange nothing until the end.  One
process per GPU (torchrun); the only collective is one gather of the final per-rank
result over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).

`partition()` is pure host logic; `gather_batch()` is the collective.  Nothing here is specific
to the compute function, so the world_size-2 gloo tests drive it with the oracle as the stand-in.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def partition(costs: Sequence[float], world_size: int) -> List[Tuple[int, int]]:
    """Contiguous ranges [start, stop) of the batch, one per rank, balancing the summed cost.

    Greedy prefix split against the ideal per-rank share; every rank gets a (possibly empty)
    range, ranges are ordered and cover 0..len(costs)."""
    n = len(costs)
    total = float(sum(costs))
    bounds = [0]
    acc = 0.0
    i = 0
    for r in range(1, world_size):
        target = total * r / world_size
        while i < n and acc + costs[i] * 0.5 <= target:
            acc += costs[i]
            i += 1
        # leave at least one job per remaining rank when possible
        i = min(i, max(n - (world_size - r), bounds[-1]))
        i = max(i, bounds[-1])
        bounds.append(i)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def shard_packed(first_idx: torch.Tensor, counts: torch.Tensor, start: int, stop: int):
    """Slice the per-mesh index vectors for jobs start..stop-1 and rebase them to the slice.

    Returns (elem_lo, elem_hi, first_rebased, counts_slice): the rank rasterizes
    packed[elem_lo:elem_hi] with first_rebased/counts_slice; pix_to_face values it produces are
    local -- add elem_lo to make them global again (`rebase_indices`)."""
    if stop <= start:
        z = first_idx.new_zeros((0,))
        return 0, 0, z, z
    f = first_idx[start:stop]
    c = counts[start:stop]
    lo = int(f[0])
    hi = int(f[-1] + c[-1])
    return lo, hi, f - lo, c


def rebase_indices(idx: torch.Tensor, offset: int) -> torch.Tensor:
    """Local primitive ids -> global ones; -1 padding stays -1."""
    return torch.where(idx >= 0, idx + offset, idx)


def gather_batch(local: torch.Tensor, sizes: Sequence[int], group=None, dst=None):
    """Collect per-rank results of shape (sizes[rank], ...) into the full batch (sum(sizes), ...).

    dst=None: `all_gather`, every rank gets the batch.  dst=r: `gather` to rank r only (returns None on the other
    ranks) -- on xGMI's point-to-point links the seven peers then send to the destination in parallel, one shard
    per link, instead of circulating all shards around a ring.  Ranks may own different numbers of jobs: shards are
    padded to max(sizes) for the collective (RCCL needs equal shapes) and trimmed after.  Not differentiable (final
    gather)."""
    if not (dist.is_available() and dist.is_initialized()):
        # one process, no process group (BASELINE configs[4] on one GPU): the batch is the shard
        if len(sizes) != 1:
            raise RuntimeError("gather_batch without a process group takes exactly one shard; got sizes=%r" % (list(sizes),))
        return local[: int(sizes[0])]
    world = dist.get_world_size(group)
    m = max(int(s) for s in sizes) if len(sizes) else 0
    tail = tuple(local.shape[1:])
    pad = local.new_zeros((m,) + tail)
    if local.shape[0]:
        pad[: local.shape[0]] = local
    if dst is None:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
    else:
        me = dist.get_rank(group)
        bufs = [torch.empty_like(pad) for _ in range(world)] if me == dst else None
        dist.gather(pad, bufs, dst=dst, group=group)
        if me != dst:
            return None
    return torch.cat([b[: int(s)] for b, s in zip(bufs, sizes)], 0)
