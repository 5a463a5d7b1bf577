"""Multi-GPU glue: reads shard across ranks with no data-path collective; the only exchange is
ONE all-reduce of the flat tally vector at the end of the job (replaces the per-thread merge of
src/TGSFilter.cpp:3208-3213 / :2673-2725 / :2586-2597 across GPUs).

backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.  The vector is a few
MB at most (17 + 512 + 8*bc_len*5 + 4*n_bins*5 words), so the all-reduce is latency-bound and
happens once per job -- it is never on the per-batch path.
"""
from __future__ import annotations

import numpy as np

from . import abi


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) of n_items for this rank."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_counters(vectors):
    """Host-side merge of several tally vectors (e.g. several contexts on one GPU)."""
    vectors = [np.asarray(v, dtype=np.uint64) for v in vectors]
    tot = vectors[0].copy()
    for v in vectors[1:]:
        rows = np.maximum(tot[abi.CTR_ROWS:abi.CTR_ROWS + 4], v[abi.CTR_ROWS:abi.CTR_ROWS + 4])
        tot += v
        tot[abi.CTR_ROWS:abi.CTR_ROWS + 4] = rows
    return tot


def allreduce_counters(ctr: np.ndarray, device=None, group=None) -> np.ndarray:
    """Sum the tally vector over all ranks (the four 'rows used' words are maxima).

    uint64 sums are done as int64 (two's complement: identical bits); every rank gets the result.
    """
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(ctr, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    # every rank must hold the same layout (same bc_len and max_read_len at tgsf_create): check before summing
    n = torch.tensor([t.numel(), -t.numel()], dtype=torch.int64, device=t.device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    if int(n[0].item()) != -int(n[1].item()):
        raise ValueError("tally vectors differ in length across ranks (%d here, %d..%d over the job): create every "
                         "context with the same bc_len and max_read_len" % (t.numel(), -int(n[1].item()), int(n[0].item())))
    rows = t[abi.CTR_ROWS:abi.CTR_ROWS + 4].clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(rows, op=dist.ReduceOp.MAX, group=group)
    t[abi.CTR_ROWS:abi.CTR_ROWS + 4] = rows
    return t.cpu().numpy().view(np.uint64).copy()
