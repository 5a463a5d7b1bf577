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


_GROUP = None


def set_group(group):
    """Process group of the tally exchange (default: the world group).  bench.py keeps a gloo world group for
    host-side barriers and hands the RCCL group in here."""
    global _GROUP
    _GROUP = group


def check_layout(n_words: int, group=None):
    """Setup-time check (once, before any batch): every rank must hold the same tally layout (same bc_len and
    max_read_len at tgsf_create), or the job's all-reduce would sum words that mean different things."""
    import torch
    import torch.distributed as dist

    group = group if group is not None else _GROUP
    dev = "cuda" if (group is not None and dist.get_backend(group) == "nccl") or (group is None and dist.get_backend() == "nccl") else "cpu"
    n = torch.tensor([n_words, -n_words], dtype=torch.int64, device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    if int(n[0].item()) != -int(n[1].item()):
        raise ValueError("tally vectors differ in length across ranks (%d here, %d..%d over the job): create every "
                         "context with the same bc_len and max_read_len" % (n_words, -int(n[1].item()), int(n[0].item())))


def allreduce_counters(ctr: np.ndarray, rank=None, world=None, device=None, group=None) -> np.ndarray:
    """The job's ONE collective: a SUM all-reduce of the tally vector; every rank gets the result.

    The four 'rows used' words are maxima, not sums: each rank carries them in a slot of its own behind the vector
    (zeros in the other ranks' slots), so the same SUM delivers every rank's values and the maximum is taken
    locally.  uint64 sums are done as int64 (two's complement: identical bits).
    """
    import torch
    import torch.distributed as dist

    group = group if group is not None else _GROUP
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    n = len(ctr)
    buf = np.zeros(n + 4 * world, dtype=np.uint64)
    buf[:n] = ctr
    buf[abi.CTR_ROWS:abi.CTR_ROWS + 4] = 0
    buf[n + 4 * rank:n + 4 * rank + 4] = ctr[abi.CTR_ROWS:abi.CTR_ROWS + 4]
    t = torch.from_numpy(buf.view(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    out = t.cpu().numpy().view(np.uint64)
    res = out[:n].copy()
    res[abi.CTR_ROWS:abi.CTR_ROWS + 4] = out[n:].reshape(world, 4).max(axis=0)
    return res
