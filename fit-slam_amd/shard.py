"""Multi-GPU sharding of the candidate list.

Candidates are independent units (DEP/src/FrontierCostsManager.cpp:74-119 carries no state from one
frontier to the next except running min/max, which the ranking step recomputes), so the frontier
batch is cut into contiguous blocks, one per rank; grid, landmark cloud and lookup table are
replicated on every GPU.  The only data-path collective is ONE all-gather of the fixed-size 32-byte
result records (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU in the tests).  Payload is
tiny (160 k candidates x 32 B = 5.1 MB in total), i.e. latency-bound: one collective, no bucketing.
"""
from __future__ import annotations

import numpy as np

RECORD_WORDS = 8          # fs_record = 8 x 32-bit words


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Rank `rank` scores candidates [lo, hi): blocks of ceil(n / world)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_capacity(n: int, world: int) -> int:
    return (n + world - 1) // world


def gather_records(local, n_total: int, group=None, async_op: bool = False):
    """All-gather per-rank record blocks into the full, order-preserving record list.

    local: torch int32 tensor [capacity, 8] (rows beyond this rank's share are padding) on the
    device the process group communicates on.  Returns a tensor [n_total, 8] on every rank; with
    async_op=True returns (tensor, work): the collective is ordered after everything already queued on the
    current stream and runs on the communicator's own stream, so the caller can queue the next batch's kernels
    right away and `work.wait()` before it touches the result (or reuses `local`).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    cap = local.shape[0]
    assert cap == shard_capacity(n_total, world), (cap, n_total, world)
    out = torch.empty((world * cap, RECORD_WORDS), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    # blocks are contiguous ranges of the candidate list, so trimming the tail padding restores the order
    if async_op:
        return out[:n_total], work
    return out[:n_total]


def records_to_numpy(t) -> np.ndarray:
    from .capi import RECORD_DTYPE
    a = t.detach().cpu().numpy().astype(np.int32, copy=False)
    return np.ascontiguousarray(a).view(RECORD_DTYPE).reshape(-1)
