"""Multi-GPU execution of the path: one process per GPU, proofs sharded by contiguous ranges, no collective
on the data path; the only exchange is the AND of the per-GPU verdict bits.
`torch.distributed` is used for that single int32 MIN all-reduce (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in CPU tests)."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous proof range [lo, hi) of `rank`: lo = rank*n/world (sizes differ by at most one)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def and_reduce(ok: bool, device=None) -> bool:
    """Logical AND of one verdict bit per rank.  Outside a process group it is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


def batch_verify_sharded(verify_fn: Callable[[np.ndarray, np.ndarray, np.ndarray, np.ndarray], bool], inst: np.ndarray,
                         commitments: np.ndarray, responses: np.ndarray, transcripts: np.ndarray, rank: int, world: int,
                         device=None) -> bool:
    """Each rank batch-verifies its own proof range (a valid batch check in its own right, with its own random
    weights: batch_verifier.rs:173-206 has no cross-proof state except the static-coefficient sums, which are
    per batch) and the verdicts are AND-ed.  `verify_fn(transcripts, inst, commitments, responses) -> bool` is
    the local batch check (zkp_amd.toolbox.batch_verify bound to an Engine on the GPU box)."""
    n = commitments.shape[0]
    lo, hi = shard_range(n, rank, world)
    ok = True
    if hi > lo:
        ok = bool(verify_fn(np.ascontiguousarray(transcripts[lo:hi]), np.ascontiguousarray(inst[:, lo:hi]),
                            np.ascontiguousarray(commitments[lo:hi]), np.ascontiguousarray(responses[lo:hi])))
    return and_reduce(ok, device)
