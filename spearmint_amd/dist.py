"""Candidate sharding across one-process-per-GPU ranks and the single
collective of the path.

Every (candidate, draw) EI evaluation is independent given the draw's factor;
the only cross-candidate step of the reference is
``np.argmax(np.mean(overall_ei, axis=1))`` (GPEIChooser.py:153).  So rank r
owns the contiguous candidate rows [lo_r, hi_r) of the grid, replicates the
(tiny) observations and hyper draws, and the ranks exchange exactly one
record each -- (best mean EI, global index) -- in ONE all-reduce:

    buf = zeros(P, 2); buf[rank] = (value, index); all_reduce(buf, SUM)

Adding zeros is exact, so after the all-reduce every rank holds all P records
bit-for-bit (a MAX all-reduce on a packed key would lose mantissa bits; RCCL
has no MAXLOC).  The final pick applies numpy's argmax rule: first NaN wins,
else the largest value, ties to the lowest global index -- contiguous shards
keep "lowest index" meaningful.  Backend "nccl" is RCCL over xGMI on ROCm;
"gloo" is used by the CPU tests.
"""
from __future__ import print_function

import numpy as np


def shard_bounds(M, world_size, rank):
    """Contiguous, balanced [lo, hi) split of M candidate rows."""
    base, extra = divmod(int(M), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def pick_best(records):
    """numpy-argmax rule over (value, global_index) records.

    records: array (P, 2) float64; index < 0 marks an empty shard."""
    best_v, best_i = None, -1
    for v, i in np.asarray(records, dtype=np.float64):
        i = int(i)
        if i < 0:
            continue
        if best_i < 0:
            best_v, best_i = v, i
            continue
        a_nan, b_nan = np.isnan(v), np.isnan(best_v)
        if a_nan or b_nan:
            better = (a_nan and not b_nan) or (a_nan and b_nan and i < best_i)
        else:
            better = (v > best_v) or (v == best_v and i < best_i)
        if better:
            best_v, best_i = v, i
    return best_i, (float(best_v) if best_i >= 0 else float("nan"))


def allreduce_best(local_value, local_index, device=None, group=None):
    """One all-reduce; returns (global_index, value), identical on every rank.

    Without an initialised process group (single process) it is the identity."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return int(local_index), float(local_value)
    if not (dist.is_available() and dist.is_initialized()):
        return int(local_index), float(local_value)
    P = dist.get_world_size(group)
    r = dist.get_rank(group)
    buf = torch.zeros((P, 2), dtype=torch.float64, device=device)
    buf[r, 0] = float(local_value)
    buf[r, 1] = float(local_index)      # exact below 2**53
    if np.isnan(local_value):
        # NaN + 0 stays NaN, which is what we want for the value column
        pass
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return pick_best(buf.cpu().numpy())
