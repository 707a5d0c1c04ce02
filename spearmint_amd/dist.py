"""Candidate sharding across one-process-per-GPU ranks and the single
collective of the path.

Every (candidate, draw) EI evaluation is independent given the draw's factor;
the only cross-candidate step of the reference is
``np.argmax(np.mean(overall_ei, axis=1))`` (GPEIChooser.py:153).  So rank r
owns the contiguous candidate rows [lo_r, hi_r) of the grid, replicates the
(tiny) observations and hyper draws, and the ranks exchange exactly one
record each -- {best mean EI (fp64), global index (int64)}, 16 bytes -- in ONE
all-gather, after which every rank holds all P records
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

    records: P pairs (value, index); index < 0 marks an empty shard."""
    best_v, best_i = None, -1
    for v, i in records:
        v, i = float(v), int(i)
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


def exchange_best(local_value, local_index, device=None, group=None):
    """The one collective of the path: every rank contributes its 16-byte record {best mean EI
    (float64), global index (int64)} to ONE all-gather (P x 16 bytes; RCCL over xGMI with backend
    "nccl"), then applies the same numpy-argmax reduction to the gathered table.  Returns
    (global_index, value), identical on every rank.  The record travels as raw bytes, so the index
    is exact over the whole int64 range.

    Without an initialised process group (single process) it is the identity."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return int(local_index), float(local_value)
    if not (dist.is_available() and dist.is_initialized()):
        return int(local_index), float(local_value)
    P = dist.get_world_size(group)
    rec = np.zeros(1, dtype=[("val", "<f8"), ("idx", "<i8")])
    rec["val"][0] = local_value
    rec["idx"][0] = local_index
    mine = torch.from_numpy(rec.view(np.uint8).copy())
    if device is not None:
        mine = mine.to(device)
    table = torch.empty(16 * P, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(table, mine, group=group)
    got = table.cpu().numpy().view([("val", "<f8"), ("idx", "<i8")])
    return pick_best([(r["val"], r["idx"]) for r in got])


allreduce_best = exchange_best   # round-1 name
