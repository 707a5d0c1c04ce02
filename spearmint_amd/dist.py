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


def exchange_records(local_value, local_index, device=None, group=None):
    """The one collective of the path: every rank contributes its 16-byte record {best mean EI
    (float64), global index (int64)} to ONE all-gather (P x 16 bytes; RCCL over xGMI with backend
    "nccl").  Returns the gathered table as a list of P (value, index) pairs in rank order -- the
    same bytes on every rank; its length is the size of the group the collective ran on.  The record travels
    as raw bytes, so the index is exact over the whole int64 range.

    Without an initialised process group (single process) the table is this rank's own record."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return [(float(local_value), int(local_index))]
    if not (dist.is_available() and dist.is_initialized()):
        return [(float(local_value), int(local_index))]
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
    return [(float(r["val"]), int(r["idx"])) for r in got]


def exchange_best(local_value, local_index, device=None, group=None):
    """exchange_records + the numpy-argmax reduction every rank applies to the gathered table:
    (global_index, value), identical on every rank."""
    return pick_best(exchange_records(local_value, local_index, device=device, group=group))


allreduce_best = exchange_best   # round-1 name


# ---- optional 2-D partition: draws x candidates ("hypers x candidates") --------------
# P = P_h x P_c ranks: rank r = rc * P_h + rh evaluates the draws of hyper shard rh for the candidates of
# shard rc.  The single collective is then an all-reduce(SUM) of the zero-padded M-vector of per-candidate EI
# sums (8 M bytes), after which EVERY rank holds sum_h EI[c, h] for every candidate and takes the argmax
# locally.  It pays when replicating the H factorisations on every rank is significant (large N, few
# candidates per rank); the price is that the sum over draws is no longer numpy's pairwise order but "local
# sums, then the reduction tree", so means can differ from the 1-D scheme in the last bits (ties and near-ties
# may resolve differently; everything else is the same number to ~1e-16 relative).
def grid_2d(world_size, hyper_shards):
    """(P_h, P_c) for `world_size` ranks; hyper_shards must divide world_size."""
    ph = int(hyper_shards)
    if ph < 1 or world_size % ph:
        raise ValueError("hyper_shards=%d does not divide world_size=%d" % (ph, world_size))
    return ph, world_size // ph


def shard_2d(M, H, world_size, rank, hyper_shards):
    """This rank's piece of the (draws x candidates) product: ((c_lo, c_hi), (h_lo, h_hi))."""
    ph, pc = grid_2d(world_size, hyper_shards)
    rh, rc = rank % ph, rank // ph
    return shard_bounds(M, pc, rc), shard_bounds(H, ph, rh)


def allreduce_ei_sums(local_sums, c_lo, M, H, device=None, group=None):
    """The one collective of the 2-D scheme.  local_sums[c - c_lo] = sum over THIS rank's draws of EI[c, h] for
    its candidates; returns (global_index, mean EI value, mean vector) -- identical on every rank -- where
    mean[c] = (sum over all ranks) / H and the index follows numpy's argmax rule."""
    full = np.zeros(int(M), dtype=np.float64)
    full[c_lo:c_lo + len(local_sums)] = local_sums
    try:
        import torch
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized()
    except ImportError:  # pragma: no cover
        live = False
    if live:
        t = torch.from_numpy(full)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        full = t.cpu().numpy()
    mean = full / float(H)
    idx = int(np.argmax(mean))
    return idx, float(mean[idx]), mean
