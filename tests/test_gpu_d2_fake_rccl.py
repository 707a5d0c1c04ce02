"""The RCCL transport of libspx with MORE THAN ONE rank on a one-GPU box.

The real librccl cannot form a communicator of several ranks on one device, so on the 1-GPU test box the code around
the collectives -- ncclCommInitAll + the ncclGroup of P ncclAllGather calls of a multi-device handle, the record table
for P > 1, the 2-D partition's ncclAllReduce, spx_comm_attach(nranks = P) -- never ran.  Here libspx binds
tests/c/fake_rccl.hip instead (SPX_RCCL_LIB): a thread-rendezvous stand-in with the NCCL 2 signatures that moves the
data on the callers' own streams, ordered by events as a collective orders them.  transport="rccl" (spx_create_multi_transport) makes
spx_create_multi take the RCCL code path for repeated device ids.  Everything must equal the one-GPU handle: the
winner, the per-candidate EI bits, the 2-D partition's all-reduced sums; a collective that fails must surface as
SPX_ERR_HIP + spx_last_error (no hang, no crash).

libspx loads its RCCL binding once per process, so every case runs in a spawned child with the environment set first.
"""
import multiprocessing
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_SRC = os.path.join(ROOT, "tests", "c", "fake_rccl.hip")
FAKE_SO = os.path.join(ROOT, "tests", "c", "libfake_rccl.so")


def _fake_lib():
    """tests/c/libfake_rccl.so: built by __graft_entry__.build() (it travels with the tree), or here."""
    if not os.path.exists(FAKE_SO) or os.path.getmtime(FAKE_SO) < os.path.getmtime(FAKE_SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared",
                               "-o", FAKE_SO, FAKE_SRC, "-lpthread"])
    return FAKE_SO


def _child(fn, env, args, q):
    os.environ.update(env)
    sys.path.insert(0, ROOT)
    try:
        q.put(("ok", fn(*args)))
    except BaseException as e:                      # the parent prints it
        import traceback
        q.put(("error", "%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc())))


def _in_child(fn, *args, **env):
    e = {"SPX_RCCL_LIB": _fake_lib()}
    e.update(env)
    ctx = multiprocessing.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_child, args=(fn, e, args, q))
    p.start()
    try:
        kind, out = q.get(timeout=280)
    finally:
        p.join(60)
        if p.is_alive():
            p.kill()                                 # this very child, nothing else
    assert kind == "ok", out
    return out


def _stats():
    import ctypes
    lib = ctypes.CDLL(FAKE_SO)
    v = [ctypes.c_int(0) for _ in range(5)]
    lib.fake_rccl_stats(*[ctypes.byref(x) for x in v])
    return dict(zip(("allgather", "allreduce", "groups", "collectives", "max_ranks"), [x.value for x in v]))


# ---- one handle over P device slots: ncclCommInitAll + the group of P all-gathers --------------------------------
def _multi_handle_case(P):
    from spearmint_amd.engine import Engine, MultiEngine, rccl_version
    from spearmint_amd.synthetic import synthetic_problem
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(200, 3001, 5, 4, 61, per_sec=True)
    eng = Engine(0)
    one = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ops = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
    out = {"version": rccl_version()}
    me = MultiEngine([0] * P, transport="rccl")
    try:
        out["transport"] = me.transport()
        many = me.ei_grid(comp, vals, cand, hypers, want_draws=True)
        out["ranks_seen"] = me.stat("ranks_seen")
        out["winner"] = (many[0], many[1]) == (one[0], one[1])
        out["mean_bits"] = bool(np.array_equal(many[2], one[2]))
        out["draw_bits"] = bool(np.array_equal(many[3], one[3]))
        mps = me.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
        out["per_sec"] = mps[0] == ops[0] and mps[1] == ops[1] and bool(np.array_equal(mps[3], ops[3]))
        # ties across shards go to the lowest global index; a NaN wins; fewer candidates than device slots
        c2 = cand.copy(); c2[2500] = c2[10]
        a = eng.ei_grid(comp, vals, c2, hypers); b = me.ei_grid(comp, vals, c2, hypers)
        out["tie"] = a[0] == b[0] and a[1] == b[1]
        c2[2900, 0] = np.nan
        out["nan"] = eng.ei_grid(comp, vals, c2, hypers)[0] == me.ei_grid(comp, vals, c2, hypers)[0] == 2900
        a = eng.ei_grid(comp, vals, cand[:2], hypers); b = me.ei_grid(comp, vals, cand[:2], hypers)
        out["few"] = a[0] == b[0] and bool(np.array_equal(a[2], b[2]))
        # repeated steps on resident data (the bench's loop): the same winner every time
        me.set_observations(comp, vals); me.set_hypers(hypers); me.set_candidates(cand)
        wins = []
        for _ in range(3):
            me.ei_step(0)
            wins.append(me.best())
        out["steps"] = all(w == (one[0], one[1]) for w in wins)
    finally:
        me.close()
        eng.close()
    out["stats"] = _stats()
    return out


@pytest.mark.parametrize("P", [2, 4, 8])
@pytest.mark.timeout(300)
def test_multi_handle_group_allgather_with_P_ranks(P):
    out = _in_child(_multi_handle_case, P)
    assert out["version"] == 22707 and out["transport"] == "rccl" and out["ranks_seen"] == P
    for key in ("winner", "mean_bits", "draw_bits", "per_sec", "tie", "nan", "few", "steps"):
        assert out[key], (key, out)
    st = out["stats"]
    # every exchange was ONE group of P ncclAllGather calls, executed as ONE collective over P ranks
    assert st["max_ranks"] == P and st["allgather"] == P * st["groups"] and st["collectives"] == st["groups"] >= 8
    assert st["allreduce"] == 0


# ---- the 2-D partition: ONE ncclAllReduce(SUM) of the EI-sum vector ------------------------------------------------
def _emulate_2d(eng, comp, vals, cand, hypers, n, ph):
    from spearmint_amd import dist as sd
    M, H = cand.shape[0], hypers.shape[0]
    full = np.zeros(M)
    blocks = np.zeros((M, H))
    for r in range(n):
        (c0, c1), (h0, h1) = sd.shard_2d(M, H, n, r, ph)
        ei = eng.ei_grid(comp, vals, cand[c0:c1], hypers[h0:h1], want_draws=True)[3]
        blocks[c0:c1, h0:h1] = ei
        part = np.zeros(M)
        part[c0:c1] = np.sum(ei, axis=1)
        full = part if r == 0 else full + part
    mean = full / float(H)
    return int(np.argmax(mean)), mean, blocks


def _partition_case(P, ph):
    from spearmint_amd.engine import Engine, MultiEngine
    from spearmint_amd.synthetic import synthetic_problem
    comp, cand, vals, hypers = synthetic_problem(150, 2111, 5, 7, 97)
    eng = Engine(0)
    me = MultiEngine([0] * P, transport="rccl")
    out = {}
    try:
        me.set_partition(ph)
        me.set_observations(comp, vals); me.set_hypers(hypers); me.set_candidates(cand)
        me.factor(); me.ei_run()
        idx, mean, blocks = _emulate_2d(eng, comp, vals, cand, hypers, P, ph)
        one = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        out["winner"] = me.best() == (idx, mean[idx]) and idx == one[0]
        out["sum_bits"] = bool(np.array_equal(me.ei_mean(), mean))      # the stand-in adds in rank order, like the emulation
        out["close"] = bool(np.allclose(me.ei_mean(), one[2], rtol=1e-14, atol=0))
        out["draw_bits"] = bool(np.array_equal(me.ei_draws(), blocks) and np.array_equal(blocks, one[3]))
        me.set_partition(1)
        me.set_hypers(hypers); me.set_candidates(cand); me.factor(); me.ei_run()
        out["back"] = me.best() == (one[0], one[1]) and bool(np.array_equal(me.ei_mean(), one[2]))
    finally:
        me.close()
        eng.close()
    out["stats"] = _stats()
    return out


@pytest.mark.parametrize("P,ph", [(2, 2), (4, 2), (8, 4)])
@pytest.mark.timeout(300)
def test_2d_partition_allreduce_with_P_ranks(P, ph):
    out = _in_child(_partition_case, P, ph)
    for key in ("winner", "sum_bits", "close", "draw_bits", "back"):
        assert out[key], (key, out)
    assert out["stats"]["allreduce"] == P and out["stats"]["max_ranks"] == P


# ---- one rank per host thread: spx_comm_attach(nranks = P) ----------------------------------------------------------
def _attach_case(P, partition):
    import threading
    from spearmint_amd import dist as sd
    from spearmint_amd.engine import Engine
    from spearmint_amd.synthetic import synthetic_problem
    comp, cand, vals, hypers = synthetic_problem(180, 4003, 6, 6, 33)
    M, H = cand.shape[0], hypers.shape[0]
    eng = Engine(0)
    one = eng.ei_grid(comp, vals, cand, hypers)
    uid = eng.comm_unique_id()
    res, errs = [None] * P, [None] * P
    ph = 2 if partition else 1

    def rank_main(r):
        try:
            e = Engine(0)
            e.comm_attach(uid, P, r)
            if partition:
                (c0, c1), (h0, h1) = sd.shard_2d(M, H, P, r, ph)
                e.set_partition(ph, M, H)
            else:
                (c0, c1), (h0, h1) = sd.shard_bounds(M, P, r), (0, H)
            e.set_observations(comp, vals); e.set_hypers(hypers[h0:h1]); e.set_candidates(cand[c0:c1], index_base=c0)
            wins = []
            for _ in range(3):                       # spx_ei_step: factor + pass + the collective, all ranks in step
                e.ei_step(0)
                wins.append(e.best())
            res[r] = (wins, e.stat("ranks_seen"), e.ei_mean(), (c0, c1))
            e.close()
        except BaseException as ex:
            errs[r] = "%s: %s" % (type(ex).__name__, ex)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    out = {"errors": [e for e in errs if e], "alive": any(t.is_alive() for t in th)}
    if not out["errors"] and not out["alive"]:
        out["ranks_seen"] = [r[1] for r in res]
        out["same_winner"] = all(w == res[0][0][0] for r in res for w in r[0])
        out["index"] = res[0][0][0][0] == one[0]
        if partition:
            out["value_close"] = abs(res[0][0][0][1] - one[1]) <= 1e-14 * abs(one[1])
            out["mean_close"] = all(np.allclose(r[2], one[2][r[3][0]:r[3][1]], rtol=1e-14, atol=0) for r in res)
        else:
            out["value_bits"] = res[0][0][0][1] == one[1]
            out["mean_bits"] = all(np.array_equal(r[2], one[2][r[3][0]:r[3][1]]) for r in res)
    eng.close()
    out["stats"] = _stats()
    return out


@pytest.mark.parametrize("P", [2, 4, 8])
@pytest.mark.timeout(300)
def test_comm_attach_with_P_ranks_allgather(P):
    out = _in_child(_attach_case, P, False)
    assert not out["errors"] and not out["alive"], out
    assert out["ranks_seen"] == [P] * P and out["same_winner"] and out["index"] and out["value_bits"] and out["mean_bits"], out
    assert out["stats"]["allgather"] == 3 * P and out["stats"]["max_ranks"] == P and out["stats"]["groups"] == 0


@pytest.mark.timeout(300)
def test_comm_attach_with_4_ranks_allreduce_partition():
    out = _in_child(_attach_case, 4, True)
    assert not out["errors"] and not out["alive"], out
    assert out["same_winner"] and out["index"] and out["value_close"] and out["mean_close"], out
    assert out["stats"]["allreduce"] == 3 * 4 and out["stats"]["allgather"] == 0


# ---- a failing collective is an error code with a message, not a hang -----------------------------------------------
def _failing_case(mode):
    from spearmint_amd.engine import Engine, MultiEngine, SpxError
    from spearmint_amd.synthetic import synthetic_problem
    comp, cand, vals, hypers = synthetic_problem(100, 900, 4, 3, 5)
    out = {}
    if mode == "initall":
        try:
            MultiEngine([0, 0], transport="rccl")
            out["error"] = None
        except SpxError as e:
            out["error"] = str(e)
        return out
    me = MultiEngine([0, 0, 0], transport="rccl")
    try:
        if mode == "allreduce":
            me.set_partition(3)
        me.set_observations(comp, vals); me.set_hypers(hypers); me.set_candidates(cand)
        try:
            me.ei_step(0)
            out["error"] = None
        except SpxError as e:
            out["error"] = str(e)
        try:
            me.best()
            out["best_after_failure"] = True
        except ValueError:
            out["best_after_failure"] = False
        # the handle is still usable once the transport works again
        os.environ.pop("FAKE_RCCL_FAIL")
        me.ei_step(0)
        e1 = Engine(0)
        ref = e1.ei_grid(comp, vals, cand, hypers)
        e1.close()
        out["recovered"] = me.best()[0] == ref[0]
    finally:
        me.close()
    return out


@pytest.mark.parametrize("mode,needle", [("allgather", "ncclAllGather"), ("groupend", "GroupEnd"), ("allreduce", "ncclAllReduce"),
                                         ("initall", "ncclCommInitAll")])
@pytest.mark.timeout(300)
def test_failing_collective_is_an_error_code(mode, needle):
    out = _in_child(_failing_case, mode, FAKE_RCCL_FAIL=mode)
    assert out["error"] and needle in out["error"] and "injected" in out["error"], out
    if mode != "initall":
        assert out["best_after_failure"] is False and out["recovered"], out


def _attach_failure_case():
    import threading
    from spearmint_amd.engine import Engine, SpxError
    from spearmint_amd.synthetic import synthetic_problem
    comp, cand, vals, hypers = synthetic_problem(100, 900, 4, 3, 5)
    eng = Engine(0)
    uid = eng.comm_unique_id()
    errs = [None, None]

    def rank_main(r):
        e = Engine(0)
        e.comm_attach(uid, 2, r)
        e.set_observations(comp, vals); e.set_hypers(hypers); e.set_candidates(cand[r * 450:(r + 1) * 450], index_base=r * 450)
        if r == 1:
            return                  # this rank never makes its call: the other one must time out, not hang
        try:
            e.ei_step(0)
        except SpxError as ex:
            errs[r] = str(ex)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    return {"alive": any(t.is_alive() for t in th), "error": errs[0]}


@pytest.mark.timeout(300)
def test_missing_rank_is_a_timeout_error_not_a_hang():
    out = _in_child(_attach_failure_case)
    assert not out["alive"] and out["error"] and "ncclAllGather" in out["error"] and "timed out" in out["error"], out
