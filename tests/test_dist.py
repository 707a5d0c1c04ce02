"""Candidate sharding + the single collective (all-gather of 16-byte records), on CPU with gloo (world_size 2)."""
import os
import socket

import numpy as np
import pytest

from spearmint_amd import dist as sd


def test_shard_bounds_cover_and_balance():
    for M, P in [(10, 3), (1000000, 8), (7, 8), (128, 2)]:
        spans = [sd.shard_bounds(M, P, r) for r in range(P)]
        assert spans[0][0] == 0 and spans[-1][1] == M
        assert all(spans[i][1] == spans[i + 1][0] for i in range(P - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_pick_best_numpy_rule():
    nan = float("nan")
    assert sd.pick_best([[1.0, 5], [3.0, 9], [3.0, 7]])[0] == 7        # tie -> lowest index
    assert sd.pick_best([[1.0, 5], [nan, 9], [3.0, 7]])[0] == 9        # NaN beats everything
    assert sd.pick_best([[nan, 12], [nan, 9]])[0] == 9                 # first NaN
    assert sd.pick_best([[0.0, -1], [2.0, 4]])[0] == 4                 # empty shard ignored
    # same answer as numpy on the concatenated vector
    rs = np.random.RandomState(0)
    for _ in range(50):
        v = rs.rand(40); v[rs.randint(40, size=3)] = v[0]
        if rs.rand() < 0.3:
            v[rs.randint(40)] = nan
        recs = []
        for r in range(4):
            lo, hi = sd.shard_bounds(40, 4, r)
            j = int(np.argmax(v[lo:hi]))
            recs.append([v[lo + j], lo + j])
        assert sd.pick_best(recs)[0] == int(np.argmax(v))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as tdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    # a fixed global EI vector; each rank scores its shard and the all-gather + numpy-argmax rule picks the winner
    v = np.random.RandomState(123).rand(1001)
    v[700] = v[100] = 2.0                      # tie across shards -> index 100 must win
    lo, hi = sd.shard_bounds(v.shape[0], world, rank)
    j = int(np.argmax(v[lo:hi]))
    out1 = sd.exchange_best(v[lo + j], lo + j)
    v2 = v.copy(); v2[900] = np.nan             # NaN in the last shard wins
    j2 = int(np.argmax(v2[lo:hi]))
    out2 = sd.exchange_best(v2[lo + j2], lo + j2)
    q.put((rank, out1, out2))
    tdist.barrier()
    tdist.destroy_process_group()


def test_exchange_best_gloo_world2():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, o1, o2 in res:
        assert o1[0] == 100 and o1[1] == 2.0
        assert o2[0] == 900 and np.isnan(o2[1])


def test_allreduce_without_group_is_identity():
    assert sd.exchange_best(1.5, 42) == (42, 1.5)


def test_strong_scaling_grid_does_not_depend_on_world_size():
    """bench.py's full-size C4 / C5 grids are defined in seeded blocks: any split over ranks
    concatenates to the same rows (so every --gpus N must report the same best_index)."""
    import bench
    cfg = dict(bench.STRONG["c5"]); cfg["M"] = 3 * bench.STRONG_BLOCK + 1234
    _, comp, vals, _ = bench.strong_problem(dict(cfg, N=32))
    full = bench.strong_rows(cfg, comp, vals, 0, cfg["M"])
    assert full.shape == (cfg["M"], cfg["D"]) and np.all((full >= 0) & (full <= 1))
    for world in (2, 3, 8):
        parts = [bench.strong_rows(cfg, comp, vals, *sd.shard_bounds(cfg["M"], world, r)) for r in range(world)]
        assert np.array_equal(np.vstack(parts), full)
    # the incumbent's jittered copies open the grid
    inc = comp[np.argmin(vals)]
    assert np.max(np.abs(full[:10] - inc)) < 0.01


def test_shard_2d_covers_the_product():
    for M, H, P, ph in [(1000, 20, 8, 2), (1000, 20, 8, 4), (77, 5, 4, 1), (50, 3, 6, 3)]:
        seen = np.zeros((M, H), dtype=int)
        for r in range(P):
            (c0, c1), (h0, h1) = sd.shard_2d(M, H, P, r, ph)
            seen[c0:c1, h0:h1] += 1
        assert np.all(seen == 1)                     # every (candidate, draw) evaluated exactly once
    with pytest.raises(ValueError):
        sd.grid_2d(8, 3)


def _worker_2d(rank, world, port, q):
    import torch.distributed as tdist
    from oracle import gp_ei_oracle as orc
    from spearmint_amd.synthetic import synthetic_problem
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    comp, cand, vals, hypers = synthetic_problem(40, 301, 3, 6, 17)
    (c0, c1), (h0, h1) = sd.shard_2d(301, 6, world, rank, 2)
    ei = orc.ei_over_hypers(comp, cand[c0:c1], vals, hypers[h0:h1])       # this rank's block of overall_ei
    idx, val, mean = sd.allreduce_ei_sums(np.sum(ei, axis=1), c0, 301, 6)
    q.put((rank, idx, val, mean))
    tdist.barrier()
    tdist.destroy_process_group()


def test_2d_partition_allreduce_gloo_world4():
    """draws x candidates over 4 ranks (2 x 2): one all-reduce(SUM) of the M-vector, every rank ends with the mean EI
    of every candidate and the reference's argmax."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import gp_ei_oracle as orc
    from spearmint_amd.synthetic import synthetic_problem
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_2d, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    comp, cand, vals, hypers = synthetic_problem(40, 301, 3, 6, 17)
    ref = np.mean(orc.ei_over_hypers(comp, cand, vals, hypers), axis=1)
    for rank, idx, val, mean in res:
        assert idx == int(np.argmax(ref))
        # (the oracle's BLAS results depend on the column blocking at the 1e-12 level)
        assert np.allclose(mean, ref, rtol=1e-9, atol=1e-300) and np.isclose(val, ref[idx], rtol=1e-9)
    assert all(np.array_equal(res[0][3], r[3]) for r in res[1:])           # bit-identical on every rank


# ---- bench.py --gpus N starts N ranks by itself ---------------------------------------------------
def _bench(args, env=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, cwd=root, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    return res.returncode, (json.loads(lines[-1]) if lines else None), res.stderr.decode()


STANDIN = {"SPX_BENCH_BACKEND": "gloo", "SPX_BENCH_SINGLE_DEVICE": "1", "SPX_BENCH_ENGINE": "tests.standin_engine:Engine"}
SMALL = ["--workload", "c2", "--candidates", "1500", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-live-traffic",
         "--skip-extras"]


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` -- the form of the command the driver runs -- with NO launcher around it: bench.py
    re-executes itself under torch.distributed.run, two ranks take part in the one all-gather (ranks_seen, counted from
    the gathered table itself), and the line says n_gpus == 2.  The engine is the oracle-backed stand-in (no GPU here);
    the GPU suite runs the same command on libspx (tests/test_gpu_d_multi.py)."""
    import bench
    from oracle import gp_ei_oracle as orc
    rc, out, err = _bench(["--gpus", "2"] + SMALL, STANDIN)
    assert rc == 0 and out is not None, err[-2000:]
    assert out["n_gpus"] == out["ranks_seen"] == 2 and out["launcher"].startswith("self")
    assert out["backend"] == "gloo" and out["engine"] == "tests.standin_engine:Engine"
    assert out["config"]["candidates_per_gpu"] == 1500 and out["value"] > 0
    # the winner over both ranks' shards == one rank over the concatenated candidates
    w = dict(bench.WORKLOADS["c2"], M=1500)
    _, comp, vals, hypers, s0 = bench.weak_problem(w, 0)
    s1 = bench.weak_problem(w, 1)[4]
    mean = np.mean(orc.ei_over_hypers(comp, np.vstack((s0, s1)), vals, hypers), axis=1)
    # (the oracle's BLAS results depend on the column blocking at the 1e-12 level: same index, value to rounding)
    assert out["best_index"] == int(np.argmax(mean)) and abs(out["best_ei"] - float(np.max(mean))) <= 1e-9 * float(np.max(mean))
    # ... and the same command with --gpus 1 is one rank
    rc1, out1, err1 = _bench(["--gpus", "1"] + SMALL, STANDIN)
    assert rc1 == 0 and out1["n_gpus"] == out1["ranks_seen"] == 1 and out1["launcher"].startswith("none"), err1[-2000:]
    m0 = np.mean(orc.ei_over_hypers(comp, s0, vals, hypers), axis=1)
    assert out1["best_index"] == int(np.argmax(m0))


def test_bench_refuses_a_line_whose_n_gpus_is_not_what_was_asked_for():
    """--gpus 2 on a node without two devices (this container has none), or inside a world of another size: a non-zero
    exit and NO JSON line -- never a line that says n_gpus 1."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert rc != 0 and out is None and "--gpus 2" in err and "GPU" in err
    rc, out, err = _bench(["--gpus", "2"] + SMALL, dict(STANDIN, WORLD_SIZE="1", RANK="0"))
    assert rc != 0 and out is None and "WORLD_SIZE=1" in err
