"""Multi-GPU pieces on the one-GPU test box:

 * spx_create_multi -- one handle over several devices, the RCCL all-gather of {EI, index} records
   inside libspx.  devices=[0] builds a real RCCL communicator of size 1 (ncclCommInitAll +
   ncclAllGather run on the hardware); devices=[0, 0, 0] puts three engines on the one GPU
   (repeated ids cannot form a communicator, so the records are staged through the host) and
   exercises sharding, the per-device threads and the final reduction.  Both must be bit-identical
   to a single engine.
 * bench.py with two ranks (one process per rank, gloo, both on GPU 0): the winner equals a
   one-rank run over the concatenated candidates, for the weak headline and for the strong-scaling
   c4 / c5 sub-records.
 * spx_ei_grad_batch against the reference's own refinement objective (tests/golden/ei_grad.npz)
   and the oracle's restatement of it.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import bench
from oracle import gp_ei_oracle as orc
from spearmint_amd.engine import Engine, MultiEngine
from spearmint_amd.synthetic import synthetic_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


# ---- spx_create_multi -----------------------------------------------------------------------------
# (first: the only test that needs two distinct GPUs -- the first multi-GPU box that runs the suite reaches it)
def test_rccl_transport_on_distinct_gpus(eng):
    """the RCCL transport on MORE than one physical GPU (the 1-GPU test box skips this; the first multi-GPU box
    that runs the suite checks it): ncclCommInitAll over the devices, ncclAllGather of the records, ncclAllReduce of the EI
    sums in the 2-D partition -- bit-equal to the one-GPU handle."""
    from spearmint_amd import engine as eng_mod
    ndev = eng_mod.device_count()
    if ndev < 2:
        pytest.skip("needs at least two GPUs (found %d)" % ndev)
    n = 4 if ndev >= 4 else 2
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(300, 20011, 6, 8, 123, per_sec=True)
    one = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ops = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
    me = MultiEngine(list(range(n)))
    try:
        assert me.transport() == "rccl"
        many = me.ei_grid(comp, vals, cand, hypers, want_draws=True)
        assert many[0] == one[0] and many[1] == one[1]
        assert np.array_equal(many[2], one[2]) and np.array_equal(many[3], one[3])
        mps = me.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
        assert mps[0] == ops[0] and np.array_equal(mps[3], ops[3])
        eng.set_observations(comp, vals); eng.set_hypers(hypers)
        me.set_observations(comp, vals); me.set_hypers(hypers)
        assert np.array_equal(me.gp_logprob(), eng.gp_logprob())       # draws sharded over the devices
        # the 2-D partition: one ncclAllReduce(SUM) of the EI-sum vector over xGMI
        me.set_partition(2)
        me.set_hypers(hypers); me.set_candidates(cand); me.factor(); me.ei_run()
        idx, mean, blocks = _emulate_2d(eng, comp, vals, cand, hypers, n, 2)
        assert me.best()[0] == idx == one[0]
        assert np.allclose(me.ei_mean(), mean, rtol=1e-14, atol=0)       # (the reduction tree's order for n = 4)
        assert np.array_equal(me.ei_draws(), blocks)
        if n == 2:
            assert np.array_equal(me.ei_mean(), mean)
    finally:
        me.close()


@pytest.mark.timeout(900)      # (the first test that maps /opt/rocm's librccl, 570 MB: 100 s on a box with a cold page cache)
def test_rccl_communicator_of_one_device(eng):
    comp, cand, vals, hypers = synthetic_problem(200, 3001, 5, 4, 61)
    one = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    me = Engine(devices=[0])
    try:
        assert me.transport() == "rccl"
        many = me.ei_grid(comp, vals, cand, hypers, want_draws=True)
        assert many[0] == one[0] and many[1] == one[1]
        assert np.array_equal(many[2], one[2]) and np.array_equal(many[3], one[3])
    finally:
        me.close()


def test_three_engines_on_one_gpu_match_single(eng):
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(200, 3001, 5, 4, 61, per_sec=True)
    one = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ops = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
    me = MultiEngine([0, 0, 0])
    try:
        assert me.transport() == "host" and eng.transport() == "none"
        many = me.ei_grid(comp, vals, cand, hypers, want_draws=True)
        assert many[0] == one[0] and many[1] == one[1]
        assert np.array_equal(many[2], one[2]) and np.array_equal(many[3], one[3])
        mps = me.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
        assert mps[0] == ops[0] and np.array_equal(mps[3], ops[3])
        # ties across shards go to the lowest global index; NaN wins
        c2 = cand.copy(); c2[2500] = c2[10]
        a = eng.ei_grid(comp, vals, c2, hypers); b = me.ei_grid(comp, vals, c2, hypers)
        assert a[0] == b[0] and np.array_equal(a[2], b[2])
        c2[2900, 0] = np.nan
        a = eng.ei_grid(comp, vals, c2, hypers); b = me.ei_grid(comp, vals, c2, hypers)
        assert a[0] == b[0] == 2900
        # fewer candidates than devices
        a = eng.ei_grid(comp, vals, cand[:2], hypers); b = me.ei_grid(comp, vals, cand[:2], hypers)
        assert a[0] == b[0] and np.array_equal(a[2], b[2])
        # building blocks across shards
        me.ei_grid(comp, vals, cand, hypers); eng.ei_grid(comp, vals, cand, hypers)
        assert np.array_equal(me.get_cross_cov(1, 900, 1500), eng.get_cross_cov(1, 900, 1500))
        assert np.array_equal(me.get_factor(2)[1], eng.get_factor(2)[1])
    finally:
        me.close()


def test_multi_shards_loglikelihood_draws_and_refinement_points(eng):
    comp, cand, vals, hypers = synthetic_problem(330, 400, 5, 7, 62)
    hypers[4, 2] = -1.0                               # one non-PD draw
    me = MultiEngine([0, 0, 0])
    try:
        for e in (eng, me):
            e.set_observations(comp, vals); e.set_hypers(hypers)
        ref = eng.gp_logprob()
        got = me.gp_logprob()                          # 7 draws over 3 engines: 3 + 2 + 2
        assert np.array_equal(got, ref) and got[4] == -np.inf
        assert me.not_pd_info()[0] == 4 == eng.not_pd_info()[0]
        with pytest.raises(np.linalg.LinAlgError):
            me.gp_logprob(raise_not_pd=True)
        # after a sharded log-likelihood the EI path must see the full hyper set again
        hypers[4, 2] = 1.0
        for e in (eng, me):
            e.set_hypers(hypers)
        lp = me.gp_logprob()
        assert np.array_equal(lp, eng.gp_logprob())
        me.set_candidates(cand); eng.set_candidates(cand)
        me.factor(); eng.factor(); me.ei_run(); eng.ei_run()
        assert me.best() == eng.best() and np.array_equal(me.ei_draws(), eng.ei_draws())
        pts = np.random.RandomState(3).rand(11, 5)
        f1, g1 = eng.ei_grad_batch(pts)
        f3, g3 = me.ei_grad_batch(pts)                 # 11 points over 3 engines
        assert np.array_equal(f1, f3) and np.array_equal(g1, g3)
    finally:
        me.close()


def test_chooser_with_ndev_uses_the_multi_handle(golden_dir, tmp_path):
    import numpy.random as npr
    from spearmint_amd.chooser import GPEIOptChooser
    g = np.load(os.path.join(golden_dir, "chooser_next.npz"))
    ch = GPEIOptChooser.init(str(tmp_path), "mcmc_iters=4,burnin=6,grid_subset=5,use_multiprocessing=0,"
                                            "gpu_refine=1,gpu_logprob=1,ndev=1")
    ch._eng = MultiEngine([0, 0])
    npr.seed(int(g["opt_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert isinstance(job, tuple) and job[0] == int(g["opt_index"])
    assert np.allclose(job[1], g["opt_point"], atol=1e-5)


# ---- bench.py, two ranks on the one GPU -----------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(900)
def test_bench_two_ranks_equals_one_rank_over_concatenated_candidates(eng):
    env = dict(os.environ, SPX_BENCH_BACKEND="gloo", SPX_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "c2", "--no-cpu-baseline",
           "--extra-steps", "1", "--c4-candidates", "30000", "--c5-candidates", "20000", "--hyper-shards", "2"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=850)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    line = [l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["c4"]["scaling"] == "strong"
    # weak headline: rank r scored its own shard; one rank over [shard0; shard1] must pick the same row
    w = bench.WORKLOADS["c2"]
    _, comp, vals, hypers, s0 = bench.weak_problem(w, 0)
    s1 = bench.weak_problem(w, 1)[4]
    idx, val, _, _ = eng.ei_grid(comp, vals, np.vstack((s0, s1)), hypers)
    assert (out["best_index"], out["best_ei"]) == (idx, val)
    # strong sub-records: the same grid whatever the number of ranks
    for name, M in (("c4", 30000), ("c5", 20000)):
        cfg = dict(bench.STRONG[name]); cfg["M"] = M
        prob, scomp, svals, shyp = bench.strong_problem(cfg)
        rows = bench.strong_rows(cfg, scomp, svals, 0, M)
        if cfg["per_sec"]:
            i1, v1, _, _ = eng.ei_per_sec_grid(scomp, svals, prob[4], rows, shyp, prob[5])
        else:
            i1, v1, _, _ = eng.ei_grid(scomp, svals, rows, shyp)
        assert (out[name]["best_index"], out[name]["best_ei"]) == (i1, v1)
        assert out[name]["config"]["candidates_total"] == M and out["%s_value" % name] > 0
    # the optional 2-D partition (here 2 draw shards x 1 candidate shard, one all-reduce of the EI sums): same winner,
    # mean EI equal up to the order of the sum over draws
    assert out["c4_2d"]["best_index"] == out["c4"]["best_index"]
    assert abs(out["c4_2d"]["best_ei"] - out["c4"]["best_ei"]) <= 1e-13 * abs(out["c4"]["best_ei"])
    # N > 1: at least 5 timed steps per sub-record, per-rank step times (a straggler shows) ...
    assert out["c4"]["steps"] >= 5 and out["c5"]["steps"] >= 5
    for key in ("c4", "c5"):
        rt = out[key]["rank_step_ms"]
        assert 0 < rt["min_ms_per_step"] <= rt["max_ms_per_step"] <= out[key]["ms_per_step"] * 1.05
    assert out["rank_step_ms"]["max_ms_per_step"] > 0
    # ... and every variant of the collective was attempted in this one run.  Two ranks on ONE device cannot form an
    # RCCL communicator: the library-collective variants must report that as an error string (not die, not hang),
    # leaving the other records intact -- on distinct GPUs they carry numbers and must agree on the winner.
    for key in ("c4_lib", "c5_lib", "c4_2d_lib"):
        rec = out[key]
        assert ("error" in rec and rec["error"]) or rec["best_index"] == out[key.split("_")[0]]["best_index"], rec
    assert "error" in out["c4_lib"] and ("nccl" in out["c4_lib"]["error"].lower() or "rccl" in out["c4_lib"]["error"].lower()
                                         or "communicator" in out["c4_lib"]["error"].lower())


@pytest.mark.timeout(900)
def test_bench_gpus_2_with_no_launcher_starts_two_ranks_on_libspx(eng):
    """`python bench.py --gpus 2` -- no torch.distributed.run around it, the form of the driver's
    command -- starts its own two ranks (here both on GPU 0, gloo), the line says n_gpus == ranks_seen == 2 (counted from
    the gathered table), and the winners are those of one rank over the same candidates."""
    env = dict(os.environ, SPX_BENCH_BACKEND="gloo", SPX_BENCH_SINGLE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c2",
           "--no-cpu-baseline", "--no-live-traffic", "--extra-steps", "1", "--c4-candidates", "30000", "--c5-candidates", "20000"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=850)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    out = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == out["ranks_seen"] == 2 and out["launcher"].startswith("self") and out["engine"] == "libspx"
    w = bench.WORKLOADS["c2"]
    _, comp, vals, hypers, s0 = bench.weak_problem(w, 0)
    s1 = bench.weak_problem(w, 1)[4]
    idx, val, _, _ = eng.ei_grid(comp, vals, np.vstack((s0, s1)), hypers)
    assert (out["best_index"], out["best_ei"]) == (idx, val)
    assert out["c4"]["ranks_seen"] == 2 and out["c5"]["ranks_seen"] == 2 and out["c4"]["n_gpus"] == 2
    # one rank, same command: the strong-scaling grids give the same winners
    cmd[cmd.index("--gpus") + 1] = "1"
    res1 = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=850)
    assert res1.returncode == 0, res1.stderr.decode()[-2000:]
    out1 = json.loads([l for l in res1.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out1["n_gpus"] == out1["ranks_seen"] == 1 and out1["launcher"].startswith("none")
    for name in ("c4", "c5"):
        assert (out[name]["best_index"], out[name]["best_ei"]) == (out1[name]["best_index"], out1[name]["best_ei"])
    # and a node with ONE GPU refuses --gpus 2 outright (no line with n_gpus 1): non-zero exit, no JSON
    from spearmint_amd import engine as eng_mod
    if eng_mod.device_count() < 2:
        env2 = {k: v for k, v in env.items() if k not in ("SPX_BENCH_BACKEND", "SPX_BENCH_SINGLE_DEVICE")}
        res2 = subprocess.run(cmd[:2] + ["--gpus", "2", "--steps", "1", "--warmup", "0"], env=env2, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert res2.returncode != 0 and not [l for l in res2.stdout.decode().splitlines() if l.startswith("{")]
        assert "refusing" in res2.stderr.decode()


# ---- spx_ei_grad_batch: the refinement objective ---------------------------------------------------
def test_ei_grad_batch_matches_reference_golden(eng, golden_dir):
    """Value + gradient of the reference's grad_optimize_ei_over_hypers (GPEIOptChooser.py:360-525,
    GPEIperSecChooser.py:322-434) as dumped from the reference itself."""
    g = np.load(os.path.join(golden_dir, "ei_grad.npz"))
    for tag in "ab":
        comp, vals, hypers, pts = g[tag + "_comp"], g[tag + "_vals"], g[tag + "_hypers"], g[tag + "_points"]
        eng.ei_grid(comp, vals, pts, hypers)             # leaves observations, draws and factors resident
        f, gr = eng.ei_grad_batch(pts)
        assert np.allclose(f, g[tag + "_f"], rtol=1e-7, atol=1e-300)
        assert np.allclose(gr, g[tag + "_g"], rtol=1e-6, atol=1e-9 * np.abs(g[tag + "_g"]).max())
    # pending branch: fantasies from the oracle's restatement of the first half of the branch
    comp, pend, vals, hypers, pts = g["p_comp"], g["p_pend"], g["p_vals"], g["p_hypers"], g["p_points"]
    H, P = hypers.shape[0], pend.shape[0]
    fb = [orc.fantasize(comp, pend, vals, hypers[h], g["p_randn"]) for h in range(H)]
    eng.set_observations(np.concatenate((comp, pend)), np.concatenate((vals, np.zeros(P))))
    eng.set_candidates(pts); eng.set_hypers(hypers); eng.factor()
    eng.set_fantasies(np.array([x[0] for x in fb]), np.array([x[1] for x in fb]))
    f, gr = eng.ei_grad_batch(pts)
    assert np.allclose(f, g["p_f"], rtol=1e-7, atol=1e-300)
    assert np.allclose(gr, g["p_g"], rtol=1e-6, atol=1e-9 * np.abs(g["p_g"]).max())
    # per second
    comp, vals, ld, hypers, th, pts = (g["s_comp"], g["s_vals"], g["s_log_durs"], g["s_hypers"],
                                       g["s_time_hypers"], g["s_points"])
    eng.ei_per_sec_grid(comp, vals, ld, pts, hypers, th)
    f, gr = eng.ei_grad_batch(pts)
    assert np.allclose(f, g["s_f"], rtol=1e-7, atol=1e-300)
    assert np.allclose(gr, g["s_g"], rtol=1e-6, atol=1e-9 * np.abs(g["s_g"]).max())


@pytest.mark.parametrize("N,D,H,seed", [(40, 2, 3, 51), (300, 7, 4, 52), (1000, 32, 3, 53)])
def test_ei_grad_batch_matches_oracle(eng, N, D, H, seed):
    comp, cand, vals, hypers = synthetic_problem(N, 50, D, H, seed)
    eng.ei_grid(comp, vals, cand, hypers)
    rs = np.random.RandomState(seed)
    pts = np.vstack((cand[:12], comp[np.argmin(vals)] + 1e-3 * rs.randn(4, D), rs.rand(5, D)))   # 21 points: 8 + 8 + 5
    f, g = eng.ei_grad_batch(pts)
    for i, x in enumerate(pts):
        f_ref, g_ref = orc.grad_optimize_ei_over_hypers(x, comp, vals, hypers)
        assert np.isclose(f[i], f_ref, rtol=1e-7, atol=1e-300)
        assert np.allclose(g[i], g_ref, rtol=1e-6, atol=1e-9 * max(np.abs(g_ref).max(), 1e-300))
    # a point's numbers do not depend on the rest of the batch; the single-point entry is the same call
    for i in (0, 7, 8, 20):
        fi, gi = eng.ei_grad_batch(pts[i:i + 1])
        assert fi[0] == f[i] and np.array_equal(gi[0], g[i])
        f1, g1 = eng.ei_grad(pts[i])
        assert f1 == f[i] and np.array_equal(g1, g[i])
    # and it is the gradient of what it says (central differences, reference scaling = 1/2)
    x = cand[7].copy()
    for d in range(min(D, 3)):
        e = np.zeros(D); e[d] = 1e-6
        fp, _ = eng.ei_grad_batch(np.vstack((x + e, x - e)))
        assert np.isclose(0.5 * (fp[0] - fp[1]) / 2e-6, g[7][d], rtol=2e-3, atol=1e-9)


def test_ei_grad_batch_with_fantasies_matches_oracle(eng):
    N, P, D, H, S = 300, 4, 6, 3, 100
    comp, cand, vals, hypers = synthetic_problem(N, 30, D, H, 71)
    rs = np.random.RandomState(71)
    pend = rs.rand(P, D)
    comp_pend = np.concatenate((comp, pend))
    fb = [orc.fantasize(comp, pend, vals, hypers[h], rs.randn(P, S)) for h in range(H)]
    fant = np.array([x[0] for x in fb]); bests = np.array([x[1] for x in fb])
    eng.set_observations(comp_pend, np.concatenate((vals, np.zeros(P))))
    eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    eng.set_fantasies(fant, bests)
    eng.ei_run()                                        # the EI pass the chooser runs first
    pts = np.vstack((cand[:9], comp[np.argmin(vals)] + 1e-3 * rs.randn(D)))
    f, g = eng.ei_grad_batch(pts)
    for i, x in enumerate(pts):
        f_ref, g_ref = 0.0, np.zeros(D)
        for h in range(H):
            e, gr = orc.grad_optimize_ei_fantasies(x, comp_pend, hypers[h], fant[h], bests[h])
            f_ref += e; g_ref = g_ref + gr
        assert np.isclose(f[i], f_ref, rtol=1e-7, atol=1e-300)
        assert np.allclose(g[i], g_ref, rtol=1e-6, atol=1e-9 * max(np.abs(g_ref).max(), 1e-300))


def test_more_than_128_fantasies(eng):
    """pending_samples above 128 (the reference accepts any): numpy's pairwise mean recursion."""
    N, P, D, H, S = 120, 3, 4, 2, 300
    comp, cand, vals, hypers = synthetic_problem(N, 500, D, H, 72)
    rs = np.random.RandomState(72)
    pend = rs.rand(P, D)
    comp_pend = np.concatenate((comp, pend))
    fb = [orc.fantasize(comp, pend, vals, hypers[h], rs.randn(P, S)) for h in range(H)]
    fant = np.array([x[0] for x in fb]); bests = np.array([x[1] for x in fb])
    eng.set_observations(comp_pend, np.concatenate((vals, np.zeros(P))))
    eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    eng.set_fantasies(fant, bests)
    eng.ei_run()
    ref = np.stack([orc.compute_ei_fantasies(comp_pend, cand, hypers[h], fant[h], bests[h]) for h in range(H)], axis=1)
    got = eng.ei_draws()
    ok = ref > 1e-280
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) <= 1e-6
    assert eng.best()[0] == orc.choose(ref)


@pytest.mark.timeout(1700)
def test_bench_in_process_mode_matches_the_default_mode(eng):
    """bench.py --in-process: the weak-scaling headline through one multi-device handle (the RCCL path of
    libspx; here a communicator of one device) must pick the same candidate as the default mode."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--in-process", "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--workload", "c2", "--c4-candidates", "30000", "--c5-candidates", "20000", "--hyper-shards", "2"]
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    out = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["config"]["transport"] == "rccl" and out["n_gpus"] == 1
    w = bench.WORKLOADS["c2"]
    _, comp, vals, hypers, s0 = bench.weak_problem(w, 0)
    idx, val, _, _ = eng.ei_grid(comp, vals, s0, hypers)
    assert (out["best_index"], out["best_ei"]) == (idx, val)
    # two engines on the one GPU (host transport): rank 1's shard appended
    cmd = cmd[:3] + ["--devices", "0,0"] + cmd[3:]
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    out = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    s1 = bench.weak_problem(w, 1)[4]
    idx, val, _, _ = eng.ei_grid(comp, vals, np.vstack((s0, s1)), hypers)
    assert out["config"]["transport"] == "host" and (out["best_index"], out["best_ei"]) == (idx, val)
    # the strong-scaling lines of this mode: the full grids sharded by the library over the handle's devices, and C4
    # in the library's 2-D partition (2 draw shards x 1 candidate shard)
    for name, M in (("c4", 30000), ("c5", 20000)):
        cfg = dict(bench.STRONG[name]); cfg["M"] = M
        prob, scomp, svals, shyp = bench.strong_problem(cfg)
        rows = bench.strong_rows(cfg, scomp, svals, 0, M)
        if cfg["per_sec"]:
            i1, v1, _, _ = eng.ei_per_sec_grid(scomp, svals, prob[4], rows, shyp, prob[5])
        else:
            i1, v1, _, _ = eng.ei_grid(scomp, svals, rows, shyp)
        assert (out[name]["best_index"], out[name]["best_ei"]) == (i1, v1) and out[name]["scaling"] == "strong"
        assert out[name]["stages_ms_max_over_devices"]["predict_gemm"] > 0
    assert out["c4_2d"]["best_index"] == out["c4"]["best_index"] and "ncclAllReduce" in out["c4_2d"]["collective"]
    assert abs(out["c4_2d"]["best_ei"] - out["c4"]["best_ei"]) <= 1e-13 * abs(out["c4"]["best_ei"])


@pytest.mark.timeout(900)
def test_process_group_communicator_attached_to_a_handle(eng):
    """spx_comm_attach: the one-process-per-GPU form of the library's collective (here a group of one rank):
    ei_run ends with ncclAllGather + the argmax rule, best() is the global winner."""
    comp, cand, vals, hypers = synthetic_problem(150, 2500, 5, 4, 91)
    ref = eng.ei_grid(comp, vals, cand, hypers)
    e = Engine(0)
    try:
        uid = e.comm_unique_id()
        assert len(uid) == 128
        e.comm_attach(uid, 1, 0)
        e.set_observations(comp, vals); e.set_hypers(hypers); e.factor()
        e.set_candidates(cand[1000:], index_base=1000)          # a shard with a global offset
        e.ei_run()
        sub = eng.ei_grid(comp, vals, cand[1000:], hypers)
        assert e.best() == (sub[0] + 1000, sub[1])
        e.set_candidates(cand)
        e.ei_run()
        assert e.best() == (ref[0], ref[1])
        with pytest.raises(ValueError):
            e.comm_attach(uid[:10], 1, 0)
    finally:
        e.close()
    # bench.py with the library collective (one rank)
    env = dict(os.environ, SPX_BENCH_COLLECTIVE="lib")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--workload", "c2",
           "--no-cpu-baseline", "--skip-extras"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    out = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert "ncclAllGather" in out["config"]["collective"]
    w = bench.WORKLOADS["c2"]
    _, c2, v2, h2, s0 = bench.weak_problem(w, 0)
    idx, val, _, _ = eng.ei_grid(c2, v2, s0, h2)
    assert (out["best_index"], out["best_ei"]) == (idx, val)


# ---- 2-D partition inside the library (spx_set_partition; SURVEY.md 8(e) "hypers x candidates") -----------------------
def _emulate_2d(eng, comp, vals, cand, hypers, n, ph, per_sec=None):
    """What dist.shard_2d + dist.allreduce_ei_sums compute for n ranks: every (candidate shard, draw shard) block on a
    plain one-GPU engine, np.sum over the block's draws, the zero-padded M-vectors added in rank order."""
    from spearmint_amd import dist as sd
    M, H = cand.shape[0], hypers.shape[0]
    full = np.zeros(M)
    blocks = np.zeros((M, H))
    for r in range(n):
        (c0, c1), (h0, h1) = sd.shard_2d(M, H, n, r, ph)
        if per_sec is None:
            ei = eng.ei_grid(comp, vals, cand[c0:c1], hypers[h0:h1], want_draws=True)[3]
        else:
            ei = eng.ei_per_sec_grid(comp, vals, per_sec[0], cand[c0:c1], hypers[h0:h1], per_sec[1][h0:h1], want_draws=True)[3]
        blocks[c0:c1, h0:h1] = ei
        part = np.zeros(M)
        part[c0:c1] = np.sum(ei, axis=1)
        full = part if r == 0 else full + part
    mean = full / float(H)
    return int(np.argmax(mean)), mean, blocks


@pytest.mark.parametrize("devs,ph", [([0, 0], 2), ([0, 0, 0, 0], 2), ([0, 0], 1), ([0, 0, 0, 0, 0, 0], 3)])
def test_2d_partition_in_the_multi_handle(eng, devs, ph):
    """draws x candidates over repeated device ids (host transport): the library shards both, keeps the per-device EI
    sums on the device, reduces them with ONE all-reduce(SUM) and takes the argmax there -- equal, bit for bit, to the
    host-side scheme of dist.allreduce_ei_sums (two contributions per candidate: the sum is exact in any order)."""
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(150, 2111, 5, 7, 97, per_sec=True)
    me = MultiEngine(devs)
    try:
        me.set_partition(ph)
        me.set_observations(comp, vals); me.set_hypers(hypers); me.set_candidates(cand)
        me.factor(); me.ei_run()
        idx, mean, blocks = _emulate_2d(eng, comp, vals, cand, hypers, len(devs), ph)
        if ph > 1:
            assert me.best() == (idx, mean[idx])
            assert np.array_equal(me.ei_mean(), mean)
        assert np.array_equal(me.ei_draws(), blocks)          # every (candidate, draw) evaluated once, same bits as 1 GPU
        one = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        assert np.array_equal(blocks, one[3]) and me.best()[0] == one[0]
        assert np.allclose(me.ei_mean(), one[2], rtol=1e-14, atol=0)
        if ph > 1:
            # per second: the time model's hyper rows are sharded with the draws
            me.set_time_model(log_durs, th)
            me.factor(); me.ei_run(1)
            idx2, mean2, _ = _emulate_2d(eng, comp, vals, cand, hypers, len(devs), ph, per_sec=(log_durs, th))
            assert me.best() == (idx2, mean2[idx2]) and np.array_equal(me.ei_mean(), mean2)
            me.set_time_model(None, None)
            me.factor()
            # the building blocks follow the partition: a draw's factor lives on the device that owns the draw
            eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.set_candidates(cand); eng.factor()
            for d in (0, hypers.shape[0] - 1):
                assert np.array_equal(me.get_factor(d)[1], eng.get_factor(d)[1])
                assert np.array_equal(me.get_cross_cov(d, 700, 900), eng.get_cross_cov(d, 700, 900))
            # a sharded log-likelihood in between must give the devices their draw shards back
            lp = me.gp_logprob()
            assert np.array_equal(lp, eng.gp_logprob())
            me.factor(); me.ei_run()
            assert me.best() == (idx, mean[idx])
            with pytest.raises(ValueError):
                me.ei_grad_batch(np.full((1, 5), 0.5))
            # not-PD: the failing draw is reported in global numbering
            bad = hypers.copy(); bad[5, 2] = -1.0
            me.set_hypers(bad)
            with pytest.raises(np.linalg.LinAlgError):
                me.factor()
            assert me.not_pd_info()[0] == 5
            # back to candidates only: the plain multi-handle behaviour
            me.set_partition(1)
            me.set_hypers(hypers); me.set_candidates(cand); me.factor(); me.ei_run()
            assert me.best() == (one[0], one[1]) and np.array_equal(me.ei_mean(), one[2])
        with pytest.raises(ValueError):
            me.set_partition(len(devs) + 1)
    finally:
        me.close()


@pytest.mark.timeout(900)
def test_allreduce_collective_on_an_attached_communicator(eng):
    """spx_set_partition on a handle with an RCCL communicator (one rank here): spx_ei_run ends with ncclAllReduce(SUM)
    of the M_total-vector instead of the all-gather of records; with one rank the result is the plain run's."""
    comp, cand, vals, hypers = synthetic_problem(150, 2500, 5, 4, 91)
    ref = eng.ei_grid(comp, vals, cand, hypers)
    e = Engine(0)
    try:
        e.comm_attach(e.comm_unique_id(), 1, 0)
        e.set_partition(1, cand.shape[0], hypers.shape[0])
        e.set_observations(comp, vals); e.set_hypers(hypers); e.factor()
        e.set_candidates(cand)
        e.ei_run()
        assert e.best() == (ref[0], ref[1]) and np.array_equal(e.ei_mean(), ref[2])
        # a shard with an offset inside a larger grid: zeros elsewhere, the index is global
        e.set_candidates(cand[1000:1800], index_base=1000)
        e.ei_run()
        sub = eng.ei_grid(comp, vals, cand[1000:1800], hypers)
        assert e.best() == (sub[0] + 1000, sub[1]) and np.array_equal(e.ei_mean(), sub[2])
        e.set_candidates(cand[2000:], index_base=2100)           # does not fit M_total
        with pytest.raises(ValueError):
            e.ei_run()
        e.set_partition(1, 0, 0)                                 # back to the all-gather of records
        e.set_candidates(cand)
        e.ei_run()
        assert e.best() == (ref[0], ref[1])
    finally:
        e.close()


def test_multi_handle_state_after_a_sharded_loglikelihood_and_options(eng):
    """(1) spx_not_pd_info after a sharded log-likelihood scans only the devices that took part -- an idle
    device's older non-PD result must not surface; (2) changing covar invalidates the multi handle's results;
    (3) spx_get_timings is the per-stage maximum over the devices."""
    comp, cand, vals, hypers = synthetic_problem(200, 600, 4, 7, 63)
    me = MultiEngine([0, 0, 0])
    try:
        me.set_observations(comp, vals)
        bad = hypers.copy(); bad[4, 2] = -1.0                    # 7 draws over 3 devices: draw 4 lands on device 1
        me.set_hypers(bad)
        assert me.gp_logprob()[4] == -np.inf and me.not_pd_info()[0] == 4
        me.set_hypers(hypers[:1])                                # one row: only device 0 takes part
        lp = me.gp_logprob(raise_not_pd=True)                    # used to raise a spurious LinAlgError (draw 1 + 0 ...)
        assert np.isfinite(lp[0]) and me.not_pd_info()[0] == -1
        me.set_hypers(hypers); me.set_candidates(cand); me.factor()
        me.set_option("timing", 1)
        me.ei_run()
        tm = me.timings()
        assert tm["predict_gemm"][0] > 0 and tm["ei_run_total"][0] >= tm["predict_gemm"][0]
        me.set_option("timing", 0)
        assert me.best()[0] >= 0
        me.set_covar("Matern32")
        with pytest.raises(ValueError):
            me.best()                                            # the factorisation and the winner are stale
        me.set_covar("Matern52")
        # a failed set_candidates leaves no half-updated shard table behind
        with pytest.raises(ValueError):
            me.set_candidates(np.zeros((10, 3)))                 # wrong D
        with pytest.raises(ValueError):
            me.ei_run()
        me.set_candidates(cand); me.factor(); me.ei_run()
        one = eng.ei_grid(comp, vals, cand, hypers)
        assert me.best() == (one[0], one[1])
    finally:
        me.close()
