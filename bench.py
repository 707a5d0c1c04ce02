#!/usr/bin/env python3
"""bench.py -- EI-candidate evaluations / second of the GP-EI hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (N > 1, no launcher: starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Whichever way it is started, the JSON line's `n_gpus` is the number of ranks that TOOK PART in the collective
(`ranks_seen`: the size of the communicator the all-gather of the 16-byte records ran on -- one record per rank in its
table; a check of the launch plumbing, not a separate measurement) and the run fails instead of printing a line
whose `n_gpus` differs from --gpus.

A "step" is one full pass of the hot path over one batch of synthetic input,
i.e. what one chooser.next() hands to the GPU: for every hyper-parameter draw
build K(X,X), factor it, then K(X*,X), the triangular solve, predictive
mean/variance and EI for every candidate of this rank's shard, the MCMC mean,
the local argmax, and (N > 1) the single collective that picks the global
best.  Inputs (observations, candidate shard, hyper draws) are resident in HBM
before the timed region starts (spx_set_* are outside it).

Headline workload (BASELINE.json configs[2], the single-GPU configuration the
metric's target is quoted on): synthetic 32-D, N_obs=2048, 200 000 candidates
per GPU, mcmc_iters=20, fp64.  Per-GPU work is fixed as N grows ("weak"
scaling): N GPUs score N x 200 000 candidates of one grid.

The same run also times, as sub-records "c4" and "c5", the two multi-GPU
configurations of BASELINE.json at their FULL size with the candidate grid split
over the WORLD_SIZE ranks ("strong" scaling: total work fixed):
  c4  N_obs=2048, 32-D, 1 000 000 candidates, mcmc_iters=20
  c5  GPEIperSec dual GP, N_obs=1024, 16-D, 500 000 candidates, mcmc_iters=20
so `c4.value` at --gpus 1, 2, 4, 8 is the north-star scaling curve.  The
candidate grid of these is generated in fixed seeded blocks, so every N scores
the same grid and must report the same best_index.  With N > 1 each of them is
timed (>= 5 steps) with BOTH forms of the collective -- the 16-byte records
through torch.distributed (`c4`, `c5`) and through libspx's own ncclAllGather
on the handle's stream (`c4_lib`, `c5_lib`; spx_comm_attach) -- and, with
--hyper-shards P_h, C4 also in the 2-D partition (`c4_2d`: host sums + torch
all-reduce; `c4_2d_lib`: spx_set_partition, ncclAllReduce of the device-resident
EI-sum vector).  Every record carries the per-rank step times (max / min); a
variant that cannot run records its error string instead of ending the run, and
all that ran must agree on the winner.  `--in-process` runs the same headline
and strong-scaling lines through ONE multi-device handle (spx_create_multi).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel
(k_predict_gemm_tri: beta = L^-1 K* as an fp64 MFMA GEMM over the lower triangle of
L^-1, with the variance/mean reduction fused in its epilogue): achieved = algorithmic flops per launch
(N^2 + 4N per (candidate, draw) evaluation) / the kernel's mean
launch duration, measured with HIP events on the library's own stream over a
second timed pass of the same steps (the headline pass runs without per-launch
events).  `roofline_hbm` is the second regime of SURVEY.md 8(d): the HBM-side
stages (the K(X*,X) write stream, EI finalize) against 8 TB/s.  `cpu_baseline`
is the reference's OWN GPEIChooser.compute_ei loop (kind "reference": the lib2to3-converted
S/chooser/GPEIChooser.py out of oracle/_ref/chooser_py3.zip) timed on this host's cores: 20 000
candidates x all draws, one warm-up on a tenth of them, median of two runs; the numpy/scipy
oracle's port of the same loop is timed once beside it (`port_value`) and the two EI matrices are
compared bit for bit (`port_equals_reference`).  Without the archive the port alone is reported
(kind "port").
"""
from __future__ import print_function

import argparse
import subprocess
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from spearmint_amd import dist as spx_dist  # noqa: E402
from spearmint_amd.engine import Engine, FLAG_PER_SEC  # noqa: E402
from spearmint_amd.synthetic import synthetic_problem  # noqa: E402

WORKLOADS = {
    # name: N_obs, candidates per GPU, D, mcmc_iters, per_sec
    "c2": dict(N=256, M=20000, D=8, H=10, per_sec=False,
               desc="C2: synthetic 8D, N_obs=256, 20k candidates/GPU, mcmc_iters=10"),
    "c3": dict(N=2048, M=200000, D=32, H=20, per_sec=False,
               desc="C3: synthetic 32D, N_obs=2048, 200k candidates/GPU, mcmc_iters=20"),
    "c4": dict(N=2048, M=125000, D=32, H=20, per_sec=False,
               desc="C4 shard: synthetic 32D, N_obs=2048, 125k candidates/GPU (1M over 8), mcmc_iters=20"),
    "c5": dict(N=1024, M=62500, D=16, H=20, per_sec=True,
               desc="C5 shard: GPEIperSec dual GP, 16D, N_obs=1024, 62.5k candidates/GPU (500k over 8), mcmc_iters=20"),
}
# full-size strong-scaling configurations (total candidates split over the ranks)
STRONG = {
    "c4": dict(N=2048, M=1000000, D=32, H=20, per_sec=False, seed=4000,
               desc="C4: synthetic 32D, N_obs=2048, 1M candidates split over the ranks, mcmc_iters=20"),
    "c5": dict(N=1024, M=500000, D=16, H=20, per_sec=True, seed=5000,
               desc="C5: GPEIperSec dual GP, 16D, N_obs=1024, 500k candidates split over the ranks, mcmc_iters=20"),
}
STRONG_BLOCK = 62500           # candidate rows per seeded block (1M = 16 blocks, 500k = 8)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD CDNA4 spec; = vector fp64 peak)
FP64_MFMA_MEASURED_TFLOPS = 77.5  # scripts/ubench_f64.hip on this pool: v_mfma_f64_16x16x4_f64, VGPR accumulators
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def pmc_traffic(workload):
    """HBM bytes per k_predict_gemm launch from the committed rocprofv3 PMC passes of this
    same command (profiles/r0X_<workload>_rocprof_summary.json, written by
    scripts/pmc_summary.py: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, FETCH_SIZE
    doubled per the gfx950 correction of MI355X_MICROARCH.md).  None when no profile exists.
    The fallback of live_traffic() below."""
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s_rocprof_summary.json" % (rnd, workload))
        try:
            with open(path) as fh:
                kernels = json.load(fh)["kernels"]
            name = [n for n in kernels if n.startswith("k_predict_gemm")][0]   # template args are part of the name
            return float(kernels[name]["hbm_bytes_per_launch_corrected"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def live_traffic(workload, timeout_s=240):
    """`roofline.traffic` measured IN this run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
    pass -- MI355X_MICROARCH.md, rocprofv3 section; no trace domain beside them) over a one-step child of this very
    command, counters of the dominant kernel only, per-launch mean, FETCH_SIZE doubled (the guide's gfx950 correction),
    WRITE_SIZE as it is.  (bytes, source text), or (None, reason) when rocprofv3 is missing or a pass fails -- the caller
    then falls back to the committed profile."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    got = {}
    work = tempfile.mkdtemp(prefix="spx_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "0",
             "--no-cpu-baseline", "--skip-extras", "--no-live-traffic", "--event-steps", "0"]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--kernel-include-regex", "k_predict_gemm", "--output-format", "csv", "-d", out,
                   "--"] + child
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, 9)          # the group this call started, nothing else
                proc.wait()
                return None, "rocprofv3 --pmc %s timed out" % counter
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for f in files:
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter and "k_predict_gemm" in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, "rocprofv3 --pmc %s produced no rows (rc %s)" % (counter, proc.returncode)
            got[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    traffic = (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0
    return traffic, ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over a one-step child of this "
                     "command, %d + %d k_predict_gemm launches, per-launch mean, FETCH_SIZE x 2 (gfx950 correction), KiB -> bytes"
                     % (got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]))


def _reference_chooser(D, H):
    """The reference's OWN GPEIChooser (S/chooser/GPEIChooser.py), converted for Python 3 by lib2to3: imported from
    /root/reference where that exists (build container), else from the archive build() wrote (oracle/_ref/chooser_py3.zip,
    shipped with the tree).  None when neither is there."""
    from oracle import ref_py3
    mods = ref_py3.load() if ref_py3.available() else ref_py3.load_shipped()
    if mods is None:
        return None
    import tempfile
    ch = mods["GPEIChooser"].GPEIChooser(tempfile.mkdtemp(prefix="spx_refbase_"), mcmc_iters=H)
    ch.D = D
    return ch


def cpu_baseline(w, m_cpu=20000, reps=2):
    """The CPU figure beside the GPU one (SURVEY.md 8(d) / BASELINE.md section 5: same N, D, H; M_cpu = 20 000
    candidates; one warm-up, then timed repetitions, median reported), on this box's host cores:

      kind "reference"  the reference's own arithmetic -- its GPEIChooser.compute_ei (GPEIChooser.py:178-208) called
                        once per hyper-parameter draw into overall_ei[:, draw], then np.argmax(np.mean(...)) exactly as
                        next() does (:143-153), the draws being the benchmark's fixed ones instead of sample_hypers';
      kind "port"       the numpy/scipy oracle's restatement of the same (the only one available without the archive);
                        always timed too (one repetition) and reported beside the reference as `port_value`."""
    from oracle import gp_ei_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        threads = max([p.get("num_threads", 1) for p in pools] or [1])
        blas = ", ".join(sorted(set("%s %s" % (p.get("internal_api", "?"), p.get("version", "")) for p in pools)))
    except Exception:
        threads, blas = os.cpu_count() or 1, "unknown"
    import scipy
    N, D, H = w["N"], w["D"], w["H"]
    m_cpu = int(min(m_cpu, w["M"]))
    comp, cand, vals, hypers = synthetic_problem(N, m_cpu, D, H, 3000)[:4]
    m_warm = max(256, m_cpu // 10)

    def port(c):
        ei = orc.ei_grid_chunked(comp, c, vals, hypers, chunk=20000)
        return orc.choose(ei), ei

    ref = _reference_chooser(D, H)
    pend = np.zeros((0, D))

    def reference(c):
        overall_ei = np.zeros((c.shape[0], H))
        for h in range(H):
            ref.mean, ref.noise, ref.amp2, ref.ls = hypers[h, 0], hypers[h, 1], hypers[h, 2], hypers[h, 3:].copy()
            overall_ei[:, h] = ref.compute_ei(comp, pend, c, vals)
        return int(np.argmax(np.mean(overall_ei, axis=1))), overall_ei

    def timed(fn, n):
        fn(cand[:m_warm])                       # warm-up: BLAS threads, page faults
        times, res = [], None
        for _ in range(max(1, n)):
            t0 = time.time()
            res = fn(cand)
            times.append(time.time() - t0)
        return float(np.median(times)), times, res

    out = {"unit": "EI evals/s", "cores": int(threads), "host_cpus": os.cpu_count(), "blas": blas,
           "numpy": np.__version__, "scipy": scipy.__version__}
    dt_p, times_p, (ip, eip) = timed(port, 1 if ref is not None else reps)
    port_txt = ("oracle.ei_grid_chunked (numpy/scipy restatement of GPEIChooser.compute_ei x H + argmax(mean)), "
                "N_obs=%d, D=%d, mcmc_iters=%d, %d candidates; warm-up on %d candidates, median of %d runs (%.1f s each)"
                % (N, D, H, m_cpu, m_warm, len(times_p), dt_p))
    if ref is None:
        out.update({"value": m_cpu * H / dt_p, "kind": "port", "reps_s": [round(t, 2) for t in times_p], "sample": port_txt})
        return out
    dt_r, times_r, (ir, eir) = timed(reference, reps)
    out.update({"value": m_cpu * H / dt_r, "kind": "reference", "reps_s": [round(t, 2) for t in times_r],
                "port_value": m_cpu * H / dt_p, "port_reps_s": [round(t, 2) for t in times_p],
                "port_equals_reference": bool(ip == ir and np.array_equal(eip, eir)),
                "sample": "the reference's own GPEIChooser.compute_ei (lib2to3-converted S/chooser/GPEIChooser.py:178-208 "
                          "+ gp.py) once per draw into overall_ei, then np.argmax(np.mean(overall_ei, axis=1)) (:143-153), "
                          "N_obs=%d, D=%d, mcmc_iters=%d, %d candidates; warm-up on %d candidates, median of %d runs "
                          "(%.1f s each); port_value: %s" % (N, D, H, m_cpu, m_warm, len(times_r), dt_r, port_txt)})
    return out


def platform_info(eng, device):
    """What the line was measured on (two boxes of one pool ran different ROCm / RCCL builds and differed by 2 %):
    HIP runtime / driver versions and the device's clocks as the runtime reports them (through libspx: spx_get_stat), the
    ROCm release of the image, torch's HIP build, and the clocks rocm-smi shows at the end of the run (bounded, optional)."""
    import subprocess
    info = {}
    if eng is not None:
        for key in ("hip_runtime_version", "hip_driver_version", "clock_khz", "mem_clock_khz", "wall_clock_khz", "n_cu",
                    "l2_bytes", "mem_bus_bits"):
            try:
                info[key] = int(eng.stat(key))
            except Exception as ex:
                info[key] = "unavailable: %s" % ex
        if isinstance(info.get("clock_khz"), int):
            info["gpu_sclk_mhz_max"] = info["clock_khz"] / 1e3
        if isinstance(info.get("mem_clock_khz"), int):
            info["gpu_mclk_mhz_max"] = info["mem_clock_khz"] / 1e3
    try:
        info["rocm_version"] = open("/opt/rocm/.info/version").read().strip()
    except Exception:
        info["rocm_version"] = None
    try:
        import torch
        info["torch"] = torch.__version__
        info["torch_hip"] = torch.version.hip
        info["device_name"] = torch.cuda.get_device_name(device)
    except Exception:
        pass
    try:
        txt = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--json"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, timeout=20).stdout.decode()
        card = list(json.loads(txt).values())[0]
        info["rocm_smi_clocks_after_run"] = {k: v for k, v in card.items() if "clk" in k.lower() or "clock" in k.lower()}
    except Exception as ex:
        info["rocm_smi_clocks_after_run"] = "unavailable: %s" % type(ex).__name__
    info["host_cpus"] = os.cpu_count()
    return info


def next_baseline(N=256, M=20000, D=8, mcmc_iters=10, burnin=10, seed=3, with_reference=True):
    """The cpu_baseline leg END TO END: one whole `GPEIOptChooser.next()` -- slice sampling of the
    hyper-parameters (burn-in + mcmc_iters draws), both EI passes over the grid, the L-BFGS-B refinement of the best 20
    candidates -- by the reference's OWN chooser (S/chooser/GPEIOptChooser.py:217-328, lib2to3-converted, numpy/scipy on this
    host's cores, use_multiprocessing=0) and by ours on libspx, from the same seeded state at a size the reference finishes
    (N_obs=256, 20 000 grid candidates, 8-D, mcmc_iters=10).  The two proposals must be the same point.  Ours is timed twice:
    a cold first call (library load, HIP context, buffer allocation) and a second chooser object on the warm process."""
    import tempfile
    import numpy.random as npr
    from oracle import ref_py3
    from spearmint_amd.chooser import GPEIOptChooser as ours_mod
    mods = (ref_py3.load() if ref_py3.available() else ref_py3.load_shipped()) if with_reference else None
    comp, cand, vals, _ = synthetic_problem(N, M, D, 1, 9)
    grid = np.vstack((comp, cand))
    values = np.concatenate((vals, np.full(M, np.nan)))
    durations = np.ones(N + M)
    complete, candidates, pending = np.arange(N), np.arange(N, N + M), np.array([], dtype=int)
    args = "mcmc_iters=%d,burnin=%d,grid_subset=20,use_multiprocessing=0" % (mcmc_iters, burnin)

    def run(mod, extra=""):
        ch = mod.init(tempfile.mkdtemp(prefix="spx_next_"), args + extra)
        npr.seed(seed)
        t0 = time.time()
        job = ch.next(grid, values, durations, candidates, pending, complete)
        run.last = ch
        return time.time() - t0, job

    def show(job):
        return {"index": int(job[0]), "point": [float(v) for v in job[1]]} if isinstance(job, tuple) else {"index": int(job)}

    cold_s, job_a = run(ours_mod)
    warm_s, job_b = run(ours_mod)
    warm_s2, job_b2 = run(ours_mod)
    ch = run.last
    st = dict(ch.sampler_stats)
    by_rows = st.pop("calls_by_rows")
    # the round-5 form of the same call on the same box, for attribution: the Python batched sampler at its fixed depth
    # and the threaded refinement (SPX_REFINE_THREADS=1) -- the same proposal
    os.environ["SPX_REFINE_THREADS"] = "1"
    try:
        run(ours_mod, ",sampler=python,lookahead=6,follow=0:0")
        r5_s, job_r5 = run(ours_mod, ",sampler=python,lookahead=6,follow=0:0")
    finally:
        os.environ.pop("SPX_REFINE_THREADS", None)
    out = {"what": "one GPEIOptChooser.next() call: %d burn-in + %d slice-sampled draws, two EI passes over the grid, L-BFGS-B "
                   "refinement of 20 candidates" % (burnin, mcmc_iters),
           "config": {"N_obs": N, "grid_candidates": M, "D": D, "chooser_args": args, "seed": seed},
           "ours": {"cold_s": cold_s, "warm_s": min(warm_s, warm_s2), "warm_runs_s": [warm_s, warm_s2], "proposal": show(job_b),
                    "engine": "libspx (HIP, fp64)",
                    "sampler": "spx_sample_hypers (native: C++ control flow + numpy MT19937 stream inside libspx), lock-step L-BFGS-B",
                    "speculation_depth": getattr(ch, "_depth_info", None), "sampler_stats": st,
                    "loglik_calls_by_rows": {str(r): c for r, c in enumerate(by_rows) if c},
                    "round5_form_warm_s": r5_s, "round5_form": "sampler=python,lookahead=6,follow=0:0 + threaded refinement",
                    "round5_form_same_proposal": show(job_r5) == show(job_b)},
           "ours_repeatable": show(job_a) == show(job_b) == show(job_b2)}
    if mods is None:
        out["reference"] = None
        return out
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    ref_s, job_r = run(mods["GPEIOptChooser"])
    out["reference"] = {"s": ref_s, "proposal": show(job_r), "cores": int(threads), "host_cpus": os.cpu_count(),
                        "what": "the reference's own GPEIOptChooser.next() (lib2to3-converted S/chooser/GPEIOptChooser.py, numpy/scipy)"}
    same = isinstance(job_b, tuple) == isinstance(job_r, tuple) and show(job_b)["index"] == show(job_r)["index"]
    if same and isinstance(job_b, tuple):
        same = bool(np.allclose(job_b[1], job_r[1], rtol=0, atol=1e-6))
    out["same_proposal"] = bool(same)
    out["speedup_warm"] = ref_s / warm_s
    return out


def emit_line(out):
    """Rank 0: everything C stdio still holds goes out FIRST (librccl prints a version banner through it, which a pipe
    buffers until the process ends), then the one JSON line.  The other ranks' stdout was sent to stderr when they started
    (main): under a launcher all ranks share one stdout.  The process then ends normally (a profiler's exit handlers run)."""
    import ctypes
    if out is not None:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


def engine_class():
    """The engine the bench drives: spearmint_amd.engine.Engine (libspx.so, no fallback).  TEST HOOK, never set by the
    driver: SPX_BENCH_ENGINE="module:Class" substitutes another class with the same methods, so that the launch / rank /
    collective plumbing of this file can be exercised where no GPU exists (tests/standin_engine.py); the JSON line then
    names it under "engine" and such a line is not a measurement."""
    spec = os.environ.get("SPX_BENCH_ENGINE", "")
    if not spec:
        return Engine, "libspx"
    import importlib
    mod, _, cls = spec.partition(":")
    return getattr(importlib.import_module(mod), cls or "Engine"), spec


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves --
    re-exec this very command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one process per
    GPU, rendezvous on 127.0.0.1, a free port) -- so that the one command the driver runs for the scaling curve yields
    N ranks, not one.  Refuses (non-zero exit, no JSON line) when the node has fewer than N devices: a line whose
    n_gpus differs from --gpus is never printed."""
    n = int(args.gpus)
    single = bool(os.environ.get("SPX_BENCH_SINGLE_DEVICE"))
    if not single and not os.environ.get("SPX_BENCH_ENGINE"):
        from spearmint_amd import engine as _e
        have = _e.device_count()
        if have < n:
            raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s); refusing to run fewer ranks than asked "
                             "for (SPX_BENCH_SINGLE_DEVICE=1 puts all ranks on device 0: a plumbing test, not a measurement)"
                             % (n, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, SPX_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes needs dmabuf IPC on this pool
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def weak_problem(w, rank):
    """Observations / hyper draws (identical on every rank) and this rank's own candidate shard of
    the weak-scaling headline workload."""
    N, M, D, H = w["N"], w["M"], w["D"], w["H"]
    prob = synthetic_problem(N, 16, D, H, 1000 * 3, near=0, per_sec=w["per_sec"])
    comp, vals, hypers = prob[0], prob[2], prob[3]
    shard = synthetic_problem(N, M, D, 1, 1000 * 3 + 17 * rank, near=(10 if rank == 0 else 0))[1]
    if rank == 0:  # jittered copies of the incumbent, as GPEIOptChooser.py:236-238
        inc = comp[np.argmin(vals)]
        shard[:10] = np.clip(inc + 1e-3 * np.random.RandomState(5).randn(10, D), 0, 1)
    return prob, comp, vals, hypers, shard


def strong_rows(cfg, comp, vals, lo, hi):
    """Rows [lo, hi) of the full candidate grid of a strong-scaling configuration.  The grid is
    defined block by block (STRONG_BLOCK rows, one seed per block), so it does not depend on how
    many ranks share it; its first 10 rows are jittered copies of the incumbent."""
    D = cfg["D"]
    out = np.empty((hi - lo, D))
    b0, b1 = lo // STRONG_BLOCK, (hi - 1) // STRONG_BLOCK
    for b in range(b0, b1 + 1):
        blk = np.random.RandomState(cfg["seed"] + 1 + b).rand(STRONG_BLOCK, D)
        if b == 0:
            inc = comp[np.argmin(vals)]
            blk[:10] = np.clip(inc + 1e-3 * np.random.RandomState(5).randn(10, D), 0, 1)
        a = max(lo, b * STRONG_BLOCK)
        z = min(hi, (b + 1) * STRONG_BLOCK)
        out[a - lo:z - lo] = blk[a - b * STRONG_BLOCK:z - b * STRONG_BLOCK]
    return out


def strong_problem(cfg):
    prob = synthetic_problem(cfg["N"], 16, cfg["D"], cfg["H"], cfg["seed"], near=0, per_sec=cfg["per_sec"])
    return prob, prob[0], prob[2], prob[3]


def main_in_process(args):
    """The weak-scaling headline through ONE multi-device handle: n x 200 000 candidates of one grid, sharded by
    the library over n GPUs driven from this process; the collective is libspx's own ncclAllGather."""
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    n = len(devs)
    w = WORKLOADS[args.workload]
    N, M, D, H = w["N"], w["M"], w["D"], w["H"]
    flags = FLAG_PER_SEC if w["per_sec"] else 0
    prob, comp, vals, hypers, shard0 = weak_problem(w, 0)
    cand = np.vstack([shard0] + [weak_problem(w, r)[4] for r in range(1, n)])
    eng = Engine(devices=devs)
    eng.set_observations(comp, vals)
    eng.set_candidates(cand)
    eng.set_hypers(hypers)
    if w["per_sec"]:
        eng.set_time_model(prob[4], prob[5])

    def run(nsteps):
        t0 = time.perf_counter()
        for _ in range(nsteps):
            eng.ei_step(flags)
            out = eng.best()
        return time.perf_counter() - t0, out

    if args.warmup:
        run(args.warmup)
    dt, best = run(args.steps)
    out = {
        "metric": "EI candidate evaluations per second (N_cand x mcmc_iters / wall time)",
        "value": n * float(M) * H * args.steps / dt, "unit": "EI evals/s", "n_gpus": n, "ranks_seen": eng.stat("ranks_seen"),
        "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["desc"], "N_obs": N, "candidates_per_gpu": M, "D": D, "mcmc_iters": H,
                   "per_sec": w["per_sec"], "mode": "one process, one multi-device handle (spx_create_multi)",
                   "devices": devs, "transport": eng.transport()},
        "best_index": best[0], "best_ei": best[1]}
    # strong scaling through the same handle: the full C4 / C5 grids sharded by the library over the n devices
    # (and, with --hyper-shards, C4 in the library's 2-D partition: one ncclAllReduce of the EI-sum vector)
    if not args.skip_extras:
        esteps = max(args.extra_steps, 5 if n > 1 else 2)
        variants = [("c4", "c4", 1), ("c5", "c5", 1)]
        if args.hyper_shards > 1:
            variants.append(("c4_2d", "c4", args.hyper_shards))
        for key, name, ph in variants:
            try:
                cfg = dict(STRONG[name])
                cfg["M"] = int(getattr(args, "%s_candidates" % name))
                sprob, scomp, svals, shyp = strong_problem(cfg)
                rows = strong_rows(cfg, scomp, svals, 0, cfg["M"])
                eng.set_partition(ph)
                eng.set_observations(scomp, svals)
                eng.set_hypers(shyp)
                eng.set_candidates(rows)
                if cfg["per_sec"]:
                    eng.set_time_model(sprob[4], sprob[5])
                else:
                    eng.set_time_model(None, None)
                fl = FLAG_PER_SEC if cfg["per_sec"] else 0
                eng.set_option("timing", 0)

                def srun(k):
                    t0 = time.perf_counter()
                    for _ in range(k):
                        eng.ei_step(fl)
                        o = eng.best()
                    return time.perf_counter() - t0, o
                srun(1)
                sdt, sbest = srun(esteps)
                eng.set_option("timing", 1)        # one more step with per-launch events: the per-stage MAX over the devices
                srun(1)
                tm = eng.timings()
                eng.set_option("timing", 0)
                out[key] = {"value": float(cfg["M"]) * cfg["H"] * esteps / sdt, "unit": "EI evals/s", "scaling": "strong",
                            "n_gpus": n, "steps": esteps, "warmup": 1, "ms_per_step": sdt / esteps * 1e3,
                            "partition": "%d draw shards x %d candidate shards" % (ph, n // ph),
                            "collective": "ncclAllReduce(SUM) of %d doubles" % cfg["M"] if ph > 1 else "ncclAllGather of 16-byte records",
                            "transport": eng.transport(),
                            "stages_ms_max_over_devices": {k: v[0] for k, v in tm.items() if v[1]},
                            "config": {"workload": cfg["desc"], "N_obs": cfg["N"], "candidates_total": cfg["M"],
                                       "D": cfg["D"], "mcmc_iters": cfg["H"], "per_sec": cfg["per_sec"]},
                            "best_index": sbest[0], "best_ei": sbest[1]}
            except Exception as e:      # a variant that cannot run here is reported, not fatal
                out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        eng.set_partition(1)
    eng.close()
    emit_line(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--candidates", type=int, default=0,
                    help="candidates per GPU of the headline instead of the workload's own (plumbing tests only: the line says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--next-baseline", action="store_true",
                    help="instead of the EI-grid bench: time one whole GPEIOptChooser.next() of the reference (CPU) and of this "
                         "library from the same seeded state (N_obs=256, 20 000 candidates, mcmc_iters=10) and print both")
    ap.add_argument("--cpu-candidates", type=int, default=20000)
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic in this run (use the committed profile)")
    ap.add_argument("--skip-extras", action="store_true", help="do not time the strong-scaling c4 / c5 sub-records")
    ap.add_argument("--extra-steps", type=int, default=2, help="timed steps of each strong-scaling sub-record (after 1 warm-up)")
    ap.add_argument("--hyper-shards", type=int, default=1,
                    help="N > 1 only: also time C4 in the optional 2-D partition of SURVEY 8(e) -- draws x candidates over "
                         "P_h x (N / P_h) ranks, one all-reduce(SUM) of the M-vector of EI sums (sub-record c4_2d)")
    ap.add_argument("--lib-timeout", type=int, default=240,
                    help="N > 1: seconds the variants whose collective runs inside libspx (c4_lib, c5_lib, c4_2d_lib) may take "
                         "together before a watchdog prints the line without them")
    ap.add_argument("--c4-candidates", type=int, default=STRONG["c4"]["M"])
    ap.add_argument("--c5-candidates", type=int, default=STRONG["c5"]["M"])
    ap.add_argument("--event-steps", type=int, default=3, help="steps of the second (per-launch HIP event) pass")
    ap.add_argument("--kstar-budget-mb", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="0 = library default")
    ap.add_argument("--gemm-waves", type=int, default=0, help="predict GEMM variant (see predict_kernels.hip)")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process, one libspx handle over --gpus devices (spx_create_multi: host thread per GPU, "
                         "RCCL all-gather inside the library) instead of one process per GPU; what the unmodified "
                         "single-process Spearmint driver uses with --method-args=ndev=N")
    ap.add_argument("--devices", default="", help="with --in-process: comma-separated device ids (default 0..gpus-1)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.next_baseline:
        return emit_line(next_baseline())
    if args.in_process:
        return main_in_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])          # does not return: exec of the launcher

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would not describe the run" % (args.gpus, world))
    if rank != 0:
        os.dup2(2, 1)       # only rank 0 owns the shared stdout (nothing of another rank may land behind the JSON line)
    EngineCls, engine_name = engine_class()

    torch = None
    tdev = None
    try:
        import torch  # plumbing only: rendezvous, barrier, the all-gather of the 16-byte records, device sync
    except ImportError:
        if world > 1:
            raise
    # Test hooks (not used by the driver): SPX_BENCH_BACKEND=gloo exercises the N > 1 code path
    # on a box where RCCL cannot run (e.g. two ranks sharing the single GPU of a dev box, with
    # SPX_BENCH_SINGLE_DEVICE=1); the default is "nccl" == RCCL over xGMI, one GPU per rank.
    backend = os.environ.get("SPX_BENCH_BACKEND", "nccl")
    if os.environ.get("SPX_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    if world > 1:
        import torch.distributed as tdist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            tdev = torch.device("cuda", local_rank)
            tdist.init_process_group(backend="nccl", device_id=tdev)
        else:
            tdist.init_process_group(backend=backend)

    def sync():
        if world > 1:
            tdist.barrier()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=tdev if tdev is not None else "cpu")
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            return float(t.item())
        return dt

    w = dict(WORKLOADS[args.workload])
    if args.candidates:
        w["M"] = int(args.candidates)
        w["desc"] += " [--candidates %d per GPU: NOT the named configuration]" % w["M"]
    N, M, D, H = w["N"], w["M"], w["D"], w["H"]
    flags = FLAG_PER_SEC if w["per_sec"] else 0
    prob, comp, vals, hypers, shard = weak_problem(w, rank)

    eng = EngineCls(local_rank)
    eng.set_observations(comp, vals)
    eng.set_candidates(shard, index_base=rank * M)
    eng.set_hypers(hypers)
    if w["per_sec"]:
        eng.set_time_model(prob[4], prob[5])
    if args.kstar_budget_mb:
        eng.set_option("kstar_budget_bytes", args.kstar_budget_mb << 20)
    if args.streams:
        eng.set_option("streams", args.streams)
    if args.gemm_waves:
        eng.set_option("gemm_waves", args.gemm_waves)

    # The collective: by default the 16-byte records travel through torch.distributed (backend nccl = RCCL);
    # SPX_BENCH_COLLECTIVE=lib attaches an RCCL communicator to the library handle instead (spx_comm_attach),
    # so that spx_ei_run itself ends with the ncclAllGather and best() is already the global winner.
    lib_collective = os.environ.get("SPX_BENCH_COLLECTIVE", "torch") == "lib"

    def attach(e):
        if not lib_collective:
            return
        if world == 1:
            e.comm_attach(e.comm_unique_id(), 1, 0)
            return
        buf = torch.zeros(128, dtype=torch.uint8, device=tdev if tdev is not None else "cpu")
        if rank == 0:
            buf.copy_(torch.frombuffer(bytearray(e.comm_unique_id()), dtype=torch.uint8))
        tdist.broadcast(buf, src=0)
        e.comm_attach(bytes(buf.cpu().numpy().tobytes()), world, rank)

    attach(eng)
    rank_times = {}
    seen = {}        # tag -> number of records the collective returned (the ranks that took part), -1 if it ever varied

    def all_ranks(x):
        """[x of rank 0, ..., x of rank P-1] on every rank (plumbing: per-rank step times, success flags)."""
        if world == 1:
            return [float(x)]
        t = torch.zeros(world, dtype=torch.float64, device=tdev if tdev is not None else "cpu")
        t[rank] = float(x)
        tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
        return [float(v) for v in t.cpu().numpy()]

    def run_steps(e, fl, nsteps, lib=None, tag=None):
        """nsteps timed steps bracketed by barrier + device sync; MAX over ranks.  lib: the collective runs inside
        libspx (a communicator is attached to `e`), else the records travel through torch.distributed."""
        lib = lib_collective if lib is None else lib
        out = None
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            e.ei_step(fl)                   # spx_factor + spx_ei_run, one host synchronisation
            idx, val = e.best()
            if lib:                         # the all-gather ran inside spx_ei_step; its table had stat("ranks_seen") records
                out, nrec = (idx, val), e.stat("ranks_seen")
            else:
                recs = spx_dist.exchange_records(val, idx, device=tdev)
                out, nrec = spx_dist.pick_best(recs), len(recs)
            seen[tag or "untagged"] = nrec if seen.get(tag or "untagged", nrec) == nrec else -1
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        mine = time.perf_counter() - t0          # this rank's own time, before the closing barrier
        sync()
        dt_all = max_over_ranks(time.perf_counter() - t0)
        if tag is not None:
            per = all_ranks(mine)
            rank_times[tag] = {"max_ms_per_step": max(per) / nsteps * 1e3, "min_ms_per_step": min(per) / nsteps * 1e3}
        return dt_all, out

    # ---- headline: weak scaling, no per-launch events inside the timed region -------------------
    if args.warmup:
        run_steps(eng, flags, args.warmup)
    dt, best = run_steps(eng, flags, args.steps, tag="headline")
    evals_per_step = float(M) * H
    value = world * evals_per_step * args.steps / dt

    # ---- second pass: the same steps with HIP events around every launch (stage breakdown, roofline)
    ev_steps = max(1, min(args.event_steps, args.steps))
    eng.set_option("timing", 1)
    dt_ev, _ = run_steps(eng, flags, ev_steps)
    tm = eng.timings()
    eng.set_option("timing", 0)

    gemm_ms, gemm_n = tm["predict_gemm"]
    flops_per_eval = float(N) * N + 4.0 * N
    roofline = roofline_hbm = None
    if gemm_n:
        avg_s = gemm_ms / gemm_n * 1e-3
        evals_per_launch = evals_per_step * ev_steps / gemm_n
        achieved = flops_per_eval * evals_per_launch / avg_s / 1e12
        traffic, traffic_src, traffic_note = None, None, None
        if world == 1:
            if not args.no_live_traffic:
                traffic, traffic_src = live_traffic(args.workload)
                if traffic is None:
                    traffic_note = traffic_src
            if traffic is None:
                traffic, traffic_src = pmc_traffic(args.workload)
        roofline = {"bound": "mfma", "kernel": "k_predict_gemm_tri", "achieved": achieved,
                    "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "frac_vs_ubench": achieved / FP64_MFMA_MEASURED_TFLOPS,
                    "traffic": traffic,
                    "traffic_source": traffic_src, "peak_measured_ubench": FP64_MFMA_MEASURED_TFLOPS,
                    "launches": gemm_n, "avg_launch_ms": gemm_ms / gemm_n,
                    "flops_per_eval": flops_per_eval, "evals_per_launch": evals_per_launch,
                    "measured_over": "%d steps with per-launch HIP events on the library's stream" % ev_steps,
                    "dtype_peak_source": "AMD MI355X spec: 78.6 TFLOP/s fp64 matrix"}
        # the WHOLE step against the same peak (N^2 solve + 2 N D sq-dist + 4 N moments + 12 N Matern flops per
        # evaluation; factorisations, K(X*,X), finalize and argmax all inside the time): the driver's own clock can check this one
        step_flops = float(N) * N + 2.0 * N * D + 16.0 * N
        roofline["whole_step"] = {"flops_per_eval": step_flops, "achieved": step_flops * evals_per_step / (dt / args.steps) / 1e12,
                                  "unit": "TFLOP/s", "frac": step_flops * evals_per_step / (dt / args.steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                  "note": "algorithmic flops of one step / its wall time (the headline pass, no per-launch events)"}
        # what one launch has to move at least: its K(X*,X) columns once (8 N bytes per evaluation) + the live half of W
        Npad = -(-N // 128) * 128
        roofline["algorithmic_bytes"] = 8.0 * Npad * evals_per_launch + 8.0 * Npad * Npad / 2.0
        if traffic:
            roofline["traffic_over_algorithmic"] = traffic / roofline["algorithmic_bytes"]
        if traffic_note:
            roofline["traffic_live_failed"] = traffic_note
        # second regime: the HBM-side stages.  K(X*,X) producer: 8 N bytes written per
        # evaluation (the staging buffer the GEMM then reads); EI finalize: the per-row-block partial
        # sums (2 x N/128 x 8 B read) + one EI written per evaluation.
        Np = -(-N // 128) * 128
        Dp_pad = 4 if D <= 4 else (8 if D <= 8 else (16 if D <= 16 else -(-D // 32) * 32))     # padded_dim() of spx_api.hip
        cov_ms, cov_n = tm["cov_cross"]
        fin_ms, fin_n = tm["ei_finalize"]
        evals_ev = evals_per_step * ev_steps
        cov_gbs = 8.0 * Np * evals_ev / (cov_ms * 1e-3) / 1e9 if cov_ms else None
        fin_bytes = (2.0 * (Np // 128) * 8.0 + 8.0) * evals_ev
        fin_gbs = fin_bytes / (fin_ms * 1e-3) / 1e9 if fin_ms else None
        roofline_hbm = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "kernel": "k_cov<0> (K(X*,X) write stream, 8*N B per evaluation)",
                        "achieved": cov_gbs, "frac": (cov_gbs / HBM_PEAK_GBS) if cov_gbs else None,
                        "launches": cov_n, "avg_launch_ms": (cov_ms / cov_n) if cov_n else None,
                        "bytes_per_eval": 8.0 * Np,
                        # what bounds k_cov: the fp64 FMA units, shared by MFMA and fp64 VALU on gfx950 -- 37 fp64 VALU
                        # instructions per element of K* (profiles/r04_k_cov_isa.md) at 39.3e12 lane-instructions / s, plus the
                        # Gram MFMAs (2 Np Dp flop per evaluation) at the 78.6 TF matrix peak, back to back
                        "fma_unit_bound_ms_per_step": (37.0 * Np * evals_per_step / 39.3e12 + 2.0 * Np * Dp_pad * evals_per_step / (FP64_MFMA_PEAK_TFLOPS * 1e12)) * 1e3,
                        "ms_per_step": (cov_ms / ev_steps) if cov_ms else None,
                        "note": "k_cov is bound by the fp64 FMA units (Matern epilogue: 37 fp64 VALU instructions per element + the "
                                "Gram MFMAs), not by its store stream: frac_of_fma_unit_bound = fma_unit_bound_ms_per_step / ms_per_step "
                                "(DESIGN.md section 4)",
                        "ei_finalize": {"achieved": fin_gbs, "frac": (fin_gbs / HBM_PEAK_GBS) if fin_gbs else None,
                                        "launches": fin_n, "avg_launch_ms": (fin_ms / fin_n) if fin_n else None,
                                        "bytes_per_eval": 2.0 * (Np // 128) * 8.0 + 8.0}}

    if roofline_hbm and roofline_hbm.get("ms_per_step"):
        roofline_hbm["frac_of_fma_unit_bound"] = roofline_hbm["fma_unit_bound_ms_per_step"] / roofline_hbm["ms_per_step"]
    out = None
    ranks_seen = seen.get("headline")
    if ranks_seen != args.gpus:      # never a line whose n_gpus is not what was asked for and what took part
        raise SystemExit("bench.py: --gpus %d but the collective returned %r records per step" % (args.gpus, ranks_seen))
    if rank == 0:
        versions = {}
        if engine_name == "libspx" and (world > 1 or lib_collective):
            try:
                from spearmint_amd import engine as _e
                versions["libspx_binding"] = _e.rccl_version()          # ncclGetVersion of the librccl libspx dlopens
            except Exception as ex:
                versions["libspx_binding"] = "unavailable: %s" % ex
        if world > 1 and backend == "nccl":
            try:
                versions["torch_backend"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as ex:
                versions["torch_backend"] = "unavailable: %s" % ex
        out = {
            "metric": "EI candidate evaluations per second (N_cand x mcmc_iters / wall time)",
            "value": value, "unit": "EI evals/s", "n_gpus": ranks_seen, "ranks_seen": ranks_seen,
            "launcher": ("self: bench.py re-executed itself under torch.distributed.run" if os.environ.get("SPX_BENCH_SELF_LAUNCHED")
                         else ("external torch.distributed.run" if world > 1 else "none (one rank)")),
            "backend": (backend if world > 1 else None), "rccl_version": versions or None, "engine": engine_name,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["desc"], "N_obs": N, "candidates_per_gpu": M, "D": D,
                       "mcmc_iters": H, "per_sec": w["per_sec"],
                       "sharding": "candidates sharded contiguously over ranks, draws replicated; "
                                   "one collective of (best EI, index) records",
                       "collective": "libspx ncclAllGather (spx_comm_attach)" if lib_collective
                                     else "torch.distributed all_gather_into_tensor"},
            "roofline": roofline, "roofline_hbm": roofline_hbm,
            "platform": platform_info(eng if engine_name == "libspx" else None, local_rank),
            "ms_per_step_with_events": dt_ev / ev_steps * 1e3,
            "stages_ms_per_step": {k: v[0] / ev_steps for k, v in tm.items() if v[1]},
            "best_index": best[0], "best_ei": best[1],
            "rank_step_ms": rank_times.get("headline"),
        }

    # ---- the metric as SURVEY 8(d) defines it: host buffers in, result out (PCIe included) ------
    if world == 1:
        def one_shot():
            if w["per_sec"]:
                return eng.ei_per_sec_grid(comp, vals, prob[4], shard, hypers, prob[5], want_mean=False)
            return eng.ei_grid(comp, vals, shard, hypers, want_mean=False)
        one_shot()
        reps = []
        for _ in range(2):
            t0 = time.perf_counter()
            r = one_shot()
            reps.append(time.perf_counter() - t0)
        out["host_inclusive_value"] = evals_per_step / min(reps)
        out["host_inclusive_ms"] = min(reps) * 1e3
        assert r[0] == best[0], "one-shot entry point and resident path disagree"

    # ---- the same headline through ONE multi-device handle over this GPU (spx_create_multi: device thread, a real RCCL
    # communicator of one rank, ncclAllGather of the record): what the n = 1 case of the in-process path costs ----------
    if world == 1 and rank == 0 and not args.skip_extras and engine_name == "libspx":
        try:
            me = Engine(devices=[local_rank])
            me.set_observations(comp, vals)
            me.set_candidates(shard)
            me.set_hypers(hypers)
            if w["per_sec"]:
                me.set_time_model(prob[4], prob[5])
            me.ei_step(flags)
            k_ip = max(2, min(args.steps, 5))
            if torch is not None and torch.cuda.is_available():
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k_ip):
                me.ei_step(flags)
                ip_best = me.best()
            dt_ip = time.perf_counter() - t0
            out["in_process_n1"] = {"value": evals_per_step * k_ip / dt_ip, "unit": "EI evals/s", "ms_per_step": dt_ip / k_ip * 1e3,
                                    "steps": k_ip, "warmup": 1, "transport": me.transport(), "ranks_seen": me.stat("ranks_seen"),
                                    "best_index": ip_best[0], "best_ei": ip_best[1],
                                    "mode": "one process, one multi-device handle over this GPU (spx_create_multi + RCCL all-gather)"}
            me.close()
            assert ip_best == (best[0], best[1]), "multi-device handle and single-GPU handle disagree"
        except Exception as ex:         # reported, not fatal (e.g. librccl missing on the box)
            out["in_process_n1"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    # ---- strong scaling at the full C4 / C5 sizes -------------------------------------------------
    # With N > 1 ranks every configuration is timed with BOTH forms of the collective in the same run -- the
    # 16-byte records through torch.distributed (backend nccl = RCCL; sub-records "c4" / "c5") and through libspx's
    # own ncclAllGather on the handle's stream (spx_comm_attach; "c4_lib" / "c5_lib") -- and, with --hyper-shards,
    # C4 in the 2-D partition both ways ("c4_2d": host-side sums + torch all-reduce; "c4_2d_lib": spx_set_partition,
    # ncclAllReduce of the device-resident EI-sum vector).  A variant that cannot run on this box (e.g. two ranks
    # sharing one GPU cannot form an RCCL communicator) records its error string; the others still count, and all
    # that ran must agree on the winner.
    def all_ok(ok):
        return min(all_ranks(1.0 if ok else 0.0)) > 0.5

    def attach_lib(e):
        """Attach an RCCL communicator over all ranks to `e`; returns an error string (same decision on every rank)."""
        err = None
        try:
            if world == 1:
                e.comm_attach(e.comm_unique_id(), 1, 0)
            else:
                buf = torch.zeros(128, dtype=torch.uint8, device=tdev if tdev is not None else "cpu")
                if rank == 0:
                    buf.copy_(torch.frombuffer(bytearray(e.comm_unique_id()), dtype=torch.uint8))
                tdist.broadcast(buf, src=0)
                e.comm_attach(bytes(buf.cpu().numpy().tobytes()), world, rank)
        except Exception as ex:
            err = "%s: %s" % (type(ex).__name__, ex)
        if not all_ok(err is None):
            return err or "another rank could not attach the communicator"
        return None

    # Order: every variant whose collective travels through torch.distributed first (c4, c5, c4_2d), THEN the ones whose
    # collective runs inside libspx (c4_lib, c5_lib, c4_2d_lib) under a watchdog: that code has run with P > 1 only on the
    # thread-rendezvous stand-in of the tests (no multi-GPU box was available to build on), and a collective that never
    # returns must not cost the run its line -- after --lib-timeout seconds rank 0 prints what it has (the variant in
    # flight marked as timed out) and every rank leaves.
    winners = {}

    def strong_variant(name, key, use_lib, esteps):
        cfg = dict(STRONG[name])
        cfg["M"] = int(getattr(args, "%s_candidates" % name))
        sprob, scomp, svals, shyp = strong_problem(cfg)
        lo, hi = spx_dist.shard_bounds(cfg["M"], world, rank)
        rows = strong_rows(cfg, scomp, svals, lo, hi)
        fl = FLAG_PER_SEC if cfg["per_sec"] else 0
        e2 = None
        err = None
        try:
            e2 = EngineCls(local_rank)
            e2.set_observations(scomp, svals)
            e2.set_candidates(rows, index_base=lo)
            e2.set_hypers(shyp)
            if cfg["per_sec"]:
                e2.set_time_model(sprob[4], sprob[5])
        except Exception as ex:
            err = "%s: %s" % (type(ex).__name__, ex)
        if not all_ok(err is None):
            err = err or "another rank failed to set the problem up"
        elif use_lib:
            err = attach_lib(e2)
        if err is None:
            run_steps(e2, fl, 1, lib=use_lib)
            sdt, sbest = run_steps(e2, fl, esteps, lib=use_lib, tag=key)
        if e2 is not None:
            e2.close()
        if rank != 0:
            return
        if err is not None:
            out[key] = {"error": err, "n_gpus": world,
                        "collective": "libspx ncclAllGather (spx_comm_attach)" if use_lib else "torch.distributed"}
            return
        sval = float(cfg["M"]) * cfg["H"] * esteps / sdt
        out[key] = {"value": sval, "unit": "EI evals/s", "scaling": "strong", "n_gpus": world,
                    "steps": esteps, "warmup": 1, "ms_per_step": sdt / esteps * 1e3,
                    "collective": "libspx ncclAllGather (spx_comm_attach)" if use_lib
                                  else "torch.distributed all_gather_into_tensor",
                    "rank_step_ms": rank_times.get(key), "ranks_seen": seen.get(key),
                    "config": {"workload": cfg["desc"], "N_obs": cfg["N"], "candidates_total": cfg["M"],
                               "D": cfg["D"], "mcmc_iters": cfg["H"], "per_sec": cfg["per_sec"]},
                    "best_index": sbest[0], "best_ei": sbest[1]}
        if key == name:
            out["%s_value" % name] = sval
        winners.setdefault(name, []).append((key, sbest[0], sbest[1]))

    # C4 in the optional 2-D (draws x candidates) partition, one all-reduce(SUM) of the EI-sum vector
    def partition_variant(key, use_lib, esteps):
        cfg = dict(STRONG["c4"])
        cfg["M"] = int(args.c4_candidates)
        sprob, scomp, svals, shyp = strong_problem(cfg)
        (lo, hi), (h0, h1) = spx_dist.shard_2d(cfg["M"], cfg["H"], world, rank, args.hyper_shards)
        rows = strong_rows(cfg, scomp, svals, lo, hi)
        part = "%d draw shards x %d candidate shards" % spx_dist.grid_2d(world, args.hyper_shards)
        e3 = EngineCls(local_rank)
        e3.set_observations(scomp, svals)
        e3.set_candidates(rows, index_base=lo)
        e3.set_hypers(shyp[h0:h1])
        err = None
        if use_lib:
            err = attach_lib(e3)
            if err is None:
                e3.set_partition(args.hyper_shards, cfg["M"], cfg["H"])

        def step_2d():
            e3.ei_step(0)
            if use_lib:                                       # ncclAllReduce + argmax ran inside spx_ei_run
                return e3.best()
            sums = np.sum(e3.ei_draws(), axis=1)              # this rank's draws, its candidates: D2H of M_local x H_local
            return spx_dist.allreduce_ei_sums(sums, lo, cfg["M"], cfg["H"], device=tdev)[:2]
        if err is None:
            step_2d()
            sync()
            t0 = time.perf_counter()
            for _ in range(esteps):
                r2 = step_2d()
            sync()
            dt2 = max_over_ranks(time.perf_counter() - t0)
        e3.close()
        if rank != 0:
            return
        if err is not None:
            out[key] = {"error": err, "n_gpus": world, "partition": part}
            return
        out[key] = {"value": float(cfg["M"]) * cfg["H"] * esteps / dt2, "unit": "EI evals/s",
                    "scaling": "strong", "n_gpus": world, "partition": part, "steps": esteps,
                    "collective": ("libspx ncclAllReduce(SUM) of %d doubles on the handle's stream (spx_set_partition)"
                                   if use_lib else "host-side sums + torch all-reduce(SUM) of %d doubles") % cfg["M"],
                    "ms_per_step": dt2 / esteps * 1e3, "best_index": int(r2[0]), "best_ei": float(r2[1])}

    # ---- the caller of the hot path end to end, at Spearmint's operating size (`bench.py --next-baseline` adds
    # the reference's own next() beside it): one warm GPEIOptChooser.next() -- native sampler (spx_sample_hypers), two EI passes,
    # lock-step refinement -- and the round-5 form of the same call on this box
    if world == 1 and rank == 0 and not args.skip_extras and engine_name == "libspx":
        try:
            nb = next_baseline(with_reference=False)
            out["next_call"] = {"what": nb["what"], "config": nb["config"], "warm_s": nb["ours"]["warm_s"], "cold_s": nb["ours"]["cold_s"],
                                "round5_form_warm_s": nb["ours"]["round5_form_warm_s"], "sampler_stats": nb["ours"]["sampler_stats"],
                                "speculation_depth": nb["ours"]["speculation_depth"], "proposal": nb["ours"]["proposal"],
                                "repeatable": nb["ours_repeatable"], "round5_form_same_proposal": nb["ours"]["round5_form_same_proposal"]}
        except Exception as ex:
            out["next_call"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if not args.skip_extras:
        esteps = max(args.extra_steps, 5) if world > 1 else args.extra_steps
        two_d = args.hyper_shards > 1 and world > 1
        for name in ("c4", "c5"):
            strong_variant(name, name, lib_collective if world == 1 else False, esteps)
        if two_d:
            partition_variant("c4_2d", False, esteps)
        if world > 1:
            import threading
            in_flight = {"key": None}

            def expired():
                if rank == 0:
                    out["lib_collective_watchdog"] = ("the library-collective variants did not finish within %d s (in flight: %s); "
                                                      "this line was printed by the watchdog" % (args.lib_timeout, in_flight["key"]))
                    if in_flight["key"] and in_flight["key"] not in out:
                        out[in_flight["key"]] = {"error": "timed out after %d s (watchdog)" % args.lib_timeout, "n_gpus": world}
                    sys.stdout.write(json.dumps(out) + "\n")
                    sys.stdout.flush()
                os._exit(0)

            dog = threading.Timer(float(args.lib_timeout), expired)
            dog.daemon = True
            dog.start()
            for name in ("c4", "c5"):
                in_flight["key"] = name + "_lib"
                strong_variant(name, name + "_lib", True, esteps)
            if two_d:
                in_flight["key"] = "c4_2d_lib"
                partition_variant("c4_2d_lib", True, esteps)
            dog.cancel()
        if rank == 0:
            for name, ws in winners.items():
                assert all(w[1:] == ws[0][1:] for w in ws), "collective variants disagree on the winner: %r" % (ws,)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w, args.cpu_candidates, args.cpu_reps)
    # teardown FIRST, the line LAST: librccl (torch's, or the one libspx binds) prints a version banner through C stdio,
    # which a pipe buffers until the process ends -- behind the JSON line if that were printed earlier
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    eng.close()
    emit_line(out if rank == 0 else None)


if __name__ == "__main__":
    main()
