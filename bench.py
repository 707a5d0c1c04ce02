#!/usr/bin/env python3
"""bench.py -- EI-candidate evaluations / second of the GP-EI hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one batch of synthetic input,
i.e. what one chooser.next() hands to the GPU: for every hyper-parameter draw
build K(X,X), factor it, then K(X*,X), the triangular solve, predictive
mean/variance and EI for every candidate of this rank's shard, the MCMC mean,
the local argmax, and (N > 1) the single RCCL all-reduce that picks the global
best.  Inputs (observations, candidate shard, hyper draws) are resident in HBM
before the timed region starts (spx_set_* are outside it).

Workload (BASELINE.json configs[2], the single-GPU configuration the metric's
target is quoted on): synthetic 32-D, N_obs=2048, 200 000 candidates per GPU,
mcmc_iters=20, fp64.  Per-GPU work is fixed as N grows ("weak" scaling):
N GPUs score N x 200 000 candidates of one grid.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel
(k_predict_gemm: beta = L^-1 K* as an fp64 MFMA GEMM with the variance/mean
reduction fused in its epilogue): achieved = algorithmic flops per launch
(N^2 + 4N per (candidate, draw) evaluation) / the kernel's mean
launch duration measured with HIP events on the library's stream inside the
timed region.  `cpu_baseline` is the numpy/scipy oracle (a port of the
reference chooser's compute_ei loop) timed on this host on a bounded sample.
"""
from __future__ import print_function

import argparse
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
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD CDNA4 spec; = vector fp64 peak)
FP64_MFMA_MEASURED_TFLOPS = 77.5  # scripts/ubench_f64.hip on this pool: v_mfma_f64_16x16x4_f64, VGPR accumulators


def pmc_traffic(workload):
    """HBM bytes per k_predict_gemm launch from the committed rocprofv3 PMC passes of this
    same command (profiles/r01_<workload>_rocprof_summary.json, written by
    scripts/pmc_summary.py: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, FETCH_SIZE
    doubled per the gfx950 correction of MI355X_MICROARCH.md).  None when no profile exists."""
    path = os.path.join(ROOT, "profiles", "r01_%s_rocprof_summary.json" % workload)
    try:
        with open(path) as fh:
            kernels = json.load(fh)["kernels"]
        name = [n for n in kernels if n.startswith("k_predict_gemm")][0]   # template args are part of the name
        return float(kernels[name]["hbm_bytes_per_launch_corrected"]), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def cpu_baseline(w, seconds_hint=15.0):
    """Time the numpy/scipy oracle (port of GPEIChooser.compute_ei x H + argmax)
    on a bounded sample of the same workload: same N, D, H; fewer candidates."""
    from oracle import gp_ei_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    N, D, H = w["N"], w["D"], w["H"]
    # ~4e3 evals/s at N=2048 -> pick M so the run is 10-30 s
    m_cpu = int(max(512, min(w["M"], seconds_hint * 4.0e3 * (2048.0 / N) ** 1.5 / H)))
    comp, cand, vals, hypers = synthetic_problem(N, m_cpu, D, H, 3000)[:4]
    t0 = time.time()
    ei = orc.ei_grid_chunked(comp, cand, vals, hypers, chunk=20000)
    orc.choose(ei)
    dt = time.time() - t0
    return {"value": m_cpu * H / dt, "unit": "EI evals/s", "cores": int(threads), "kind": "port",
            "sample": "oracle.ei_grid_chunked (numpy/scipy restatement of GPEIChooser.compute_ei x H + "
                      "argmax(mean)), N_obs=%d, D=%d, mcmc_iters=%d, %d candidates, %.1f s, 1 run"
                      % (N, D, H, m_cpu, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kstar-budget-mb", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="0 = library default")
    ap.add_argument("--gemm-waves", type=int, default=0, help="predict GEMM variant: 4 or 8 waves per workgroup")
    ap.add_argument("--host-inclusive", action="store_true",
                    help="also time the one-shot host-buffer entry point (PCIe H2D/D2H included)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    torch = None
    tdev = None
    try:
        import torch  # plumbing only: barrier / all-reduce / device sync
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

    w = WORKLOADS[args.workload]
    N, M, D, H = w["N"], w["M"], w["D"], w["H"]
    flags = FLAG_PER_SEC if w["per_sec"] else 0

    # identical observations / hyper draws on every rank, own candidate shard
    prob = synthetic_problem(N, 16, D, H, 1000 * 3, near=0, per_sec=w["per_sec"])
    comp, vals, hypers = prob[0], prob[2], prob[3]
    shard = synthetic_problem(N, M, D, 1, 1000 * 3 + 17 * rank, near=(10 if rank == 0 else 0))[1]
    if rank == 0:  # jittered copies of the incumbent, as GPEIOptChooser.py:236-238
        inc = comp[np.argmin(vals)]
        shard[:10] = np.clip(inc + 1e-3 * np.random.RandomState(5).randn(10, D), 0, 1)

    eng = Engine(local_rank)
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

    def sync():
        if world > 1:
            tdist.barrier()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    def step():
        eng.factor()
        eng.ei_run(flags)
        idx, val = eng.best()
        return spx_dist.allreduce_best(val, idx, device=tdev)

    for _ in range(args.warmup):
        step()
    eng.set_option("timing", 1)   # HIP events around every launch, on the library's own stream
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev if tdev is not None else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())

    tm = eng.timings()
    evals_per_step = float(M) * H
    value = world * evals_per_step * args.steps / dt

    gemm_ms, gemm_n = tm["predict_gemm"]
    flops_per_eval = float(N) * N + 4.0 * N
    roofline = None
    if gemm_n:
        avg_s = gemm_ms / gemm_n * 1e-3
        evals_per_launch = evals_per_step * args.steps / gemm_n
        achieved = flops_per_eval * evals_per_launch / avg_s / 1e12
        traffic, traffic_src = pmc_traffic(args.workload) if world == 1 else (None, None)
        roofline = {"bound": "mfma", "kernel": "k_predict_gemm", "achieved": achieved,
                    "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_source": traffic_src, "peak_measured_ubench": FP64_MFMA_MEASURED_TFLOPS,
                    "launches": gemm_n, "avg_launch_ms": gemm_ms / gemm_n,
                    "flops_per_eval": flops_per_eval, "evals_per_launch": evals_per_launch,
                    "dtype_peak_source": "AMD MI355X spec: 78.6 TFLOP/s fp64 matrix"}

    if rank == 0:
        out = {
            "metric": "EI candidate evaluations per second (N_cand x mcmc_iters / wall time)",
            "value": value, "unit": "EI evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["desc"], "N_obs": N, "candidates_per_gpu": M, "D": D,
                       "mcmc_iters": H, "per_sec": w["per_sec"],
                       "sharding": "candidates sharded contiguously over ranks, draws replicated; "
                                   "one all-reduce of (best EI, index) records"},
            "roofline": roofline,
            "stages_ms_per_step": {k: v[0] / args.steps for k, v in tm.items() if v[1]},
            "best_index": best[0], "best_ei": best[1],
        }
        if world == 1 and args.host_inclusive and not w["per_sec"]:
            t0 = time.perf_counter()
            eng.ei_grid(comp, vals, shard, hypers, want_mean=False)
            out["host_inclusive_value"] = evals_per_step / (time.perf_counter() - t0)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
