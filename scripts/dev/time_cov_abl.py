"""Dev tool: what bounds k_cov<0> (K(X*,X))?  Per-stage HIP-event times of a C3 step with the in-tree library and with
ablation builds of it (make COV_ABL=1|2|3 TARGET=../../_variants/libspx_covablN.so: no store stream / no correlation
function / no Gram MFMAs -- WRONG results, timing only).
    python scripts/dev/time_cov_abl.py [_variants/libspx_covabl1.so ...]"""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    from spearmint_amd.engine import Engine
    from spearmint_amd.synthetic import synthetic_problem
    lib = None if sys.argv[2] == "-" else sys.argv[2]
    N, M, D, H = 2048, 57344, 32, 20          # two chunks of the C3 plan
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 3000)
    eng = Engine(0, lib=lib)
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
    eng.ei_step(0)
    eng.set_option("timing", 1)
    for _ in range(3):
        eng.ei_step(0)
    tm = eng.timings()
    print(json.dumps({k: (v[0] / 3, v[1] // 3) for k, v in tm.items() if v[1]}))
    sys.exit(0)
for lib in ["-"] + sys.argv[1:]:
    o = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", lib]).decode().strip().splitlines()[-1]
    d = json.loads(o)
    print("%-40s cov_cross %.3f ms / step (%d launches, %.4f ms each)   predict_gemm %.2f   step %.2f" % (
        lib, d["cov_cross"][0], d["cov_cross"][1], d["cov_cross"][0] / d["cov_cross"][1], d["predict_gemm"][0], d["ei_run_total"][0]))
