"""Dev tool: spx_gp_logprob wall time per call with the in-tree libspx.so against variant builds of it (paths given).
   python scripts/dev/lib_ab.py _variants/libspx_x.so | NAME=VALUE [...]      (each variant runs in its own process; NAME=VALUE:
   the in-tree library with that environment variable)"""
import os, sys, subprocess, json
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import time
    import numpy as np
    from spearmint_amd.engine import Engine
    from spearmint_amd.synthetic import synthetic_problem
    eng = Engine(0, lib=sys.argv[2] if sys.argv[2] != "-" and "=" not in sys.argv[2] else None)
    out = {}
    for N, D in ((2048, 32), (1000, 16), (512, 8), (4096, 32)):
        for H in (1, 2, 4, 6, 8, 12, 20, 32):
            if N == 4096 and H > 4: continue
            comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
            eng.set_observations(comp, vals)
            eng.set_hypers(hypers); r = eng.gp_logprob()
            best = 1e9
            for rep in range(3):
                t = time.time()
                for _ in range(20):
                    eng.set_hypers(hypers); eng.gp_logprob()
                best = min(best, (time.time() - t) / 20 * 1e3)
            out["%d/%d" % (N, H)] = (best, float(np.sum(r)))
    print(json.dumps(out))
    sys.exit(0)
libs = ["-"] + sys.argv[1:]
res = []
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if "=" in lib: env[lib.split("=")[0]] = lib.split("=")[1]
        o = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", lib], env=env).decode().strip().splitlines()[-1]
        res.append((lib, json.loads(o)))
keys = list(res[0][1].keys())
for k in keys:
    line = "N/H %-8s" % k
    for lib in libs:
        ts = [r[1][k][0] for r in res if r[0] == lib]
        same = all(r[1][k][1] == res[0][1][k][1] for r in res)
        line += "  %s %.3f ms" % (os.path.basename(lib), min(ts))
    print(line + ("" if same else "   <-- RESULTS DIFFER"))
