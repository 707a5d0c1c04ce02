"""Dev tool: stage times of spx_gp_logprob (k_cov = cov_self) for SPX_COV_RPW values given in the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in ((2048, 32), (1000, 16), (256, 8)):
    out = []
    for H in (1, 6, 12):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
        eng.set_option("timing", 1)
        for _ in range(10):
            eng.set_hypers(hypers); eng.gp_logprob()
        tm = eng.timings(); eng.set_option("timing", 0)
        out.append("H=%d cov %.1f us chol %.1f us" % (H, tm["cov_self"][0] / 10 * 1e3, tm["chol_diag"][0] / 10 * 1e3))
    print("rpw=%s N=%d: " % (os.environ.get("SPX_COV_RPW", "-"), N) + "  ".join(out))
