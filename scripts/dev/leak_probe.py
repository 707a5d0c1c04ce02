"""Dev: which call leaves device memory behind after spx_destroy (GPU box).

    python scripts/dev/leak_probe.py [reps] [opt=val ...]      e.g.  leak_probe.py 6 lean_flow=0 ei_flow=0

Per op: MiB of device memory NOT returned per create/op/destroy lifetime (hipMemGetInfo), measured after
two settling lifetimes of the same op.  The last line repeats the cycle of tests/test_gpu_z_robustness.py
::test_handles_release_their_device_memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd import sobol
from spearmint_amd.synthetic import synthetic_problem

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
comp, cand, vals, hypers, ld, th = synthetic_problem(300, 20000, 6, 3, 77, per_sec=True)
rs = np.random.RandomState(0)
fant, fb = rs.randn(3, 300, 9), rs.randn(3, 9)


def mk():
    e = Engine(0)
    for k, v in opts:
        e.set_option(k, int(v))
    return e


def resident(e):
    e.set_observations(comp, vals); e.set_candidates(cand); e.set_hypers(hypers); e.factor()


ops = {
    "create only": lambda e: None,
    "set_observations": lambda e: e.set_observations(comp, vals),
    "factor": resident,
    "ei_grid": lambda e: e.ei_grid(comp, vals, cand, hypers, want_draws=True),
    "per_sec": lambda e: e.ei_per_sec_grid(comp, vals, ld, cand, hypers, th),
    "grad": lambda e: (e.ei_grid(comp, vals, cand, hypers), e.ei_grad_batch(cand[:5])),
    "fantasies": lambda e: (resident(e), e.set_fantasies(fant, fb), e.ei_run(), e.ei_grad_batch(cand[:3])),
    "logprob": lambda e: (e.set_observations(comp, vals), e.set_hypers(hypers), e.gp_logprob()),
    "sobol": lambda e: e.sobol_grid(sobol.load_dirs("bf40"), 8, 50000, 1),
}


import ctypes
_hip = ctypes.CDLL("libamdhip64.so")


def free():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert _hip.hipDeviceSynchronize() == 0 and _hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return int(f.value)


def full_cycle():
    e = mk()
    e.ei_grid(comp, vals, cand, hypers, want_draws=True)
    e.ei_per_sec_grid(comp, vals, ld, cand, hypers, th)
    e.ei_grad_batch(cand[:5])
    resident(e)
    e.set_fantasies(fant, fb)
    e.ei_run(); e.ei_grad_batch(cand[:3])
    e.set_hypers(hypers); e.gp_logprob()
    e.sobol_grid(sobol.load_dirs("bf40"), 8, 50000, 1)
    e.close()


print("options:", opts, "reps:", reps, flush=True)
e = mk(); e.ei_grid(comp, vals, cand, hypers); e.close()
for name, op in ops.items():
    for _ in range(2):
        e = mk(); op(e); e.close()
    f0 = free()
    for _ in range(reps):
        e = mk(); op(e); e.close()
    print("%-18s %8.3f MiB per lifetime" % (name, (f0 - free()) / reps / 2.0 ** 20), flush=True)
import time
for _ in range(4):
    full_cycle()
f0 = free()
trace = []
for k in range(24):
    full_cycle(); d = (f0 - free()) / 2.0 ** 20; trace.append(d)
    if d > 1.0:     # a reading below the settled figure: how long does it last without any further call?
        t0 = time.time(); back = None
        for _ in range(100):
            time.sleep(0.01)
            if (f0 - free()) / 2.0 ** 20 < 1.0:
                back = time.time() - t0; break
        print("  lifetime %d: reading %.1f MiB low; %s" % (k, d, "back after %.0f ms idle" % (back * 1e3) if back is not None
                                                              else "still low after 1 s idle (back after the next lifetime?)"))
print("full cycle x24: %.3f MiB after the last; per lifetime readings: %s" % (trace[-1], " ".join("%.1f" % t for t in trace)))
