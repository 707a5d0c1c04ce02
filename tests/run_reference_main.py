#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- runs the reference's own primary driver, `spearmint/spearmint/main.py`, in a process of its
own (as `bin/spearmint` does: `python main.py <options> <expt>/config.pb`), against the chooser modules that lie in the
tree's `chooser/` directory.

`--tree` is an unpacked `oracle/_ref/main_py3.zip` (bin/, examples/, spearmint/: the reference's driver files converted for Python 3 by
`oracle/ref_py3.convert_main_tree`; nothing of it is in the repository).  With `--engine hip|oracle` the three files
`chooser/GPEI{,Opt,perSec}Chooser.py` of that tree are first REPLACED by our shims `dropin/chooser/*.py` and the repo root
joins PYTHONPATH -- the recipe of INTEGRATION.md section 3 for main.py -- and nothing else of the tree is touched;
`--engine reference` leaves the reference's own choosers in place (how tests/golden/main_loop.npz was made).
`oracle` additionally swaps the HIP engine class for the test-only oracle engine (a CPU box has no GPU).

Two modes:

  --mode main      runpy.run_path(main.py, run_name="__main__") with sys.argv = the options: the reference's literal
                   `main()` loop (main.py:147-180) until --max-finished-jobs is reached.  Deterministic for
                   --max-concurrent=1 (the chooser is asked only when nothing is pending).
  --mode dispatch  the six set-up lines of main() (main.py:152-172: check_experiment_dirs, import_module('chooser.' +
                   method).init, import_module('driver.' + driver).init) and then the reference's unmodified
                   `attempt_dispatch` (main.py:187-284), called at the points of a fixed schedule in which jobs are held
                   RUNNING while the next dispatch happens -- so the pending branch (--max-concurrent=2), the "(id,
                   candidate)" tuple return -> ExperimentGrid.add_to_grid (main.py:258-260) and the "maximum number of
                   jobs pending" return fire at known steps.  The hold is a gate around runner.run_python_job in the
                   forked job process (it waits for a file); the job itself -- runner.job_runner, ExperimentGrid's lock
                   and pickle, examples/braninpy/branin.py -- runs as it is.

The global numpy RNG is seeded once before the chooser is created (main.py never seeds it).  On exit a JSON record is
written: the job ids in dispatch order, their grid rows and values from the reference's own `expt-grid.pkl`, the files
of the experiment directory.
"""
import argparse
import importlib
import json
import os
import pickle
import runpy
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = ("GPEIChooser.py", "GPEIOptChooser.py", "GPEIperSecChooser.py")


def install_shims(tree):
    """INTEGRATION.md section 3, main.py: "replace the three files"."""
    for name in SHIMS:
        shutil.copy(os.path.join(ROOT, "dropin", "chooser", name), os.path.join(tree, "spearmint", "chooser", name))


def read_grid(expt_dir):
    """The reference's grid pickle (ExperimentGrid.py:165-183; moved into place atomically, so readable without its lock)."""
    for _ in range(200):
        try:
            with open(os.path.join(expt_dir, "expt-grid.pkl"), "rb") as fh:
                return pickle.load(fh)
        except (EOFError, FileNotFoundError, pickle.UnpicklingError):
            time.sleep(0.01)
    raise RuntimeError("expt-grid.pkl unreadable")


def wait_status(expt_dir, job_id, wanted, timeout=120.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        st = read_grid(expt_dir)["status"]
        if job_id < len(st) and int(st[job_id]) in wanted:
            return int(st[job_id])
        time.sleep(0.01)
    raise RuntimeError("job %d never reached status %s" % (job_id, wanted))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", required=True)
    ap.add_argument("--engine", choices=("hip", "oracle", "reference"), required=True)
    ap.add_argument("--mode", choices=("main", "dispatch"), required=True)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--schedule", default="", help="dispatch mode: comma list of d (dispatch), rN (release the N-th job "
                                                   "dispatched and wait until it is complete), wN (wait until the "
                                                   "N-th job is running)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="-- followed by main.py's own command line")
    a = ap.parse_args()
    argv = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest

    os.environ["PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION"] = "python"
    S = os.path.join(a.tree, "spearmint")       # the tree holds bin/, examples/, spearmint/ like the reference's spearmint/
    if a.engine != "reference":
        install_shims(a.tree)
    # sys.path as `PYTHONPATH=${DIR}/..[:repo] python ${DIR}/../spearmint/main.py` builds it (bin/spearmint)
    sys.path[0:0] = [S, a.tree] + ([ROOT] if a.engine != "reference" else [])
    made = []
    if a.engine == "oracle":
        sys.path.append(ROOT) if ROOT not in sys.path else None
        import spearmint_amd.engine as eng
        from tests.helpers import OracleEngine

        def make(*args, **kw):
            made.append("oracle")
            return OracleEngine()
        eng.Engine = make
    elif a.engine == "hip":
        import spearmint_amd.engine as eng
        real = eng.Engine

        def make(*args, **kw):
            e = real(*args, **kw)
            made.append(type(e).__module__ + "." + type(e).__name__)
            return e
        eng.Engine = make

    import numpy as np
    np.random.seed(a.seed)
    expt_config = argv[-1]
    expt_dir = os.path.dirname(os.path.realpath(expt_config))
    order = []

    if a.mode == "main":
        class Tee(object):                    # helpers.log writes main.py's messages to sys.stderr: keep a copy
            def __init__(self, real):
                self.real, self.text = real, []

            def write(self, s):
                self.text.append(s)
                return self.real.write(s)

            def flush(self):
                self.real.flush()
        tee = sys.stderr = Tee(sys.stderr)
        sys.argv = [os.path.join(S, "main.py")] + argv
        try:
            runpy.run_path(os.path.join(S, "main.py"), run_name="__main__")
        finally:
            sys.stderr = tee.real
        log_text = "".join(tee.text)
        method = [x for x in argv if x.startswith("--method=")][0].split("=", 1)[1]
        chooser_file = sys.modules["chooser." + method].__file__
    else:
        ref_main = importlib.import_module("main")
        runner = importlib.import_module("runner")
        gate_dir = os.path.join(a.tree, "gates")
        os.makedirs(gate_dir, exist_ok=True)
        run_job = runner.run_python_job

        def gated(job):                       # runs in the forked job process (driver/local.py:16)
            while not os.path.exists(os.path.join(gate_dir, "%d" % job.id)):
                time.sleep(0.005)
            return run_job(job)
        runner.run_python_job = gated

        sys.argv = [os.path.join(S, "main.py")] + argv
        options, args = ref_main.parse_args()
        ref_main.check_experiment_dirs(expt_dir)                                        # main.py:160
        module = importlib.import_module("chooser." + options.chooser_module)           # main.py:163
        chooser = module.init(expt_dir, options.chooser_args)                           # main.py:164
        driver = importlib.import_module("driver." + options.driver).init()             # main.py:170-171
        chooser_file = module.__file__
        steps = []
        for tok in [t for t in a.schedule.split(",") if t]:
            if tok == "d":
                before = read_grid(expt_dir)["status"].copy() if os.path.exists(os.path.join(expt_dir, "expt-grid.pkl")) else None
                more = ref_main.attempt_dispatch(expt_config, expt_dir, chooser, driver, options)
                after = read_grid(expt_dir)["status"]
                if before is None:
                    new = [i for i in range(len(after)) if after[i] in (1, 2)]
                else:
                    new = [i for i in range(len(after)) if after[i] in (1, 2) and (i >= len(before) or before[i] == 0)]
                assert len(new) <= 1, new
                order.extend(new)
                steps.append({"step": "d", "returned": bool(more), "job": (new[0] if new else None),
                              "pending_before": ([int(i) for i in np.nonzero((before == 1) | (before == 2))[0]] if before is not None else []),
                              "complete_before": (int(np.sum(before == 3)) if before is not None else 0)})
            elif tok[0] == "w":
                wait_status(expt_dir, order[int(tok[1:])], (2, 3))
            elif tok[0] == "r":
                job = order[int(tok[1:])]
                wait_status(expt_dir, job, (2,))
                open(os.path.join(gate_dir, "%d" % job), "w").close()
                wait_status(expt_dir, job, (3, -1))
            else:
                raise ValueError(tok)
        del chooser                           # the reference's GPEIChooser writes its state pickle in __del__

    import multiprocessing
    for p in multiprocessing.active_children():
        p.join(60)
    g = read_grid(expt_dir)
    if a.mode == "main":
        import re
        order = [int(m) for m in re.findall(r"selected job (\d+) from the grid", log_text)]       # main.py:262
        steps = {"dispatches": log_text.count("-" * 40), "held_back": log_text.count("Maximum number of jobs")}
    touched = [i for i in range(len(g["status"])) if g["status"][i] != 0]
    rec = {"engine": a.engine, "engines_made": made, "chooser_file": chooser_file, "order": order, "steps": steps,
           "grid_rows": int(g["grid"].shape[0]),
           "status": {str(i): int(g["status"][i]) for i in touched},
           "values": {str(i): (float(g["values"][i]) if np.isfinite(g["values"][i]) else None) for i in touched},
           "points": {str(i): [float(x) for x in g["grid"][i]] for i in touched},
           "files": sorted(os.listdir(expt_dir))}
    with open(a.out, "w") as fh:
        json.dump(rec, fh)


if __name__ == "__main__":
    main()
