"""Dev: native sampler (spx_sample_hypers) vs the Python batched sampler on libspx, random problems: the same hyper rows bit for
bit and the same generator state.   python scripts/dev/fuzz_sampler.py [n=40] [seed=1]"""
import os, sys, tempfile, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
import spearmint_amd.chooser._base as b
b.log = lambda *a: None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
eng = None
for t in range(n):
    name = ["GPEIChooser", "GPEIOptChooser", "GPEIperSecChooser"][rs.randint(3)]
    mod = importlib.import_module("spearmint_amd.chooser." + name)
    o = importlib.import_module("spearmint_amd.chooser." + name); o.log = lambda *a: None
    N, D = int(rs.choice([16, 17, 40, 64, 65, 100, 128, 200, 256, 300, 513, 700])), int(rs.randint(1, 13))
    noiseless = int(rs.rand() < 0.3)
    comp = rs.rand(N, D); vals = np.sin(3 * comp).sum(axis=1) + 0.05 * rs.randn(N); durs = np.log(1.0 + 3.0 * comp[:, 0] + rs.rand(N))
    la, fp, fh = int(rs.randint(1, 10)), int(rs.randint(0, 8)), int(rs.randint(0, 4))
    out = {}
    for tag, extra in (("python", "sampler=python,lookahead=%d,follow=0:0" % la), ("native", "sampler=native,lookahead=%d,follow=%d:%d" % (la, fp, fh))):
        ch = mod.init(tempfile.mkdtemp(), "mcmc_iters=2,noiseless=%d,%s" % (noiseless, extra))
        if eng is not None:
            ch._eng = eng
        if name == "GPEIperSecChooser":
            ch._real_init(D, vals, np.exp(durs))
        else:
            ch._real_init(D, vals)
        eng = ch.engine()
        npr.seed(1000 + t)
        rows = []
        try:
            for _ in range(3):
                if hasattr(ch, "hyper_samples"):
                    ch.hyper_samples = []
                if name == "GPEIperSecChooser":
                    ch.sample_hypers(comp, vals, durs)
                    rows.append(np.concatenate((ch.current_hyper_row(), [ch.time_mean, ch.time_noise, ch.time_amp2], ch.time_ls)))
                else:
                    ch.sample_hypers(comp, vals)
                    rows.append(ch.current_hyper_row().copy())
            err = None
        except Exception as ex:
            err = type(ex).__name__
        out[tag] = (np.array(rows), npr.get_state(), err)
    a, c = out["python"], out["native"]
    same = a[2] == c[2] and a[0].shape == c[0].shape and np.array_equal(a[0], c[0]) and np.array_equal(a[1][1], c[1][1]) and a[1][2:] == c[1][2:]
    bad += not same
    print("%-18s N=%3d D=%2d noiseless=%d la=%d follow=%d:%d  %s%s" % (name, N, D, noiseless, la, fp, fh, "same chain" if same else "DIFFERENT", (" (both raised %s)" % a[2]) if a[2] else ""), flush=True)
print("%d of %d differ" % (bad, n))
sys.exit(1 if bad else 0)
