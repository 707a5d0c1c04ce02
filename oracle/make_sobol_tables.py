"""TEST/BUILD INFRASTRUCTURE -- derive the Sobol direction tables and the Sobol golden vectors by
RUNNING the reference (needs /root/reference; never runs on the GPU box).

  spearmint_amd/data/sobol_dirs_bf40.npy    (40, 30) uint32    spearmint-lite/sobol_lib.py
  spearmint_amd/data/sobol_dirs_jk1111.npy  (1111, 30) uint32  spearmint/spearmint/sobol_lib.py
      = the module-global `v` after the reference's own initialisation for dim_num = dim_max
        (initial direction numbers -> Bratley-Fox recurrence -> scaling by 2^(29-b)); numbers only.
  tests/golden/sobol.npz                    i4_sobol_generate outputs for a few (m, n, skip)

Usage:  python oracle/make_sobol_tables.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_py3  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (table, m, n, skip)        (called in this order: exercises the reference's
    "bf_2x1000_s1": ("bf40", 2, 1000, 1),          # restart / skip-forward / continue branches)
    "bf_40x64_s1": ("bf40", 40, 64, 1),
    "bf_7x50_s0": ("bf40", 7, 50, 0),
    "bf_5x33_s1000": ("bf40", 5, 33, 1000),
    "bf_5x10_s1020": ("bf40", 5, 10, 1020),
    "bf_3x6_sneg": ("bf40", 3, 6, -2),
    "jk_2x1000_s1": ("jk1111", 2, 1000, 1),
    "jk_1111x8_s1": ("jk1111", 1111, 8, 1),
    "jk_32x200_s12345": ("jk1111", 32, 200, 12345),
    "jk_33x17_s7": ("jk1111", 33, 17, 7),
    "jk_8x300_s1": ("jk1111", 8, 300, 1),
}


def dirs_of(mod, dim_max):
    mod.i4_sobol_generate(dim_max, 1, 1)          # forces initialisation of every row
    v = np.array(mod.v)
    assert v.shape == (dim_max, 30) and mod.maxcol == 30 and mod.recipd == 2.0 ** -30
    assert np.all(v == np.floor(v)) and v.min() >= 0 and v.max() < 2 ** 30
    return v.astype(np.uint32)


def main():
    mods = {"bf40": ref_py3.load_lite_sobol(), "jk1111": ref_py3.load()["sobol_lib"]}
    os.makedirs(os.path.join(ROOT, "spearmint_amd", "data"), exist_ok=True)
    for name, dim_max in (("bf40", 40), ("jk1111", 1111)):
        d = dirs_of(mods[name], dim_max)
        np.save(os.path.join(ROOT, "spearmint_amd", "data", "sobol_dirs_%s.npy" % name), d)
        print(name, d.shape, d.dtype)
    out = {}
    for key, (table, m, n, skip) in CASES.items():
        out[key] = np.array(mods[table].i4_sobol_generate(m, n, skip))
        out[key + "_args"] = np.array([m, n, skip], dtype=np.int64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sobol.npz"), **out)
    print("wrote tests/golden/sobol.npz with %d cases" % len(CASES))


if __name__ == "__main__":
    main()
