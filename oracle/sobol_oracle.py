"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Sobol grid generator
(SURVEY.md 8(f) row 4).  Only tests/ may import this; the product path is the HIP kernel
k_sobol_grid behind spx_sobol_grid.

Follows  spearmint/spearmint/sobol_lib.py:125-157 (i4_sobol_generate), :158-13787 (i4_sobol)
and      spearmint-lite/sobol_lib.py:124-156, :157-431  (same generator, 40-dimension tables).

The reference keeps the running vector in module globals and advances it one Gray-code step per
call (lastq ^= v[:, lo0(seed)], sobol_lib.py i4_sobol tail); whatever the call history, the
point it returns for a seed s >= 0 is

    x_s[d] = ( XOR over the set bits b of gray(s) = s ^ (s >> 1) of  V[d][b] ) * 2^-30

with V the scaled direction integers it builds at initialisation (v[d][b] * 2^(29-b), 30
columns, recipd = 2^-30).  i4_sobol_generate(m, n, skip) asks for seeds skip-1 ... skip+n-2,
and i4_sobol maps a negative seed to 0.  V is a table (Bratley & Fox, ACM TOMS 659, for the
40-dimension file; Joe & Kuo's numbers for the 1111-dimension file): it is read from
spearmint_amd/data/sobol_dirs_*.npy, which oracle/make_sobol_tables.py derives by running the
reference's own initialisation.  Pinned by tests/golden/sobol.npz (reference outputs).
"""
import numpy as np

NCOL = 30


def point_seeds(n, skip):
    """seed of the j-th generated point, j = 0..n-1 (sobol_lib.py i4_sobol_generate loop)."""
    s = int(skip) - 1 + np.arange(int(n), dtype=np.int64)
    return np.maximum(s, 0)


def i4_sobol_generate(m, n, skip, dirs):
    """r (m, n) float64 exactly as the reference returns it.  dirs: (dim_max, 30) uint32."""
    dirs = np.asarray(dirs, dtype=np.uint32)
    if not (1 <= m <= dirs.shape[0]):
        raise ValueError("dimension %d outside 1..%d" % (m, dirs.shape[0]))
    s = point_seeds(n, skip)
    if n and s.max() >= (1 << NCOL):
        raise ValueError("too many points for 30 direction columns")
    gray = (s ^ (s >> 1)).astype(np.uint32)
    x = np.zeros((m, int(n)), dtype=np.uint32)
    for b in range(NCOL):
        on = ((gray >> np.uint32(b)) & np.uint32(1)).astype(bool)
        if on.any():
            x[:, on] ^= dirs[:m, b][:, None]
    return x.astype(np.float64) * 2.0 ** -NCOL
