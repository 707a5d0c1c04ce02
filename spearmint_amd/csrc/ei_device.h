// EI from the predictive moments: shared by k_ei_finalize / k_ei_finalize_fant (predict_kernels.hip) and the fused small-N
// kernel (fused_kernels.hip) -- one source, the same bits.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// EI from the partial sums.  Phi follows scipy.special.ndtr (cephes): erf for
// |x|/sqrt2 < 1/sqrt2, erfc otherwise, so the lower tail keeps relative
// accuracy; phi = exp(-u^2/2)/sqrt(2 pi) as scipy.stats.norm.pdf.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double ndtr_dev(double a)
{
#pragma clang fp contract(off)
    const double x = a * 0.70710678118654752440;
    const double z = fabs(x);
    if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
    double y = 0.5 * erfc(z);
    if (x > 0) y = 1.0 - y;
    return y;
}

__device__ __forceinline__ double ei_dev(double func_m, double func_v, double best)
{
#pragma clang fp contract(off)
    const double func_s = sqrt(func_v);  // NaN for func_v < 0, as np.sqrt
    const double u = (best - func_m) / func_s;
    const double ncdf = ndtr_dev(u);
    const double npdf = exp(-(u * u) / 2.0) / 2.50662827463100050242;  // sqrt(2 pi)
    return func_s * (u * ncdf + npdf);
}

