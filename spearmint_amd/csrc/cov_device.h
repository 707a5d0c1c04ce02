// Device helpers of the covariance epilogue (correlation functions of the supported kernels on W values in lock step).
// Included by cov_kernels.hip (k_cov) and chol_kernels.hip (k_lean_flow builds its own tiles of K(X,X)): one source, the
// same instruction sequences, the same bits.
#pragma once
#include "common.h"

#define SQRT5 2.23606797749978969641  // == np.sqrt(5.0) (gp.py:32)
#define SQRT3 1.73205080756887719318  // == np.sqrt(3.0) (gp.py:31)

// The epilogue below runs once per covariance entry (4e8 times per draw at C3).  What bounds
// k_cov on gfx950 is the fp64 FMA units, which the fp64 MFMA and the fp64 VALU instructions
// appear to share (vector and matrix fp64 peaks are the same 78.6 TFLOP/s, and a v_mfma_f64_16x16x4 holds
// the unit for 64 cycles): PMC at C3 shows the unit 31 % busy with the Gram MFMAs + 44 % with
// VALU work, and the kernel time did not move with the stores removed, with a balanced
// one-round grid, with the chains interleaved four ways, or with the next tile's MFMAs
// software-pipelined under the epilogue -- only with fewer instructions.  So the helpers carry
// no special-case selects: matern52_corr clamps r^2 into a range where they need none and
// restores NaN / inf inputs with one fma at the end.  They work on W independent values in
// lock step (every stage is a loop over W) so that consecutive instructions are independent.

// sqrt(x) for x in [1e-300, 1e300], ~1 ulp: v_rsq_f64 seed + two coupled Newton steps.
template <int W>
__device__ __forceinline__ void sqrt_pos(const double (&x)[W], double (&out)[W])
{
    double g[W], hh[W], r[W], d[W];
#pragma unroll
    for (int w = 0; w < W; ++w) hh[w] = __builtin_amdgcn_rsq(x[w]);
#pragma unroll
    for (int w = 0; w < W; ++w) g[w] = x[w] * hh[w];
#pragma unroll
    for (int w = 0; w < W; ++w) hh[w] = 0.5 * hh[w];
#pragma unroll
    for (int w = 0; w < W; ++w) r[w] = fma(-hh[w], g[w], 0.5);
#pragma unroll
    for (int w = 0; w < W; ++w) g[w] = fma(g[w], r[w], g[w]);
#pragma unroll
    for (int w = 0; w < W; ++w) hh[w] = fma(hh[w], r[w], hh[w]);
#pragma unroll
    for (int w = 0; w < W; ++w) d[w] = fma(-g[w], g[w], x[w]);
#pragma unroll
    for (int w = 0; w < W; ++w) out[w] = fma(d[w], hh[w], g[w]);
}

// exp(-t) for t in [0, 800], ~1 ulp: n = rint(-t log2 e) by the 1.5 * 2^52 shift (its low word
// is n as an integer), Cody-Waite reduction with a two-part ln 2, degree-13 Taylor polynomial
// on |f| <= ln2/2, scale by 2^n (v_ldexp_f64 rounds into the denormals and to 0 below them, as
// exp does).
template <int W>
__device__ __forceinline__ void exp_neg(const double (&t)[W], double (&out)[W])
{
    const double SHIFT = 6755399441055744.0;   // 1.5 * 2^52
    double sh[W], f[W], p[W];
#pragma unroll
    for (int w = 0; w < W; ++w) sh[w] = fma(-t[w], 1.4426950408889634074, SHIFT);
#pragma unroll
    for (int w = 0; w < W; ++w) f[w] = fma(sh[w] - SHIFT, -6.93147180369123816490e-01, -t[w]);   // ln2_hi
#pragma unroll
    for (int w = 0; w < W; ++w) f[w] = fma(sh[w] - SHIFT, -1.90821492927058770002e-10, f[w]);     // ln2_lo
#pragma unroll
    for (int w = 0; w < W; ++w) p[w] = fma(1.6059043836821613e-10, f[w], 2.08767569878681e-09);   // 1/13!, 1/12!
#define SPX_EXP_STEP(C_)                              \
    _Pragma("unroll") for (int w = 0; w < W; ++w) p[w] = fma(p[w], f[w], C_)
    SPX_EXP_STEP(2.505210838544172e-08);    // 1/11!
    SPX_EXP_STEP(2.755731922398589e-07);    // 1/10!
    SPX_EXP_STEP(2.7557319223985893e-06);   // 1/9!
    SPX_EXP_STEP(2.48015873015873e-05);     // 1/8!
    SPX_EXP_STEP(1.984126984126984e-04);    // 1/7!
    SPX_EXP_STEP(1.3888888888888889e-03);   // 1/6!
    SPX_EXP_STEP(8.333333333333333e-03);    // 1/5!
    SPX_EXP_STEP(4.1666666666666664e-02);   // 1/4!
    SPX_EXP_STEP(1.6666666666666666e-01);   // 1/3!
    SPX_EXP_STEP(0.5);
    SPX_EXP_STEP(1.0);
    SPX_EXP_STEP(1.0);
#undef SPX_EXP_STEP
#pragma unroll
    for (int w = 0; w < W; ++w) out[w] = __builtin_ldexp(p[w], __double2loint(sh[w]));
}

// Matern-5/2 correlation from the Gram term and the two squared norms (gp.py:34-54, :120-127):
//   r2 = np.maximum(-t, 0) is taken as clamp(-t, 1e-300, 1.28e5): below 1e-300 the result is
//   exactly 1 either way; above 1.28e5, sqrt5 r > 800 and exp underflows to 0 either way.
//   The clamps are v_max/v_min, which drop NaN, so non-finite t (NaN or inf inputs, where the
//   reference yields NaN: np.maximum propagates NaN, and inf * exp(-inf) = NaN) is restored by
//   the closing fma(t, 0, .) -- 0 for finite t, NaN otherwise.
//   The polynomial (1 + sqrt5 r) + (5/3) r2 keeps the reference's association; sqrt and exp are
//   ~1 ulp device implementations.
// (the *_t forms take t = (gram - s1) - s2 already formed: the fused small-N EI kernel runs values of DIFFERENT rows in
// lock step; per value the instruction sequence -- and so the bits -- are the same)
template <int W>
__device__ __forceinline__ void matern52_corr_t(const double (&t)[W], double (&out)[W])
{
#pragma clang fp contract(off)
    double r2[W], r[W], sr[W], e[W];
#pragma unroll
    for (int w = 0; w < W; ++w) r2[w] = __builtin_fmin(__builtin_fmax(-t[w], 1e-300), 1.28e5);
    sqrt_pos<W>(r2, r);
#pragma unroll
    for (int w = 0; w < W; ++w) sr[w] = SQRT5 * r[w];
    exp_neg<W>(sr, e);
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const double poly = (1.0 + sr[w]) + (5.0 / 3.0) * r2[w];
        out[w] = fma(t[w], 0.0, poly * e[w]);
    }
}
template <int W>
__device__ __forceinline__ void matern52_corr(const double (&g)[W], double s1, const double (&s2)[W],
                                              double (&out)[W])
{
#pragma clang fp contract(off)
    double t[W];
#pragma unroll
    for (int w = 0; w < W; ++w) t[w] = (g[w] - s1) - s2[w];
    matern52_corr_t<W>(t, out);
}

// Matern-3/2 (gp.py:107-113): r = sqrt(dist2); (1 + sqrt3 r) exp(-sqrt3 r).  Same clamps as above
// (sqrt3 r > 800 beyond r2 = 2.1e5).
template <int W>
__device__ __forceinline__ void matern32_corr_t(const double (&t)[W], double (&out)[W])
{
#pragma clang fp contract(off)
    double r2[W], r[W], sr[W], e[W];
#pragma unroll
    for (int w = 0; w < W; ++w) r2[w] = __builtin_fmin(__builtin_fmax(-t[w], 1e-300), 2.1e5);
    sqrt_pos<W>(r2, r);
#pragma unroll
    for (int w = 0; w < W; ++w) sr[w] = SQRT3 * r[w];
    exp_neg<W>(sr, e);
#pragma unroll
    for (int w = 0; w < W; ++w) out[w] = fma(t[w], 0.0, (1.0 + sr[w]) * e[w]);
}
template <int W>
__device__ __forceinline__ void matern32_corr(const double (&g)[W], double s1, const double (&s2)[W],
                                              double (&out)[W])
{
#pragma clang fp contract(off)
    double t[W];
#pragma unroll
    for (int w = 0; w < W; ++w) t[w] = (g[w] - s1) - s2[w];
    matern32_corr_t<W>(t, out);
}

// squared exponential (gp.py:95-100; SE :87-93 is the same with unit length scales):
// exp(-0.5 dist2); 0.5 r2 > 800 underflows to 0 either way.
template <int W>
__device__ __forceinline__ void ardse_corr_t(const double (&t)[W], double (&out)[W])
{
#pragma clang fp contract(off)
    double hr[W], e[W];
#pragma unroll
    for (int w = 0; w < W; ++w) hr[w] = 0.5 * __builtin_fmin(__builtin_fmax(-t[w], 0.0), 1600.0);
    exp_neg<W>(hr, e);
#pragma unroll
    for (int w = 0; w < W; ++w) out[w] = fma(t[w], 0.0, e[w]);
}
template <int W>
__device__ __forceinline__ void ardse_corr(const double (&g)[W], double s1, const double (&s2)[W],
                                           double (&out)[W])
{
#pragma clang fp contract(off)
    double t[W];
#pragma unroll
    for (int w = 0; w < W; ++w) t[w] = (g[w] - s1) - s2[w];
    ardse_corr_t<W>(t, out);
}

template <int KIND, int W>
__device__ __forceinline__ void corr_of_kind(const double (&g)[W], double s1, const double (&s2)[W],
                                             double (&out)[W])
{
    if (KIND == SPX_COV_MATERN52) matern52_corr<W>(g, s1, s2, out);
    else if (KIND == SPX_COV_MATERN32) matern32_corr<W>(g, s1, s2, out);
    else ardse_corr<W>(g, s1, s2, out);
}


template <int KIND, int W>
__device__ __forceinline__ void corr_of_kind_t(const double (&t)[W], double (&out)[W])
{
    if (KIND == SPX_COV_MATERN52) matern52_corr_t<W>(t, out);
    else if (KIND == SPX_COV_MATERN32) matern32_corr_t<W>(t, out);
    else ardse_corr_t<W>(t, out);
}
