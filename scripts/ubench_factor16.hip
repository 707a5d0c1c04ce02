// Dev micro-benchmark: cycles per pivot of the in-wave 16x16 Cholesky (+ inverse) built from rank-1
// v_mfma_f64_16x16x4 updates (csrc/chol_kernels.hip: factor16_mfma), in several variants.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_factor16.hip -o scripts/ubench_factor16 && scripts/ubench_factor16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double readlane_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// V: 0 = factor + inverse (production), 1 = factor only, 2 = like 0 without the d > 0 select before rsq,
//    3 = like 2 with mask multipliers instead of selects, 4 = factor only, rsq without refinement (chain floor probe)
template <int V>
__device__ __forceinline__ void factor16(d4& C, d4& X, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { X[r] = (q + 4 * r == c) ? 1.0 : 0.0; U[r] = 0.0; }
    double mk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mk[k] = (q == k) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int kq = j & 3, rj = j >> 2;
        double d = readlane_f64(C[rj], j + 16 * kq);
        if (V == 0 || V == 1) {
            if (!(d > 0.0)) { if (!bad) bad = j + 1; d = 1.0; }
        }
        const double y0 = __builtin_amdgcn_rsq(d);
        double rinv;
        if (V == 4) rinv = y0;
        else {
            const double e0 = fma(-d * y0, y0, 1.0);
            rinv = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        }
        if (V >= 2) { if (!(rinv == rinv) && !bad) bad = j + 1; }   // off the chain: NaN for d <= 0 / NaN
        const bool grp = (q == kq);
        double b, lcol;
        if (V == 3) { const double rm = rinv * mk[kq]; b = C[rj] * rm; lcol = b; }
        else { lcol = C[rj] * rinv; b = grp ? lcol : 0.0; }
        C = MFMA_F64(-b, b, C);
        if (V == 0 || V == 2 || V == 3) {
            const double xs = X[rj] * rinv;
            const double bX = (V == 3) ? xs * mk[kq] : (grp ? xs : 0.0);
            const double aX = (grp && c > j) ? -lcol : 0.0;
            X = MFMA_F64(aX, bX, X);
            X[rj] = grp ? xs : X[rj];
        }
        double sd = d * rinv;
        sd = fma(fma(-sd, sd, d), 0.5 * rinv, sd);
        const double keep = (c == j) ? sd : ((c > j) ? lcol : 0.0);
        U[rj] = grp ? keep : U[rj];
    }
}

// row of 16 lanes q -> row q + 1 for q = 0, 2 (v_permlane16_swap_b32: odd rows of the first operand <-> even rows of the second)
__device__ __forceinline__ double from_even_row(double v)
{
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]);
}

// V5: TWO pivots per MFMA (k-slots j & 3 and (j & 3) + 1): every value is formed by the same operations as in V0 -- the
// second pivot's row is brought up to date by one explicit fma (what the rank-1 MFMA of the first pivot would have done
// to it) -- so the factor and the inverse are the same bits IF the MFMA adds its k-slots in ascending order, one rounded
// fma each (checked below against V0).
template <>
__device__ __forceinline__ void factor16<5>(d4& C, d4& Xout, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) { X[r] = (q + 4 * r == c) ? 1.0 : 0.0; U[r] = 0.0; Xout[r] = 0.0; }
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const int qa = j & 3, rj = j >> 2;
        const bool ga = (q == qa), gb = (q == qa + 1);
        double d0 = readlane_f64(C[rj], j + 16 * qa);
        const double e = readlane_f64(C[rj], j + 1 + 16 * qa);
        const double d1 = readlane_f64(C[rj], j + 1 + 16 * (qa + 1));
        if (!(d0 > 0.0)) { if (!bad) bad = j + 1; d0 = 1.0; }
        const double y0 = __builtin_amdgcn_rsq(d0);
        const double e0 = fma(-d0 * y0, y0, 1.0);
        const double rinv0 = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        const double l10 = e * rinv0;
        double d1p = fma(-l10, l10, d1);
        if (!(d1p > 0.0)) { if (!bad) bad = j + 2; d1p = 1.0; }
        const double y1 = __builtin_amdgcn_rsq(d1p);
        const double e1 = fma(-d1p * y1, y1, 1.0);
        const double rinv1 = fma(y1 * e1, fma(0.375, e1, 0.5), y1);
        const double l0 = C[rj] * rinv0;                 // (meaningful in group qa: column j of L)
        const double x0 = X[rj] * rinv0;                 // (group qa: row j of the inverse, final)
        const double l0n = from_even_row(l0), x0n = from_even_row(x0);
        const double l1 = fma(-l10, l0n, C[rj]) * rinv1; // (group qa + 1: column j + 1 of L)
        const double x1 = fma(-l10, x0n, X[rj]) * rinv1;
        const double bC = ga ? l0 : (gb ? l1 : 0.0);
        C = MFMA_F64(-bC, bC, C);
        const double bX = ga ? x0 : (gb ? x1 : 0.0);
        const double aX = (ga && c > j) ? -l0 : ((gb && c > j + 1) ? -l1 : 0.0);
        X = MFMA_F64(aX, bX, X);
        Xout[rj] = ga ? x0 : (gb ? x1 : Xout[rj]);
        double sd0 = d0 * rinv0;
        sd0 = fma(fma(-sd0, sd0, d0), 0.5 * rinv0, sd0);
        double sd1 = d1p * rinv1;
        sd1 = fma(fma(-sd1, sd1, d1p), 0.5 * rinv1, sd1);
        const double keep0 = (c == j) ? sd0 : ((c > j) ? l0 : 0.0);
        const double keep1 = (c == j + 1) ? sd1 : ((c > j + 1) ? l1 : 0.0);
        U[rj] = ga ? keep0 : (gb ? keep1 : U[rj]);
    }
}


// V11 = V5 without the selects in front of the two rsq (a failed pivot is recorded; its draw's numbers are NaN instead of finite garbage)
template <>
__device__ __forceinline__ void factor16<11>(d4& C, d4& Xout, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) { X[r] = (q + 4 * r == c) ? 1.0 : 0.0; U[r] = 0.0; Xout[r] = 0.0; }
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const int qa = j & 3, rj = j >> 2;
        const bool ga = (q == qa), gb = (q == qa + 1);
        double d0 = readlane_f64(C[rj], j + 16 * qa);
        const double e = readlane_f64(C[rj], j + 1 + 16 * qa);
        const double d1 = readlane_f64(C[rj], j + 1 + 16 * (qa + 1));
        if (!(d0 > 0.0) && !bad) bad = j + 1;
        const double y0 = __builtin_amdgcn_rsq(d0);
        const double e0 = fma(-d0 * y0, y0, 1.0);
        const double rinv0 = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        const double l10 = e * rinv0;
        double d1p = fma(-l10, l10, d1);
        if (!(d1p > 0.0) && !bad) bad = j + 2;
        const double y1 = __builtin_amdgcn_rsq(d1p);
        const double e1 = fma(-d1p * y1, y1, 1.0);
        const double rinv1 = fma(y1 * e1, fma(0.375, e1, 0.5), y1);
        const double l0 = C[rj] * rinv0;                 // (meaningful in group qa: column j of L)
        const double x0 = X[rj] * rinv0;                 // (group qa: row j of the inverse, final)
        const double l0n = from_even_row(l0), x0n = from_even_row(x0);
        const double l1 = fma(-l10, l0n, C[rj]) * rinv1; // (group qa + 1: column j + 1 of L)
        const double x1 = fma(-l10, x0n, X[rj]) * rinv1;
        const double bC = ga ? l0 : (gb ? l1 : 0.0);
        C = MFMA_F64(-bC, bC, C);
        const double bX = ga ? x0 : (gb ? x1 : 0.0);
        const double aX = (ga && c > j) ? -l0 : ((gb && c > j + 1) ? -l1 : 0.0);
        X = MFMA_F64(aX, bX, X);
        Xout[rj] = ga ? x0 : (gb ? x1 : Xout[rj]);
        double sd0 = d0 * rinv0;
        sd0 = fma(fma(-sd0, sd0, d0), 0.5 * rinv0, sd0);
        double sd1 = d1p * rinv1;
        sd1 = fma(fma(-sd1, sd1, d1p), 0.5 * rinv1, sd1);
        const double keep0 = (c == j) ? sd0 : ((c > j) ? l0 : 0.0);
        const double keep1 = (c == j + 1) ? sd1 : ((c > j + 1) ? l1 : 0.0);
        U[rj] = ga ? keep0 : (gb ? keep1 : U[rj]);
    }
}



// V7 = V5 with the NEXT pair's pivot block (d0, e, d1) formed ahead of the MFMA that updates it: the three entries by the
// two rounded fmas the matrix pipe applies to them (k-slot order), on v_readlane values of the two columns just formed --
// the scalar recurrence of pair j + 2 then starts without waiting for the MFMA of pair j to retire.
template <>
__device__ __forceinline__ void factor16<7>(d4& C, d4& Xout, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) { X[r] = (q + 4 * r == c) ? 1.0 : 0.0; U[r] = 0.0; Xout[r] = 0.0; }
    double d0 = readlane_f64(C[0], 0), e = readlane_f64(C[0], 1), d1 = readlane_f64(C[0], 1 + 16);
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const int qa = j & 3, rj = j >> 2;
        const bool ga = (q == qa), gb = (q == qa + 1);
        if (!(d0 > 0.0)) { if (!bad) bad = j + 1; d0 = 1.0; }
        const double y0 = __builtin_amdgcn_rsq(d0);
        const double e0 = fma(-d0 * y0, y0, 1.0);
        const double rinv0 = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        const double l10 = e * rinv0;
        double d1p = fma(-l10, l10, d1);
        if (!(d1p > 0.0)) { if (!bad) bad = j + 2; d1p = 1.0; }
        const double y1 = __builtin_amdgcn_rsq(d1p);
        const double e1 = fma(-d1p * y1, y1, 1.0);
        const double rinv1 = fma(y1 * e1, fma(0.375, e1, 0.5), y1);
        const double l0 = C[rj] * rinv0;
        const double x0 = X[rj] * rinv0;
        const double l0n = from_even_row(l0), x0n = from_even_row(x0);
        const double l1 = fma(-l10, l0n, C[rj]) * rinv1;
        const double x1 = fma(-l10, x0n, X[rj]) * rinv1;
        double nd0 = 0.0, ne = 0.0, nd1 = 0.0;
        if (j + 2 < 16) {
            const int jn = j + 2, qn = jn & 3, rn = jn >> 2;
            const double a0 = readlane_f64(l0, 16 * qa + jn), a1 = readlane_f64(l0, 16 * qa + jn + 1);
            const double b0 = readlane_f64(l1, 16 * (qa + 1) + jn), b1 = readlane_f64(l1, 16 * (qa + 1) + jn + 1);
            const double c00 = readlane_f64(C[rn], jn + 16 * qn), c01 = readlane_f64(C[rn], jn + 1 + 16 * qn);
            const double c11 = readlane_f64(C[rn], jn + 1 + 16 * (qn + 1));
            nd0 = fma(-b0, b0, fma(-a0, a0, c00));
            ne = fma(-b0, b1, fma(-a0, a1, c01));
            nd1 = fma(-b1, b1, fma(-a1, a1, c11));
        }
        const double bC = ga ? l0 : (gb ? l1 : 0.0);
        C = MFMA_F64(-bC, bC, C);
        const double bX = ga ? x0 : (gb ? x1 : 0.0);
        const double aX = (ga && c > j) ? -l0 : ((gb && c > j + 1) ? -l1 : 0.0);
        X = MFMA_F64(aX, bX, X);
        Xout[rj] = ga ? x0 : (gb ? x1 : Xout[rj]);
        double sd0 = d0 * rinv0;
        sd0 = fma(fma(-sd0, sd0, d0), 0.5 * rinv0, sd0);
        double sd1 = d1p * rinv1;
        sd1 = fma(fma(-sd1, sd1, d1p), 0.5 * rinv1, sd1);
        const double keep0 = (c == j) ? sd0 : ((c > j) ? l0 : 0.0);
        const double keep1 = (c == j + 1) ? sd1 : ((c > j + 1) ? l1 : 0.0);
        U[rj] = ga ? keep0 : (gb ? keep1 : U[rj]);
        d0 = nd0; e = ne; d1 = nd1;
    }
}

// all-gather over the four 16-lane rows: v of row a -> out[a] in every row (one v_permlane16_swap + two v_permlane32_swap per dword)
__device__ __forceinline__ void gather_rows(double v, double (&out)[4])
{
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // a[0] = rows (0,0,2,2), a[1] = rows (1,1,3,3)
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const auto a0 = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);   // [0] = row 0 everywhere, [1] = row 2
    const auto a1 = __builtin_amdgcn_permlane32_swap(a[1], a[1], false, false);   // [0] = row 1, [1] = row 3
    const auto b0 = __builtin_amdgcn_permlane32_swap(b[0], b[0], false, false);
    const auto b1 = __builtin_amdgcn_permlane32_swap(b[1], b[1], false, false);
    out[0] = __hiloint2double(b0[0], a0[0]);
    out[1] = __hiloint2double(b1[0], a1[0]);
    out[2] = __hiloint2double(b0[1], a0[1]);
    out[3] = __hiloint2double(b1[1], a1[1]);
}
__device__ __forceinline__ double rsq_refined(double d)
{
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e0 = fma(-d * y0, y0, 1.0);
    return fma(y0 * e0, fma(0.375, e0, 0.5), y0);
}
__device__ __forceinline__ double sqrt_from(double d, double rinv)
{
    double sd = d * rinv;
    return fma(fma(-sd, sd, d), 0.5 * rinv, sd);
}

// V8 = V5 with fewer instructions per pair (the wave is bound by issue, ~100 instructions per pair, as much as by latency):
// the inverse's MFMA takes the factor's A operand as it is (rows <= j + 1 of X are dead: nothing reads what the update does
// to them); U and the inverse's rows are the MFMA operands themselves, picked up once per register (two pairs); the
// diagonal's sqrt(d) refinement -- off the chain -- is done once, vectorised, after the last pivot, from the pivots kept
// lane by lane (same formula on the same values: same bits).
template <>
__device__ __forceinline__ void factor16<8>(d4& C, d4& Xout, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = (q + 4 * r == c) ? 1.0 : 0.0;
    double dvec = 1.0, bCa = 0.0, bXa = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const int qa = j & 3, rj = j >> 2;
        const bool ga = (q == qa);
        double d0 = readlane_f64(C[rj], j + 16 * qa);
        const double e = readlane_f64(C[rj], j + 1 + 16 * qa);
        const double d1 = readlane_f64(C[rj], j + 1 + 16 * (qa + 1));
        if (!(d0 > 0.0)) { if (!bad) bad = j + 1; d0 = 1.0; }
        const double y0 = __builtin_amdgcn_rsq(d0);
        const double e0 = fma(-d0 * y0, y0, 1.0);
        const double rinv0 = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        const double l10 = e * rinv0;
        double d1p = fma(-l10, l10, d1);
        if (!(d1p > 0.0)) { if (!bad) bad = j + 2; d1p = 1.0; }
        const double y1 = __builtin_amdgcn_rsq(d1p);
        const double e1 = fma(-d1p * y1, y1, 1.0);
        const double rinv1 = fma(y1 * e1, fma(0.375, e1, 0.5), y1);
        const double l0 = C[rj] * rinv0;
        const double x0 = X[rj] * rinv0;
        const double l0n = from_even_row(l0), x0n = from_even_row(x0);
        const double l1 = fma(-l10, l0n, C[rj]) * rinv1;
        const double x1 = fma(-l10, x0n, X[rj]) * rinv1;
        // groups qa / qa + 1 carry the two columns; the other two k-slots are zero
        const bool live = (q >> 1) == (qa >> 1);
        const double bC = live ? (ga ? l0 : l1) : 0.0;
        const double bX = live ? (ga ? x0 : x1) : 0.0;
        C = MFMA_F64(-bC, bC, C);
        X = MFMA_F64(-bC, bX, X);
        dvec = (c == j) ? d0 : ((c == j + 1) ? d1p : dvec);
        if (qa == 0) { bCa = bC; bXa = bX; }
        else {
            U[rj] = (q < 2) ? bCa : bC;
            Xout[rj] = (q < 2) ? bXa : bX;
        }
    }
    // U[r] (lane (q, c)) = L[c][q + 4 r]: the column entries above, the diagonal refined to ~0.5 ulp, zeros below; the
    // inverse's rows are zero right of the diagonal by construction (X starts as the identity)
    const double rv = rsq_refined(dvec);
    const double sdv = sqrt_from(dvec, rv);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q + 4 * r;
        U[r] = (c > row) ? U[r] : ((c == row) ? sdv : 0.0);
    }
}

// V9 / V10: the pair's second pivot in closed form.  For the 2 x 2 pivot block [[a, b], [b, c]]: l11^2 = c - b^2 / a =
// det / a, so 1 / l11 = sqrt(a) rsq(det) with det = fma(a, c, -b^2): the second reciprocal square root starts two levels
// behind the first instead of behind all of it (rsq, its refinement, l10, the pivot's own fma) -- the pair's dependent chain
// is ~13 levels instead of ~20, at ~25 cycles a level.  Not the same bits as V0 (the second column of a pair is scaled by
// sqrt(a) rsq(det), ~2 ulp, instead of rsq(c - l10^2), ~1 ulp); same backward error.  V10: no select in front of the rsq
// (a failed pivot is recorded, its numbers are garbage as the result is anyway).
template <int V>
__device__ __forceinline__ void factor16_cf(d4& C, d4& Xout, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = (q + 4 * r == c) ? 1.0 : 0.0;
    double dvec = 1.0, bCa = 0.0, bXa = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const int qa = j & 3, rj = j >> 2;
        const bool ga = (q == qa);
        double a = readlane_f64(C[rj], j + 16 * qa);
        const double b = readlane_f64(C[rj], j + 1 + 16 * qa);
        const double cc = readlane_f64(C[rj], j + 1 + 16 * (qa + 1));
        if (V == 9) { if (!(a > 0.0)) { if (!bad) bad = j + 1; a = 1.0; } }
        else if (!(a > 0.0) && !bad) bad = j + 1;
        double det = fma(a, cc, -(b * b));
        if (V == 9) { if (!(det > 0.0)) { if (!bad) bad = j + 2; det = 1.0; } }
        else if (!(det > 0.0) && !bad) bad = j + 2;
        const double rinv0 = rsq_refined(a);
        const double rdet = rsq_refined(det);
        const double sd0 = a * rinv0;
        const double rinv1 = sd0 * rdet;
        const double l10 = b * rinv0;
        const double l0 = C[rj] * rinv0;
        const double x0 = X[rj] * rinv0;
        const double l0n = from_even_row(l0), x0n = from_even_row(x0);
        const double l1 = fma(-l10, l0n, C[rj]) * rinv1;
        const double x1 = fma(-l10, x0n, X[rj]) * rinv1;
        const bool live = (q >> 1) == (qa >> 1);
        const double bC = live ? (ga ? l0 : l1) : 0.0;
        const double bX = live ? (ga ? x0 : x1) : 0.0;
        C = MFMA_F64(-bC, bC, C);
        X = MFMA_F64(-bC, bX, X);
        const double d1p = fma(-l10, l10, cc);        // (off the chain: the diagonal entry is the refined sqrt of this)
        dvec = (c == j) ? a : ((c == j + 1) ? d1p : dvec);
        if (qa == 0) { bCa = bC; bXa = bX; }
        else {
            U[rj] = (q < 2) ? bCa : bC;
            Xout[rj] = (q < 2) ? bXa : bX;
        }
    }
    const double rv = rsq_refined(dvec);
    const double sdv = sqrt_from(dvec, rv);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q + 4 * r;
        U[r] = (c > row) ? U[r] : ((c == row) ? sdv : 0.0);
    }
}
template <> __device__ __forceinline__ void factor16<9>(d4& C, d4& Xout, d4& U, int lane, int& bad) { factor16_cf<9>(C, Xout, U, lane, bad); }
template <> __device__ __forceinline__ void factor16<10>(d4& C, d4& Xout, d4& U, int lane, int& bad) { factor16_cf<10>(C, Xout, U, lane, bad); }

// V6: FOUR pivots per pair of MFMAs (all k-slots).  The four rows j .. j + 3 of the block sit in the four lane rows of one
// register; an all-gather gives every lane the four entries of ITS column, the 4x4 pivot block comes by v_readlane, and
// every lane runs the little recurrence for its column -- each value by the operations the one-pivot form applies to it.
template <>
__device__ __forceinline__ void factor16<6>(d4& C, d4& Xout, d4& U, int lane, int& bad)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) { X[r] = (q + 4 * r == c) ? 1.0 : 0.0; U[r] = 0.0; Xout[r] = 0.0; }
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        const int rj = j >> 2;
        // the 4x4 pivot block (lower triangle), before any update of this round
        double d00 = readlane_f64(C[rj], j + 0 + 16 * 0);
        const double d10 = readlane_f64(C[rj], j + 0 + 16 * 1), d11 = readlane_f64(C[rj], j + 1 + 16 * 1);
        const double d20 = readlane_f64(C[rj], j + 0 + 16 * 2), d21 = readlane_f64(C[rj], j + 1 + 16 * 2), d22 = readlane_f64(C[rj], j + 2 + 16 * 2);
        const double d30 = readlane_f64(C[rj], j + 0 + 16 * 3), d31 = readlane_f64(C[rj], j + 1 + 16 * 3), d32 = readlane_f64(C[rj], j + 2 + 16 * 3),
                     d33 = readlane_f64(C[rj], j + 3 + 16 * 3);
        double cc[4], xx[4];
        gather_rows(C[rj], cc);
        gather_rows(X[rj], xx);
        if (!(d00 > 0.0)) { if (!bad) bad = j + 1; d00 = 1.0; }
        const double r0 = rsq_refined(d00);
        const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
        double p1 = fma(-l10, l10, d11);
        if (!(p1 > 0.0)) { if (!bad) bad = j + 2; p1 = 1.0; }
        const double r1 = rsq_refined(p1);
        const double l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
        double p2 = fma(-l21, l21, fma(-l20, l20, d22));
        if (!(p2 > 0.0)) { if (!bad) bad = j + 3; p2 = 1.0; }
        const double r2 = rsq_refined(p2);
        const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * r2;
        double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
        if (!(p3 > 0.0)) { if (!bad) bad = j + 4; p3 = 1.0; }
        const double r3 = rsq_refined(p3);
        // this lane's column of L (entries j .. j + 3 of row c of L ... as columns l_a[c]) and of the inverse's rows
        const double l0 = cc[0] * r0;
        const double l1 = fma(-l10, l0, cc[1]) * r1;
        const double l2 = fma(-l21, l1, fma(-l20, l0, cc[2])) * r2;
        const double l3 = fma(-l32, l2, fma(-l31, l1, fma(-l30, l0, cc[3]))) * r3;
        const double x0 = xx[0] * r0;
        const double x1 = fma(-l10, x0, xx[1]) * r1;
        const double x2 = fma(-l21, x1, fma(-l20, x0, xx[2])) * r2;
        const double x3 = fma(-l32, x2, fma(-l31, x1, fma(-l30, x0, xx[3]))) * r3;
        const double lq = (q == 0) ? l0 : (q == 1) ? l1 : (q == 2) ? l2 : l3;
        const double xq = (q == 0) ? x0 : (q == 1) ? x1 : (q == 2) ? x2 : x3;
        C = MFMA_F64(-lq, lq, C);
        const double aX = (c > j + q) ? -lq : 0.0;
        X = MFMA_F64(aX, xq, X);
        Xout[rj] = xq;
        const double sdq = (q == 0) ? sqrt_from(d00, r0) : (q == 1) ? sqrt_from(p1, r1) : (q == 2) ? sqrt_from(p2, r2) : sqrt_from(p3, r3);
        U[rj] = (c == j + q) ? sdq : ((c > j + q) ? lq : 0.0);
    }
}

template <int V>
__global__ void bench(const double* S, double* out, long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    d4 C0;
    for (int r = 0; r < 4; ++r) C0[r] = S[(q + 4 * r) * 16 + c];
    d4 C, X, U;
    int bad = 0;
    double acc = 0.0;
    long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        C = C0;
        C[0] += acc * 1e-300;          // serialise the repetitions
        factor16<V>(C, X, U, lane, bad);
        acc += U[3] + X[3];
    }
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    for (int r = 0; r < 4; ++r) {
        out[((q + 4 * r) * 16 + c) * 2 + 0] = U[r];
        out[((q + 4 * r) * 16 + c) * 2 + 1] = X[r];
    }
    if (lane == 0) out[512] = acc + bad;
}


// V12: the inverse on a HELPER wave (another SIMD of the CU: scripts/ubench_neigh.hip shows that it does not slow the chain wave).
// The chain wave (wave 0) runs V5 without anything that belongs to X and posts, per pair, the MFMA operand bC (one double per
// lane) and the three scalars (rinv0, rinv1, l10) in an LDS mailbox, a sequence word last; the helper (wave 1) polls the word,
// forms x0, x1 and the inverse's rank-2 MFMA exactly as V5 does.  Timed: the chain wave alone, and until the helper is done too.
__global__ void bench_two(const double* S, double* out, long long* cyc, int reps)
{
    __shared__ double mb_vec[8][64];
    __shared__ double mb_sc[8][4];
    __shared__ volatile int mb_seq[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    d4 C0;
    for (int r = 0; r < 4; ++r) C0[r] = S[(q + 4 * r) * 16 + c];
    if (threadIdx.x < 8) mb_seq[threadIdx.x] = 0;
    __syncthreads();
    d4 C, X, U, Xout;
    int bad = 0;
    double acc = 0.0;
    long long t0 = clock64(), t1 = 0;
    for (int it = 0; it < reps; ++it) {
        const int tag = it + 1;
        if (wave == 0) {
            C = C0;
            C[0] += acc * 1e-300;
#pragma unroll
            for (int r = 0; r < 4; ++r) U[r] = 0.0;
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const int qa = j & 3, rj = j >> 2;
                const bool ga = (q == qa), gb = (q == qa + 1);
                double d0 = readlane_f64(C[rj], j + 16 * qa);
                const double e = readlane_f64(C[rj], j + 1 + 16 * qa);
                const double d1 = readlane_f64(C[rj], j + 1 + 16 * (qa + 1));
                if (!(d0 > 0.0)) { if (!bad) bad = j + 1; d0 = 1.0; }
                const double y0 = __builtin_amdgcn_rsq(d0);
                const double e0 = fma(-d0 * y0, y0, 1.0);
                const double rinv0 = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
                const double l10 = e * rinv0;
                double d1p = fma(-l10, l10, d1);
                if (!(d1p > 0.0)) { if (!bad) bad = j + 2; d1p = 1.0; }
                const double y1 = __builtin_amdgcn_rsq(d1p);
                const double e1 = fma(-d1p * y1, y1, 1.0);
                const double rinv1 = fma(y1 * e1, fma(0.375, e1, 0.5), y1);
                const double l0 = C[rj] * rinv0;
                const double l0n = from_even_row(l0);
                const double l1 = fma(-l10, l0n, C[rj]) * rinv1;
                const double bC = ga ? l0 : (gb ? l1 : 0.0);
                mb_vec[j >> 1][lane] = bC;
                if (lane < 3) mb_sc[j >> 1][lane] = (lane == 0) ? rinv0 : ((lane == 1) ? rinv1 : l10);
                asm volatile("" ::: "memory");      // (a wave's LDS operations are served in order: no wait between the data and the word)
                if (lane == 0) mb_seq[j >> 1] = tag;
                C = MFMA_F64(-bC, bC, C);
                double sd0 = d0 * rinv0;
                sd0 = fma(fma(-sd0, sd0, d0), 0.5 * rinv0, sd0);
                double sd1 = d1p * rinv1;
                sd1 = fma(fma(-sd1, sd1, d1p), 0.5 * rinv1, sd1);
                const double keep0 = (c == j) ? sd0 : ((c > j) ? l0 : 0.0);
                const double keep1 = (c == j + 1) ? sd1 : ((c > j + 1) ? l1 : 0.0);
                U[rj] = ga ? keep0 : (gb ? keep1 : U[rj]);
            }
            acc += U[3];
            if (it == reps - 1) t1 = clock64();
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { X[r] = (q + 4 * r == c) ? 1.0 : 0.0; Xout[r] = 0.0; }
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const int qa = j & 3, rj = j >> 2, p = j >> 1;
                const bool ga = (q == qa), gb = (q == qa + 1);
                while (mb_seq[p] != tag) { }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const double bC = mb_vec[p][lane];
                const double rinv0 = mb_sc[p][0], rinv1 = mb_sc[p][1], l10 = mb_sc[p][2];
                const double x0 = X[rj] * rinv0;
                const double x0n = from_even_row(x0);
                const double x1 = fma(-l10, x0n, X[rj]) * rinv1;
                const double bX = ga ? x0 : (gb ? x1 : 0.0);
                const double aX = (ga && c > j) ? -bC : ((gb && c > j + 1) ? -bC : 0.0);
                X = MFMA_F64(aX, bX, X);
                Xout[rj] = ga ? x0 : (gb ? x1 : Xout[rj]);
            }
        }
        __syncthreads();       // (the end of the phase: both waves done)
    }
    long long t2 = clock64();
    if (threadIdx.x == 0) { cyc[0] = t2 - t0; cyc[1] = t1; }
    if (wave == 0) for (int r = 0; r < 4; ++r) out[((q + 4 * r) * 16 + c) * 2 + 0] = U[r];
    if (wave == 1) for (int r = 0; r < 4; ++r) out[((q + 4 * r) * 16 + c) * 2 + 1] = Xout[r];
    if (threadIdx.x == 0) out[512] = acc + bad;
}

static double g_ref[513];
template <int V>
void run(const double* dS, double* dOut, long long* dCyc, const double* hS, const char* name)
{
    const int reps = 200;
    hipLaunchKernelGGL(bench<V>, dim3(1), dim3(64), 0, 0, dS, dOut, dCyc, reps);
    hipLaunchKernelGGL(bench<V>, dim3(1), dim3(64), 0, 0, dS, dOut, dCyc, reps);
    hipDeviceSynchronize();
    long long cyc; double out[513];
    hipMemcpy(&cyc, dCyc, 8, hipMemcpyDeviceToHost);
    hipMemcpy(out, dOut, sizeof out, hipMemcpyDeviceToHost);
    // check U^T U == S
    double err = 0.0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += out[(k * 16 + i) * 2] * out[(k * 16 + j) * 2];
        err = fmax(err, fabs(s - hS[i * 16 + j]));
    }
    // against V0, bit for bit: U (all of it: zeros below the diagonal) and the lower triangle of X = L^-1 (out[(row * 16 + col) * 2 + {0, 1}])
    if (V == 0) for (int i = 0; i < 512; ++i) g_ref[i] = out[i];
    int ndiff = 0;
    for (int row = 0; row < 16; ++row)
        for (int col = 0; col < 16; ++col) {
            if (out[(row * 16 + col) * 2] != g_ref[(row * 16 + col) * 2]) ++ndiff;
            if (col <= row && out[(row * 16 + col) * 2 + 1] != g_ref[(row * 16 + col) * 2 + 1]) ++ndiff;
        }
    double errx = 0.0;     // |X U^T - I| over the lower triangle of X
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double sx = 0.0;
        for (int k = 0; k < 16; ++k) sx += (k <= i ? out[(i * 16 + k) * 2 + 1] : 0.0) * out[(j * 16 + k) * 2];
        errx = fmax(errx, fabs(sx - (i == j)));
    }
    printf("%-44s %7.1f cycles / 16x16 block  = %5.1f / pivot   |U^T U - S| = %.2e  |X L - I| = %.1e   values of U / tril(X) differing from V0: %d\n", name,
           (double)cyc / reps, (double)cyc / reps / 16, err, errx, ndiff);
}

int main()
{
    double hS[256], A[16 * 40];
    unsigned s = 12345;
    for (int i = 0; i < 16 * 40; ++i) { s = s * 1664525u + 1013904223u; A[i] = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double v = (i == j) ? 0.1 : 0.0;
        for (int k = 0; k < 40; ++k) v += A[i * 40 + k] * A[j * 40 + k];
        hS[i * 16 + j] = v;
    }
    double *dS, *dOut; long long* dCyc;
    hipMalloc(&dS, sizeof hS); hipMalloc(&dOut, 513 * 8); hipMalloc(&dCyc, 8);
    hipMemcpy(dS, hS, sizeof hS, hipMemcpyHostToDevice);
    run<0>(dS, dOut, dCyc, hS, "V0 factor + inverse (production)");
    run<1>(dS, dOut, dCyc, hS, "V1 factor only");
    run<2>(dS, dOut, dCyc, hS, "V2 factor + inverse, no select before rsq");
    run<3>(dS, dOut, dCyc, hS, "V3 = V2 with mask multipliers");
    run<4>(dS, dOut, dCyc, hS, "V4 factor only, raw rsq (chain floor probe)");
    run<5>(dS, dOut, dCyc, hS, "V5 factor + inverse, TWO pivots per MFMA");
    run<6>(dS, dOut, dCyc, hS, "V6 factor + inverse, FOUR pivots per MFMA");
    run<7>(dS, dOut, dCyc, hS, "V7 = V5 + next pivot block ahead of the MFMA");
    run<8>(dS, dOut, dCyc, hS, "V8 = V5 with fewer instructions per pair");
    run<9>(dS, dOut, dCyc, hS, "V9 = V8, second pivot in closed form");
    run<10>(dS, dOut, dCyc, hS, "V10 = V9 without selects before the rsq");
    run<11>(dS, dOut, dCyc, hS, "V11 = V5 without selects before the rsq");
    {
        const int reps = 200;
        long long* dC2; hipMalloc(&dC2, 16);
        hipLaunchKernelGGL(bench_two, dim3(1), dim3(128), 0, 0, dS, dOut, dC2, reps);
        hipLaunchKernelGGL(bench_two, dim3(1), dim3(128), 0, 0, dS, dOut, dC2, reps);
        hipDeviceSynchronize();
        long long cy[2]; double out[513];
        hipMemcpy(cy, dC2, 16, hipMemcpyDeviceToHost);
        hipMemcpy(out, dOut, sizeof out, hipMemcpyDeviceToHost);
        int ndiff = 0;
        for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col) {
                if (out[(row * 16 + col) * 2] != g_ref[(row * 16 + col) * 2]) ++ndiff;
                if (col <= row && out[(row * 16 + col) * 2 + 1] != g_ref[(row * 16 + col) * 2 + 1]) ++ndiff;
            }
        printf("V12 inverse on a helper wave (LDS mailbox)   %7.1f cycles / 16x16 block incl. the barrier that ends the phase = %5.1f / pivot   values of U / tril(X) differing from V0: %d\n",
               (double)cy[0] / reps, (double)cy[0] / reps / 16, ndiff);
    }
    return 0;
}
