// ARD Matern-5/2 covariance kernels for gfx950 (hand-written HIP, wave64).
//
// Reference arithmetic being reproduced (spearmint/spearmint/gp.py:34-54 dist2,
// :120-127 Matern52; chooser cov GPEIChooser.py:117-122):
//     xx  = x / ls                               (true division)
//     r2  = max(-((xx1 . (2 xx2)^T - |xx1|^2) - |xx2|^2), 0) ; r2 = |r2| ; r = sqrt(r2)
//     k   = (1 + sqrt5 r + 5/3 r2) * exp(-sqrt5 r)
//     K_self  = amp2 (k + 1e-6 I) + noise I      K_cross = amp2 k
//
// The pairwise Gram term xx1 . (2 xx2)^T is an fp64 MFMA GEMM
// (v_mfma_f64_16x16x4_f64, contraction over the padded input dimension), the
// norm / Matern part is the epilogue on the accumulator registers, so the
// squared-distance matrix never exists in memory.
#include "common.h"

#ifndef SPX_COV_HOIST
#define SPX_COV_HOIST 1   // keep the column-side fragments in registers across row tiles
#endif

// ---------------------------------------------------------------------------
// x / ls, row norms.   One thread per (row, draw).
//   xs[h][row][d]  = factor * (x[row][d] / ls[h][d])     (0 for pad rows / dims)
//   xs2 (optional) = 2 * xs  (exact: the pre-doubled second operand of gp.py:50, same launch)
//   sumsq[h][row]  = sum_d (x[row][d] / ls[h][d])^2
// ---------------------------------------------------------------------------
#define SR_DC 32   // feature columns per LDS pass
// ROWS rows per workgroup: 256, or 64 when the launch would otherwise be a handful of workgroups (the log-likelihood
// path: N = 2048 x 6 draws was 48 workgroups and 20 us; the arithmetic per row is the same either way)
// (the body of k_scale_rows for row block bx of draw h: also the first part of k_lean_prologue)
template <int ROWS>
__device__ __forceinline__ void scale_rows_body(
    const double* __restrict__ x, int64_t n, int64_t n_pad, int D, int Dp,
    const double* __restrict__ ls, int ls_stride, double factor,
    double* __restrict__ xs, double* __restrict__ sumsq, double* __restrict__ xs2, int bx, int h)
{
#pragma clang fp contract(off)
    // ROWS rows per workgroup, staged through LDS so that both the read of x (rows of D doubles)
    // and the write of xs (rows of Dp doubles) are contiguous across the wave; each thread then
    // owns one row and accumulates its squared norm left to right.
    __shared__ double T[ROWS][SR_DC + 1];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)bx * ROWS;
    const int rows = (int)((n_pad - row0 < ROWS) ? (n_pad - row0) : ROWS);
    const int64_t row = row0 + tid;
    const double* lsh = ls + (size_t)h * ls_stride;
    double* o = xs + ((size_t)h * n_pad + row0) * Dp;
    double* o2 = xs2 ? xs2 + ((size_t)h * n_pad + row0) * Dp : nullptr;
    double acc = 0.0;
    for (int d0 = 0; d0 < Dp; d0 += SR_DC) {
        const int dc = (Dp - d0 < SR_DC) ? (Dp - d0) : SR_DC;                    // output columns
        const int dr = (D - d0 < 0) ? 0 : ((D - d0 < SR_DC) ? (D - d0) : SR_DC);  // real columns
        for (int e = tid; e < rows * dr; e += 256) {
            const int r = e / dr, c = e - r * dr;
            T[r][c] = (row0 + r < n) ? x[(size_t)(row0 + r) * D + d0 + c] : 0.0;
        }
        __syncthreads();
        if (tid < rows) {
            if (row < n) {
                for (int c = 0; c < dr; ++c) {
                    const double v = T[tid][c] / lsh[d0 + c];
                    acc = acc + v * v;
                    T[tid][c] = factor * v;
                }
                for (int c = dr; c < dc; ++c) T[tid][c] = 0.0;
            } else {
                for (int c = 0; c < dc; ++c) T[tid][c] = 0.0;
            }
        }
        __syncthreads();
        for (int e = tid; e < rows * dc; e += 256) {
            const int r = e / dc, c = e - r * dc;
            o[(size_t)r * Dp + d0 + c] = T[r][c];
            if (o2) o2[(size_t)r * Dp + d0 + c] = 2.0 * T[r][c];
        }
        __syncthreads();
    }
    if (tid < rows) sumsq[(size_t)h * n_pad + row] = acc;
}

template <int ROWS>
__global__ __launch_bounds__(256) void k_scale_rows(
    const double* __restrict__ x, int64_t n, int64_t n_pad, int D, int Dp,
    const double* __restrict__ ls, int ls_stride, double factor,
    double* __restrict__ xs, double* __restrict__ sumsq, double* __restrict__ xs2, int* __restrict__ zero_ints, int n_zero)
{
    // (the first kernel of a factorisation also clears its not-PD flags: one stream operation fewer per call)
    if (zero_ints && blockIdx.x == 0 && blockIdx.y == 0)
        for (int e = threadIdx.x; e < n_zero; e += 256) zero_ints[e] = 0;
    scale_rows_body<ROWS>(x, n, n_pad, D, Dp, ls, ls_stride, factor, xs, sumsq, xs2, blockIdx.x, blockIdx.y);
}

// The log-likelihood path's two prologue launches as ONE (round 5): the first `nsb` workgroups of a draw scale the observations
// (k_scale_rows<ROWS>, with the pre-doubled copy), the others write the right-hand-side block row in tile storage -- row 0 =
// vals - mean, the rest 0 -- and clear the draw's not-PD flag and (k_lean_step_ps) its hand-off flags: k_lean_rhs_init's
// body, value for value (chol_kernels.hip).  The two halves do not depend on each other; a call of spx_gp_logprob is a chain
// of small dependent launches, and each one costs ~4 us whatever it does (profiles: scripts/dev/trace_small_lp.sh).
template <int ROWS>
__global__ __launch_bounds__(256) void k_lean_prologue(
    const double* __restrict__ x, int64_t n, int64_t n_pad, int D, int Dp,
    const double* __restrict__ ls, int ls_stride,
    double* __restrict__ xs, double* __restrict__ sumsq, double* __restrict__ xs2, int nsb,
    const double* __restrict__ vals, const double* __restrict__ htab, double* __restrict__ rhs,
    int* __restrict__ info, int* __restrict__ flags)
{
    const int h = blockIdx.y;
    if ((int)blockIdx.x < nsb) {
        scale_rows_body<ROWS>(x, n, n_pad, D, Dp, ls, ls_stride, 1.0, xs, sumsq, xs2, blockIdx.x, h);
        return;
    }
    const int bx = blockIdx.x - nsb;
    const int Np = (int)n_pad, N = (int)n;
    const int idx = bx * 256 + threadIdx.x;      // over [nblk][4096]
    if (bx == 0) {
        if (threadIdx.x == 0) info[h] = 0;
        if (flags && (int)threadIdx.x < Np / SPX_NB) flags[h * (Np / SPX_NB) + threadIdx.x] = 0;
    }
    if (idx >= SPX_NB * Np) return;
    const int J = idx >> 12, e = idx & 4095;
    const int t = (e >> 1) & 255, q = ((e >> 9) << 1) | (e & 1);   // thread slot, value q = nt * 4 + r
    const int wave = t >> 6, lane = t & 63, r = q & 3, nt = q >> 2;
    const int rowi = 16 * wave + (lane >> 4) + 4 * r, col = J * SPX_NB + 16 * nt + (lane & 15);
    double v = 0.0;
    if (rowi == 0 && col < N) v = vals[col] - htab[h * SPX_HT + 0];
    rhs[(size_t)h * SPX_NB * Np + idx] = v;
}

void launch_lean_prologue(hipStream_t s, const double* x, int64_t n, int64_t n_pad, int D, int Dp, const double* ls,
                          int ls_stride, int nh, double* xs, double* sumsq, double* xs2, const double* vals,
                          const double* htab, double* rhs, int* info, int* flags)
{
    const int nrhs = (int)((SPX_NB * n_pad + 255) / 256);
    if (((n_pad + 255) / 256) * nh < 512) {
        const int nsb = (int)((n_pad + 63) / 64);
        hipLaunchKernelGGL(k_lean_prologue<64>, dim3(nsb + nrhs, nh), dim3(256), 0, s, x, n, n_pad, D, Dp, ls, ls_stride, xs, sumsq,
                           xs2, nsb, vals, htab, rhs, info, flags);
    } else {
        const int nsb = (int)((n_pad + 255) / 256);
        hipLaunchKernelGGL(k_lean_prologue<256>, dim3(nsb + nrhs, nh), dim3(256), 0, s, x, n, n_pad, D, Dp, ls, ls_stride, xs, sumsq,
                           xs2, nsb, vals, htab, rhs, info, flags);
    }
}

void launch_scale_rows(hipStream_t s, const double* x, int64_t n, int64_t n_pad, int D, int Dp,
                       const double* ls, int ls_stride, int nh, double factor,
                       double* xs, double* sumsq, double* xs2, int* zero_ints, int n_zero)
{
    if (((n_pad + 255) / 256) * nh < 512) {
        dim3 grid((unsigned)((n_pad + 63) / 64), nh);
        hipLaunchKernelGGL(k_scale_rows<64>, grid, dim3(256), 0, s, x, n, n_pad, D, Dp, ls, ls_stride, factor, xs, sumsq, xs2, zero_ints, n_zero);
    } else {
        dim3 grid((unsigned)((n_pad + 255) / 256), nh);
        hipLaunchKernelGGL(k_scale_rows<256>, grid, dim3(256), 0, s, x, n, n_pad, D, Dp, ls, ls_stride, factor, xs, sumsq, xs2, zero_ints, n_zero);
    }
}

// compute units of the current device (cached per device id: one process may drive several GPUs)
static int spx_cov_cus()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

#include "cov_device.h"   // sqrt_pos / exp_neg / *_corr: shared with the log-likelihood path's in-kernel covariance (chol_kernels.hip)

// The rows [jbeg, jend) x the 64 columns from c0 of draw h: the body of k_cov (one call per workgroup) and of k_cov_flat
// (a workgroup's share of the launch, run by run).
template <int MODE, int QC, int KIND>
__device__ __forceinline__ void cov_run(
    const double* __restrict__ Xs, const double* __restrict__ s1,
    const double* __restrict__ Cs, const double* __restrict__ s2,
    const double* __restrict__ htab, const double* __restrict__ alpha,
    double* __restrict__ out, int N, int Np, int Mc, int Dp, int nchunks, int64_t ldo, int h, int c0, int jbeg, int jend)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int Q = Dp >> 2;
    const double* Xh = Xs + (size_t)h * Np * Dp;
    const double* Ch = Cs + (size_t)h * Mc * Dp;
    const double* s1h = s1 + (size_t)h * Np;
    const double noise = htab[h * SPX_HT + 1];
    const double amp2 = htab[h * SPX_HT + 2];

    double s2v[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) s2v[nt] = s2[(size_t)h * Mc + c0 + 16 * nt + li];

    double bf[4][QC];
    if (nchunks == 1 && SPX_COV_HOIST) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const double* p = Ch + (size_t)(c0 + 16 * nt + li) * Dp + g * Q;
#pragma unroll
            for (int q = 0; q < QC; ++q) bf[nt][q] = p[q];
        }
    }

    // MODE 2 accumulates sum_j k[j][c] alpha[j] for this lane's column(s)
    double colsum[4] = {0.0, 0.0, 0.0, 0.0};

    // Matern epilogue of one 16 x 64 tile on the accumulator layout: row = j0 + g + 4 r,
    // col = c0 + 16 nt + li
    auto epilogue = [&](const d4 (&acc)[4], int j0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + g + 4 * r;
            const double s1v = s1h[j];
            double av = 0.0;
            if (MODE == 2) av = alpha[(size_t)h * Np + j];
            const double amp_j = (j < N) ? amp2 : 0.0;
            double gv[4], cv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) gv[nt] = acc[nt][r];
#if defined(SPX_COV_ABL) && SPX_COV_ABL == 2   // dev, timing only (WRONG results): no correlation function
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = (gv[nt] - s1v) - s2v[nt];
#else
            corr_of_kind<KIND, 4>(gv, s1v, s2v, cv);
#endif
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int c = c0 + 16 * nt + li;
                const double corr = cv[nt];
                if (MODE == 0) {
                    // pad rows (j >= N) are written as 0 (amp_j = 0); a NaN column stays NaN there,
                    // which is harmless: that candidate's result is NaN anyway
#if defined(SPX_COV_ABL) && SPX_COV_ABL == 1   // dev, timing only (WRONG results): no store stream (a value that never occurs)
                    if (corr == 123.456)
#endif
                    out[((size_t)h * Np + j) * ldo + c] = amp_j * corr;
                } else if (MODE == 1 || MODE == 3) {
#pragma clang fp contract(off)
                    const double eye = (j == c) ? 1.0 : 0.0;
                    double v = amp2 * (corr + 1e-6 * eye) + noise * eye;
                    if (j >= N || c >= N) v = eye;
                    if (MODE == 1) {
                        out[((size_t)h * Np + j) * ldo + c] = v;
                    } else {
                        // tile-major, accumulator order inside a 64x64 tile as 8 planes of [256][2]
                        // (chol_kernels.hip, log-likelihood path): value q = 4 nt + r of thread t
                        const size_t tile = (size_t)(j >> 6) * (Np >> 6) + (c >> 6);
                        const int t = ((j >> 4) & 3) * 64 + lane, q = nt * 4 + r;
                        out[(size_t)h * Np * Np + tile * 4096 + (((q >> 1) * 256 + t) * 2 + (q & 1))] = v;
                    }
                } else {
                    // pad rows have alpha == 0 and finite corr
                    colsum[nt] += (amp2 * corr) * av;
                }
            }
        }
    };

    // MODE 0/1: a workgroup covers `rows_per_wg` rows (a multiple of 128) so that the prologue
    // (column-side fragment and norm loads, ~1-2 us of latency) is amortised over many row tiles
    // (MODE 3 relies on j0 being a multiple of 16 and c0 of 64: both hold for every launch geometry below)
    for (int j0 = jbeg + wave * 16; j0 < jend; j0 += 64) {
        // tile-major output feeds the right-looking factorisation, which only ever reads the tiles on and below
        // the diagonal: the 64 x 64 tiles above it are not computed
        if (MODE == 3 && (j0 >> 6) < (c0 >> 6)) continue;
        d4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int ch = 0; ch < nchunks; ++ch) {
            double af[QC];
            const double* pa = Xh + (size_t)(j0 + li) * Dp + g * Q + ch * QC;
#pragma unroll
            for (int q = 0; q < QC; ++q) af[q] = pa[q];
            if (nchunks > 1 || !SPX_COV_HOIST) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const double* p = Ch + (size_t)(c0 + 16 * nt + li) * Dp + g * Q + ch * QC;
#pragma unroll
                    for (int q = 0; q < QC; ++q) bf[nt][q] = p[q];
                }
            }
#if defined(SPX_COV_ABL) && SPX_COV_ABL == 3   // dev, timing only (WRONG results): no Gram MFMAs
#pragma unroll
            for (int q = 0; q < QC; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt][q & 3] += af[q] * bf[nt][q];
#else
#pragma unroll
            for (int q = 0; q < QC; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA_F64(af[q], bf[nt][q], acc[nt]);
#endif
        }
        epilogue(acc, j0);
    }

    if (MODE == 2) {
        __shared__ double red[4][64];
        const double mean = htab[h * SPX_HT + 0];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            double v = colsum[nt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) red[wave][16 * nt + li] = v;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int c = threadIdx.x;
            const double dot = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
            out[(size_t)h * Mc + c0 + c] = exp(dot + mean);
        }
    }
}

template <int MODE, int QC, int KIND>
__global__ __launch_bounds__(256, 2) void k_cov(
    const double* __restrict__ Xs, const double* __restrict__ s1,
    const double* __restrict__ Cs, const double* __restrict__ s2,
    const double* __restrict__ htab, const double* __restrict__ alpha,
    double* __restrict__ out, int N, int Np, int Mc, int Dp, int nchunks, int64_t ldo, int rows_per_wg, int live_rows)
{
    // live_rows (MODE 0, a multiple of 16, >= N): the rows from there on are padding whose consumer skips them
    // (k_predict_gemm_tri<true>): they are not computed and not written
    const int jtop = (MODE == 0 && live_rows > 0) ? live_rows : Np;
    const int jbeg = (MODE == 2) ? 0 : blockIdx.y * rows_per_wg;
    const int jend = (MODE == 2) ? Np : min(jtop, jbeg + rows_per_wg);
    cov_run<MODE, QC, KIND>(Xs, s1, Cs, s2, htab, alpha, out, N, Np, Mc, Dp, nchunks, ldo, blockIdx.z, blockIdx.x * 64, jbeg, jend);
}

// K(X*,X) of a LARGE launch (MODE 0), flat: the launch's work -- units of 128 rows x 64 columns, numbered draw-major, column
// block next, row chunk fastest -- is dealt out in equal contiguous shares to a number of workgroups that is a whole multiple
// of what the chip holds (2 per CU), so every residency round is full.  k_cov's grid at C3 was 1 792 workgroups on 512 places:
// 3.5 rounds, the fourth half empty (13 % of the launch), 1 120 times per step.  Same arithmetic per element, same bits.
template <int QC, int KIND>
__global__ __launch_bounds__(256, 2) void k_cov_flat(
    const double* __restrict__ Xs, const double* __restrict__ s1,
    const double* __restrict__ Cs, const double* __restrict__ s2,
    const double* __restrict__ htab, double* __restrict__ out, int N, int Np, int Mc, int Dp, int nchunks, int64_t ldo, int nh,
    int live_rows)
{
    const int jtop = live_rows > 0 ? live_rows : Np;        // (as in k_cov)
    const int nrc = (jtop + 127) >> 7, ncb = Mc >> 6;       // row chunks that hold live rows
    const int64_t per_h = (int64_t)ncb * nrc, U = per_h * nh;
    int64_t u = U * blockIdx.x / gridDim.x;
    const int64_t u1 = U * (blockIdx.x + 1) / gridDim.x;
    while (u < u1) {
        const int h = (int)(u / per_h);
        const int rem = (int)(u - (int64_t)h * per_h);
        const int cb = rem / nrc, rc = rem - cb * nrc;
        const int run = (int)min((int64_t)(nrc - rc), u1 - u);       // row chunks of this column block that are ours
        cov_run<0, QC, KIND>(Xs, s1, Cs, s2, htab, nullptr, out, N, Np, Mc, Dp, nchunks, ldo, h, cb * 64, rc * 128,
                             min(jtop, (rc + run) * 128));
        u += run;
    }
}

template <int MODE, int KIND>
static void launch_cov_kind(hipStream_t s, const double* Xs, const double* s1, const double* Cs,
                            const double* s2, const double* htab, const double* alpha, double* out,
                            int N, int Np, int Mc, int Dp, int nh, int64_t ldo, int live_rows, bool allow_flat)
{
    const int Q = Dp / 4;
    int rows_per_wg = (Np >= 1024) ? 512 : ((Np >= 256) ? 256 : 128);
#ifdef SPX_DEV_KNOBS   // (make DEV_KNOBS=1: development builds only; the shipped library reads no SPX_COV_* variable)
    static const char* rpw = getenv("SPX_COV_RPW");
    if (rpw && MODE == 3) rows_per_wg = atoi(rpw);
#endif
    const int row_top = (MODE == 0 && live_rows > 0) ? live_rows : Np;     // (MODE 0: rows past live_rows are left alone)
    dim3 grid(Mc / 64, (MODE == 2) ? 1 : (row_top + rows_per_wg - 1) / rows_per_wg, nh);
    dim3 block(256);
    if (MODE == 0) {
        // a launch of several residency rounds: whole rounds, equal shares (k_cov_flat); `places` = 2 workgroups per CU
        // (handle option "cov_flat" = 0: always the 3-D grid -- bit-identical, kept for A/B runs and its test)
        const int64_t places = 2 * (int64_t)spx_cov_cus(), wgs = (int64_t)grid.x * grid.y * grid.z;
        const int64_t units = (int64_t)nh * (Mc / 64) * (((live_rows > 0 ? live_rows : Np) + 127) / 128);
        if (wgs > places && units >= 4 * places && allow_flat) {
            // shares of at most 16 units (2048 rows x 64 columns): as many whole rounds as that takes
            const int64_t rounds = (units + 16 * places - 1) / (16 * places);
            const dim3 fgrid((unsigned)(places * rounds));
#define SPX_COVF_LAUNCH(QC_)                                                                                   \
    hipLaunchKernelGGL((k_cov_flat<QC_, KIND>), fgrid, block, 0, s, Xs, s1, Cs, s2, htab, out, N, Np, Mc, Dp, Q / QC_, ldo, nh, live_rows)
            if (Q == 1) SPX_COVF_LAUNCH(1);
            else if (Q == 2) SPX_COVF_LAUNCH(2);
            else if (Q == 4) SPX_COVF_LAUNCH(4);
            else SPX_COVF_LAUNCH(8);
#undef SPX_COVF_LAUNCH
            return;
        }
    }
#define SPX_COV_LAUNCH(QC_)                                                                        \
    hipLaunchKernelGGL((k_cov<MODE, QC_, KIND>), grid, block, 0, s, Xs, s1, Cs, s2, htab, alpha, out, N, \
                       Np, Mc, Dp, Q / QC_, ldo, rows_per_wg, live_rows)
    if (Q == 1) SPX_COV_LAUNCH(1);
    else if (Q == 2) SPX_COV_LAUNCH(2);
    else if (Q == 4) SPX_COV_LAUNCH(4);
    else SPX_COV_LAUNCH(8);
#undef SPX_COV_LAUNCH
}

template <int MODE>
static void launch_cov_mode(hipStream_t s, int kind, const double* Xs, const double* s1, const double* Cs,
                            const double* s2, const double* htab, const double* alpha, double* out,
                            int N, int Np, int Mc, int Dp, int nh, int64_t ldo, int live_rows = 0, bool allow_flat = true)
{
    if (kind == SPX_COV_MATERN32)
        launch_cov_kind<MODE, SPX_COV_MATERN32>(s, Xs, s1, Cs, s2, htab, alpha, out, N, Np, Mc, Dp, nh, ldo, live_rows, allow_flat);
    else if (kind == SPX_COV_ARDSE)
        launch_cov_kind<MODE, SPX_COV_ARDSE>(s, Xs, s1, Cs, s2, htab, alpha, out, N, Np, Mc, Dp, nh, ldo, live_rows, allow_flat);
    else
        launch_cov_kind<MODE, SPX_COV_MATERN52>(s, Xs, s1, Cs, s2, htab, alpha, out, N, Np, Mc, Dp, nh, ldo, live_rows, allow_flat);
}

void launch_cov_cross(hipStream_t s, const double* Xs, const double* s1, const double* Cs,
                      const double* s2, const double* htab, double* Kst, int N, int Np, int Mc,
                      int Dp, int nh, int kind, int live_rows, bool allow_flat)
{
    launch_cov_mode<0>(s, kind, Xs, s1, Cs, s2, htab, nullptr, Kst, N, Np, Mc, Dp, nh, Mc, live_rows, allow_flat);
}

// X2s = 2 * Xs (the reference multiplies the second operand by 2, gp.py:50)
void launch_cov_self(hipStream_t s, const double* Xs, const double* s1, const double* X2s,
                     const double* htab, double* K, int N, int Np, int Dp, int nh, bool tiled, int kind)
{
    if (tiled) launch_cov_mode<3>(s, kind, Xs, s1, X2s, s1, htab, nullptr, K, N, Np, Np, Dp, nh, Np);
    else launch_cov_mode<1>(s, kind, Xs, s1, X2s, s1, htab, nullptr, K, N, Np, Np, Dp, nh, Np);
}

void launch_cross_mean(hipStream_t s, const double* Xs, const double* s1, const double* Cs,
                       const double* s2, const double* htab, const double* alpha, double* out,
                       int N, int Np, int Mc, int Dp, int nh, int kind)
{
    launch_cov_mode<2>(s, kind, Xs, s1, Cs, s2, htab, alpha, out, N, Np, Mc, Dp, nh, Mc);
}
