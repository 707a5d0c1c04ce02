// The EI pass for SMALL observation counts (N <= 128: where Spearmint lives -- S/main.py:83-85, a grid of 20 000
// candidates, mcmc_iters = 10, tens of observations) as ONE kernel: K(X,X*) tile -> beta = W K* -> sum beta^2, sum beta gamma
// -> EI, per (candidate, draw), with nothing of K* or beta in memory ("fully fused path, 8 D + 8 bytes
// per evaluation").  Reference arithmetic: gp.py:34-54,120-127 (K*), GPEIChooser.py:195-206 (solve_triangular as the
// product with W = L^-1, moments, EI).
//
// The general path (k_cov<0> -> HBM -> k_predict_gemm_tri -> partial sums -> k_ei_finalize) is three launches per group of
// draws and writes / re-reads 8 N bytes per evaluation; at N = 256 it runs at 0.60 of the fp64 peak, below that the launches
// and the staging dominate.  What makes the fusion cheap on gfx950:
//   * v_mfma_f64_16x16x4's ACCUMULATOR layout (reg r of lane (g, li) = element [g + 4 r][li]) is, register by register,
//     its B-OPERAND layout (lane (g, li) supplies B[k = k0 + g][n = li]) for the contraction steps k0 = 4 r: the 16 x 16
//     Gram tile, once the correlation function has been applied to its accumulator registers, IS the K* operand of four
//     MFMA steps of the solve.  No LDS staging, no cross-lane traffic, for K*.
//   * so a wavefront is self-contained: it owns 32 candidates (two 16-column tiles) for ALL rows of W -- accumulators
//     acc[8 row tiles][2] = 128 VGPRs -- and walks the 16-row tiles kt of K*: Gram MFMAs (K = padded D), epilogue on
//     8 values in lock step, then for r = 0..3 the MFMAs of step k0 = 16 kt + 4 r against the row tiles t >= kt of W (the
//     tiles above the diagonal of the lower-triangular W are never multiplied; tiles beyond ceil(N / 16) are padding and
//     are skipped altogether -- their beta is exactly 0).
//   * W^T of the draw (128 x 128, K-major, row stride 144 doubles = the bank-conflict-free stride of the predict GEMM) sits
//     in LDS, loaded once per workgroup; a workgroup (8 waves, one per CU: 144 KB of LDS) then strides over candidate blocks.
// Bits: every value follows the general path's instruction sequences -- K* through k_cov's Gram order and cov_device.h,
// beta through the same MFMA chain (k ascending in fours), the column sums in k_predict_gemm_tri's order (wave row wm =
// row tiles of parity wm: fma over (mt, r), + xor 16, + xor 32, then wm 0 + wm 1), EI through ei_device.h -- so the result
// equals the general path's bit for bit (tests/test_gpu_a_parity.py::test_fused_small_n_*).
#include "common.h"
#include "cov_device.h"
#include "ei_device.h"

#define FU_LDW 144                 // LDS row stride of W^T (doubles): rows k0 + g, g = 0, 1 fall on disjoint bank halves
#define FU_NP 128
#define FU_CPW 16                  // candidates per wave pass
#ifndef FU_WAVES
#define FU_WAVES 8
#endif
#define FU_THREADS (64 * FU_WAVES)
#define FU_PASSES 8                // tiles per wave, at most (their sums wait in registers: 4 lane groups x 2 slots)
#define FU_CPB (FU_CPW * FU_WAVES) // candidates per workgroup pass

// One pass of a wave: 16 candidates (cb .. cb + 15) against the NTL live row tiles of W; out: sum beta^2 and sum beta gamma
// of this lane's candidate (the same value in the four lanes (g, li) of a candidate).  NTL is a template parameter so that
// the whole pass is straight-line code: with run-time tile guards every MFMA sat in a basic block of its own, behind its
// LDS read and a wait, and the four lock-step epilogue chains were split into four dependent ones.
// ONECH: the padded input dimension fits one chunk of QC features per k-slot (D <= 16): no chunk loop, so the pass is ONE
// basic block and the scheduler can start a tile's operand loads under the previous tile's MFMAs.
template <int QC, int KIND, int NTL, bool ONECH>
__device__ __forceinline__ void fused_pass(const double* __restrict__ Ws, const double* __restrict__ gl,
                                           const double* __restrict__ Xh, const double* __restrict__ s1h,
                                           const double* __restrict__ pc /* this lane's candidate row + g Q */, double s2v,
                                           double amp2, int N, int Dp, int nchunks, int g, int li, double& ss_out,
                                           double& bg_out)
{
#pragma clang fp contract(off)
    const int Q = Dp >> 2;
    double bf[QC];
    if (ONECH || nchunks == 1) {
#pragma unroll
        for (int q = 0; q < QC; ++q) bf[q] = pc[q];
    }
    d4 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < NTL; ++kt) {
        const int j0 = 16 * kt;
        // ---- Gram tile (k_cov's order: chunks of QC features per k-slot, q ascending) ----
        d4 gr = (d4){0.0, 0.0, 0.0, 0.0};
        if (ONECH) {
            double af[QC];
            const double* pa = Xh + (size_t)(j0 + li) * Dp + g * Q;
#pragma unroll
            for (int q = 0; q < QC; ++q) af[q] = pa[q];

#pragma unroll
            for (int q = 0; q < QC; ++q) gr = MFMA_F64(af[q], bf[q], gr);
        } else {
            for (int ch = 0; ch < nchunks; ++ch) {
                double af[QC];
                const double* pa = Xh + (size_t)(j0 + li) * Dp + g * Q + ch * QC;
#pragma unroll
                for (int q = 0; q < QC; ++q) af[q] = pa[q];
                if (nchunks > 1) {
#pragma unroll
                    for (int q = 0; q < QC; ++q) bf[q] = pc[ch * QC + q];
                }
#pragma unroll
                for (int q = 0; q < QC; ++q) gr = MFMA_F64(af[q], bf[q], gr);
            }
        }
        // ---- correlation function on the accumulator registers: rows j0 + g + 4 r, the four of them in lock step ----
        double t4[4], c4[4], kv[4];      // kv[r] = K*[j0 + g + 4 r][cb + li]
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] = (gr[r] - s1h[j0 + g + 4 * r]) - s2v;
        corr_of_kind_t<KIND, 4>(t4, c4);
#pragma unroll
        for (int r = 0; r < 4; ++r) kv[r] = ((j0 + g + 4 * r < N) ? amp2 : 0.0) * c4[r];
        // ---- the tile is the B operand of steps k0 = j0 + 4 r: beta += W[:, k0 .. k0 + 3] K*[k0 .. k0 + 3, :] ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double* wr = Ws + (j0 + 4 * r + g) * FU_LDW + li;
#pragma unroll
            for (int t = kt; t < NTL; ++t) acc[t] = MFMA_F64(wr[16 * t], kv[r], acc[t]);
        }
    }
    // ---- column sums in k_predict_gemm_tri's order: wave row wm = row tiles of parity wm, fma over (mt, r), + xor 16,
    //      + xor 32, then wm 0 + wm 1 (tiles beyond NTL hold exact zeros there: fma(0, 0, s) = s, skipped here) ----
    double ssw[2], bgw[2];
#pragma unroll
    for (int wm = 0; wm < 2; ++wm) {
        double ss = 0.0, bg = 0.0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            if (2 * mt + wm < NTL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = acc[2 * mt + wm][r];
                    ss = fma(v, v, ss);
                    bg = fma(v, gl[16 * (2 * mt + wm) + g + 4 * r], bg);
                }
            }
        ss += __shfl_xor(ss, 16);
        bg += __shfl_xor(bg, 16);
        ss += __shfl_xor(ss, 32);
        bg += __shfl_xor(bg, 32);
        ssw[wm] = ss;
        bgw[wm] = bg;
    }
    ss_out = 0.0;                                  // (k_ei_finalize: 0.0 + the one row block's partial sums)
    bg_out = 0.0;
    ss_out += ssw[0] + ssw[1];
    bg_out += bgw[0] + bgw[1];
}

template <int QC, int KIND, bool ONECH>
__global__ __launch_bounds__(FU_THREADS, 1) void k_ei_fused128(
    const double* __restrict__ WT /*[nh][128][128]*/, const double* __restrict__ gamma /*[nh][128]*/,
    const double* __restrict__ Xs /*[nh][128][Dp]*/, const double* __restrict__ s1 /*[nh][128]*/,
    const double* __restrict__ Cs /*[nh][Mc][Dp]*/, const double* __restrict__ s2 /*[nh][Mc]*/,
    const double* __restrict__ htab, const double* __restrict__ time_m /*[nh][Mc] or null*/, double best,
    double* __restrict__ ei_draw /*[H][Mp]*/, double* __restrict__ mom_m, double* __restrict__ mom_v,
    int N, int Mc, int Dp, int nchunks, int wgs_per_draw, int64_t c0g, int64_t M, int64_t Mp)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ws = smem;                          // [16 ntl][FU_LDW]
    double* gl = smem + FU_NP * FU_LDW;         // [128] gamma of the draw
    double* s1l = gl + FU_NP;                   // [128] row norms of the draw's scaled observations
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int ntl = (N + 15) >> 4;              // live 16-row tiles (1..8)
    const int Q = Dp >> 2;

    {   // W^T of the draw: rows k < 16 ntl (the rest multiplies K* rows that are exactly zero)
        const double* Wg = WT + (size_t)h * FU_NP * FU_NP;
        for (int e = tid; e < 16 * ntl * (FU_NP / 2); e += FU_THREADS) {
            const int row = e >> 6, c2 = e & 63;
            *reinterpret_cast<d2*>(Ws + row * FU_LDW + 2 * c2) = *reinterpret_cast<const d2*>(Wg + (size_t)row * FU_NP + 2 * c2);
        }
        if (tid < FU_NP) {
            gl[tid] = gamma[(size_t)h * FU_NP + tid];
            s1l[tid] = s1[(size_t)h * FU_NP + tid];
        }
    }
    __syncthreads();

    const double* Xh = Xs + (size_t)h * FU_NP * Dp;
    const double* Ch = Cs + (size_t)h * Mc * Dp;
    const double* s1h = s1l;
    const double amp2 = htab[h * SPX_HT + 2];

    // This workgroup's share of the draw's 16-candidate tiles: [tile0, tile1); wave w takes tile0 + w, + FU_WAVES, ... --
    // at most FU_PASSES of them (the launcher sees to that).  The sums of pass p stay in registers of the lane group
    // g = p & 3 (slot p >> 2), and EI is finished AFTER the loop, by all 64 lanes at once: with EI's erf / erfc / exp inside
    // the loop the compiler keeps their constants in registers across it and spills the accumulators.
    const int tiles = Mc >> 4;
    const int tile0 = (int)((int64_t)tiles * blockIdx.x / wgs_per_draw);
    const int tile1 = (int)((int64_t)tiles * (blockIdx.x + 1) / wgs_per_draw);
    double kss[2] = {0.0, 0.0}, kbg[2] = {0.0, 0.0};
#pragma unroll 1
    for (int ps = 0; ps < FU_PASSES; ++ps) {
        const int tile = tile0 + wave + FU_WAVES * ps;
        if (tile >= tile1) break;               // (wave-uniform; no barrier inside the loop)
        const int cb = 16 * tile;
        const double s2v = s2[(size_t)h * Mc + cb + li];
        const double* pc = Ch + (size_t)(cb + li) * Dp + g * Q;
        double ss, bg;
        switch (ntl) {
#define SPX_FU_CASE(NTL_) case NTL_: fused_pass<QC, KIND, NTL_, ONECH>(Ws, gl, Xh, s1h, pc, s2v, amp2, N, Dp, nchunks, g, li, ss, bg); break;
            SPX_FU_CASE(1) SPX_FU_CASE(2) SPX_FU_CASE(3) SPX_FU_CASE(4) SPX_FU_CASE(5) SPX_FU_CASE(6) SPX_FU_CASE(7)
            default: fused_pass<QC, KIND, 8, ONECH>(Ws, gl, Xh, s1h, pc, s2v, amp2, N, Dp, nchunks, g, li, ss, bg); break;
#undef SPX_FU_CASE
        }
        const bool mine = (ps & 3) == g;
        if (ps < 4) { kss[0] = mine ? ss : kss[0]; kbg[0] = mine ? bg : kbg[0]; }
        else        { kss[1] = mine ? ss : kss[1]; kbg[1] = mine ? bg : kbg[1]; }
    }
    const double mean = htab[h * SPX_HT + 0];
    const double prior_v = htab[h * SPX_HT + 3];
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int tile = tile0 + wave + FU_WAVES * (4 * slot + g);
        const int c = 16 * tile + li;
        if (tile < tile1 && c0g + c < M) {
            const double func_m = kbg[slot] + mean;
            const double func_v = prior_v - kss[slot];
            double ei = ei_dev(func_m, func_v, best);
            if (time_m) ei = ei / time_m[(size_t)h * Mc + c];
            const size_t o = (size_t)h * Mp + c0g + c;
            ei_draw[o] = ei;
            if (mom_m) {
                mom_m[o] = func_m;
                mom_v[o] = func_v;
            }
        }
    }
}

template <int KIND>
static void launch_fused_kind(hipStream_t s, const double* WT, const double* gamma, const double* Xs, const double* s1,
                              const double* Cs, const double* s2, const double* htab, const double* time_m, double best,
                              double* ei_draw, double* mom_m, double* mom_v, int N, int Mc, int Dp, int nh, int64_t c0,
                              int64_t M, int64_t Mp, int n_cu)
{
    const int Q = Dp / 4;
    const size_t lds = (size_t)(FU_NP * FU_LDW + 2 * FU_NP) * sizeof(double);   // 146 KB: one workgroup per CU
    const int tiles = Mc / 16;
    // Workgroups per draw.  Every workgroup loads W once (a few us) and its waves take up to FU_PASSES tiles each: the
    // count that minimises (rounds of the chip) x (tiles per wave + the W load, 0.4 of a tile's time); ties: fewer.
    const int gmin = (tiles + FU_WAVES * FU_PASSES - 1) / (FU_WAVES * FU_PASSES);
    int per = gmin;
    double best_cost = 1e300;
    for (int gq = gmin; gq <= gmin + 2 * n_cu && gq <= tiles; ++gq) {
        const int64_t rounds = ((int64_t)gq * nh + n_cu - 1) / n_cu;
        const int per_wave = ((tiles + gq - 1) / gq + FU_WAVES - 1) / FU_WAVES;
        const double cost = (double)rounds * (per_wave + 0.4);
        if (cost < best_cost) { best_cost = cost; per = gq; }
    }
    if (per < 1) per = 1;
    dim3 grid(per, nh);
#define SPX_FU_LAUNCH(QC_, ONE_)                                                                                          \
    do {                                                                                                                  \
        SPX_LDS_ATTR((k_ei_fused128<QC_, KIND, ONE_>), lds);                                  \
        hipLaunchKernelGGL((k_ei_fused128<QC_, KIND, ONE_>), grid, dim3(FU_THREADS), lds, s, WT, gamma, Xs, s1, Cs, s2, htab,   \
                           time_m, best, ei_draw, mom_m, mom_v, N, Mc, Dp, Q / QC_, per, c0, M, Mp);                      \
    } while (0)
    if (Q == 1) SPX_FU_LAUNCH(1, true);
    else if (Q == 2) SPX_FU_LAUNCH(2, true);
    else if (Q == 4) SPX_FU_LAUNCH(4, true);
    // (Q = 8 as one chunk of 8 was tried: -8 % at D = 32, but two of its three instantiations spill 1-3 registers)
    else SPX_FU_LAUNCH(4, false);
#undef SPX_FU_LAUNCH
}

// EI of every (candidate of the chunk, draw): N <= 128 (Np = 128), no fantasies.  ei_draw[h][c0 + c].
void launch_ei_fused128(hipStream_t s, int kind, const double* WT, const double* gamma, const double* Xs, const double* s1,
                        const double* Cs, const double* s2, const double* htab, const double* time_m, double best,
                        double* ei_draw, double* mom_m, double* mom_v, int N, int Mc, int Dp, int nh, int64_t c0,
                        int64_t M, int64_t Mp, int n_cu)
{
    if (kind == SPX_COV_MATERN32)
        launch_fused_kind<SPX_COV_MATERN32>(s, WT, gamma, Xs, s1, Cs, s2, htab, time_m, best, ei_draw, mom_m, mom_v, N, Mc, Dp, nh, c0, M, Mp, n_cu);
    else if (kind == SPX_COV_ARDSE)
        launch_fused_kind<SPX_COV_ARDSE>(s, WT, gamma, Xs, s1, Cs, s2, htab, time_m, best, ei_draw, mom_m, mom_v, N, Mc, Dp, nh, c0, M, Mp, n_cu);
    else
        launch_fused_kind<SPX_COV_MATERN52>(s, WT, gamma, Xs, s1, Cs, s2, htab, time_m, best, ei_draw, mom_m, mom_v, N, Mc, Dp, nh, c0, M, Mp, n_cu);
}
