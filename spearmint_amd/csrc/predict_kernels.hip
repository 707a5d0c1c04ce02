// Predictive variance / mean / EI kernels for gfx950.
//
// Reference (spearmint/spearmint/chooser/GPEIChooser.py):
//     beta   = spla.solve_triangular(obsv_chol, cand_cross, lower=True)   :195
//     func_m = np.dot(cand_cross.T, alpha) + self.mean                    :198
//     func_v = self.amp2*(1+1e-6) - np.sum(beta**2, axis=0)               :199
//     EI     = func_s*(u*ncdf + npdf)                                     :202-206
//     best_cand = np.argmax(np.mean(overall_ei, axis=1))                  :153
//
// k_predict_gemm is the dominant kernel of the whole path: beta = W K* as an
// fp64 MFMA GEMM (W = L^-1 from chol_kernels.hip, both operands K-major in
// memory) whose epilogue reduces the accumulator tile to the two numbers per
// candidate the posterior needs,
//     sum_i beta[i][c]^2            and        sum_i beta[i][c] gamma[i]
// (gamma = W (vals - mean), so the second one equals cand_cross^T alpha).
// beta itself is never written to memory.
#include "common.h"
#include "np_sum.h"
#include "ei_device.h"

#define BM SPX_BM
#define BN SPX_BN
#define BK SPX_BK
#define LDT 144  // LDS row stride (doubles): 128 + 16, so rows k and k+1 of a fragment land on disjoint bank halves

// ---------------------------------------------------------------------------
// C[i][c] = sum_{j <= i-block end} WT[j][i] * Kst[j][c]
//   tile 128(i) x 128(c), 4 waves as 2x2, each wave 64x64 = 4x4 MFMA tiles,
//   K loop over j in steps of 16, register-staged double-buffered LDS.
// Grid: 1-D, XCD-aware (see the block -> tile map below), heaviest row blocks first.
// part_ss / part_bg: [nrb][part_nh][Mc] (all draws of a candidate chunk; this launch fills draws part_h0 ...)
// ---------------------------------------------------------------------------
// NW = waves per workgroup: 4 (wave tile 64x64) or 8 (wave tile 32x64).
// STG = how the operand tiles reach LDS: 0 registers + ds_write (prefetch distance one tile),
//       1 LDS-DMA (global_load_lds, no staging registers, no ds_write).
// ABL > 0: timing-only ablations for performance analysis (wrong results): 1 = no in-loop
// global loads / LDS stores, 2 = also no barriers, 3 = also operand fragments read once,
// 4 (with STG 1) = every DMA fetches the first tile again (cache-resident loads).
template <int NW, int STG, int ABL>
__global__ __launch_bounds__(64 * NW, NW / 2) void k_predict_gemm(
    const double* __restrict__ WT, const double* __restrict__ Kst,
    const double* __restrict__ gamma, double* __restrict__ part_ss,
    double* __restrict__ part_bg, int Np, int Mc, int nh, int ncb, int nrb,
    int part_nh /*draws per chunk in part_ss / part_bg*/, int part_h0 /*first draw of this launch there*/,
    const double* __restrict__ gammaS /*[nh][S][Np] or null*/, int S,
    double* __restrict__ part_bgS /*[nrb][2][nh][S][Mc]*/)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                      // [2][BK][LDT]
    double* Bs = smem + 2 * BK * LDT;       // [2][BK][LDT]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int MT = 16 / NW;         // 16-row MFMA tiles per wave (rows per wave = 16 MT)
    constexpr int NQ = 16 / NW;         // staging: tile rows (16-byte loads) per thread per operand tile
    constexpr int WROWS = 16 * MT;

    // Block -> tile map, XCD-aware.  Workgroup b is dispatched to XCD b % 8 (each XCD
    // has a private 4 MiB L2).  All row blocks ib of one candidate tile (h, cb) read the
    // same K* columns, so they are given to the same XCD (cb % 8 == xcd), and row blocks
    // are issued strictly longest-K-loop-first inside every XCD: that order is what
    // bounds the tail of the launch.  (Measured alternatives that co-schedule the row
    // blocks of a tile to share K* in L2 cut FETCH_SIZE by 9-37 % but lengthen the
    // launch by 1.5-8 %: the kernel is MFMA-bound, not fabric-bound -- DESIGN.md.)
    const int ncbx = (ncb + 7) >> 3;               // candidate tiles per XCD
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int per_rb = ncbx * nh;
    const int ib = nrb - 1 - slot / per_rb;
    const int rem = slot % per_rb;
    const int h = rem / ncbx;
    const int cb = (rem % ncbx) * 8 + xcd;
    if (cb >= ncb) return;                          // uniform per workgroup

    const double* Ag = WT + (size_t)h * Np * Np + (size_t)ib * BM;
    const double* Bg = Kst + (size_t)h * Np * Mc + (size_t)cb * BN;
    const int nk = (ib + 1) * (BM / BK);

    // global->LDS staging map: 4 x 16 B per thread per operand tile (16 rows x 128 doubles)
    const int lrow = tid >> 6;        // + NW q
    const int lcol = (tid & 63) * 2;  // doubles
    typedef __attribute__((address_space(3))) void lds_void_t;
    d4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};

  if constexpr (STG == 2) {
    // ---- LDS-DMA staging, three LDS buffers of 8 rows, two tiles in flight --------------------
    // Step kt issues the DMA of tile kt+2, computes tile kt, then waits only until tile kt+1 has
    // landed (counted vmcnt) and passes a raw s_barrier: a __syncthreads() would drain the
    // in-flight DMA with vmcnt(0).  Buffer (kt+2)%3 was last read in step kt-1, which every
    // wave finished before the barrier that ended it.
    constexpr int BD = 8, RQ = BD / NW;
    static_assert(NW == 4, "DMA pipeline is written for 4 waves");
    double* A3 = smem;                    // [3][BD][LDT]
    double* B3 = smem + 3 * BD * LDT;     // [3][BD][LDT]
    const int nk2 = (ib + 1) * (BM / BD);
#define SPX_DMA3(KT_, BUF_)                                                                           \
    {                                                                                                 \
        const size_t j0_ = (size_t)(KT_) * BD;                                                        \
        _Pragma("unroll") for (int q = 0; q < RQ; ++q) {                                              \
            const int row = wave + NW * q;                                                            \
            __builtin_amdgcn_global_load_lds(Ag + (j0_ + row) * Np + 2 * lane,                        \
                                             (lds_void_t*)(A3 + (BUF_) * BD * LDT + row * LDT), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds(Bg + (j0_ + row) * Mc + 2 * lane,                        \
                                             (lds_void_t*)(B3 + (BUF_) * BD * LDT + row * LDT), 16, 0, 0); \
        }                                                                                             \
    }
    SPX_DMA3(0, 0)
    SPX_DMA3(1, 1)                        // nk2 >= 16
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");     // tile 0 landed (2 RQ = 4 younger DMAs may fly)
    int b0 = 0, b1 = 1, b2 = 2;           // buffers of tiles kt, kt+1, kt+2
    for (int kt = 0; kt < nk2; ++kt) {
        const bool more = (kt + 2 < nk2);
        if (more) SPX_DMA3(kt + 2, b2)
        const double* Ac = A3 + b0 * BD * LDT + WROWS * wm + li;
        const double* Bc = B3 + b0 * BD * LDT + 64 * wn + li;
#pragma unroll
        for (int k0 = 0; k0 < BD; k0 += 4) {
            double a[MT], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = Bc[(k0 + g) * LDT + 16 * t];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = Ac[(k0 + g) * LDT + 16 * t];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA_F64(a[mt], b[nt], acc[mt][nt]);
        }
        if (more) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const int t = b0; b0 = b1; b1 = b2; b2 = t;
    }
#undef SPX_DMA3
  } else {
    d2 ra[NQ], rb[NQ];
    // LDS-DMA: one instruction moves one 1 KiB tile row (64 lanes x 16 B) straight into LDS at a
    // wave-uniform base (+ lane * 16 B); rows stay padded to LDT because every row is its own
    // instruction.  Wave w moves rows w, w + NW, ... of both operand tiles.
#define SPX_DMA_TILE(KT_, BUF_)                                                                            \
    {                                                                                                      \
        const size_t j0_ = (ABL == 4) ? 0 : (size_t)(KT_) * BK;   /* ABL 4: always the first tile */       \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                   \
            const int row = wave + NW * q;                                                                 \
            __builtin_amdgcn_global_load_lds(Ag + (j0_ + row) * Np + 2 * lane,                             \
                                             (lds_void_t*)(As + (BUF_) * BK * LDT + row * LDT), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds(Bg + (j0_ + row) * Mc + 2 * lane,                             \
                                             (lds_void_t*)(Bs + (BUF_) * BK * LDT + row * LDT), 16, 0, 0); \
        }                                                                                                  \
    }
    if (STG == 1) {
        SPX_DMA_TILE(0, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = lrow + NW * q;
            ra[q] = *reinterpret_cast<const d2*>(Ag + (size_t)row * Np + lcol);
            rb[q] = *reinterpret_cast<const d2*>(Bg + (size_t)row * Mc + lcol);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = lrow + NW * q;
            *reinterpret_cast<d2*>(As + row * LDT + lcol) = ra[q];
            *reinterpret_cast<d2*>(Bs + row * LDT + lcol) = rb[q];
        }
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (STG == 1 && (ABL == 0 || ABL == 4) && kt + 1 < nk) SPX_DMA_TILE(kt + 1, cur ^ 1)
        const bool more = (STG == 0) && (ABL == 0) && (kt + 1 < nk);
        if (more) {
            const size_t j0 = (size_t)(kt + 1) * BK;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const size_t row = j0 + lrow + NW * q;
                ra[q] = *reinterpret_cast<const d2*>(Ag + row * Np + lcol);
                rb[q] = *reinterpret_cast<const d2*>(Bg + row * Mc + lcol);
            }
        }
        const double* Ac = As + cur * BK * LDT + WROWS * wm + li;
        const double* Bc = Bs + cur * BK * LDT + 64 * wn + li;
#pragma unroll
        for (int k0 = 0; k0 < BK; k0 += 4) {
            double a[MT], b[4];
            const int kk = (ABL == 3) ? 0 : k0;
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = Bc[(kk + g) * LDT + 16 * t];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = Ac[(kk + g) * LDT + 16 * t];
            if (ABL == 3) {  // keep the values opaque so the loads are hoisted but the MFMAs stay
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(b[t]));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA_F64(a[mt], b[nt], acc[mt][nt]);
        }
        if (more) {
            double* An = As + (cur ^ 1) * BK * LDT;
            double* Bn = Bs + (cur ^ 1) * BK * LDT;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int row = lrow + NW * q;
                *reinterpret_cast<d2*>(An + row * LDT + lcol) = ra[q];
                *reinterpret_cast<d2*>(Bn + row * LDT + lcol) = rb[q];
            }
        }
        if (STG == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA rows have landed
        if (ABL < 2 || ABL == 4) __syncthreads();
        if (ABL == 0 || ABL == 4) cur ^= 1;
    }
#undef SPX_DMA_TILE

  }
    // ---- epilogue: column sums of C^2 and C*gamma over this block's 128 rows ----
    // accumulator layout: acc[mt][nt][r] = C[WROWS wm + 16 mt + g + 4 r][64 wn + 16 nt + li]
    const double* gh = gamma + (size_t)h * Np + (size_t)ib * BM + WROWS * wm;
    double gam[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) gam[mt][r] = gh[16 * mt + g + 4 * r];

    double* red = smem;  // [NW/2 (wm)][128][2]; safe: every wave passed the last barrier of the K loop
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        double ss = 0.0, bg = 0.0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = acc[mt][nt][r];
                ss = fma(v, v, ss);
                bg = fma(v, gam[mt][r], bg);
            }
        ss += __shfl_xor(ss, 16);
        bg += __shfl_xor(bg, 16);
        ss += __shfl_xor(ss, 32);
        bg += __shfl_xor(bg, 32);
        if (g == 0) {
            const int c = 64 * wn + 16 * nt + li;
            red[(wm * BN + c) * 2 + 0] = ss;
            red[(wm * BN + c) * 2 + 1] = bg;
        }
    }
    __syncthreads();
    if (tid < BN) {
        double ss = red[tid * 2 + 0], bg = red[tid * 2 + 1];
#pragma unroll
        for (int w = 1; w < NW / 2; ++w) {          // row groups in order
            ss += red[(w * BN + tid) * 2 + 0];
            bg += red[(w * BN + tid) * 2 + 1];
        }
        const size_t o = ((size_t)ib * part_nh + part_h0 + h) * Mc + (size_t)cb * BN + tid;
        part_ss[o] = ss;
        part_bg[o] = bg;
    }

    // Pending-experiment fantasies (GPEIChooser.py:253-258): the predictive mean is needed
    // against S right-hand sides, func_m[c][s] = sum_i beta[i][c] Gamma_s[i] + mean.  Each wave
    // reduces its 64 rows; the two row halves (wm) are written separately and summed, in
    // fixed order, by k_ei_finalize_fant.
    if (NW == 4 && S > 0) {
        const double* gS = gammaS + (size_t)h * S * Np + (size_t)ib * BM + 64 * wm;
        double* outS = part_bgS + ((((size_t)ib * 2 + wm) * nh + h) * S) * Mc + (size_t)cb * BN + 64 * wn;
        for (int sidx = 0; sidx < S; ++sidx) {
            const double* gs = gS + (size_t)sidx * Np;
            double gv[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[mt][r] = gs[16 * mt + g + 4 * r];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                double bg = 0.0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bg = fma(acc[mt][nt][r], gv[mt][r], bg);
                bg += __shfl_xor(bg, 16);
                bg += __shfl_xor(bg, 32);
                if (g == 0) outS[(size_t)sidx * Mc + 16 * nt + li] = bg;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_predict_gemm_tri: the production tiling (128 x 128, 2 x 2 waves, LDS-DMA staging) that also skips the
// structurally zero part of W's diagonal block.  In the K step that covers columns 16 d .. 16 d + 15 of the
// diagonal block of row block ib, the 16-row tiles t < d of W are zero.  Two things make skipping them pay:
//   * the two wave rows own the 16-row tiles ALTERNATELY (wave row wm: tiles wm, wm + 2, wm + 4, wm + 6), so
//     both lose tiles at the same pace (live tiles per wave 4,4 / 3,4 / 3,3 / 2,3 / 2,2 / 1,2 / 1,1 / 0,1 for
//     d = 0..7) -- with contiguous halves the lower wave row keeps all its tiles until d = 4 and the barrier
//     makes the idle upper one wait for it;
//     (that, and not the branches, is why skipping never paid with the contiguous layout);
//   * the code of a full step is untouched: a diagonal step takes a second copy of the step body.
// Full steps run the same code as k_predict_gemm; a diagonal step takes a second copy of the step body whose
// MFMA groups are entered at the first live tile (wave-uniform jump).  The K order is unchanged; the row
// ownership changes the order of the epilogue's floating-point sums, nothing else.  What did NOT work on the
// way here (C3, ms per launch, 1.85-1.87 for k_predict_gemm on the same box): the same skip with the four waves
// side by side (128 x 32 each, every wave the same triangular profile) 1.89 even before skipping; one
// straight-line copy of the step per skip count 1.99 (256 VGPRs); the short steps spread between the full ones
// 1.83-1.85; this form 1.80.
__global__ __launch_bounds__(256, 2) void k_predict_gemm_tri(
    const double* __restrict__ WT, const double* __restrict__ Kst,
    const double* __restrict__ gamma, double* __restrict__ part_ss,
    double* __restrict__ part_bg, int Np, int Mc, int nh, int ncb, int nrb,
    int part_nh, int part_h0, const double* __restrict__ gammaS, int S, double* __restrict__ part_bgS)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                      // [2][BK][LDT]
    double* Bs = smem + 2 * BK * LDT;       // [2][BK][LDT]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the skip below is a scalar jump
    const int g = lane >> 4, li = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int NW = 4, NQ = 4;

    const int ncbx = (ncb + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int per_rb = ncbx * nh;
    const int ib = nrb - 1 - slot / per_rb;
    const int rem = slot % per_rb;
    const int h = rem / ncbx;
    const int cb = (rem % ncbx) * 8 + xcd;
    if (cb >= ncb) return;

    const double* Ag = WT + (size_t)h * Np * Np + (size_t)ib * BM;
    const double* Bg = Kst + (size_t)h * Np * Mc + (size_t)cb * BN;
    const int nk = (ib + 1) * (BM / BK);
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned loff = (unsigned)lane * 16u;   // this lane's 16 bytes of a 1 KiB tile row
    d4 acc[4][4];        // acc[mt][nt]: rows 16 (2 mt + wm) + g + 4 r, columns 64 wn + 16 nt + li
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};

    // LDS-DMA of one 1 KiB tile row in the SGPR-base form (wave-uniform row address + this lane's 32-bit byte
    // offset), written as asm because the builtin always materialises a 64-bit VGPR address: that is one
    // 64-bit VALU add per DMA instruction, and every VALU instruction in this loop costs MFMA issue time.
#define SPX_DMA_ROW(GPTR_, LPTR_)                                                                          \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                          \
                 :: "v"(loff), "s"(GPTR_), "s"((unsigned)(size_t)(lds_void_t*)(LPTR_)) : "memory")   /* m0 is reserved: the compiler never keeps anything in it */
#define SPX_DMA_TILE(KT_, BUF_)                                                                            \
    {                                                                                                      \
        const size_t j0_ = (size_t)(KT_) * BK;                                                             \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                   \
            const int row = wave + NW * q;                                                                 \
            SPX_DMA_ROW(Ag + (j0_ + row) * Np, As + (BUF_) * BK * LDT + row * LDT);                        \
            SPX_DMA_ROW(Bg + (j0_ + row) * Mc, Bs + (BUF_) * BK * LDT + row * LDT);                        \
        }                                                                                                  \
    }
    SPX_DMA_TILE(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nfull = nk - BM / BK;           // the last 8 K steps cover the diagonal block
    int cur = 0;
    // ONE loop over all K steps, on purpose: with the rectangular part peeled into a loop of its own (the
    // exact loop of k_predict_gemm) the compiler's schedule of both loops gets worse (2.03 ms per launch at
    // C3 instead of 1.80).
    for (int j = 0; j < nk; ++j) {
        // diagonal step d = j - nfull covers columns 16 d .. 16 d + 15 of the diagonal block; first live tile
        // of this wave: tiles 2 mt + wm >= d  <=>  mt >= (d - wm + 1) >> 1   (4: none)
        const int m0 = (j >= nfull) ? ((j - nfull - wm + 1) >> 1) : 0;
        if (j + 1 < nk) SPX_DMA_TILE(j + 1, cur ^ 1)
        const double* Ac = As + cur * BK * LDT + 16 * wm + li;
        const double* Bc = Bs + cur * BK * LDT + 64 * wn + li;
        if (m0 == 0) {
#pragma unroll
            for (int k0 = 0; k0 < BK; k0 += 4) {
                double a[4], b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = Bc[(k0 + g) * LDT + 16 * t];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = Ac[(k0 + g) * LDT + 32 * t];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA_F64(a[mt], b[nt], acc[mt][nt]);
            }
            // (the wait + barrier are repeated in both branches so that the compiler can, as in
            // k_predict_gemm, sink the last MFMA group of a full step below them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
#define SPX_TRI_G4(MT_)                                                      \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[MT_][nt] = MFMA_F64(a[MT_], b[nt], acc[MT_][nt]);
#pragma unroll
            for (int k0 = 0; k0 < BK; k0 += 4) {
                double a[4], b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = Bc[(k0 + g) * LDT + 16 * t];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = Ac[(k0 + g) * LDT + 32 * t];
                switch (m0) {            // one copy of the MFMAs, entered at the first live tile
                    case 1: SPX_TRI_G4(1)
                    case 2: SPX_TRI_G4(2)
                    case 3: SPX_TRI_G4(3)
                    default: break;
                }
            }
#undef SPX_TRI_G4
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        cur ^= 1;
    }
#undef SPX_DMA_TILE
#undef SPX_DMA_ROW

    // ---- epilogue: column sums of C^2 and C*gamma over this wave row's 64 rows, then the two wave rows ----
    const double* gh = gamma + (size_t)h * Np + (size_t)ib * BM + 16 * wm;
    double gam[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) gam[mt][r] = gh[32 * mt + g + 4 * r];
    double* red = smem;  // [2 (wm)][128][2]; safe: every wave passed the last barrier of the K loop
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        double ss = 0.0, bg = 0.0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = acc[mt][nt][r];
                ss = fma(v, v, ss);
                bg = fma(v, gam[mt][r], bg);
            }
        ss += __shfl_xor(ss, 16);
        bg += __shfl_xor(bg, 16);
        ss += __shfl_xor(ss, 32);
        bg += __shfl_xor(bg, 32);
        if (g == 0) {
            const int c = 64 * wn + 16 * nt + li;
            red[(wm * BN + c) * 2 + 0] = ss;
            red[(wm * BN + c) * 2 + 1] = bg;
        }
    }
    __syncthreads();
    if (tid < BN) {
        const size_t o = ((size_t)ib * part_nh + part_h0 + h) * Mc + (size_t)cb * BN + tid;
        part_ss[o] = red[tid * 2 + 0] + red[(BN + tid) * 2 + 0];
        part_bg[o] = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
    }
    if (S > 0) {   // pending-experiment fantasies: one partial per wave row, summed by k_ei_finalize_fant
        const double* gS = gammaS + (size_t)h * S * Np + (size_t)ib * BM + 16 * wm;
        double* outS = part_bgS + ((((size_t)ib * 2 + wm) * nh + h) * S) * Mc + (size_t)cb * BN + 64 * wn;
        for (int sidx = 0; sidx < S; ++sidx) {
            const double* gs = gS + (size_t)sidx * Np;
            double gv[4][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[mt][r] = gs[32 * mt + g + 4 * r];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                double bg = 0.0;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bg = fma(acc[mt][nt][r], gv[mt][r], bg);
                bg += __shfl_xor(bg, 16);
                bg += __shfl_xor(bg, 32);
                if (g == 0) outS[(size_t)sidx * Mc + 16 * nt + li] = bg;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_predict_gemm_tail (round 5): the LAST row block of a problem whose observation count is not a multiple of 128.
// N is padded to the GEMM's 128-row tiles with an identity: pad rows of K* are zero, pad rows of W are unit vectors.  Of the
// last row block only the first lt = ceil(N / 16) - 8 ib 16-row tiles hold observations, and of its K steps only the first
// nlive = ceil(N / 16) multiply anything but zeros.  k_predict_gemm_tri computes all of it (N = 129: 16 K steps x 8 row tiles
// for ONE live row; averaged over N, a third of a mid-sized problem's GEMM is padding).  This kernel is k_predict_gemm_tri for
// that block with both bounds applied -- same tile, same wave rows owning the 16-row tiles alternately, same LDS-DMA pipeline,
// the K steps in the same order, the zero tiles of the diagonal steps skipped the same way -- so every live accumulator holds
// the bits of the padded computation, the dead ones are the exact zeros they would have been, and the epilogue's sums (dead
// tiles left out of the fma chains: fma(0, g, s) = s) are the same numbers.  The launcher gives row blocks 0 .. nrb - 2 to
// k_predict_gemm_tri as before; K(X*,X) leaves the rows from 16 nlive on unwritten.  ML = ceil(lt / 2) <= 3: the
// accumulator tiles per wave (lt = 7 is left to the padded path: one tile of sixteen is not worth a kernel).
template <int ML>
__global__ __launch_bounds__(256, 2) void k_predict_gemm_tail(
    const double* __restrict__ WT, const double* __restrict__ Kst,
    const double* __restrict__ gamma, double* __restrict__ part_ss,
    double* __restrict__ part_bg, int Np, int Mc, int nh, int ncb, int ib, int nlive,
    int part_nh, int part_h0, const double* __restrict__ gammaS, int S, double* __restrict__ part_bgS)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                      // [2][BK][LDT]
    double* Bs = smem + 2 * BK * LDT;       // [2][BK][LDT]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int NW = 4, NQ = 4;

    const int ncbx = (ncb + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int h = slot / ncbx;
    const int cb = (slot % ncbx) * 8 + xcd;
    if (cb >= ncb) return;

    const int lt = nlive - 8 * ib;                 // live 16-row tiles of this block: 1 .. 2 ML
    const int mlive = (lt - wm + 1) >> 1;          // ... of this wave row (tiles 2 mt + wm < lt); wave row 1 may have none
    const double* Ag = WT + (size_t)h * Np * Np + (size_t)ib * BM;
    const double* Bg = Kst + (size_t)h * Np * Mc + (size_t)cb * BN;
    const int nk = nlive;                          // (<= 8 (ib + 1): the K steps of live rows of K*)
    const int nfull = 8 * ib;                      // from here on: the diagonal block
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned loff = (unsigned)lane * 16u;
    d4 acc[ML][4];
#pragma unroll
    for (int mt = 0; mt < ML; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};

#define SPX_DMA_ROW(GPTR_, LPTR_)                                                                          \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                          \
                 :: "v"(loff), "s"(GPTR_), "s"((unsigned)(size_t)(lds_void_t*)(LPTR_)) : "memory")
#define SPX_DMA_TILE(KT_, BUF_)                                                                            \
    {                                                                                                      \
        const size_t j0_ = (size_t)(KT_) * BK;                                                             \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                   \
            const int row = wave + NW * q;                                                                 \
            SPX_DMA_ROW(Ag + (j0_ + row) * Np, As + (BUF_) * BK * LDT + row * LDT);                        \
            SPX_DMA_ROW(Bg + (j0_ + row) * Mc, Bs + (BUF_) * BK * LDT + row * LDT);                        \
        }                                                                                                  \
    }
    SPX_DMA_TILE(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int j = 0; j < nk; ++j) {
        // diagonal step d = j - nfull: the tiles 2 mt + wm < d of W are zero (as in k_predict_gemm_tri)
        const int m0 = (j >= nfull) ? ((j - nfull - wm + 1) >> 1) : 0;
        if (j + 1 < nk) SPX_DMA_TILE(j + 1, cur ^ 1)
        const double* Ac = As + cur * BK * LDT + 16 * wm + li;
        const double* Bc = Bs + cur * BK * LDT + 64 * wn + li;
        if (m0 < mlive) {
#pragma unroll
            for (int k0 = 0; k0 < BK; k0 += 4) {
                double b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = Bc[(k0 + g) * LDT + 16 * t];
#pragma unroll
                for (int mt = 0; mt < ML; ++mt)
                    if (mt >= m0 && mt < mlive) {
                        const double a = Ac[(k0 + g) * LDT + 32 * mt];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA_F64(a, b[nt], acc[mt][nt]);
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#undef SPX_DMA_TILE
#undef SPX_DMA_ROW

    // ---- epilogue: k_predict_gemm_tri's sums with the dead tiles (exact zeros there) left out of the chains ----
    const double* gh = gamma + (size_t)h * Np + (size_t)ib * BM + 16 * wm;
    double gam[ML][4];
#pragma unroll
    for (int mt = 0; mt < ML; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) gam[mt][r] = (mt < mlive) ? gh[32 * mt + g + 4 * r] : 0.0;
    double* red = smem;  // [2 (wm)][128][2]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        double ss = 0.0, bg = 0.0;
#pragma unroll
        for (int mt = 0; mt < ML; ++mt)
            if (mt < mlive) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = acc[mt][nt][r];
                    ss = fma(v, v, ss);
                    bg = fma(v, gam[mt][r], bg);
                }
            }
        ss += __shfl_xor(ss, 16);
        bg += __shfl_xor(bg, 16);
        ss += __shfl_xor(ss, 32);
        bg += __shfl_xor(bg, 32);
        if (g == 0) {
            const int c = 64 * wn + 16 * nt + li;
            red[(wm * BN + c) * 2 + 0] = ss;
            red[(wm * BN + c) * 2 + 1] = bg;
        }
    }
    __syncthreads();
    if (tid < BN) {
        const size_t o = ((size_t)ib * part_nh + part_h0 + h) * Mc + (size_t)cb * BN + tid;
        part_ss[o] = red[tid * 2 + 0] + red[(BN + tid) * 2 + 0];
        part_bg[o] = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
    }
    if (S > 0) {   // pending-experiment fantasies: one partial per wave row, summed by k_ei_finalize_fant
        const double* gS = gammaS + (size_t)h * S * Np + (size_t)ib * BM + 16 * wm;
        double* outS = part_bgS + ((((size_t)ib * 2 + wm) * nh + h) * S) * Mc + (size_t)cb * BN + 64 * wn;
        for (int sidx = 0; sidx < S; ++sidx) {
            const double* gs = gS + (size_t)sidx * Np;
            double gv[ML][4];
#pragma unroll
            for (int mt = 0; mt < ML; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[mt][r] = (mt < mlive) ? gs[32 * mt + g + 4 * r] : 0.0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                double bg = 0.0;
#pragma unroll
                for (int mt = 0; mt < ML; ++mt)
                    if (mt < mlive) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) bg = fma(acc[mt][nt][r], gv[mt][r], bg);
                    }
                bg += __shfl_xor(bg, 16);
                bg += __shfl_xor(bg, 32);
                if (g == 0) outS[(size_t)sidx * Mc + 16 * nt + li] = bg;
            }
        }
    }
}

// what the production GEMM does with a padded observation count: 0 = nothing to skip (or not this variant), else
// nlive = ceil(N / 16): the last row block goes to k_predict_gemm_tail, K(X*,X) stops at row 16 nlive
int predict_gemm_padding_plan(int variant, int N, int Np)
{
    if (!(variant == 0 || variant == 32)) return 0;
    const int nlive = (N + 15) / 16;
    const int nrb = Np / BM;
    const int lt = nlive - 8 * (nrb - 1);          // live 16-row tiles of the last row block
    if (lt < 1 || lt > 6) return 0;
    // The short block is a launch of its own in front of the others (it cannot share k_predict_gemm_tri's registers), which
    // costs a launch boundary: about a tenth of the pass.  What it saves is (8 - lt) / 8 of the last block's (8 nrb)-step K
    // loop out of 4 nrb (nrb + 1) steps in all = (8 - lt) / (4 (nrb + 1)).  Measured (profiles/r05_padding_skip.log: EI step,
    // 20 000 x 10 / 100 000 x 10): N = 129 ... 160 -31 %, 257 -21 %, 300 -14 %, 400 -16 %, 900 -7 %, 1300 -5 %; but N = 600
    // +-0, 1500 +6 %, 2000 +3 % where the saving is under a tenth.  Taken from 0.12 up.
    return (25 * (8 - lt) >= 12 * (nrb + 1)) ? nlive : 0;
}

// Variants (spx_set_option "gemm_waves", per handle): 0 / 32 = production (k_predict_gemm_tri: 4 waves, LDS-DMA
// staging, zero tiles of the diagonal block skipped -- measured fastest; round 6 measured a form whose workgroups walk over
// several tiles, prefetching the next tile's first K slab during the epilogue: same bits, 6-10 % slower at every size,
// scripts/dev/attic/k_predict_gemm_walk.hip.txt, profiles/r06_time_walk.log), 14 = the same without the skipping
// (k_predict_gemm; production until round 2), 4 / 8 = 4 / 8 waves with register staging, 18 = 8 waves with LDS-DMA, 24 = LDS-DMA
// with three 8-row buffers and two tiles in flight.  They agree to rounding (the 8-wave kernels sum the
// row groups of the epilogue in another order).  The
// timing-only ablations 41..44 (WRONG results, for performance analysis) exist only in a library
// built with -DSPX_ABLATIONS (make ABLATIONS=1); the shipped library rejects them.
bool predict_gemm_variant_ok(int v)
{
    switch (v) {
        case 0: case 4: case 8: case 14: case 18: case 24: case 32: return true;
#ifdef SPX_ABLATIONS
        case 41: case 42: case 43: case 44: return true;
#endif
        default: return false;
    }
}

template <int NW, int STG, int ABL>
static void launch_gemm_variant(hipStream_t s, int grid, size_t lds, const double* WT, const double* Kst,
                                const double* gamma, double* part_ss, double* part_bg, int Np, int Mc, int nh,
                                int ncb, int nrb, int part_nh, int part_h0, const double* gammaS, int S,
                                double* part_bgS)
{
    // attribute set per launch (not cached): it is per device, and one process may drive several GPUs
    SPX_LDS_ATTR((k_predict_gemm<NW, STG, ABL>), lds);
    hipLaunchKernelGGL((k_predict_gemm<NW, STG, ABL>), dim3(grid), dim3(64 * NW), lds, s, WT, Kst, gamma, part_ss,
                       part_bg, Np, Mc, nh, ncb, nrb, part_nh, part_h0, gammaS, S, part_bgS);
}

// nlive: predict_gemm_padding_plan()'s figure when the caller wants the padding of N skipped (and has told K(X*,X) so), else 0
void launch_predict_gemm(hipStream_t s, int variant, const double* WT, const double* Kst, const double* gamma,
                         double* part_ss, double* part_bg, int Np, int Mc, int nh, int part_nh, int part_h0,
                         const double* gammaS, int S, double* part_bgS, int nlive)
{
    const int ncb = Mc / BN, nrb = Np / BM;
    const size_t lds = (size_t)(4 * BK * LDT) * sizeof(double);
    const int grid = 8 * ((ncb + 7) / 8) * nrb * nh;
#define SPX_GO(NW_, STG_, ABL_)                                                                               \
    launch_gemm_variant<NW_, STG_, ABL_>(s, grid, lds, WT, Kst, gamma, part_ss, part_bg, Np, Mc, nh, ncb, nrb, \
                                         part_nh, part_h0, gammaS, S, part_bgS)
    // 0 = production: k_predict_gemm_tri.  The rectangular kernels have the fantasy epilogue in their 4-wave
    // form only (14 = the round-1 production kernel: 4 waves, LDS-DMA staging).
    const int v = (S > 0 && variant != 0 && variant != 32) ? 14 : variant;
    if (v == 0 || v == 32) {
        const bool tail = nlive > 0 && predict_gemm_padding_plan(v, 16 * nlive, Np) == nlive;   // (the caller planned with it)
        const int nrb_main = tail ? nrb - 1 : nrb;
        if (tail) {   // the short block first: it is the one that would otherwise start last and run alone
            const int lt = nlive - 8 * (nrb - 1), ml = (lt + 1) / 2;
            const int tgrid = 8 * ((ncb + 7) / 8) * nh;
#define SPX_TAIL(ML_)                                                                                                   \
    do {                                                                                                                \
        SPX_LDS_ATTR(k_predict_gemm_tail<ML_>, lds);                                                                    \
        hipLaunchKernelGGL(k_predict_gemm_tail<ML_>, dim3(tgrid), dim3(256), lds, s, WT, Kst, gamma, part_ss, part_bg, Np, Mc, \
                           nh, ncb, nrb - 1, nlive, part_nh, part_h0, gammaS, S, part_bgS);                              \
    } while (0)
            if (ml == 1) SPX_TAIL(1);
            else if (ml == 2) SPX_TAIL(2);
            else SPX_TAIL(3);
#undef SPX_TAIL
        }
        if (nrb_main > 0) {
            const int mgrid = 8 * ((ncb + 7) / 8) * nrb_main * nh;
            SPX_LDS_ATTR(k_predict_gemm_tri, lds);
            hipLaunchKernelGGL(k_predict_gemm_tri, dim3(mgrid), dim3(256), lds, s, WT, Kst, gamma, part_ss, part_bg, Np, Mc,
                               nh, ncb, nrb_main, part_nh, part_h0, gammaS, S, part_bgS);
        }
        return;
    }
    switch (v) {
        case 8:  SPX_GO(8, 0, 0); break;
        case 18: SPX_GO(8, 1, 0); break;
        case 24: SPX_GO(4, 2, 0); break;
        case 4:  SPX_GO(4, 0, 0); break;
#ifdef SPX_ABLATIONS
        case 41: SPX_GO(4, 0, 1); break;
        case 42: SPX_GO(4, 0, 2); break;
        case 43: SPX_GO(4, 0, 3); break;
        case 44: SPX_GO(4, 1, 4); break;
#endif
        default: SPX_GO(4, 1, 0); break;   // 14: 4 waves, LDS-DMA staging, no skipping (production until round 2)
    }
#undef SPX_GO
}

// one thread per (candidate of the chunk, draw of the group)
__global__ __launch_bounds__(256) void k_ei_finalize(
    const double* __restrict__ part_ss, const double* __restrict__ part_bg,
    const double* __restrict__ htab, const double* __restrict__ time_m /*[nh][Mc] or null*/,
    double best, double* __restrict__ ei_draw /*[H][Mp]*/, double* __restrict__ mom_m,
    double* __restrict__ mom_v, int nrb, int Mc, int nh, int64_t c0, int64_t M, int64_t Mp, int h0)
{
#pragma clang fp contract(off)
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y;
    if (c >= Mc || c0 + c >= M) return;
    double ss = 0.0, bg = 0.0;
    for (int ib = 0; ib < nrb; ++ib) {
        const size_t o = ((size_t)ib * nh + h) * Mc + c;
        ss += part_ss[o];
        bg += part_bg[o];
    }
    const double mean = htab[h * SPX_HT + 0];
    const double prior_v = htab[h * SPX_HT + 3];  // amp2*(1+1e-6)
    const double func_m = bg + mean;
    const double func_v = prior_v - ss;
    double ei = ei_dev(func_m, func_v, best);
    if (time_m) ei = ei / time_m[(size_t)h * Mc + c];
    const size_t o = (size_t)(h0 + h) * Mp + c0 + c;
    ei_draw[o] = ei;
    if (mom_m) {
        mom_m[o] = func_m;
        mom_v[o] = func_v;
    }
}

void launch_ei_finalize(hipStream_t s, const double* part_ss, const double* part_bg,
                        const double* htab, const double* time_m, double best, double* ei_draw,
                        double* mom_m, double* mom_v, int nrb, int Mc, int nh, int64_t c0,
                        int64_t M, int64_t Mp, int h0)
{
    hipLaunchKernelGGL(k_ei_finalize, dim3((Mc + 255) / 256, nh), dim3(256), 0, s, part_ss, part_bg,
                       htab, time_m, best, ei_draw, mom_m, mom_v, nrb, Mc, nh, c0, M, Mp, h0);
}


// EI against S fantasies, averaged over S in numpy's pairwise order (np.mean(ei, axis=1) on the (M, S) array of
// GPEIChooser.py:261-266), in two launches since round 4: every (candidate, draw, fantasy) EI value by a thread of its own
// (k_ei_fant_values; the single-launch form walked the S fantasies of a candidate in ONE thread -- 39 workgroups per launch
// at S = 100, 338 ms of an 807 ms pass at C3 size), then the ordered sum over S per candidate (k_ei_fant_mean: np_sum.h, the
// same additions in the same order as the streaming form it replaces: eight accumulators, the tree, the tail).
__global__ __launch_bounds__(256) void k_ei_fant_values(
    const double* __restrict__ part_ss, const double* __restrict__ part_bgS,
    const double* __restrict__ htab, const double* __restrict__ bests /*[nh][S]*/, int nrb, int Mc, int nh,
    int S, int64_t c0, int64_t M, double* __restrict__ ei_s /*[nh][S][Mc]*/)
{
#pragma clang fp contract(off)
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y / S, sidx = blockIdx.y - h * S;
    if (c >= Mc || c0 + c >= M) return;
    double ss = 0.0;
    for (int ib = 0; ib < nrb; ++ib) ss += part_ss[((size_t)ib * nh + h) * Mc + c];
    const double mean = htab[h * SPX_HT + 0];
    const double func_v = htab[h * SPX_HT + 3] - ss;
    double bg = 0.0;
    for (int ib = 0; ib < nrb; ++ib) {
        const size_t o0 = ((((size_t)ib * 2 + 0) * nh + h) * S + sidx) * Mc + c;
        const size_t o1 = ((((size_t)ib * 2 + 1) * nh + h) * S + sidx) * Mc + c;
        bg += part_bgS[o0] + part_bgS[o1];
    }
    ei_s[((size_t)h * S + sidx) * Mc + c] = ei_dev(bg + mean, func_v, bests[(size_t)h * S + sidx]);
}

__global__ __launch_bounds__(256) void k_ei_fant_mean(const double* __restrict__ ei_s, const double* __restrict__ time_m,
                                                      double* __restrict__ ei_draw, int Mc, int S, int64_t c0, int64_t M,
                                                      int64_t Mp, int h0)
{
#pragma clang fp contract(off)
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y;
    if (c >= Mc || c0 + c >= M) return;
    const double res = np_pairwise(ei_s + (size_t)h * S * Mc + c, Mc, S);
    double out = (0.0 + res) / (double)S;
    if (time_m) out = out / time_m[(size_t)h * Mc + c];
    ei_draw[(size_t)(h0 + h) * Mp + c0 + c] = out;
}

void launch_ei_finalize_fant(hipStream_t s, const double* part_ss, const double* part_bgS,
                             const double* htab, const double* bests, const double* time_m,
                             double* ei_draw, int nrb, int Mc, int nh, int S, int64_t c0, int64_t M,
                             int64_t Mp, int h0, double* ei_s)
{
    hipLaunchKernelGGL(k_ei_fant_values, dim3((Mc + 255) / 256, nh * S), dim3(256), 0, s, part_ss, part_bgS, htab, bests, nrb,
                       Mc, nh, S, c0, M, ei_s);
    hipLaunchKernelGGL(k_ei_fant_mean, dim3((Mc + 255) / 256, nh), dim3(256), 0, s, ei_s, time_m, ei_draw, Mc, S, c0, M, Mp, h0);
}

// ---------------------------------------------------------------------------
// mean over draws in numpy's summation order (np.mean(overall_ei, axis=1) on a
// C-contiguous (M, H) array = 0.0 + pairwise_sum(row) then / H; numpy
// loops_utils.h.src pairwise sum: <8 sequential, <=128 eight accumulators,
// else split in halves rounded to multiples of 8).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mean_over_draws(const double* __restrict__ ei_draw,
                                                         double* __restrict__ ei_mean, int64_t M,
                                                         int64_t Mp, int H)
{
#pragma clang fp contract(off)
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= M) return;
    const double s = 0.0 + np_pairwise(ei_draw + c, Mp, H);
    ei_mean[c] = s / (double)H;
}

void launch_mean_over_draws(hipStream_t s, const double* ei_draw, double* ei_mean, int64_t M,
                            int64_t Mp, int H)
{
    hipLaunchKernelGGL(k_mean_over_draws, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s,
                       ei_draw, ei_mean, M, Mp, H);
}

// 2-D partition (draws x candidates): this rank's share of sum_h EI[c, h] -- the sum over ITS draws in
// numpy's order (np.sum(overall_ei[:, h0:h1], axis=1)) -- written into its segment of the zero-padded M-vector that
// the single all-reduce(SUM) then completes; and the division by the TOTAL number of draws afterwards.
__global__ __launch_bounds__(256) void k_sum_over_draws(const double* __restrict__ ei_draw,
                                                        double* __restrict__ out, int64_t M, int64_t Mp, int H)
{
#pragma clang fp contract(off)
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= M) return;
    out[c] = 0.0 + np_pairwise(ei_draw + c, Mp, H);
}

void launch_sum_over_draws(hipStream_t s, const double* ei_draw, double* out, int64_t M, int64_t Mp, int H)
{
    hipLaunchKernelGGL(k_sum_over_draws, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, ei_draw, out, M, Mp, H);
}

__global__ __launch_bounds__(256) void k_div_scalar(double* __restrict__ v, int64_t n, double denom)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = v[i] / denom;
}

void launch_div_scalar(hipStream_t s, double* v, int64_t n, double denom)
{
    hipLaunchKernelGGL(k_div_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n, denom);
}

// ---------------------------------------------------------------------------
// argmax with numpy semantics: the first NaN wins; otherwise the first maximum.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool better(double av, int64_t ai, double bv, int64_t bi)
{
    if (bi < 0) return ai >= 0;
    if (ai < 0) return false;
    const bool an = (av != av), bn = (bv != bv);
    if (an || bn) {
        if (an && bn) return ai < bi;
        return an;
    }
    if (av > bv) return true;
    if (av < bv) return false;
    return ai < bi;
}

#define ARGMAX_BLOCKS 1024
int argmax_blocks(int64_t M)
{
    int64_t b = (M + 255) / 256;
    return (int)(b < ARGMAX_BLOCKS ? b : ARGMAX_BLOCKS);
}

__device__ __forceinline__ void block_argmax(double& v, int64_t& idx)
{
    __shared__ double sv[256];
    __shared__ int64_t si[256];
    sv[threadIdx.x] = v;
    si[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            if (better(sv[threadIdx.x + s], si[threadIdx.x + s], sv[threadIdx.x], si[threadIdx.x])) {
                sv[threadIdx.x] = sv[threadIdx.x + s];
                si[threadIdx.x] = si[threadIdx.x + s];
            }
        }
        __syncthreads();
    }
    v = sv[0];
    idx = si[0];
}

__global__ __launch_bounds__(256) void k_argmax_stage1(const double* __restrict__ v, int64_t M,
                                                       double* __restrict__ bv, int64_t* __restrict__ bi)
{
    double best = 0.0;
    int64_t besti = -1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const double x = v[i];
        if (better(x, i, best, besti)) { best = x; besti = i; }
    }
    block_argmax(best, besti);
    if (threadIdx.x == 0) { bv[blockIdx.x] = best; bi[blockIdx.x] = besti; }
}

// mean over draws and the first argmax stage in one launch (the EI pass's tail: one launch fewer per step; at small N a
// launch is 5 us of a 140 us step).  ei_mean[c] is written exactly as k_mean_over_draws writes it.
__global__ __launch_bounds__(256) void k_mean_argmax_stage1(const double* __restrict__ ei_draw, double* __restrict__ ei_mean,
                                                            int64_t M, int64_t Mp, int H, double* __restrict__ bv,
                                                            int64_t* __restrict__ bi)
{
#pragma clang fp contract(off)
    double best = 0.0;
    int64_t besti = -1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const double s = 0.0 + np_pairwise(ei_draw + i, Mp, H);
        const double x = s / (double)H;
        ei_mean[i] = x;
        if (better(x, i, best, besti)) { best = x; besti = i; }
    }
    block_argmax(best, besti);
    if (threadIdx.x == 0) { bv[blockIdx.x] = best; bi[blockIdx.x] = besti; }
}

// host_mirror (optional, pinned host memory the device can write): {double value, int64 index, int info[n_info]} -- the
// winner and the factorisation's not-PD flags reach the host with the kernel's own stores, no copy commands behind it
__global__ __launch_bounds__(256) void k_argmax_stage2(const double* __restrict__ bv,
                                                       const int64_t* __restrict__ bi, int nb,
                                                       double* __restrict__ ov, int64_t* __restrict__ oi,
                                                       double* __restrict__ host_mirror, const int* __restrict__ info, int n_info)
{
    double best = 0.0;
    int64_t besti = -1;
    for (int i = threadIdx.x; i < nb; i += 256)
        if (better(bv[i], bi[i], best, besti)) { best = bv[i]; besti = bi[i]; }
    block_argmax(best, besti);
    if (threadIdx.x == 0) { *ov = best; *oi = besti; }
    if (host_mirror) {
        if (threadIdx.x == 0) {
            host_mirror[0] = best;
            reinterpret_cast<int64_t*>(host_mirror)[1] = besti;
        }
        int* io = reinterpret_cast<int*>(host_mirror + 2);
        for (int i = threadIdx.x; i < n_info; i += 256) io[i] = info[i];
    }
}

void launch_argmax(hipStream_t s, const double* v, int64_t M, double* blk_val, int64_t* blk_idx,
                   double* out_val, int64_t* out_idx, double* host_mirror, const int* info, int n_info)
{
    const int nb = argmax_blocks(M);
    hipLaunchKernelGGL(k_argmax_stage1, dim3(nb), dim3(256), 0, s, v, M, blk_val, blk_idx);
    hipLaunchKernelGGL(k_argmax_stage2, dim3(1), dim3(256), 0, s, blk_val, blk_idx, nb, out_val, out_idx, host_mirror, info, n_info);
}

void launch_mean_argmax(hipStream_t s, const double* ei_draw, double* ei_mean, int64_t M, int64_t Mp, int H, double* blk_val,
                        int64_t* blk_idx, double* out_val, int64_t* out_idx, double* host_mirror, const int* info, int n_info)
{
    const int nb = argmax_blocks(M);
    hipLaunchKernelGGL(k_mean_argmax_stage1, dim3(nb), dim3(256), 0, s, ei_draw, ei_mean, M, Mp, H, blk_val, blk_idx);
    hipLaunchKernelGGL(k_argmax_stage2, dim3(1), dim3(256), 0, s, blk_val, blk_idx, nb, out_val, out_idx, host_mirror, info, n_info);
}
