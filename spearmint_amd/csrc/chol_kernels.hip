// Batched blocked left-looking Cholesky + triangular inverse for gfx950.
//
// Replaces, for all hyper-parameter draws at once,
//     obsv_chol = spla.cholesky(obsv_cov, lower=True)        GPEIChooser.py:191
//     alpha     = spla.cho_solve((obsv_chol, True), vals-mean)          :194
// and prepares  W = L^-1  so that  spla.solve_triangular(L, K*) (:195) becomes
// the dense fp64 MFMA GEMM  W K*  of predict_kernels.hip.
//
// Matrices: [nh][Np][Np] row-major, Np a multiple of 128, block size NB = 64.
// Left-looking, one block column k at a time (two launches per k):
//   k_chol_diag   1 workgroup / draw : S = K_kk - L_k,:k L_k,:k^T (MFMA, panels
//                 staged in LDS); the 64x64 block is factored with 16x16
//                 sub-blocks: each diagonal sub-block is factored and inverted by
//                 one wavefront on the matrix pipe (factor16_mfma: a rank-1 MFMA
//                 per pivot), sub-panel and trailing updates by MFMA; the inverse
//                 of the block by block forward substitution (MFMA).
//   k_lean_step   (log-likelihood path, tile-major storage) one right-looking
//                 update step for every trailing tile, the diagonal block factored
//                 by the workgroup that owns it; k_lean_step2: two steps per pass
//                 for batches that outgrow the Infinity Cache; k_lean_trsm: the
//                 panel's triangular solve.
//   k_chol_panel  1 workgroup / (row block > k, draw):
//                 L_rk = (K_rk - L_r,:k L_k,:k^T) L_kk^-T          (MFMA)
// k_trinv: W = L^-1 by block columns (one launch), stored transposed
//   WT[j][i] = W[i][j]  so the predict GEMM reads both operands K-major.
#include "common.h"
#include "cov_device.h"

#define NB SPX_NB
#define LDP 66   // LDS row stride (doubles) for MFMA operand tiles: 16 rows x 2 cols hit 32 distinct 8-byte banks

// An MFMA operand out of LDS as ONE ds_read_b64.  Left to itself the compiler pairs the reads of neighbouring contraction
// steps (32 bytes apart) into ds_read2_b64, and that instruction is serviced in 16-lane groups over 32 banks instead of
// 32-lane halves over 64: with the operand pattern [16 rows li][k-slot g] at a row stride of 66 (or 18) doubles the lanes
// li and li + 8 of a group share a bank -- a 2-way conflict, 16 LDS cycles per pair of values where two ds_read_b64 take 4.
// (k_lean_flow in round 3: 276 ds_read2_b64, SQ_LDS_BANK_CONFLICT = 44 % of SQ_LDS_IDX_ACTIVE, matrix pipes 52 % busy
// with two workgroups per CU -- LDS-bound.)  A volatile access is never merged; the address-space cast keeps it a DS
// instruction (a plain volatile generic pointer would become flat_load).
typedef const volatile double __attribute__((address_space(3))) * lds_cvd_p;
__device__ __forceinline__ double lds_ld(const double* p) { return *(lds_cvd_p)p; }

__device__ __forceinline__ double readlane_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// copy a 64x64 tile (row stride ld in global) into LDS [64][LDP]; 256 threads
__device__ __forceinline__ void tile_to_lds(const double* __restrict__ g, size_t ld, double* lds)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = threadIdx.x + 256 * q;  // double2 units
        const int row = idx >> 5, c2 = idx & 31;
        const d2 v = *reinterpret_cast<const d2*>(g + (size_t)row * ld + 2 * c2);
        *reinterpret_cast<d2*>(lds + row * LDP + 2 * c2) = v;
    }
}

// the same copy split in two so the global loads of step p+1 fly while step p computes
struct TileRegs { d2 v[8]; };
__device__ __forceinline__ void tile_load(const double* __restrict__ g, size_t ld, TileRegs& t)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = threadIdx.x + 256 * q;
        t.v[q] = *reinterpret_cast<const d2*>(g + (size_t)(idx >> 5) * ld + 2 * (idx & 31));
    }
}
__device__ __forceinline__ void tile_store(const TileRegs& t, double* lds)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = threadIdx.x + 256 * q;
        *reinterpret_cast<d2*>(lds + (idx >> 5) * LDP + 2 * (idx & 31)) = t.v[q];
    }
}

// acc[nt] (+)= sign * A_lds[16w+li][:] . B_lds[16nt+li][:]^T over the 64-deep tile
__device__ __forceinline__ void mma_tile_64(const double* A_lds, const double* B_lds, d4 acc[4],
                                            int wave, int g, int li, bool negate)
{
#pragma unroll 4
    for (int k0 = 0; k0 < NB; k0 += 4) {
        double a = lds_ld(A_lds + (16 * wave + li) * LDP + k0 + g);
        if (negate) a = -a;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const double b = lds_ld(B_lds + (16 * nt + li) * LDP + k0 + g);
            acc[nt] = MFMA_F64(a, b, acc[nt]);
        }
    }
}

// ---------------------------------------------------------------------------
// Cholesky factor AND inverse of one symmetric 16x16 block, by one wavefront, on the matrix pipe.
// The block sits in the MFMA accumulator layout (reg r of lane (c, q) = element [q + 4r][c]); so
// does X, which starts as the identity.  Pivot j (row j = q + 4r with q = j & 3, r = j >> 2):
//   d = C[j][j]  (v_readlane),  rinv = d^-1/2  (v_rsq_f64 + one third-order step, ~1 ulp);
//   the lanes of group q = j & 3 hold row j of C -- which, C being symmetric, is column j -- in
//   reg j >> 2, exactly where the A / B operands of k-slot j & 3 are read from, so
//       C <- C - l l^T             l = C[:, j] rinv      (column j of L)
//       X <- X - l_{>j} (x_j rinv) x_j = row j of X      (row operations that turn I into L^-1)
//   are ONE v_mfma_f64_16x16x4 each (three k-slots zero) with operands straight out of the
//   accumulator registers: no cross-lane traffic except the one readlane of the pivot.
// The dependent chain per pivot is readlane -> rsq -> 5 fma -> 2 mul -> MFMA, against ~1100
// dependent VALU / readlane instructions for the register-row formulation it replaces (65 % of
// k_chol_diag's 30 us: 8.5k cycles per 16x16 block).
// Out: U[r] = (L^T)[q + 4r][c] (upper, zeros below the diagonal), X = L^-1 (lower).
// Rows and columns < j of C are dead after their pivot (they collect garbage; nothing reads them).
// (Splitting the two MFMAs over two wavefronts -- the factor wave publishing l and rinv through an LDS
// mailbox, per pivot or per four pivots, the inverse wave spinning on a sequence word -- was measured:
// the factor wave alone runs 300 cycles per pivot next to its busy neighbours, and the inverse wave
// trails by ~1.1k cycles, 5.9k per sub-block either way against 5.85k for this single-wave form.)
// Round 4: TWO pivots per pair of MFMAs.  Pivots j and j + 1 (j even) live in the neighbouring k-slots j & 3 and (j & 3) + 1
// of the same register, so one v_mfma_f64_16x16x4 applies both rank-1 updates.  What the second pivot needs of the first
// one's update -- its own row, C[j+1][:] - l_{j+1,j} l_j^T -- is one explicit fma per lane on the column l_j brought over
// from the neighbouring lane row (v_permlane16_swap_b32: rows 0 -> 1, 2 -> 3), and the pivot itself is
// fma(-l_{j+1,j}, l_{j+1,j}, C[j+1][j+1]) on three v_readlane values.  The matrix pipe adds its k-slots in ascending
// order, one rounded fma each, so every value is formed by the same operations as with one MFMA per pivot: the factor and
// the inverse are the same bits (scripts/ubench_factor16.hip compares the two forms value by value), 287 cycles per pivot
// instead of 331 -- each MFMA a pivot issues costs ~100 cycles of the in-order wave, and there are now half as many.
// row of 16 lanes q -> row q + 1 for q = 0, 2
__device__ __forceinline__ double from_even_row(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
__device__ __forceinline__ void factor16_mfma(d4& C, d4& Xo, d4& U, int lane, int& bad, int pivot_base)
{
    const int c = lane & 15, q = lane >> 4;
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        X[r] = (q + 4 * r == c) ? 1.0 : 0.0;
        Xo[r] = 0.0;
        U[r] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const int qa = j & 3, rj = j >> 2;
        const bool ga = (q == qa), gb = (q == qa + 1);
        double d0 = readlane_f64(C[rj], j + 16 * qa);
        const double e = readlane_f64(C[rj], j + 1 + 16 * qa);            // C[j][j+1]
        const double d1 = readlane_f64(C[rj], j + 1 + 16 * (qa + 1));
        if (!(d0 > 0.0)) {  // also catches NaN, like LAPACK dpotrf's (ajj <= 0 || isnan)
            if (!bad) bad = pivot_base + j + 1;
            d0 = 1.0;
        }
        const double y0 = __builtin_amdgcn_rsq(d0);
        const double e0 = fma(-d0 * y0, y0, 1.0);
        const double rinv0 = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        const double l10 = e * rinv0;                                    // L[j+1][j]
        double d1p = fma(-l10, l10, d1);                                 // the pivot j + 1 after pivot j's update
        if (!(d1p > 0.0)) {
            if (!bad) bad = pivot_base + j + 2;
            d1p = 1.0;
        }
        const double y1 = __builtin_amdgcn_rsq(d1p);
        const double e1 = fma(-d1p * y1, y1, 1.0);
        const double rinv1 = fma(y1 * e1, fma(0.375, e1, 0.5), y1);
        const double l0 = C[rj] * rinv0;                                 // lane row qa: column j of L
        const double x0 = X[rj] * rinv0;                                 // lane row qa: row j of the inverse (final)
        const double l0n = from_even_row(l0), x0n = from_even_row(x0);
        const double l1 = fma(-l10, l0n, C[rj]) * rinv1;                 // lane row qa + 1: column j + 1 of L
        const double x1 = fma(-l10, x0n, X[rj]) * rinv1;                 // lane row qa + 1: row j + 1 of the inverse
        const double bC = ga ? l0 : (gb ? l1 : 0.0);
        C = MFMA_F64(-bC, bC, C);
        const double bX = ga ? x0 : (gb ? x1 : 0.0);
        const double aX = (ga && c > j) ? -l0 : ((gb && c > j + 1) ? -l1 : 0.0);
        X = MFMA_F64(aX, bX, X);
        // rows j, j + 1 of X are final: collected in Xo, not written back into the accumulator (a VALU write to an MFMA
        // destination right behind the MFMA stalls the in-order wave)
        Xo[rj] = ga ? x0 : (gb ? x1 : Xo[rj]);
        // the diagonal entries sqrt(d) to ~0.5 ulp (off the critical path)
        double sd0 = d0 * rinv0;
        sd0 = fma(fma(-sd0, sd0, d0), 0.5 * rinv0, sd0);
        double sd1 = d1p * rinv1;
        sd1 = fma(fma(-sd1, sd1, d1p), 0.5 * rinv1, sd1);
        const double keep0 = (c == j) ? sd0 : ((c > j) ? l0 : 0.0);
        const double keep1 = (c == j + 1) ? sd1 : ((c > j + 1) ? l1 : 0.0);
        U[rj] = ga ? keep0 : (gb ? keep1 : U[rj]);
    }
}

// ---------------------------------------------------------------------------
// Factor the 64x64 block held (full, symmetric) in S and invert the factor; 256 threads (a larger workgroup may
// call it too: waves 4 and up only take part in the barriers).
//   S   [64][LDP]  in: the updated diagonal block; out: L_kk in the lower triangle
//   XT  [64][LDP]  out: (L_kk^-1)^T
//   T16 [4][16][18] scratch: inverses of the 16x16 diagonal sub-blocks
// then write L_kk (upper part zero) to Lkk (row stride ldl; may be null) and L_kk^-1 to Dk ([64][64]).
// Blocked with 16x16 sub-blocks b = 0..3; per round b
//   phase 1  wave 0 factors and inverts sub-block (b,b) on the matrix pipe (factor16_mfma); waves 1-3
//            finish the trailing updates of round b-1 -- every tile but (b,b), which wave 0 took
//            itself -- and the off-diagonal blocks of row b-1 of the inverse
//   phase 2  sub-panel: P_ti = S(ti,b) Linv_b^T for the tiles below (MFMA), wave 0 the first one
//   phase 3  wave 0 alone: S(b+1,b+1) -= P_{b+1} P_{b+1}^T, the one tile its next factorisation needs
// so the critical wave goes factor -> sub-panel tile -> trailing tile -> factor with two barriers per
// round.  Inverse by block forward substitution:
//   X_jj = Linv16_j ;  X_ij = -Linv16_i * sum_{p=j}^{i-1} L_ip X_pj          (i > j)
// XT holds X transposed so that X_pj is read in the MFMA B-operand pattern; the accumulator of the
// first product is itself in B-operand layout for the second.
#define DIAG_T16_DOUBLES (4 * 16 * 18)

__device__ __forceinline__ d4 inv_partial(const double* S, const double* XT, int i, int j, int g, int li)
{
    d4 t4 = (d4){0.0, 0.0, 0.0, 0.0};
    for (int pb = j; pb < i; ++pb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const double av = lds_ld(S + (16 * i + li) * LDP + 16 * pb + 4 * ks + g);
            const double bv = lds_ld(XT + (16 * j + li) * LDP + 16 * pb + 4 * ks + g);
            t4 = MFMA_F64(av, bv, t4);
        }
    }
    return t4;
}
// X_ij = -Linv16_i t4: into XT (transposed, for the products of the rows below) and straight to Dk (row-major
// [64][64]) from the registers of the wave that computed it
// A store other workgroups of the SAME launch read (k_lean_step_ps hands the inverse of the diagonal block to the
// panel workgroups while it is being built): relaxed, agent scope = a write-through `sc1` store, visible at the
// device's coherence point once the issuing wave's vmcnt reaches zero (MI355X_MICROARCH.md, inter-workgroup
// visibility: "8-B agent atomics both sides").
template <bool PUB>
__device__ __forceinline__ void dk_store(double* p, double v)
{
    if (PUB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <bool PUB>
__device__ __forceinline__ void inv_finish(const d4& t4, double* XT, const double* T16, double* __restrict__ Dk,
                                           int i, int j, int g, int li)
{
    d4 o4 = (d4){0.0, 0.0, 0.0, 0.0};
    const double* Ti = T16 + i * 16 * 18;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) o4 = MFMA_F64(-lds_ld(Ti + li * 18 + 4 * ks + g), t4[ks], o4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        XT[(16 * j + li) * LDP + 16 * i + g + 4 * r] = o4[r];
        dk_store<PUB>(Dk + (16 * i + g + 4 * r) * NB + 16 * j + li, o4[r]);
    }
    if (PUB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drained before the barrier that precedes the flag
}

// S(ti,tj) -= P_ti P_tj^T with the sub-panel tiles of column b
__device__ __forceinline__ void trail_tile(double* S, int ti, int tj, int b0, int g, int li)
{
    d4 c4;
#pragma unroll
    for (int r = 0; r < 4; ++r) c4[r] = S[(16 * ti + g + 4 * r) * LDP + 16 * tj + li];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const double av = -lds_ld(S + (16 * ti + li) * LDP + b0 + 4 * ks + g);
        const double bv = lds_ld(S + (16 * tj + li) * LDP + b0 + 4 * ks + g);
        c4 = MFMA_F64(av, bv, c4);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(16 * ti + g + 4 * r) * LDP + 16 * tj + li] = c4[r];
}

// the same for the tile wave 0 factors next, (t,t): the result stays in its registers (the accumulator layout is the
// layout factor16_mfma takes; the block's place in S is rewritten by the factor before anybody reads it)
__device__ __forceinline__ d4 trail_tile_regs(const double* S, int t, int b0, int g, int li)
{
    d4 c4;
#pragma unroll
    for (int r = 0; r < 4; ++r) c4[r] = S[(16 * t + g + 4 * r) * LDP + 16 * t + li];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const double av = -lds_ld(S + (16 * t + li) * LDP + b0 + 4 * ks + g);
        const double bv = lds_ld(S + (16 * t + li) * LDP + b0 + 4 * ks + g);
        c4 = MFMA_F64(av, bv, c4);
    }
    return c4;
}

#ifdef FLOW_STAMPS   // dev (k_lean_flow's time line, below): the diagonal item's own eight stamps; 4 .. 7 from inside flow_trsm / diag_block
__shared__ long long* s_fstamp;
#define FSTAMP2(j) do { if (threadIdx.x == 0 && s_fstamp) s_fstamp[j] = wall_clock64(); } while (0)
#else
#define FSTAMP2(j)
#endif
#ifdef SPX_DIAG_STAMPS   // dev: phase time stamps for scripts/ubench_diag.hip
__shared__ long long g_stamp[32];
#define STAMP(i) do { if (threadIdx.x == 0) g_stamp[i] = clock64(); } while (0)
#else
#define STAMP(i)
#endif
// PUB: block row b of the inverse (X_b0 .. X_bb) is published to the other workgroups of the launch as soon as it is
// complete -- write-through stores, drained, then *flag = b + 1 (after the barrier that follows the last store of the
// row) -- so that the panel solve of this block column proceeds behind the pivots instead of behind a launch boundary.
// direct (k_lean_flow's chain): wave 0 brings sub-block (0,0) in its registers (c0) and starts on it at once -- the other
// waves have stored rows 16 .. 63 of the block into S, nobody has synchronised yet; the barrier that S, XT and T16 need
// before wave 0 writes into them stands behind its first sub-block's pivots instead of in front of them.
template <bool PUB = false>
__device__ __forceinline__ void diag_block(double* S, double* XT, double* T16, int* info_h, int pivot_base,
                                           double* __restrict__ Lkk, size_t ldl, double* __restrict__ Dk,
                                           double* __restrict__ diag_out = nullptr, int* flag = nullptr, int flag_base = 0,
                                           bool direct = false, const d4* c0 = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    int bad = 0;
    d4 t4 = (d4){0.0, 0.0, 0.0, 0.0};     // waves 1-3: inner sum of the last row of the inverse (column wave - 1)
    d4 Cn = (d4){0.0, 0.0, 0.0, 0.0};     // wave 0: the sub-block of the next round, updated (phase 3)
    __shared__ int s_pub[5];              // PUB: waves 1-3 that have drained block row b - 1 of the inverse (first used behind a barrier)
    if (PUB && threadIdx.x < 5) s_pub[threadIdx.x] = 0;
    STAMP(0);
    if (wave > 0 && wave < 4) {
        // The inverse goes to global memory block by block from the registers of the wave that computes it; the
        // blocks above the block diagonal are zero: (0,1) (0,2) | (0,3) (1,2) | (1,3) (2,3) for waves 1 | 2 | 3.
        const int zi0 = (wave == 3) ? 1 : 0, zj0 = (wave == 1) ? 1 : 3;
        const int zi1 = (wave == 1) ? 0 : (wave == 2 ? 1 : 2), zj1 = (wave == 3) ? 3 : 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Dk[(16 * zi0 + g + 4 * r) * NB + 16 * zj0 + li] = 0.0;
            Dk[(16 * zi1 + g + 4 * r) * NB + 16 * zj1 + li] = 0.0;
        }
    }
    for (int b = 0; b < 4; ++b) {
        const int b0 = 16 * b;
        double* Tb = T16 + b * 16 * 18;
        // ---- phase 1 ----
        d4 C, X, U;
        if (wave == 0) {
            if (b > 0) C = Cn;                                  // (from phase 3 of the round before)
            else if (direct) C = *c0;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) C[r] = S[(g + 4 * r) * LDP + li];
            }
            factor16_mfma(C, X, U, lane, bad, pivot_base + b0);
        }
        if (b == 0 && direct) __syncthreads();                  // everybody is done with what S, XT and T16 held before
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = g + 4 * r;                      // U[row][li] = L[li][row]
                S[(b0 + li) * LDP + b0 + row] = U[r];           // lower triangle incl. diagonal, zeros above
                Tb[row * 18 + li] = X[r];                       // Linv16[row][col = li]
                XT[(b0 + li) * LDP + b0 + row] = X[r];          // XT[col][row] = X[row][col]
            }
        } else if (b > 0 && wave < 4) {
            // row b-1 of the inverse first: wave j + 1 owns block column j (all X_pj of a column come from one wave)
            if (wave - 1 < b - 1) inv_finish<PUB>(inv_partial(S, XT, b - 1, wave - 1, g, li), XT, T16, Dk, b - 1, wave - 1, g, li);
            if (wave == 3) {
                // the diagonal block of row b-1 of the inverse, from T16 to global memory -- here, where waves 1-3 have
                // slack, not by wave 0 from its registers (four global stores in the critical wave cost ~200 cycles per
                // round) and not in phase 2 (whose barrier would wait for the stores to drain when they are published)
                const double* Tp = T16 + (b - 1) * 16 * 18;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dk_store<PUB>(Dk + (b0 - 16 + g + 4 * r) * NB + b0 - 16 + li, Tp[(g + 4 * r) * 18 + li]);
                if (PUB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // block row b - 1 of the inverse is complete once the three waves' stores are drained: the LAST of them to get
            // here publishes rows 0 .. b - 1 -- now, while wave 0 is still busy with sub-block b, not behind the barrier that
            // ends its pivots (the solve that follows this block in k_lean_flow's chain is a quarter behind otherwise)
            if (PUB && lane == 0 && __hip_atomic_fetch_add(s_pub + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 2)
                __hip_atomic_store(flag, flag_base + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // trailing tiles of round b-1 other than (b,b): (ti,tj), b-1 < tj <= ti, dealt to waves 1-3
            int idx = 0;
            for (int ti = b; ti < 4; ++ti)
                for (int tj = b; tj <= ti; ++tj) {
                    if (ti == b && tj == b) continue;
                    if ((idx++ % 3) == wave - 1) trail_tile(S, ti, tj, b0 - 16, g, li);
                }
            // while wave 0 factors the last sub-block: the inner sum of the LAST row, so that only the product
            // with Linv16_3 is left behind the last pivot (reads this wave's own XT stores: same wave, in order)
            if (b == 3) t4 = inv_partial(S, XT, 3, wave - 1, g, li);
        }
        STAMP(1 + 4 * b);
        if (PUB && b == 0) FSTAMP2(7);
        __syncthreads();
        STAMP(2 + 4 * b);
        // ---- phase 2: sub-panel, rows of tile ti = b+1+wave: P <- P Linv16^T ----
        {
            const int ti = b + 1 + wave;
            if (ti < 4) {
                d4 c4 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const double av = lds_ld(S + (16 * ti + li) * LDP + b0 + 4 * ks + g);
                    const double bv = lds_ld(Tb + li * 18 + 4 * ks + g);
                    c4 = MFMA_F64(av, bv, c4);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(16 * ti + g + 4 * r) * LDP + b0 + li] = c4[r];
            }
        }
        if (b < 3) __syncthreads();   // round 3 has no sub-panel: nothing was written
        STAMP(3 + 4 * b);
        // ---- phase 3: the one trailing tile the next factorisation needs (its operands are wave 0's own) ----
        if (wave == 0 && b < 3) Cn = trail_tile_regs(S, b + 1, b0, g, li);
        STAMP(4 + 4 * b);
    }
    if (wave == 0 && lane == 0 && bad) {
        if (PUB) {
            // several diagonal blocks of a draw can be in flight in one launch (k_lean_flow), on different XCDs: the
            // lowest failed pivot wins, at device scope (a hand-off timeout, < 0, stays)
            const int old = atomicCAS(info_h, 0, bad);
            if (old > bad) atomicMin(info_h, bad);
        } else if (*info_h == 0) {
            *info_h = bad;
        }
    }
    // last row of the inverse: one product with Linv16_3 per block (rows 0-2 went out as they were built)
    if (wave == 3) {         // the last diagonal block of the inverse
        const double* Tp = T16 + 3 * 16 * 18;
#pragma unroll
        for (int r = 0; r < 4; ++r) dk_store<PUB>(Dk + (48 + g + 4 * r) * NB + 48 + li, Tp[(g + 4 * r) * 18 + li]);
    }
    // the log-likelihood path only needs the diagonal of L_kk (diag_out; S is complete since the last barrier).  PUB: it
    // leaves written through and drained (inv_finish below) BEFORE the block's last flag, so that a workgroup of the same
    // launch that has seen the flag -- the one that reduces the log-likelihood, k_lean_flow's `fused` form -- reads it
    if (PUB && diag_out && wave == 3) dk_store<true>(diag_out + lane, S[lane * LDP + lane]);
    if (wave > 0 && wave < 4) inv_finish<PUB>(t4, XT, T16, Dk, 3, wave - 1, g, li);
    if (PUB) {   // the last block row (inv_finish drained this wave's stores).  Wave 0 joins the count behind its not-PD
                 // report above (its atomicCAS has returned): whoever sees the flag sees info too
        if (lane == 0 && __hip_atomic_fetch_add(s_pub + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 3)
            __hip_atomic_store(flag, flag_base + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    STAMP(17);
    if (!PUB && diag_out && threadIdx.x < NB) diag_out[threadIdx.x] = S[threadIdx.x * LDP + threadIdx.x];
    if (Lkk) {
        for (int idx = threadIdx.x; idx < NB * NB / 2; idx += blockDim.x) {
            const int row = idx >> 5, col = (idx & 31) * 2;
            d2 lv;
            lv[0] = (col <= row) ? S[row * LDP + col] : 0.0;
            lv[1] = (col + 1 <= row) ? S[row * LDP + col + 1] : 0.0;
            *reinterpret_cast<d2*>(Lkk + (size_t)row * ldl + col) = lv;
        }
    }
    STAMP(18);
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_chol_diag(double* __restrict__ Lm, double* __restrict__ Dinv,
                                                   int* __restrict__ info, int Np, int k, int updated)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* P = smem;                  // [64][LDP] staging tile of the row panel
    double* S = P + NB * LDP;          // [64][LDP] the diagonal block, factored in place
    double* XT = S + NB * LDP;         // [64][LDP] (L_kk^-1)^T
    double* T16 = XT + NB * LDP;       // [4][16][18] inverses of the 16x16 diagonal sub-blocks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x;
    const int nblk = Np / NB;
    double* Lh = Lm + (size_t)h * Np * Np;
    const size_t kb0 = (size_t)k * NB;

    // S = K_kk - sum_p L_kp L_kp^T
    d4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[nt][r] = Lh[(kb0 + 16 * wave + g + 4 * r) * Np + kb0 + 16 * nt + li];
    // steps p < k-1 were already applied in place by the previous panel launch (k_chol_panel,
    // diag_pre); only p = k-1, whose tile that launch produced, is left.
    if (k > 0 && !updated) {
        tile_to_lds(Lh + kb0 * Np + (size_t)(k - 1) * NB, Np, P);
        __syncthreads();
        mma_tile_64(P, P, acc, wave, g, li, true);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(16 * wave + g + 4 * r) * LDP + 16 * nt + li] = acc[nt][r];
    __syncthreads();
    diag_block(S, XT, T16, info + h, (int)kb0, Lh + kb0 * Np + kb0, (size_t)Np,
               Dinv + ((size_t)h * nblk + k) * NB * NB);
}

void launch_chol_diag(hipStream_t s, double* L, double* Dinv, int* info, int Np, int k, int nh, int updated)
{
    const size_t lds = (size_t)(3 * NB * LDP + DIAG_T16_DOUBLES) * sizeof(double);   // 112 KB > the 64 KB default
    SPX_LDS_ATTR(k_chol_diag, lds);
    hipLaunchKernelGGL(k_chol_diag, dim3(nh), dim3(256), lds, s, L, Dinv, info, Np, k, updated);
}

// ---------------------------------------------------------------------------
// With pre != 0, workgroup x = 0 of every draw does not compute a panel tile: it applies the
// first k steps of the NEXT diagonal block's update, K_{k+1,k+1} -= sum_{p<k} L_{k+1,p} L_{k+1,p}^T
// (in place; those tiles are final), so that k_chol_diag(k+1) -- alone on the chip, the critical
// path of the factorisation -- is left with the single step p = k instead of k + 1 steps.  Same
// accumulation order as before, so the factor is bit-identical.
// With rhs != null, the last workgroup x of every draw carries the right-hand side along: rhs is
// a [nh][64][Np] buffer whose row 0 holds r^T = (vals - mean)^T (rows 1..63 zero), treated as one
// more row block below the matrix -- the Cholesky of [[K, r], [r^T, .]] has y^T = (L^-1 r)^T in
// that row.  The forward solve of the log-likelihood then costs no extra pass over L and no
// latency of its own (it ran 2.4 ms as a separate sequential kernel at N=2048).
__global__ __launch_bounds__(256, 2) void k_chol_panel(double* __restrict__ Lm,
                                                    const double* __restrict__ Dinv, int Np, int k, int pre,
                                                    double* __restrict__ rhs, int nsteps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* A = smem;                  // [2][64][LDP]
    double* B = smem + 2 * NB * LDP;   // [2][64][LDP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int nblk = Np / NB;
    const bool diag_pre = pre && blockIdx.x == 0;
    const bool is_rhs = rhs && blockIdx.x == gridDim.x - 1;
    const int rb = k + 1 + (int)blockIdx.x - (pre && !diag_pre ? 1 : 0);
    double* Lh = Lm + (size_t)h * Np * Np;
    const size_t rb0 = (size_t)rb * NB;
    // this tile's 64 rows: block row rb of the matrix, or the right-hand-side rows
    double* Ar = is_rhs ? rhs + (size_t)h * NB * Np : Lh + rb0 * Np;
    // the other operand's rows (and this tile's columns): block row k, or rb itself for diag_pre
    const size_t kb0 = diag_pre ? rb0 : (size_t)k * NB;

    d4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[nt][r] = Ar[(size_t)(16 * wave + g + 4 * r) * Np + kb0 + 16 * nt + li];
    if (nsteps > 0) {   // the k steps of this tile's left-looking update
        TileRegs ta, tb;
        tile_load(Ar, Np, ta);
        tile_load(Lh + kb0 * Np, Np, tb);
        for (int p = 0; p < nsteps; ++p) {
            double* Ac = A + (p & 1) * NB * LDP;
            double* Bc = B + (p & 1) * NB * LDP;
            tile_store(ta, Ac);
            tile_store(tb, Bc);
            __syncthreads();
            if (p + 1 < nsteps) {
                tile_load(Ar + (size_t)(p + 1) * NB, Np, ta);
                tile_load(Lh + kb0 * Np + (size_t)(p + 1) * NB, Np, tb);
            }
            mma_tile_64(Ac, Bc, acc, wave, g, li, true);
        }
    }
    if (diag_pre) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ar[(size_t)(16 * wave + g + 4 * r) * Np + kb0 + 16 * nt + li] = acc[nt][r];
        return;
    }
    __syncthreads();
    // L_rk = S L_kk^-T :  out[i][n] = sum_q S[i][q] Dinv[n][q]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(16 * wave + g + 4 * r) * LDP + 16 * nt + li] = acc[nt][r];
    tile_to_lds(Dinv + ((size_t)h * nblk + k) * NB * NB, NB, B);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
    mma_tile_64(A, B, acc, wave, g, li, false);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Ar[(size_t)(16 * wave + g + 4 * r) * Np + kb0 + 16 * nt + li] = acc[nt][r];
}

void launch_chol_panel(hipStream_t s, double* L, const double* Dinv, int Np, int k, int nh, double* rhs)
{
    const int nblk = Np / NB;
    const int nrows = nblk - k - 1;
    // k = 0: the next diagonal block has no earlier steps
    const int pre = (k > 0 && nrows > 0) ? 1 : 0;
    const int nx = nrows + pre + (rhs ? 1 : 0);
    if (nx <= 0) return;
    const size_t lds = (size_t)(4 * NB * LDP) * sizeof(double);                // 135 KB
    SPX_LDS_ATTR(k_chol_panel, lds);
    hipLaunchKernelGGL(k_chol_panel, dim3(nx, nh), dim3(256), lds, s, L, Dinv, Np, k, pre, rhs,
                       k);
}

// ---------------------------------------------------------------------------
// Log-likelihood path (spx_gp_logprob, up to 32 draws): right-looking factorisation of 64x64 tiles.
//
// Storage.  The matrix of a draw is kept TILE-MAJOR: tile (I, J) at ((I nblk + J) * 4096) doubles,
// and inside a tile in MFMA accumulator order, as 8 planes of [256 threads][2 doubles]: value
// q = 4 nt + r of thread t = 64 wave + lane -- the element (row 16 wave + (lane >> 4) + 4 r, column
// 16 nt + (lane & 15)) -- sits at ((q >> 1) * 256 + t) * 2 + (q & 1).  Every kernel below moves a tile
// with eight 16-byte accesses per thread, each of them 1 KiB contiguous across the wavefront; with a
// row-major matrix the same accumulator traffic is 8 bytes per lane in 128-byte row segments, and
// timing-only ablations showed those loads, not the MFMAs, bound the trailing update (DESIGN.md
// section 8).  k_cov writes the same layout (launch_cov_self, tiled); the right-hand side rows are one
// more block row.
//
// k_lean_step: launch k applies update step k-1 to every remaining lower tile,
//   A_ij -= L_i,k-1 L_j,k-1^T      for k <= j <= i   (and the right-hand-side rows),
// one 64-deep MFMA step per tile, in place -- and the workgroup that owns the diagonal tile (k, k),
// which is thereby complete, goes straight on to factor and invert it (diag_block), so that the
// serial part of the factorisation (the 64x64 diagonal blocks, ~15 us each) runs beside the bulk of
// the update instead of in a launch of its own: two launches per block column (this one and the
// triangular solve of the panel, k_lean_trsm) instead of three.  A workgroup walks up to LEAN_CH
// tiles of one block row: the row operand stays in LDS, the next tile's operand and accumulator are
// in flight while the current one computes.  The steps reach each tile in the order p = 0, 1, ... as
// in the left-looking kernel of the EI path, so the factor is bit-identical.
// Grid (draws, rows, column chunks) over the trailing tiles; k = 0: only the diagonal workgroups.
#ifndef LEAN_CH
#define LEAN_CH 4
#endif
#define LEAN_TILE (NB * NB)

// a tile in accumulator order <-> the [64][LDP] MFMA operand layout in LDS
__device__ __forceinline__ void acc_tile_to_lds(const d4 (&t)[4], double* lds, int wave, int g, int li)
{
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(16 * wave + g + 4 * r) * LDP + 16 * nt + li] = t[nt][r];
}
__device__ __forceinline__ void load_tile(const double* __restrict__ tile, d4 (&t)[4])
{
    const d2* p = reinterpret_cast<const d2*>(tile) + threadIdx.x;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const d2 lo = p[(2 * nt) * 256], hi = p[(2 * nt + 1) * 256];
        t[nt] = (d4){lo[0], lo[1], hi[0], hi[1]};
    }
}
__device__ __forceinline__ void store_tile(double* __restrict__ tile, const d4 (&t)[4])
{
    d2* p = reinterpret_cast<d2*>(tile) + threadIdx.x;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        p[(2 * nt) * 256] = (d2){t[nt][0], t[nt][1]};
        p[(2 * nt + 1) * 256] = (d2){t[nt][2], t[nt][3]};
    }
}

__global__ __launch_bounds__(256, 2) void k_lean_step(double* __restrict__ Lt, double* __restrict__ Dinv,
                                                   int* __restrict__ info, double* __restrict__ rhs,
                                                   double* __restrict__ diagL, int Np, int k, int col_only)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* A = smem;              // [64][LDP]  L_i,k-1 (or the right-hand-side rows' block k-1); then S
    double* B = smem + NB * LDP;   // [64][LDP]  L_j,k-1; then XT
    double* T16 = B + NB * LDP;    // [4][16][18]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    // Workgroups are dispatched in linear block order, x fastest: with the draws on x, the diagonal
    // workgroups of ALL draws (row 0, chunk 0) start first and the bulk of the update fills in behind
    // them; with the draws on z the last draw's diagonal block started only after the other draws'
    // tiles had been handed out and stuck out at the end of the launch.
    const int h = blockIdx.x;
    const int nblk = Np / NB;
    const bool is_rhs = rhs && blockIdx.y == gridDim.y - 1;
    const int i = k + blockIdx.y;
    // this workgroup's tiles: block row i (or the right-hand-side rows), block columns j0 .. j1 - 1
    const int j0 = k + blockIdx.z * LEAN_CH;
    // col_only (lazy updates, odd k): only block columns k and k + 1 take step k-1 now -- column k is then complete
    // and column k + 1 will need just ONE more step before its diagonal block is factored, so no diagonal block ever
    // waits for two -- and the columns to their right take the steps k-1 and k together in the next launch (k_lean_step2)
    const int j1 = min(col_only ? j0 + 2 : j0 + LEAN_CH, is_rhs ? nblk : i + 1);
    if (j0 >= j1) return;
    double* Lh = Lt + (size_t)h * Np * Np;
    double* row = is_rhs ? rhs + (size_t)h * nblk * LEAN_TILE : Lh + (size_t)i * nblk * LEAN_TILE;   // tiles (i, .)
    const int kp = k > 0 ? k - 1 : 0;
    d4 acc[4], accn[4], tb[4];
    load_tile(row + (size_t)j0 * LEAN_TILE, accn);
    if (k > 0) {
        load_tile(Lh + ((size_t)j0 * nblk + kp) * LEAN_TILE, tb);
        d4 ta[4];
        load_tile(row + (size_t)kp * LEAN_TILE, ta);
        acc_tile_to_lds(ta, A, wave, g, li);      // the row operand, once
    }
    for (int j = j0; j < j1; ++j) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = accn[nt];
        if (k > 0) {
            acc_tile_to_lds(tb, B, wave, g, li);
            __syncthreads();
            if (j + 1 < j1) {              // the next tile's operand and accumulator fly while this one computes
                load_tile(Lh + ((size_t)(j + 1) * nblk + kp) * LEAN_TILE, tb);
                load_tile(row + (size_t)(j + 1) * LEAN_TILE, accn);
            }
            mma_tile_64(A, B, acc, wave, g, li, true);
        }
        if (!is_rhs && i == k && j == k) {
            // the serial part of the factorisation (this is the only tile of this workgroup)
            __syncthreads();           // every wave is done reading A / B
            acc_tile_to_lds(acc, A, wave, g, li);
            __syncthreads();
            diag_block(A, B, T16, info + h, k * NB, nullptr, 0, Dinv + ((size_t)h * nblk + k) * NB * NB,
                       diagL + (size_t)h * Np + (size_t)k * NB);
            return;
        }
        if (k == 0) return;
        store_tile(row + (size_t)j * LEAN_TILE, acc);
        __syncthreads();               // B is rewritten for the next tile
    }
}

// k_lean_step_ps: k_lean_step with the PANEL SOLVE of block column k inside the same launch -- one launch per block
// column, no redundant products (a form in which every workgroup recomputed the panel operands it needs, k_lean_fused,
// was measured and dropped: scripts/dev/attic).  The workgroup that owns the first chunk of block row i > k
// (and of the right-hand-side rows) keeps its updated tile (i,k) and, once its chunk is done, forms
//     L_ik = R_ik Dinv_k^T
// block column by block column of the result, BEHIND the pivots of the diagonal workgroup: diag_block<true> publishes
// block row b of Dinv_k (write-through stores, drained, then a flag) as soon as it is complete -- three of the four
// rows before the last 16 pivots have run -- and out columns 16 b .. 16 b + 15 need exactly the rows 0 .. b.  What is
// left behind the last pivot is a flag hand-off, a 6 KB read and a quarter of a tile product, instead of a launch
// boundary, a 64 KB read and a whole one.  The products skip the structurally zero upper blocks of Dinv_k (adding
// exact zeros: same bits as k_lean_trsm).  One lane per waiting workgroup polls (relaxed agent-scope load + s_sleep,
// bounded); the diagonal workgroups are the first of the grid, so they are resident before anyone waits for them.
// flags: [nh][nblk] ints, zeroed by the caller.
#define PS_SPIN_LIMIT (1 << 22)

__global__ __launch_bounds__(256, 2) void k_lean_step_ps(double* __restrict__ Lt, double* __restrict__ Dinv,
                                                      int* __restrict__ info, double* __restrict__ rhs,
                                                      double* __restrict__ diagL, int* __restrict__ flags, int Np, int k)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* A = smem;              // [64][LDP]  L_i,k-1; then S (diagonal workgroup) / R_ik (panel solve)
    double* B = smem + NB * LDP;   // [64][LDP]  L_j,k-1; then XT / the published rows of Dinv_k
    double* T16 = B + NB * LDP;    // [4][16][18]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x;      // draws on x: the diagonal workgroups of all draws are dispatched first
    const int nblk = Np / NB;
    const bool is_rhs = blockIdx.y == gridDim.y - 1;
    const int i = k + blockIdx.y;
    const int j0 = k + blockIdx.z * LEAN_CH;
    const int j1 = (k == 0) ? j0 + 1 : min(j0 + LEAN_CH, is_rhs ? nblk : i + 1);   // k = 0: nothing to apply, tile (i,0) only
    if (j0 >= j1) return;
    double* Lh = Lt + (size_t)h * Np * Np;
    double* row = is_rhs ? rhs + (size_t)h * nblk * LEAN_TILE : Lh + (size_t)i * nblk * LEAN_TILE;   // tiles (i, .)
    double* Dk = Dinv + ((size_t)h * nblk + k) * NB * NB;
    int* flag = flags + (size_t)h * nblk + k;
    const bool solves = (j0 == k) && (is_rhs || i > k);      // this workgroup owns tile (i,k): it solves it too
    const int kp = k > 0 ? k - 1 : 0;
    d4 acc[4], accn[4], tb[4], rk[4];
    load_tile(row + (size_t)j0 * LEAN_TILE, accn);
    if (k > 0) {
        load_tile(Lh + ((size_t)j0 * nblk + kp) * LEAN_TILE, tb);
        d4 ta[4];
        load_tile(row + (size_t)kp * LEAN_TILE, ta);
        acc_tile_to_lds(ta, A, wave, g, li);      // the row operand, once
    }
    for (int j = j0; j < j1; ++j) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = accn[nt];
        if (k > 0) {
            acc_tile_to_lds(tb, B, wave, g, li);
            __syncthreads();
            if (j + 1 < j1) {              // the next tile's operand and accumulator fly while this one computes
                load_tile(Lh + ((size_t)(j + 1) * nblk + kp) * LEAN_TILE, tb);
                load_tile(row + (size_t)(j + 1) * LEAN_TILE, accn);
            }
            mma_tile_64(A, B, acc, wave, g, li, true);
        }
        if (!is_rhs && i == k && j == k) {
            // the serial part of the factorisation (this is the only tile of this workgroup)
            __syncthreads();           // every wave is done reading A / B
            acc_tile_to_lds(acc, A, wave, g, li);
            __syncthreads();
            diag_block<true>(A, B, T16, info + h, k * NB, nullptr, 0, Dk, diagL + (size_t)h * Np + (size_t)k * NB, flag);
            return;
        }
        if (solves && j == k) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) rk[nt] = acc[nt];       // R_ik stays here; its tile is written once, solved
        } else if (k > 0) {
            store_tile(row + (size_t)j * LEAN_TILE, acc);
        }
        if (k > 0) __syncthreads();    // B is rewritten for the next tile
    }
    if (!solves) return;
    // ---- panel solve of tile (i,k), block column by block column, behind the diagonal workgroup ----
    acc_tile_to_lds(rk, A, wave, g, li);
    d4 out[4];
    for (int b = 0; b < 4; ++b) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= b && ++spins < PS_SPIN_LIMIT)
                __builtin_amdgcn_s_sleep(8);
            if (spins >= PS_SPIN_LIMIT && info[h] == 0) info[h] = -1;   // never on a healthy device: not a hang, an error
        }
        __syncthreads();
        // rows 16 b .. 16 b + 15 of Dinv_k, columns 0 .. 16 b + 15, past this CU's L1 (the producer wrote them through)
        // (16 x 16 (b + 1) doubles = b + 1 per thread; all loads in flight before the first LDS store)
        double dv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m <= b) {
                const int e = threadIdx.x + 256 * m;
                dv[m] = __hip_atomic_load(Dk + (16 * b + e / (16 * (b + 1))) * NB + e % (16 * (b + 1)), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m <= b) {
                const int e = threadIdx.x + 256 * m;
                B[(16 * b + e / (16 * (b + 1))) * LDP + e % (16 * (b + 1))] = dv[m];
            }
        __syncthreads();
        out[b] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < 16 * (b + 1); k0 += 4)
            out[b] = MFMA_F64(lds_ld(A + (16 * wave + li) * LDP + k0 + g), lds_ld(B + (16 * b + li) * LDP + k0 + g), out[b]);
    }
    store_tile(row + (size_t)k * LEAN_TILE, out);
}

void launch_lean_step_ps(hipStream_t s, double* Lt, double* Dinv, int* info, double* rhs, double* diagL, int* flags,
                         int Np, int k, int nh)
{
    const int n = Np / NB - k;
    if (n <= 0) return;
    const size_t lds = (size_t)(2 * NB * LDP + DIAG_T16_DOUBLES) * sizeof(double);   // 74.5 KB: two workgroups per CU
    SPX_LDS_ATTR(k_lean_step_ps, lds);
    // rows k .. nblk-1 and the right-hand-side rows; k = 0 has nothing to apply: one chunk (tile (i,0)) per row
    const dim3 grid(nh, n + 1, k == 0 ? 1 : (n + LEAN_CH - 1) / LEAN_CH);
    hipLaunchKernelGGL(k_lean_step_ps, grid, dim3(256), lds, s, Lt, Dinv, info, rhs, diagL, flags, Np, k);
}

// ---------------------------------------------------------------------------
// k_lean_flow: the WHOLE factorisation -- of the log-likelihood path (spx_gp_logprob, with the right-hand-side rows) and
// of the EI path (spx_factor, without) -- in ONE launch, as a data-flow program.
//
// Every dependency of the blocked factorisation -- diagonal block -> panel tile -> update -> next diagonal block --
// is a hand-off inside the launch (the mechanism k_lean_step_ps proved: write-through stores, drained, a flag;
// readers past their L1), so no dependency waits for a launch boundary or for unrelated tiles of the same step.
// A workgroup owns two neighbouring tiles (i, hi-1), (i, hi) of a block row (chunks are aligned to the RIGHT end of
// the row, so that the tile next to the diagonal and the diagonal tile always share a workgroup) and processes them
// LEFT-looking, accumulators in registers:
//   0. the tiles of K(X,X) + noise are built where they are consumed (flow_cov_tile; or loaded, if k_cov ran);
//   1. history: for k < hi-1 both tiles take step k as soon as L_ik, L_hi-1,k and L_hi,k are published;
//   2. tile (i, hi-1): the diagonal block (if it is the diagonal tile) or its panel solve, block column by block
//      column behind the pivots of the diagonal workgroup of that column, then PUBLISHED (tile + flag);
//   3. tile (i, hi): step hi-1 with the tile just solved, then the diagonal block or the panel solve.
// Per tile the steps arrive in the order 0, 1, 2, ... through the same MFMA chains as everywhere else: same bits.
// The dependent chain of a block column is   diagonal block -> (hand-off) -> last quarter of ONE panel solve ->
// one tile product -> next diagonal block,   all of the last three in the same workgroup.
// Deadlock freedom does not depend on residency: work items are numbered column-major (last column ascending, block
// row ascending; draws fastest), every wait is for a tile of a LOWER-numbered item (columns to the left in the same
// row or in a row above; the diagonal item of the same column, a row above), and a workgroup takes its item from a
// ticket counter when it starts -- whoever is waited for is running or done.  Waits are one lane polling (relaxed agent-scope loads,
// s_sleep), bounded: a timeout is an error (info < 0), never a hang.
// Flags carry the call's generation (no memset per call): Lflag[h][row][col] = 8 gen once tile (row, col) of L is
// published; Dflag[h][col] = 8 gen + b once block rows 0 .. b-1 of Dinv_col are.  One scale for both kinds: the batch
// size and the matrix size change between calls and with them which word is which flag -- whatever an earlier call
// left anywhere is below 8 gen.
// The call follows the chain of diagonal blocks, and a diagonal block runs a third slower beside a neighbour on its
// CU whose products keep the matrix pipes busy: small launches ask for so much LDS that every workgroup has a CU to
// itself; in large ones a workgroup yields while its neighbour is the next link of a chain (flow_step, cu_busy).
#define FLOW_SPIN_LIMIT (1 << 20)
// (the limit in force: the launch's own figure -- option flow_spin_limit, a test sets 1 to see a time-out -- or the default)
__shared__ int s_flow_spin_limit;
#ifdef FLOW_STAMPS   // dev (make FLOW_STAMPS=1; scripts/dev/flow_timeline.py): wall-clock stamps of every diagonal item,
                     // [draw][column][4] = item start, history done, diagonal block start, diagonal block end
__device__ long long g_flow_stamps[32 * 64 * 8];
extern "C" void spx_dev_flow_stamps(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flow_stamps), sizeof(g_flow_stamps)); }
// (and the shader clock at the start and the end of every diagonal block: cycles per microsecond of the launch)
__device__ long long g_flow_clk[32 * 64 * 2];
extern "C" void spx_dev_flow_clk(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flow_clk), sizeof(g_flow_clk)); }
#define FCLK(h, c, j) do { if (threadIdx.x == 0 && (h) < 32) g_flow_clk[((h) * 64 + (c)) * 2 + (j)] = clock64(); } while (0)
#define FSTAMP(h, c, j) do { if (threadIdx.x == 0 && (h) < 32) g_flow_stamps[((h) * 64 + (c)) * 8 + (j)] = wall_clock64(); } while (0)
#else
#define FSTAMP(h, c, j)
#define FCLK(h, c, j)
#endif
#define FLOW_BATCH 8        // history steps looked at per poll (3 flags each: 24 lanes)

// a tile past the non-coherent caches (sc1: device scope), 16 bytes per lane and access like load_tile / store_tile
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define SPX_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const double* tile)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(tile), 0, LEAN_TILE * 8, 0x00020000);
}
__device__ __forceinline__ void load_tile_sc1(const double* __restrict__ tile, d4 (&t)[4])
{
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const d2 lo = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)threadIdx.x * 16 + (2 * nt) * 4096, 0, SPX_SC1));
        const d2 hi = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)threadIdx.x * 16 + (2 * nt + 1) * 4096, 0, SPX_SC1));
        t[nt] = (d4){lo[0], lo[1], hi[0], hi[1]};
    }
}
// the same into eight 16-byte pieces (plane p = values 2p, 2p + 1 of the accumulator order): a tile that is only on its
// way to LDS is not assembled into 32-byte accumulator registers (the register allocator pays for that with copies)
__device__ __forceinline__ void load_tile_sc1_p(const double* __restrict__ tile, d2 (&t)[8])
{
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile);
#pragma unroll
    for (int p = 0; p < 8; ++p)
        t[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)threadIdx.x * 16 + p * 4096, 0, SPX_SC1));
}
__device__ __forceinline__ void load_tile_p(const double* __restrict__ tile, d2 (&t)[8])     // (a tile of an earlier launch)
{
    const d2* p = reinterpret_cast<const d2*>(tile) + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = p[q * 256];
}
__device__ __forceinline__ void planes_to_lds(const d2 (&t)[8], double* lds, int wave, int g, int li)
{
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(16 * wave + g + 4 * r) * LDP + 16 * nt + li] = t[2 * nt + (r >> 1)][r & 1];
}
// the planes of block column nt of a tile (no drain: the caller waits for vmcnt(0) before it raises the flag)
__device__ __forceinline__ void store_tile_quarter_sc1(double* __restrict__ tile, const d4& v, int nt)
{
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, (d2){v[0], v[1]}), rs, (int)threadIdx.x * 16 + (2 * nt) * 4096, 0, SPX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, (d2){v[2], v[3]}), rs, (int)threadIdx.x * 16 + (2 * nt + 1) * 4096, 0, SPX_SC1);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// one lane waits until *f >= want (bounded); everybody leaves together
__device__ __forceinline__ void flow_wait(const int* f, int want, int* info_h)
{
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < s_flow_spin_limit) {
            __builtin_amdgcn_s_sleep(8);
            // somebody else gave up already: do not queue a second timeout behind the first
            if ((spins & 1023) == 0 && __hip_atomic_load(info_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) break;
        }
        if (spins >= s_flow_spin_limit) __hip_atomic_store(info_h, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// one lane waits until *f >= want (bounded) and tells everybody what it saw: the progress of a diagonal block
// beyond the row waited for lets the caller go on without asking again
__device__ __forceinline__ int flow_wait_value(const int* f, int want, int* info_h, int* s_val)
{
    if (threadIdx.x == 0) {
        int spins = 0, v;
        while ((v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want && ++spins < s_flow_spin_limit) {
            __builtin_amdgcn_s_sleep(8);
            if ((spins & 1023) == 0 && __hip_atomic_load(info_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) break;
        }
        if (spins >= s_flow_spin_limit) __hip_atomic_store(info_h, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_val = v < want ? want : v;      // gave up: go on (the call fails with info < 0)
    }
    __syncthreads();
    return *s_val;
}

// a1 -= X X^T for the 16 columns of the solve held in Q ([64][18]): the wave's row block against the first NTL row blocks
template <int NTL>
__device__ __forceinline__ void syrk_quarter(const double* Q, d4 (&a1)[4], int wave, int g, int li)
{
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
        const double a = -lds_ld(Q + (16 * wave + li) * 18 + k0 + g);
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) a1[nt] = MFMA_F64(a, lds_ld(Q + (16 * nt + li) * 18 + k0 + g), a1[nt]);
    }
}

// rows 16 b .. 16 b + 15 of Dinv_col, columns 0 .. 16 b + 15: b + 1 values per thread, past the L1
__device__ __forceinline__ void dinv_rows_load(const double* __restrict__ Dk, int b, double (&dv)[4])
{
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m <= b) {
            const int e = threadIdx.x + 256 * m;
            dv[m] = __hip_atomic_load(Dk + (16 * b + e / (16 * (b + 1))) * NB + e % (16 * (b + 1)), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
        }
}

// panel solve of the tile held in R (LDS, [64][LDP]) against Dinv_col, block column by block column behind the
// diagonal workgroup's progress flag; B is scratch for the published rows.  The solved quarters leave for `dst` as
// they are formed.  When the diagonal block is further along than the row asked for (for most tiles it is long
// done), the remaining rows are fetched without asking again, each during the products of the one before.
// DIAG: see below, Q = [64][18] scratch.
template <bool DIAG>
__device__ __forceinline__ void flow_trsm(const double* R, double* B, const double* __restrict__ Dk, const int* dflag,
                                          int gen, int* info_h, int* s_val, d4 (&out)[4], double* __restrict__ dst, double* Q,
                                          d4 (&a1)[4], int wave, int g, int li)
{
    int have = 0;                          // block rows of Dinv_col known to be published
    double dv[4], dn[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (DIAG && b == 3) FSTAMP2(4);
        if (b >= have) {
            have = flow_wait_value(dflag, 8 * gen + b + 1, info_h, s_val) - 8 * gen;
            dinv_rows_load(Dk, b, dv);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) dv[m] = dn[m];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m <= b) {
                const int e = threadIdx.x + 256 * m;
                B[(16 * b + e / (16 * (b + 1))) * LDP + e % (16 * (b + 1))] = dv[m];
            }
        __syncthreads();
        if (DIAG && b == 3) FSTAMP2(5);
        if (b + 1 < 4 && b + 1 < have) dinv_rows_load(Dk, b + 1, dn);
        out[b] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < 16 * (b + 1); k0 += 4)
            out[b] = MFMA_F64(lds_ld(R + (16 * wave + li) * LDP + k0 + g), lds_ld(B + (16 * b + li) * LDP + k0 + g), out[b]);
        store_tile_quarter_sc1(dst, out[b], b);      // on its way while the next quarter waits
        if (DIAG) {
            // the owner of the next diagonal block: step lo of tile (i,i), a1 -= L_i,lo L_i,lo^T, follows the solve
            // quarter by quarter (k ascending per accumulator, like mma_tile_64: same bits) -- behind the last pivot
            // of the block above there is a quarter of a solve and a quarter of a product, not a whole one.  Only the
            // blocks on and below the diagonal: diag_block reads nothing above it.  Wave 0 -- which factors sub-block
            // (0,0) next -- needs its own rows of Q and nothing else: its block goes first, in front of the barrier.
#pragma unroll
            for (int r = 0; r < 4; ++r) Q[(16 * wave + g + 4 * r) * 18 + li] = out[b][r];
            if (wave == 0) syrk_quarter<1>(Q, a1, wave, g, li);
            __syncthreads();
            if (wave == 1) syrk_quarter<2>(Q, a1, wave, g, li);
            else if (wave == 2) syrk_quarter<3>(Q, a1, wave, g, li);
            else if (wave == 3) syrk_quarter<4>(Q, a1, wave, g, li);
            if (b < 3) __syncthreads();    // Q is rewritten by the next quarter (which may not wait any more)
        }
    }
    if (DIAG) FSTAMP2(6);
}

// one history step of a chunk: a0 -= L_i,k L_lo,k^T, a1 -= L_i,k L_hi,k^T (DIAG: L_hi,k is L_i,k).  In: tA = L_i,k and
// tB = L_lo,k (requested earlier); MORE: out, the same for step k + 1, requested during this step's products.
// `busy` (two workgroups per CU): this CU's count of workgroups inside a diagonal block.  A diagonal block's dependent MFMA
// chain runs a third slower beside a neighbour whose products keep the matrix pipes busy, and the whole call follows
// that chain -- so a workgroup about to start a step's products while its neighbour factors a diagonal block waits for
// it (14 us at most, on one CU per draw at a time; the word is read one step ahead, behind the products, so that
// asking costs nothing).
#define FLOW_YIELD_LIMIT 4096
// (the other places where an item that is not a diagonal item starts a stretch of work: thread 0 waits, the others meet
// it at the next barrier)
__device__ __forceinline__ void flow_yield(const int* busy)
{
    if (busy && threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 && ++spins < FLOW_YIELD_LIMIT)
            __builtin_amdgcn_s_sleep(16);
    }
}
template <bool DIAG, bool MORE>
__device__ __forceinline__ void flow_step(double* A, double* B, const double* pi, const double* pl, const double* ph,
                                          d2 (&tA)[8], d2 (&tB)[8], d4 (&a0)[4], d4 (&a1)[4], int wave, int g, int li,
                                          const int* busy, int& busy_seen)
{
    if (busy && threadIdx.x == 0 && busy_seen > 0) {
        int spins = 0;
        while (__hip_atomic_load(busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 && ++spins < FLOW_YIELD_LIMIT)
            __builtin_amdgcn_s_sleep(16);
    }
    planes_to_lds(tA, A, wave, g, li);
    planes_to_lds(tB, B, wave, g, li);
    __syncthreads();
    if (busy && threadIdx.x == 0) busy_seen = __hip_atomic_load(busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MORE) load_tile_sc1_p(pi + LEAN_TILE, tA);
    if (!DIAG) load_tile_sc1_p(ph, tB);
    else if (MORE) load_tile_sc1_p(pl + LEAN_TILE, tB);
    mma_tile_64(A, B, a0, wave, g, li, true);
    if (DIAG) {
        mma_tile_64(A, A, a1, wave, g, li, true);
    } else {
        __syncthreads();
        planes_to_lds(tB, B, wave, g, li);
        __syncthreads();
        if (MORE) load_tile_sc1_p(pl + LEAN_TILE, tB);
        mma_tile_64(A, B, a1, wave, g, li, true);
    }
    __syncthreads();                                                         // A and B are rewritten next step
}

// Tile (I, J) of K(X,X) + noise, built where it is consumed: the Gram term on the matrix pipe, the correlation function
// on the accumulators, in the accumulator layout of the tile storage -- k_cov's MODE 3 body for one 64x64 tile (same
// helpers, same order of the contraction: the same bits), so that the log-likelihood path has no covariance launch and
// the matrix never travels through memory before it is factored.  Xh / X2h: x / ls and 2 x / ls of the draw ([Np][Dp]),
// s1h: row norms.
struct FlowCov {
    const double* Xs; const double* X2s; const double* s1; const double* htab;   // null Xs: the tiles are in memory (k_cov ran)
    int N, Dp, kind;
    // `fused` (comp != null; the log-likelihood call as ONE launch, round 6): x / ls, the row norms and the right-hand-side
    // rows are formed where they are consumed -- no prologue launch -- and the last right-hand-side item of a draw reduces
    // -sum log diag L - 0.5 |y|^2 into pinned host memory -- no reduction launch
    const double* comp; const double* hyp; const double* vals;   // observations [N][D], raw hyper rows [nh][hs], values [N]
    int D, hs;
    double* lp_out; int* info_out;
};

template <int KIND>
__device__ __forceinline__ void flow_cov_tile_kind(const double* __restrict__ Xh, const double* __restrict__ X2h,
                                                   const double* __restrict__ s1h, double amp2, double noise, int N, int Dp,
                                                   int I, int J, d4 (&acc)[4])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int j0 = NB * I + 16 * wave, c0 = NB * J, Q = Dp >> 2;
    double s2v[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        s2v[nt] = s1h[c0 + 16 * nt + li];
        acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    const double* pa = Xh + (size_t)(j0 + li) * Dp + g * Q;
    const double* pb = X2h + (size_t)(c0 + li) * Dp + g * Q;
    // the fragments of up to 8 contraction steps are requested together (one round of memory latency per tile for
    // D <= 32), the products follow in k_cov's order
    for (int q0 = 0; q0 < Q; q0 += 8) {
        double af[8], bf[4][8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q0 + q < Q) {
                af[q] = pa[q0 + q];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bf[nt][q] = pb[(size_t)16 * nt * Dp + q0 + q];
            }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q0 + q < Q) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA_F64(af[q], bf[nt][q], acc[nt]);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + g + 4 * r;
        const double s1v = s1h[j];
        double gv[4], cv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) gv[nt] = acc[nt][r];
        corr_of_kind<KIND, 4>(gv, s1v, s2v, cv);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma clang fp contract(off)
            const int c = c0 + 16 * nt + li;
            const double eye = (j == c) ? 1.0 : 0.0;
            double v = amp2 * (cv[nt] + 1e-6 * eye) + noise * eye;
            if (j >= N || c >= N) v = eye;
            acc[nt][r] = v;
        }
    }
}

// fused form: block b of the observations scaled by the draw's length scales, in LDS -- dst[r][c] = factor * x[64 b + r][c]
// / ls[c] ([64][LDP], columns D .. Dp-1 and rows >= N zero) and nrm[r] = sum_c (x / ls)^2, value for value what
// k_scale_rows writes to Xs / X2s / s1 (true division, the norm accumulated left to right by one thread per row)
__device__ __forceinline__ void flow_scale_block(const FlowCov& cv, const double* __restrict__ lsh, int b, double* dst,
                                                 double* nrm, double factor)
{
#pragma clang fp contract(off)
    const int D = cv.D, Dp = cv.Dp, N = cv.N;
    for (int e = threadIdx.x; e < NB * Dp; e += 256) {
        const int r = e / Dp, c = e - r * Dp;
        const int row = NB * b + r;
        dst[r * LDP + c] = (row < N && c < D) ? cv.comp[(size_t)row * D + c] / lsh[c] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        double acc = 0.0;
        double* d = dst + threadIdx.x * LDP;
        for (int c = 0; c < D; ++c) {
            const double v = d[c];
            acc = acc + v * v;
            d[c] = factor * v;
        }
        nrm[threadIdx.x] = acc;
    }
    __syncthreads();
}

// flow_cov_tile_kind with both operands in LDS (Ar: rows of block I scaled, Bc: rows of block J scaled and doubled; nr / nc
// their norms): the same contraction order, the same epilogue -- the same bits
template <int KIND>
__device__ __forceinline__ void flow_cov_tile_lds(const double* Ar, const double* Bc, const double* nr, const double* nc,
                                                  double amp2, double noise, int N, int Dp, int I, int J, d4 (&acc)[4])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int j0 = NB * I + 16 * wave, c0 = NB * J, Q = Dp >> 2;
    double s2v[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        s2v[nt] = nc[16 * nt + li];
        acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    const double* pa = Ar + (16 * wave + li) * LDP + g * Q;
    const double* pb = Bc + li * LDP + g * Q;
    for (int q0 = 0; q0 < Q; q0 += 8) {
        double af[8], bf[4][8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q0 + q < Q) {
                af[q] = pa[q0 + q];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bf[nt][q] = pb[16 * nt * LDP + q0 + q];
            }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q0 + q < Q) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA_F64(af[q], bf[nt][q], acc[nt]);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + g + 4 * r;
        const double s1v = nr[16 * wave + g + 4 * r];
        double gv[4], cv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) gv[nt] = acc[nt][r];
        corr_of_kind<KIND, 4>(gv, s1v, s2v, cv);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma clang fp contract(off)
            const int c = c0 + 16 * nt + li;
            const double eye = (j == c) ? 1.0 : 0.0;
            double v = amp2 * (cv[nt] + 1e-6 * eye) + noise * eye;
            if (j >= N || c >= N) v = eye;
            acc[nt][r] = v;
        }
    }
}

__device__ __forceinline__ void flow_cov_tile_fused(const FlowCov& cv, int h, const double* Ar, const double* Bc, const double* nr,
                                                    const double* nc, int I, int J, d4 (&acc)[4])
{
    const double noise = cv.htab[h * SPX_HT + 1], amp2 = cv.htab[h * SPX_HT + 2];
    if (cv.kind == SPX_COV_MATERN32) flow_cov_tile_lds<SPX_COV_MATERN32>(Ar, Bc, nr, nc, amp2, noise, cv.N, cv.Dp, I, J, acc);
    else if (cv.kind == SPX_COV_ARDSE) flow_cov_tile_lds<SPX_COV_ARDSE>(Ar, Bc, nr, nc, amp2, noise, cv.N, cv.Dp, I, J, acc);
    else flow_cov_tile_lds<SPX_COV_MATERN52>(Ar, Bc, nr, nc, amp2, noise, cv.N, cv.Dp, I, J, acc);
}

// fused form: tile J of the right-hand-side block row -- row 0 = vals - mean (0 for pad entries), rows 1 .. 63 zero --
// in the accumulator layout (k_lean_rhs_init's values)
__device__ __forceinline__ void flow_rhs_tile(const FlowCov& cv, int h, int J, d4 (&acc)[4])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const double mean = cv.htab[h * SPX_HT + 0];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = J * NB + 16 * nt + li;
        acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
        if (wave == 0 && g == 0 && col < cv.N) acc[nt][0] = cv.vals[col] - mean;
    }
}

// fused form, the last right-hand-side item of draw h (every diagonal block and every other right-hand-side tile of the draw
// is published: it has waited for all of them): lp = -sum log diag(L) - 0.5 |y|^2 in k_lean_logprob's summation order, or
// -inf if not positive definite, into pinned host memory -- and the draw's not-PD flag goes back to zero for the next call
__device__ __forceinline__ void flow_logprob(const FlowCov& cv, int h, int Np, const double* __restrict__ diagL,
                                             const double* __restrict__ rhs_h, int* info_h, double* red)
{
    double sl = 0.0, sq = 0.0;
    for (int i = threadIdx.x; i < cv.N; i += 256) {
        sl += log(__hip_atomic_load(diagL + (size_t)h * Np + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const double gi = __hip_atomic_load(rhs_h + (size_t)(i >> 6) * LEAN_TILE + ((((i & 63) >> 4) * 2) * 256 + (i & 15)) * 2,
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sq += gi * gi;
    }
    red[threadIdx.x] = sl;
    red[256 + threadIdx.x] = sq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[threadIdx.x] += red[threadIdx.x + s];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int bad = __hip_atomic_load(info_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cv.lp_out[h] = bad ? -__builtin_inf() : (-red[0] - 0.5 * red[256]);
        // (the flag is what the host polls -- option lean_poll: it leaves behind the value, at system scope)
        __hip_atomic_store(cv.info_out + h, bad, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(info_h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void flow_cov_tile(const FlowCov& cv, int h, int Np, int I, int J, d4 (&acc)[4])
{
    const double* Xh = cv.Xs + (size_t)h * Np * cv.Dp;
    const double* X2h = cv.X2s + (size_t)h * Np * cv.Dp;
    const double* s1h = cv.s1 + (size_t)h * Np;
    const double noise = cv.htab[h * SPX_HT + 1], amp2 = cv.htab[h * SPX_HT + 2];
    if (cv.kind == SPX_COV_MATERN32) flow_cov_tile_kind<SPX_COV_MATERN32>(Xh, X2h, s1h, amp2, noise, cv.N, cv.Dp, I, J, acc);
    else if (cv.kind == SPX_COV_ARDSE) flow_cov_tile_kind<SPX_COV_ARDSE>(Xh, X2h, s1h, amp2, noise, cv.N, cv.Dp, I, J, acc);
    else flow_cov_tile_kind<SPX_COV_MATERN52>(Xh, X2h, s1h, amp2, noise, cv.N, cv.Dp, I, J, acc);
}

// the tiles (i, lo), (i, hi) of one work item (lo = hi - 1; lo < 0: tile (i, 0) alone); DIAG: (i, hi) is the diagonal tile
template <bool DIAG>
__device__ __forceinline__ void flow_chunk(double* A, double* B, double* T16, double* __restrict__ row, double* __restrict__ Lh,
                                           double* __restrict__ Dh, const int* lf, int* lf_row, int* df, int* info_h,
                                           double* __restrict__ diag_out, int i, int lo, int hi, int nblk, int gen,
                                           const FlowCov& cov, int h, bool is_rhs, int* busy, const double* lp_diag = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const bool two = lo >= 0;
    __shared__ int s_n, s_val;
    d4 a0[4], a1[4], st[4];
#ifdef FLOW_STAMPS
    if (threadIdx.x == 0) s_fstamp = (DIAG && h < 32) ? g_flow_stamps + ((size_t)h * 64 + i) * 8 : nullptr;
    __syncthreads();
#endif
    if (DIAG) FSTAMP(h, i, 0);
    if (!DIAG) { flow_yield(busy); __syncthreads(); }
    if (cov.comp && !is_rhs) {
        // fused: the operands of the two Gram tiles in LDS (A: rows of block i; B: doubled rows of block lo, then of block hi)
        const double* lsh = cov.hyp + (size_t)h * cov.hs + 3;
        flow_scale_block(cov, lsh, i, A, T16, 1.0);
        if (two) {
            flow_scale_block(cov, lsh, lo, B, T16 + NB, 2.0);
            flow_cov_tile_fused(cov, h, A, B, T16, T16 + NB, i, lo, a0);
            __syncthreads();
        }
        flow_scale_block(cov, lsh, hi, B, T16 + NB, 2.0);
        flow_cov_tile_fused(cov, h, A, B, T16, T16 + NB, i, hi, a1);
        __syncthreads();                                                     // A, B and T16 are free again
    } else if (cov.comp) {
        if (two) flow_rhs_tile(cov, h, lo, a0);
        flow_rhs_tile(cov, h, hi, a1);
    } else if (cov.Xs && !is_rhs) {
        if (two) flow_cov_tile(cov, h, nblk * NB, i, lo, a0);
        flow_cov_tile(cov, h, nblk * NB, i, hi, a1);
    } else {
        if (two) load_tile(row + (size_t)lo * LEAN_TILE, a0);
        load_tile(row + (size_t)hi * LEAN_TILE, a1);
    }
    // ---- 1. history: steps k < lo (a chunk with history has two tiles) ----
    // Step k needs L_ik (an earlier chunk of this row), L_lo,k and -- unless (i,hi) is the diagonal tile, whose second
    // operand is L_ik again -- L_hi,k (rows above).  One wave looks at the flags of the next FLOW_BATCH steps at once (a
    // lane per flag) and the workgroup then takes every step that is ready without asking again, the tiles of step
    // k + 1 in flight during the products of step k: far behind the diagonal workgroups -- where most of the work is
    // -- a step costs its two products; next to them, one look and one round of loads instead of three of each.
    const int first = two ? lo : 0;
    int busy_seen = 0;
#pragma unroll 1
    for (int k = 0; k < first;) {
        if (wave == 0) {
            const int sl = lane / 3, w = lane - 3 * sl, kk = k + sl;
            const bool live = sl < FLOW_BATCH && kk < first && !(w == 2 && DIAG);
            const int* fp = (w == 0) ? lf_row + kk : lf + (size_t)(w == 1 ? lo : hi) * nblk + kk;
            int n = 0, spins = 0;
            for (;;) {
                const int f = live ? __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
                const unsigned long long late = ~__ballot(f >= 8 * gen);
                n = late ? (__ffsll((long long)late) - 1) / 3 : FLOW_BATCH;     // steps whose flags are all up, from k on
                if (n > 0) break;
                if (++spins >= s_flow_spin_limit) {
                    if (lane == 0) __hip_atomic_store(info_h, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    n = 1;                 // gave up: go on (the call fails with info < 0)
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if ((spins & 1023) == 0 && __hip_atomic_load(info_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) { n = 1; break; }
            }
            if (lane == 0) s_n = min(n, min(FLOW_BATCH, first - k));
        }
        __syncthreads();
        const int n = s_n;
        // two tiles in flight, each requested one product before it is needed
        d2 tA[8], tB[8];
        const double* pi = row + (size_t)k * LEAN_TILE;                      // L_i,k   L_lo,k   L_hi,k
        const double* pl = Lh + ((size_t)lo * nblk + k) * LEAN_TILE;
        const double* ph = Lh + ((size_t)hi * nblk + k) * LEAN_TILE;
        load_tile_sc1_p(pi, tA);
        load_tile_sc1_p(pl, tB);
#pragma unroll 1
        for (int s = 0; s + 1 < n; ++s, pi += LEAN_TILE, pl += LEAN_TILE, ph += LEAN_TILE)
            flow_step<DIAG, true>(A, B, pi, pl, ph, tA, tB, a0, a1, wave, g, li, busy, busy_seen);
        flow_step<DIAG, false>(A, B, pi, pl, ph, tA, tB, a0, a1, wave, g, li, busy, busy_seen);
        k += n;
    }
    // ---- 2. tile (i, lo): always a panel tile (lo < hi <= i) ----
    // (from here to the end of its diagonal block this item is the next link of the draw's chain: the neighbour yields)
    if (DIAG) FSTAMP(h, i, 1);
    if (DIAG && busy && threadIdx.x == 0) atomicAdd(busy, 1);
    if (!DIAG) flow_yield(busy);
    if (two) {
        acc_tile_to_lds(a0, A, wave, g, li);
        flow_trsm<DIAG>(A, B, Dh + (size_t)lo * NB * NB, df + lo, gen, info_h, &s_val, st, row + (size_t)lo * LEAN_TILE, T16, a1, wave, g, li);
        if (!DIAG) {
            // step lo of tile (i, hi): with the tile just solved as the row operand
            __syncthreads();                                                 // every wave is done with R
            acc_tile_to_lds(st, A, wave, g, li);
            flow_wait(lf + (size_t)hi * nblk + lo, 8 * gen, info_h);             // L_hi,lo (row hi's own diagonal chunk)
            d4 tb[4];
            load_tile_sc1(Lh + ((size_t)hi * nblk + lo) * LEAN_TILE, tb);
            acc_tile_to_lds(tb, B, wave, g, li);
            __syncthreads();
            mma_tile_64(A, B, a1, wave, g, li, true);
        }
        drain_stores();                                                      // L_i,lo is out (written through) ...
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(lf_row + lo, 8 * gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... and says so
    }
    // ---- 3. tile (i, hi) ----
    if (!DIAG) flow_yield(busy);
    if (!DIAG) {
        acc_tile_to_lds(a1, A, wave, g, li);
        __syncthreads();
    } else if (wave != 0) {
        // (wave 0 starts on sub-block (0,0) from its registers -- diag_block, direct; every reader of A's and T16's earlier
        // contents is behind a barrier of flow_trsm's)
        acc_tile_to_lds(a1, A, wave, g, li);
    }
    if (DIAG) FSTAMP(h, i, 2);
    if (DIAG) FCLK(h, i, 0);
    if (DIAG) {
        // (the EI path keeps L_ii -- row-major in its tile's place -- for spx_get_factor; the log-likelihood path only its diagonal)
        diag_block<true>(A, B, T16, info_h, i * NB, diag_out ? nullptr : row + (size_t)i * LEAN_TILE, NB, Dh + (size_t)i * NB * NB, diag_out,
                         df + i, 8 * gen, true, &a1[0]);
        if (busy && threadIdx.x == 0) atomicAdd(busy, -1);
        FSTAMP(h, i, 3);
        FCLK(h, i, 1);
    } else {
        flow_trsm<false>(A, B, Dh + (size_t)hi * NB * NB, df + hi, gen, info_h, &s_val, st, row + (size_t)hi * LEAN_TILE, T16, a1, wave, g, li);
        drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(lf_row + hi, 8 * gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // fused: the right-hand-side item of the last column pair is the last link of the draw -- it reduces the log-likelihood
        if (is_rhs && cov.lp_out && hi == nblk - 1) flow_logprob(cov, h, nblk * NB, lp_diag, row, info_h, A);
    }
}

__global__ __launch_bounds__(256, 2) void k_lean_flow(double* __restrict__ Lt, double* __restrict__ Dinv,
                                                   int* __restrict__ info, double* __restrict__ rhs,
                                                   double* __restrict__ diagL, int* __restrict__ lflags,
                                                   int* __restrict__ dflags, unsigned* __restrict__ tickets,
                                                   int Np, int nh, int gen, FlowCov cov, int* __restrict__ cu_busy,
                                                   int spin_limit)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* A = smem;              // [64][LDP]
    double* B = smem + NB * LDP;   // [64][LDP]
    double* T16 = B + NB * LDP;    // [4][16][18]
    // Work is handed out by TICKET, not by blockIdx: a workgroup that holds ticket t is running, and every ticket
    // below t was taken by a workgroup that is running or done -- the order the deadlock argument above needs, by
    // construction rather than by the dispatcher's habits.  Draws fastest: the diagonal workgroups of all draws first.
    // (tickets[0]: the counter; tickets[1]: workgroups that are done -- the last one to leave puts both back to zero for
    // the next launch, so the host keeps no count that a failed launch could put out of step)
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) { s_ticket = atomicAdd(tickets, 1u); s_flow_spin_limit = spin_limit; }
    __syncthreads();
    const int h = (int)(s_ticket % (unsigned)nh);
    const int nblk = Np / NB;
    // ticket / nh -> work item, COLUMN-major: by last column hi ascending, block row ascending within a column.  Block row
    // i (the right-hand-side rows: i = nblk, tiles 0 .. nblk-1) is cut into pairs of columns from its right end, so its
    // items have hi = last, last - 2, ...: the items of column hi are the rows i >= hi of hi's parity (and the
    // right-hand-side rows when nblk - 1 has it).  An item's history is as long as its column index and is wanted when
    // the diagonal workgroups get there: in this order the workgroups that hold a place on the chip are the ones whose
    // columns come next -- they run through their history at full speed and leave -- where a row-major order fills the
    // chip with the right-hand ends of a few rows, each waiting a block column's time for every step.
    int hi = 0, i;
    {
        int c = (int)(s_ticket / (unsigned)nh);
        for (;; ++hi) {
            const int nrow = (nblk - 1 - hi) / 2 + 1;                        // rows hi, hi + 2, ... <= nblk - 1
            const int cnt = nrow + ((rhs && ((nblk - 1 - hi) & 1) == 0) ? 1 : 0);   // + the right-hand-side rows (if any)
            if (c < cnt) { i = (c < nrow) ? hi + 2 * c : nblk; break; }
            c -= cnt;
        }
    }
    const bool is_rhs = (i == nblk);
    const int lo = hi - 1;                                               // this item: columns hi - 1 (if >= 0) and hi
    double* Lh = Lt + (size_t)h * Np * Np;
    double* row = is_rhs ? rhs + (size_t)h * nblk * LEAN_TILE : Lh + (size_t)i * nblk * LEAN_TILE;   // tiles (i, .)
    int* info_h = info + h;
    const int* lf = lflags + (size_t)h * (nblk + 1) * nblk;     // [row][col]
    int* lf_row = lflags + ((size_t)h * (nblk + 1) + i) * nblk;
    int* df = dflags + (size_t)h * nblk;
    double* Dh = Dinv + (size_t)h * nblk * NB * NB;

    // this CU's word of cu_busy (null: one workgroup per CU, nobody to yield to): XCC_ID[3:0] and HW_ID[15:8] = SE, SH, CU
    int* busy = nullptr;
    if (cu_busy) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        const unsigned cu = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);
        busy = cu_busy + ((xcc & 15u) << 8 | (cu & 255u));
    }
    const bool diag = !is_rhs && hi == i;
    // the two kinds of chunk as two straight-line bodies (one body with the distinction inside costs the register
    // allocator 110 registers more than either)
    if (diag) flow_chunk<true>(A, B, T16, row, Lh, Dh, lf, lf_row, df, info_h, diagL ? diagL + (size_t)h * Np + (size_t)i * NB : nullptr, i, lo, hi, nblk, gen, cov, h, is_rhs, busy);
    else flow_chunk<false>(A, B, T16, row, Lh, Dh, lf, lf_row, df, info_h, nullptr, i, lo, hi, nblk, gen, cov, h, is_rhs, busy, diagL);
    if (threadIdx.x == 0 && atomicAdd(tickets + 1, 1u) == gridDim.x - 1) {   // everybody else has left (and long since drawn a ticket)
        __hip_atomic_store(tickets + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

void launch_lean_flow(hipStream_t s, double* Lt, double* Dinv, int* info, double* rhs, double* diagL, int* lflags,
                      int* dflags, unsigned* tickets, int Np, int nh, int gen, bool alone,
                      const double* Xs, const double* X2s, const double* s1, const double* htab, int N, int Dp, int kind,
                      int* cu_busy, int spin_limit, const FlowFused* fused)
{
    FlowCov cov{Xs, X2s, s1, htab, N, Dp, kind, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr};
    if (fused) {
        cov.comp = fused->comp; cov.hyp = fused->hyp; cov.vals = fused->vals; cov.D = fused->D; cov.hs = fused->hs;
        cov.lp_out = fused->lp_out; cov.info_out = fused->info_out;
    }
    const int nblk = Np / NB;
    int ny = rhs ? (nblk + 1) / 2 : 0;                         // the right-hand-side rows (the EI path has none)
    for (int i = 0; i < nblk; ++i) ny += (i + 2) / 2;
    // 74.5 KB: two workgroups per CU.  `alone`: ask for more than half of a CU's 160 KB, so that every workgroup has its
    // CU to itself -- a diagonal block's dependent MFMA chain runs a third slower beside a neighbour whose products
    // keep the matrix pipes busy (each of its MFMAs then waits for one of theirs to drain), and as long as the chip is
    // not short of workgroups the whole call follows the diagonal blocks (spx_api.hip decides)
    size_t lds = (size_t)(2 * NB * LDP + DIAG_T16_DOUBLES) * sizeof(double);
    if (alone) lds = 96 * 1024;
    // the attribute is per function and device, not per launch: always the larger figure, so that two handles (threads)
    // with different batch sizes cannot lower it under each other's launch.  A failure shows as a launch error
    // (checked by the caller's hipGetLastError after the launch).
    SPX_LDS_ATTR(k_lean_flow, 96 * 1024);
    hipLaunchKernelGGL(k_lean_flow, dim3(nh * ny), dim3(256), lds, s, Lt, Dinv, info, rhs, diagL, lflags, dflags, tickets,
                       Np, nh, gen, cov, alone ? nullptr : cu_busy, spin_limit > 0 ? spin_limit : FLOW_SPIN_LIMIT);
}

// k_lean_step2 (even k >= 2): the steps k-2 and k-1 for every remaining tile right of block column k (which needs
// only step k-1: see col_only in k_lean_step) in ONE pass -- the accumulator tiles
// of a workgroup's chunk stay in registers across both steps, so the trailing matrix is read and written once
// per two block columns instead of once per column (its traffic is what bounds the update beyond a few draws).
// Same order of steps per tile, same MFMA chain: bit-identical factor.  The workgroup of the diagonal tile goes
// on to factor it, as in k_lean_step.
__global__ __launch_bounds__(256, 2) void k_lean_step2(double* __restrict__ Lt, double* __restrict__ Dinv,
                                                    int* __restrict__ info, double* __restrict__ rhs,
                                                    double* __restrict__ diagL, int Np, int k)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* A = smem;              // [64][LDP]  row operand of the current step; then S
    double* B = smem + NB * LDP;   // [64][LDP]  column operand; then XT
    double* T16 = B + NB * LDP;    // [4][16][18]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x;
    const int nblk = Np / NB;
    const bool is_rhs = rhs && blockIdx.y == gridDim.y - 1;
    const int i = k + blockIdx.y;
    const int j0 = k + blockIdx.z * LEAN_CH;
    const int j1 = min(j0 + LEAN_CH, is_rhs ? nblk : i + 1);
    if (j0 >= j1) return;
    const int nc = j1 - j0;
    double* Lh = Lt + (size_t)h * Np * Np;
    double* row = is_rhs ? rhs + (size_t)h * nblk * LEAN_TILE : Lh + (size_t)i * nblk * LEAN_TILE;
    d4 acc[LEAN_CH][4], ta[4], tb[4];
#pragma unroll
    for (int c = 0; c < LEAN_CH; ++c)
        if (c < nc) load_tile(row + (size_t)(j0 + c) * LEAN_TILE, acc[c]);
    for (int pass = 0; pass < 2; ++pass) {
        const int p = k - 2 + pass;
        if (pass == 0 && j0 == k && nc == 1) continue;   // (uniform) a chunk that is block column k alone: step k-1 only
        load_tile(row + (size_t)p * LEAN_TILE, ta);
        load_tile(Lh + ((size_t)j0 * nblk + p) * LEAN_TILE, tb);
        acc_tile_to_lds(ta, A, wave, g, li);      // (the barrier that ended the previous pass freed A and B)
#pragma unroll
        for (int c = 0; c < LEAN_CH; ++c)
            if (c < nc) {
                acc_tile_to_lds(tb, B, wave, g, li);
                __syncthreads();
                if (c + 1 < nc) load_tile(Lh + ((size_t)(j0 + c + 1) * nblk + p) * LEAN_TILE, tb);
                // block column k already took step k-2 in the previous (odd) launch
                if (!(pass == 0 && j0 + c == k)) mma_tile_64(A, B, acc[c], wave, g, li, true);
                __syncthreads();
            }
    }
    if (!is_rhs && i == k && j0 == k) {
        acc_tile_to_lds(acc[0], A, wave, g, li);
        __syncthreads();
        diag_block(A, B, T16, info + h, k * NB, nullptr, 0, Dinv + ((size_t)h * nblk + k) * NB * NB,
                   diagL + (size_t)h * Np + (size_t)k * NB);
        return;
    }
#pragma unroll
    for (int c = 0; c < LEAN_CH; ++c)
        if (c < nc) store_tile(row + (size_t)(j0 + c) * LEAN_TILE, acc[c]);
}

// lazy = 1: updates are applied two steps at a time (k_lean_step2 at even k >= 2; at odd k only block column k is
// brought up to date); lazy = 0: every launch applies one step to every remaining tile.
void launch_lean_step(hipStream_t s, double* Lt, double* Dinv, int* info, double* rhs, double* diagL, int Np, int k,
                      int nh, int lazy)
{
    const int n = Np / NB - k;
    if (n <= 0) return;
    const size_t lds = (size_t)(2 * NB * LDP + DIAG_T16_DOUBLES) * sizeof(double);   // 74.5 KB: two workgroups per CU
    if (lazy && k >= 2 && (k & 1) == 0) {
        SPX_LDS_ATTR(k_lean_step2, lds);
        hipLaunchKernelGGL(k_lean_step2, dim3(nh, n + (rhs ? 1 : 0), (n + LEAN_CH - 1) / LEAN_CH), dim3(256), lds, s, Lt,
                           Dinv, info, rhs, diagL, Np, k);
        return;
    }
    const int col_only = (lazy && (k & 1)) ? 1 : 0;
    SPX_LDS_ATTR(k_lean_step, lds);
    const dim3 grid = (k == 0) ? dim3(nh, 1, 1)
                               : dim3(nh, n + (rhs ? 1 : 0), col_only ? 1 : (n + LEAN_CH - 1) / LEAN_CH);
    hipLaunchKernelGGL(k_lean_step, grid, dim3(256), lds, s, Lt, Dinv, info, (k == 0) ? nullptr : rhs, diagL, Np, k,
                       col_only);
}

// L_ik = A_ik L_kk^-T for the tiles below the diagonal block of column k and for the right-hand-side
// rows (last workgroup): one 64x64x64 tile product, in place.  This launch sits on the critical path of
// every block column; the product is spread over SIXTEEN wavefronts (1024 threads, a 16x16 output block and
// 16 MFMAs each): waves 0-3 bring the tile in, waves 4-7 the inverse of the diagonal block, and the
// operand fragments of a wave are a quarter as many LDS reads.  (It does NOT shorten the matrix-pipe time: a CU has
// four matrix pipes, one per SIMD, and a 64x64x64 fp64 product is 256 v_mfma_f64_16x16x4 of 64 cycles each =
// 1.7 us of a CU however many waves issue them -- measured in round 3 with the corner workgroup of a
// two-columns-per-launch variant, scripts/dev/attic/pair_columns_kernels.hip.txt.)
__global__ __launch_bounds__(1024) void k_lean_trsm(double* __restrict__ Lt, const double* __restrict__ Dinv,
                                                    double* __restrict__ rhs, int Np, int k)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* A = smem;              // [64][LDP]
    double* B = smem + NB * LDP;   // [64][LDP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int nblk = Np / NB;
    const bool is_rhs = blockIdx.x == gridDim.x - 1;
    const int i = k + 1 + blockIdx.x;
    double* tile = is_rhs ? rhs + ((size_t)h * nblk + k) * LEAN_TILE
                          : Lt + (size_t)h * Np * Np + ((size_t)i * nblk + k) * LEAN_TILE;
    if (wave < 4) {
        d4 t[4];
        load_tile(tile, t);        // threadIdx.x < 256: the tile's own thread slots
        acc_tile_to_lds(t, A, wave, g, li);
    } else if (wave < 8) {
        const double* Dk = Dinv + ((size_t)h * nblk + k) * NB * NB;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = (threadIdx.x - 256) + 256 * q;  // double2 units
            const int row = idx >> 5, c2 = idx & 31;
            *reinterpret_cast<d2*>(B + row * LDP + 2 * c2) = *reinterpret_cast<const d2*>(Dk + (size_t)row * NB + 2 * c2);
        }
    }
    __syncthreads();
    // wave (wr, wc): out[16 wr + .][16 wc + .] = sum_q A[16 wr + .][q] Dinv[16 wc + .][q]
    const int wr = wave >> 2, wc = wave & 3;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < NB; k0 += 4) {
        const double a = lds_ld(A + (16 * wr + li) * LDP + k0 + g);
        const double bv = lds_ld(B + (16 * wc + li) * LDP + k0 + g);
        acc = MFMA_F64(a, bv, acc);
    }
    // element (16 wr + g + 4 r, 16 wc + li) = value q = 4 wc + r of tile thread t = 64 wr + lane
    d2* p = reinterpret_cast<d2*>(tile) + (64 * wr + lane);
    p[(2 * wc) * 256] = (d2){acc[0], acc[1]};
    p[(2 * wc + 1) * 256] = (d2){acc[2], acc[3]};
}

void launch_lean_trsm(hipStream_t s, double* Lt, const double* Dinv, double* rhs, int Np, int k, int nh)
{
    const int nrows = Np / NB - k - 1;
    const size_t lds = (size_t)(2 * NB * LDP) * sizeof(double);   // 67.6 KB
    SPX_LDS_ATTR(k_lean_trsm, lds);
    hipLaunchKernelGGL(k_lean_trsm, dim3(nrows + 1, nh), dim3(1024), lds, s, Lt, Dinv, rhs, Np, k);
}

// the right-hand-side block row in tile storage: row 0 = vals - mean (0 for pad entries), rows 1..63 = 0
// (also zeroes the draw's not-PD flag and, for k_lean_step_ps, its hand-off flags: two memsets fewer per call)
__global__ __launch_bounds__(256) void k_lean_rhs_init(const double* __restrict__ vals,
                                                       const double* __restrict__ htab,
                                                       double* __restrict__ rhs, int N, int Np,
                                                       int* __restrict__ info, int* __restrict__ flags)
{
    const int h = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;      // over [nblk][4096]
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) info[h] = 0;
        if (flags && threadIdx.x < Np / NB) flags[h * (Np / NB) + threadIdx.x] = 0;
    }
    if (idx >= NB * Np) return;
    const int J = idx >> 12, e = idx & 4095;
    const int t = (e >> 1) & 255, q = ((e >> 9) << 1) | (e & 1);   // thread slot, value q = nt * 4 + r
    const int wave = t >> 6, lane = t & 63, r = q & 3, nt = q >> 2;
    const int rowi = 16 * wave + (lane >> 4) + 4 * r, col = J * NB + 16 * nt + (lane & 15);
    double v = 0.0;
    if (rowi == 0 && col < N) v = vals[col] - htab[h * SPX_HT + 0];
    rhs[(size_t)h * NB * Np + idx] = v;
}

void launch_lean_rhs_init(hipStream_t s, const double* vals, const double* htab, double* rhs, int N, int Np, int nh,
                          int* info, int* flags)
{
    hipLaunchKernelGGL(k_lean_rhs_init, dim3((NB * Np + 255) / 256, nh), dim3(256), 0, s, vals, htab, rhs, N, Np, info, flags);
}

// lp = -sum log diag(L) - 0.5 |y|^2 (GPEIChooser.py:284) from the diagonal the diag blocks left in diagL
// and y = row 0 of the right-hand-side tiles; -inf if not PD
__global__ __launch_bounds__(256) void k_lean_logprob(const double* __restrict__ diagL,
                                                      const double* __restrict__ rhs,
                                                      const int* __restrict__ info,
                                                      double* __restrict__ out, int* __restrict__ info_out, int N, int Np)
{
    __shared__ double red[2][256];
    const int h = blockIdx.x;
    double sl = 0.0, sq = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        sl += log(diagL[(size_t)h * Np + i]);
        // element (row 0, column i): tile i / 64, thread t = i & 15 (wave 0, g 0), value q = 4 nt, nt = (i & 63) >> 4
        const double gi = rhs[(size_t)h * NB * Np + (size_t)(i >> 6) * LEAN_TILE + ((((i & 63) >> 4) * 2) * 256 + (i & 15)) * 2];
        sq += gi * gi;
    }
    red[0][threadIdx.x] = sl;
    red[1][threadIdx.x] = sq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {     // (out / info_out may be pinned host memory: one store each)
        const int bad = info[h];
        out[h] = bad ? -__builtin_inf() : (-red[0][0] - 0.5 * red[1][0]);
        if (info_out) info_out[h] = bad;
    }
}

void launch_lean_logprob(hipStream_t s, const double* diagL, const double* rhs, const int* info, double* out, int* info_out,
                         int N, int Np, int nh)
{
    hipLaunchKernelGGL(k_lean_logprob, dim3(nh), dim3(256), 0, s, diagL, rhs, info, out, info_out, N, Np);
}

// ---------------------------------------------------------------------------
// W = L^-1, block column jb per workgroup, output transposed: WT[j][i] = W[i][j].
//   W_jj = Dinv_j ;  W_ij = -Dinv_i * sum_{p=j}^{i-1} L_ip W_pj     (i > j)
// computed as transposes so both stores and loads are row-contiguous:
//   Tt[n][i'] = sum_p sum_q WT[j0+n][p0+q] L[i0+i'][p0+q]
//   WT[j0+n][i0+i''] = - sum_{i'} Tt[n][i'] Dinv_i[i''][i']
// The zeros of WT left of the diagonal block (entries the predict GEMM reads inside its 128-wide row blocks) are written by
// this kernel itself, by the workgroup that owns the rows: the caller does NOT clear the buffer.
// ---------------------------------------------------------------------------
// TILED: L in the tile-major storage of the log-likelihood path (k_lean_flow factored it): tile (i, p) at
// (i nblk + p) * 4096 doubles, accumulator order inside
template <bool TILED>
__global__ __launch_bounds__(256, 2) void k_trinv(const double* __restrict__ Lm,
                                               const double* __restrict__ Dinv,
                                               double* __restrict__ WT, int Np, int nh)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* A = smem;                  // [2][64][LDP]
    double* B = smem + 2 * NB * LDP;   // [2][64][LDP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int nblk = Np / NB;
    // Block column jb costs (nblk-jb)(nblk-jb-1)/2 tile steps.  Workgroups are numbered column-major
    // over (draw, column): all draws' column 0 first, then column 1, ... -- longest first for the
    // dispatcher, and consecutive ids (which land on consecutive XCDs) carry equal work.  With a
    // (column, draw) grid the id modulo 8 was the column modulo 8, so one XCD got every draw's
    // heaviest column: 4.7 ms instead of 1.5 ms at N=2048, H=20.
    const int jb = blockIdx.x / nh;
    const int h = blockIdx.x - jb * nh;
    const double* Lh = Lm + (size_t)h * Np * Np;
    double* Wh = WT + (size_t)h * Np * Np;
    const double* Dh = Dinv + (size_t)h * nblk * NB * NB;
    const size_t j0 = (size_t)jb * NB;

    // diagonal block: WT[j0+n][j0+i] = Dinv_j[i][n]
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
        const int n = idx >> 6, i = idx & 63;
        Wh[(j0 + n) * Np + j0 + i] = Dh[(size_t)jb * NB * NB + i * NB + n];
    }
    // W is lower triangular: WT[j0 + n][i] = 0 for i < j0 -- written here, by the workgroup that owns these rows, instead
    // of a memset of the whole [nh][Np][Np] buffer before the launch (twice the bytes, and one more stream operation)
    for (int idx = threadIdx.x; idx < NB * (int)(j0 / 2); idx += 256) {
        const int n = idx / (int)(j0 / 2), i2 = idx - n * (int)(j0 / 2);
        *reinterpret_cast<d2*>(Wh + (j0 + n) * Np + 2 * i2) = (d2){0.0, 0.0};
    }
    for (int ib = jb + 1; ib < nblk; ++ib) {
        const size_t i0 = (size_t)ib * NB;
        d4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
        __syncthreads();  // orders the previous iteration's WT stores (and LDS reads) before what follows
        {
            TileRegs ta, tb;
            d2 tp[8];
            const double* Lrow = Lh + (size_t)ib * nblk * LEAN_TILE;          // TILED: the tiles (ib, .)
            tile_load(Wh + j0 * Np + (size_t)jb * NB, Np, ta);
            if (TILED) load_tile_p(Lrow + (size_t)jb * LEAN_TILE, tp);
            else tile_load(Lh + i0 * Np + (size_t)jb * NB, Np, tb);
            for (int p = jb; p < ib; ++p) {
                double* Ac = A + ((p - jb) & 1) * NB * LDP;
                double* Bc = B + ((p - jb) & 1) * NB * LDP;
                tile_store(ta, Ac);
                if (TILED) planes_to_lds(tp, Bc, wave, g, li);
                else tile_store(tb, Bc);
                __syncthreads();
                if (p + 1 < ib) {
                    tile_load(Wh + j0 * Np + (size_t)(p + 1) * NB, Np, ta);
                    if (TILED) load_tile_p(Lrow + (size_t)(p + 1) * LEAN_TILE, tp);
                    else tile_load(Lh + i0 * Np + (size_t)(p + 1) * NB, Np, tb);
                }
                mma_tile_64(Ac, Bc, acc, wave, g, li, false);
            }
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(16 * wave + g + 4 * r) * LDP + 16 * nt + li] = acc[nt][r];
        tile_to_lds(Dh + (size_t)ib * NB * NB, NB, B);
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
        mma_tile_64(A, B, acc, wave, g, li, true);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Wh[(j0 + 16 * wave + g + 4 * r) * Np + i0 + 16 * nt + li] = acc[nt][r];
        __threadfence_block();
    }
}

void launch_trinv(hipStream_t s, const double* L, const double* Dinv, double* WT, int Np, int nh, bool tiled)
{
    const size_t lds = (size_t)(4 * NB * LDP) * sizeof(double);
    if (tiled) {
        SPX_LDS_ATTR((k_trinv<true>), lds);
        hipLaunchKernelGGL(k_trinv<true>, dim3((Np / NB) * nh), dim3(256), lds, s, L, Dinv, WT, Np, nh);
    } else {
        SPX_LDS_ATTR((k_trinv<false>), lds);
        hipLaunchKernelGGL(k_trinv<false>, dim3((Np / NB) * nh), dim3(256), lds, s, L, Dinv, WT, Np, nh);
    }
}

// ---------------------------------------------------------------------------
// gamma = W (vals - mean)   (one thread per row i, coalesced over i)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gamma(const double* __restrict__ WT, size_t wt_stride,
                                               const double* __restrict__ vals, size_t vals_stride,
                                               const double* __restrict__ htab, int htab_stride,
                                               double* __restrict__ gamma, int N, int Np)
{
    // htab == nullptr: plain product W * rhs (no mean subtraction)
    // batch entry b = blockIdx.y: its own right-hand side, and (strides permitting) its own W / mean
    __shared__ double r[256];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double* Wh = WT + (size_t)b * wt_stride;
    const double mean = htab ? htab[b * htab_stride + 0] : 0.0;
    const double* vh = vals + (size_t)b * vals_stride;
    double acc = 0.0;
    const int jmax = blockIdx.x * 256 + 255;  // rows this block needs: j <= i
    for (int jb = 0; jb <= jmax; jb += 256) {
        __syncthreads();
        const int jj = jb + threadIdx.x;
        r[threadIdx.x] = (jj < N) ? (vh[jj] - mean) : 0.0;
        __syncthreads();
        const int jn = (i < Np) ? min(256, i - jb + 1) : 0;
        // the sum runs in the order j = 0, 1, 2, ... (one fma each: these bits are part of every EI value), but its loads
        // need not: 16 of them are in flight at a time -- with the compiler's 4 the launch was one memory round trip per
        // four rows (11 us at N = 128, 21 us at N = 256, one launch per factorisation)
        const double* wp = Wh + (size_t)jb * Np + i;
        int t = 0;
        for (; t + 16 <= jn; t += 16) {
            double w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = wp[(size_t)(t + u) * Np];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = fma(w[u], r[t + u], acc);
        }
        for (; t < jn; ++t) acc = fma(wp[(size_t)t * Np], r[t], acc);
    }
    if (i < Np) gamma[(size_t)b * Np + i] = acc;
}

void launch_gamma(hipStream_t s, const double* WT, const double* vals, const double* htab,
                  double* gamma, int N, int Np, int nh)
{
    // one right-hand side (vals) shared by nh draws, each with its own W and mean
    hipLaunchKernelGGL(k_gamma, dim3(Np / 256 + (Np % 256 != 0), nh), dim3(256), 0, s, WT,
                       (size_t)Np * Np, vals, (size_t)0, htab, SPX_HT, gamma, N, Np);
}

// t_h = W_h rhs_h for nh draws, rhs [nh][Np] (pad rows must be zero)
void launch_gemv_lower(hipStream_t s, const double* WT, const double* rhs, double* out, int Np, int nh)
{
    hipLaunchKernelGGL(k_gamma, dim3(Np / 256 + (Np % 256 != 0), nh), dim3(256), 0, s, WT,
                       (size_t)Np * Np, rhs, (size_t)Np, (const double*)nullptr, 0, out, Np, Np);
}

// S right-hand sides (fantasy columns, [S][n] contiguous) against ONE draw's W and mean
void launch_gamma_multi(hipStream_t s, const double* WT_h, const double* rhs, const double* htab_h,
                        double* gamma, int N, int Np, int S)
{
    hipLaunchKernelGGL(k_gamma, dim3(Np / 256 + (Np % 256 != 0), S), dim3(256), 0, s, WT_h,
                       (size_t)0, rhs, (size_t)N, htab_h, 0, gamma, N, Np);
}

// alpha = W^T gamma  == K^-1 (vals - mean);  one wavefront per row j of WT
__global__ __launch_bounds__(256) void k_alpha(const double* __restrict__ WT,
                                               const double* __restrict__ gamma,
                                               double* __restrict__ alpha, int Np)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y;
    const int j = blockIdx.x * 4 + wave;
    const double* row = WT + ((size_t)h * Np + j) * Np;
    const double* gh = gamma + (size_t)h * Np;
    double acc = 0.0;
    for (int i = (j & ~63) + lane; i < Np; i += 64)
        if (i >= j) acc += row[i] * gh[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) alpha[(size_t)h * Np + j] = acc;
}

void launch_alpha(hipStream_t s, const double* WT, const double* gamma, double* alpha, int Np, int nh)
{
    hipLaunchKernelGGL(k_alpha, dim3(Np / 4, nh), dim3(256), 0, s, WT, gamma, alpha, Np);
}

// rhs[h][0][j] = vals[j] - mean_h (0 for pad entries), rows 1..63 = 0: the right-hand-side row
// block that k_chol_panel carries through the factorisation (log-likelihood path)
__global__ __launch_bounds__(256) void k_rhs_init(const double* __restrict__ vals,
                                                  const double* __restrict__ htab,
                                                  double* __restrict__ rhs, int N, int Np)
{
    const int h = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;      // over [64][Np]
    if (idx >= NB * Np) return;
    const int row = idx / Np, j = idx - row * Np;
    double v = 0.0;
    if (row == 0 && j < N) v = vals[j] - htab[h * SPX_HT + 0];
    rhs[(size_t)h * NB * Np + idx] = v;
}

void launch_rhs_init(hipStream_t s, const double* vals, const double* htab, double* rhs, int N, int Np, int nh)
{
    hipLaunchKernelGGL(k_rhs_init, dim3((NB * Np + 255) / 256, nh), dim3(256), 0, s, vals, htab, rhs, N, Np);
}

// lp = -sum log diag(L) - 0.5 |gamma|^2   (GPEIChooser.py:284); -inf if not PD
__global__ __launch_bounds__(256) void k_logprob(const double* __restrict__ Lm,
                                                 const double* __restrict__ gamma,
                                                 const int* __restrict__ info,
                                                 double* __restrict__ out, int N, int Np, size_t gstride)
{
    __shared__ double red[2][256];
    const int h = blockIdx.x;
    double sl = 0.0, sq = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        sl += log(Lm[((size_t)h * Np + i) * Np + i]);
        const double gi = gamma[(size_t)h * gstride + i];
        sq += gi * gi;
    }
    red[0][threadIdx.x] = sl;
    red[1][threadIdx.x] = sq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[h] = info[h] ? -__builtin_inf() : (-red[0][0] - 0.5 * red[1][0]);
}

void launch_logprob(hipStream_t s, const double* L, const double* gamma, size_t gstride, const int* info,
                    double* out, int Np, int nh)
{
    // N is recovered on the host side: pad rows have L_ii = 1 (log 0) and gamma = 0
    hipLaunchKernelGGL(k_logprob, dim3(nh), dim3(256), 0, s, L, gamma, info, out, Np, Np, gstride);
}
