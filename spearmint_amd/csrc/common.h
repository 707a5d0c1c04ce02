// Shared device/host definitions for libspx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) * B(4x16) + C.
//   A: lane l supplies A[i = l & 15][k = l >> 4]
//   B: lane l supplies B[k = l >> 4][n = l & 15]
//   C/D: reg r of lane l is element [row = (l >> 4) + 4 r][col = l & 15]
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// geometry shared by host and device code
#define SPX_NB 64        // Cholesky / inverse block size
#define SPX_BM 128       // predict GEMM: rows (observations) per workgroup tile
#define SPX_BN 128       // predict GEMM: candidates per workgroup tile
#define SPX_BK 16        // predict GEMM: contraction depth per LDS stage
#define SPX_PADN 128     // observations are padded to a multiple of this

// correlation function of the GP (gp.py:87-132; spx.h SPX_COVAR_*).  SE is ARDSE with unit length scales
// (gp.py:88): the API layer substitutes the length scales, the kernels see ARDSE.
#define SPX_COV_MATERN52 0
#define SPX_COV_MATERN32 1
#define SPX_COV_ARDSE 2

// device hyper table row: [mean, noise, amp2, amp2*(1+1e-6)]
#define SPX_HT 4

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// The launchers below return nothing: a kernel that needs more dynamic LDS than the default asks for it right before its
// launch, and a refusal is NOTED (per host thread, spx_api.hip) and reported by the API's next launch check (LAUNCHCHK) as
// SPX_ERR_HIP naming the kernel -- not left to surface as an opaque launch failure.
void spx_note_attr_error(const char* kernel, size_t lds_bytes, hipError_t e);
#define SPX_LDS_ATTR(fn, bytes)                                                                                            \
    do {                                                                                                                   \
        hipError_t ae_ = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             (int)(bytes));                                                                \
        if (ae_ != hipSuccess) spx_note_attr_error(#fn, (size_t)(bytes), ae_);                                             \
    } while (0)

// ---- launchers (defined in the *.hip translation units) ---------------------
// cov_kernels.hip
void launch_scale_rows(hipStream_t s, const double* x, int64_t n, int64_t n_pad, int D, int Dp,
                       const double* ls /*[nh][ls_stride]*/, int ls_stride, int nh, double factor,
                       double* xs /*[nh][n_pad][Dp]*/, double* sumsq /*[nh][n_pad]*/,
                       double* xs2 = nullptr /*optional: 2 * xs, same launch*/,
                       int* zero_ints = nullptr /*optional: n_zero ints cleared by the same launch*/, int n_zero = 0);
// the log-likelihood path's prologue in ONE launch: launch_scale_rows (factor 1, with the doubled copy) + launch_lean_rhs_init
void launch_lean_prologue(hipStream_t s, const double* x, int64_t n, int64_t n_pad, int D, int Dp, const double* ls,
                          int ls_stride, int nh, double* xs, double* sumsq, double* xs2, const double* vals,
                          const double* htab, double* rhs, int* info, int* flags);
void launch_cov_self(hipStream_t s, const double* Xs, const double* s1, const double* X2s,
                     const double* htab, double* K, int N, int Np, int Dp, int nh, bool tiled = false,
                     int kind = SPX_COV_MATERN52);
void launch_cov_cross(hipStream_t s, const double* Xs, const double* s1, const double* Cs,
                      const double* s2, const double* htab, double* Kst, int N, int Np, int Mc,
                      int Dp, int nh, int kind = SPX_COV_MATERN52, int live_rows = 0 /*> 0: rows from here on are NOT written*/,
                      bool allow_flat = true /*false: always the 3-D grid (option cov_flat)*/);
void launch_cross_mean(hipStream_t s, const double* Xs, const double* s1, const double* Cs,
                       const double* s2, const double* htab, const double* alpha, double* out,
                       int N, int Np, int Mc, int Dp, int nh, int kind = SPX_COV_MATERN52);

// chol_kernels.hip
void launch_chol_diag(hipStream_t s, double* L, double* Dinv, int* info, int Np, int k, int nh, int updated = 0);
void launch_chol_panel(hipStream_t s, double* L, const double* Dinv, int Np, int k, int nh, double* rhs = nullptr);
// log-likelihood path, tile-major storage (chol_kernels.hip)
void launch_lean_step(hipStream_t s, double* Lt, double* Dinv, int* info, double* rhs, double* diagL, int Np, int k,
                      int nh, int lazy);
void launch_lean_trsm(hipStream_t s, double* Lt, const double* Dinv, double* rhs, int Np, int k, int nh);
void launch_lean_step_ps(hipStream_t s, double* Lt, double* Dinv, int* info, double* rhs, double* diagL, int* flags,
                         int Np, int k, int nh);
// (k_lean_flow's `fused` form -- the log-likelihood call as one launch: where the raw inputs and the pinned outputs are)
struct FlowFused {
    const double* comp; const double* hyp; const double* vals; int D, hs;
    double* lp_out; int* info_out;
};
void launch_lean_flow(hipStream_t s, double* Lt, double* Dinv, int* info, double* rhs, double* diagL, int* lflags,
                      int* dflags, unsigned* tickets, int Np, int nh, int gen, bool alone,
                      const double* Xs, const double* X2s, const double* s1, const double* htab, int N, int Dp, int kind,
                      int* cu_busy, int spin_limit = 0, const FlowFused* fused = nullptr);
void launch_lean_rhs_init(hipStream_t s, const double* vals, const double* htab, double* rhs, int N, int Np, int nh,
                          int* info, int* flags);
void launch_lean_logprob(hipStream_t s, const double* diagL, const double* rhs, const int* info, double* out, int* info_out,
                         int N, int Np, int nh);
void launch_trinv(hipStream_t s, const double* L, const double* Dinv, double* WT, int Np, int nh, bool tiled = false);
void launch_gamma(hipStream_t s, const double* WT, const double* vals, const double* htab,
                  double* gamma, int N, int Np, int nh);
void launch_gamma_multi(hipStream_t s, const double* WT_h, const double* rhs, const double* htab_h,
                        double* gamma, int N, int Np, int S);
void launch_gemv_lower(hipStream_t s, const double* WT, const double* rhs, double* out, int Np, int nh);
void launch_alpha(hipStream_t s, const double* WT, const double* gamma, double* alpha, int Np, int nh);
void launch_rhs_init(hipStream_t s, const double* vals, const double* htab, double* rhs, int N, int Np, int nh);
void launch_logprob(hipStream_t s, const double* L, const double* gamma, size_t gstride, const int* info,
                    double* out, int Np, int nh);

// refine_kernels.hip
#define SPX_REFINE_PB 8   // points (right-hand sides) that share one pass over W in the refinement kernels
void launch_point_cov(hipStream_t s, const double* Xs, const double* s1, const double* hyp,
                      const double* htab, const double* x, double* kvec, double* dkdr2, int N, int Np,
                      int D, int Dp, int nh, int P, int kind = SPX_COV_MATERN52);
void launch_trimv_multi(hipStream_t s, const double* WT, const double* rhs, double* out, int Np, int nh, int P);
void launch_trimvT_multi(hipStream_t s, const double* WT, const double* rhs, double* out, int Np, int nh, int P);
void launch_point_finish(hipStream_t s, const double* Xs, const double* hyp, const double* htab,
                         const double* alpha, const double* kvec, const double* dkdr2,
                         const double* tvec, const double* zvec, const double* x, double best,
                         double* out, int N, int Np, int D, int Dp, int nh, int P, const double* kt,
                         const double* dkt, int S, const double* gammaS, const double* alphaS,
                         const double* bests, double* uvec);

// fused_kernels.hip: the whole EI pass of a chunk for N <= 128 in one launch (no K* / beta in memory)
void launch_ei_fused128(hipStream_t s, int kind, const double* WT, const double* gamma, const double* Xs, const double* s1,
                        const double* Cs, const double* s2, const double* htab, const double* time_m, double best,
                        double* ei_draw, double* mom_m, double* mom_v, int N, int Mc, int Dp, int nh, int64_t c0,
                        int64_t M, int64_t Mp, int n_cu);

// sobol_kernels.hip
void launch_sobol_grid(hipStream_t s, const uint32_t* dirs, int dim, int64_t n, int64_t skip, double* out);

// predict_kernels.hip
void launch_predict_gemm(hipStream_t s, int variant, const double* WT, const double* Kst, const double* gamma,
                         double* part_ss, double* part_bg, int Np, int Mc, int nh, int part_nh, int part_h0,
                         const double* gammaS = nullptr, int S = 0, double* part_bgS = nullptr, int nlive = 0);
bool predict_gemm_variant_ok(int v);
int predict_gemm_padding_plan(int variant, int N, int Np);
void launch_ei_finalize_fant(hipStream_t s, const double* part_ss, const double* part_bgS,
                             const double* htab, const double* bests, const double* time_m,
                             double* ei_draw, int nrb, int Mc, int nh, int S, int64_t c0, int64_t M,
                             int64_t Mp, int h0, double* ei_s);
void launch_ei_finalize(hipStream_t s, const double* part_ss, const double* part_bg,
                        const double* htab, const double* time_m, double best, double* ei_draw,
                        double* mom_m, double* mom_v, int nrb, int Mc, int nh, int64_t c0,
                        int64_t M, int64_t Mp, int h0);
void launch_mean_over_draws(hipStream_t s, const double* ei_draw, double* ei_mean, int64_t M,
                            int64_t Mp, int H);
void launch_sum_over_draws(hipStream_t s, const double* ei_draw, double* out, int64_t M, int64_t Mp, int H);
void launch_div_scalar(hipStream_t s, double* v, int64_t n, double denom);
void launch_argmax(hipStream_t s, const double* v, int64_t M, double* blk_val, int64_t* blk_idx,
                   double* out_val, int64_t* out_idx, double* host_mirror = nullptr, const int* info = nullptr, int n_info = 0);
void launch_mean_argmax(hipStream_t s, const double* ei_draw, double* ei_mean, int64_t M, int64_t Mp, int H, double* blk_val,
                        int64_t* blk_idx, double* out_val, int64_t* out_idx, double* host_mirror = nullptr,
                        const int* info = nullptr, int n_info = 0);
int argmax_blocks(int64_t M);
