// Internal definitions shared by spx_api.hip (single-GPU engine + C ABI) and spx_multi.hip
// (several GPUs behind one handle).  Not installed; include/spx.h is the public surface.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/spx.h"
#include "common.h"

std::string& spx_err_slot();            // thread-local last-error text
int spx_fail(int code, const char* fmt, ...);
#define fail spx_fail

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipGetLastError(); /* reported now: the runtime's sticky copy is cleared */   \
            return fail(SPX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                  \
        }                                                                                     \
    } while (0)

// after a batch of launches: a noted hipFuncSetAttribute refusal first (common.h: SPX_LDS_ATTR), then the runtime's own error
std::string& spx_attr_err_slot();
#define LAUNCHCHK()                                                             \
    do {                                                                        \
        if (!spx_attr_err_slot().empty()) {                                     \
            std::string m_ = spx_attr_err_slot();                               \
            spx_attr_err_slot().clear();                                        \
            (void)hipGetLastError();                                            \
            return fail(SPX_ERR_HIP, "%s", m_.c_str());                         \
        }                                                                       \
        HIPCHK(hipGetLastError());                                              \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool owned = true;      // false: p points into another DevBuf (alias_of): never freed here
    void alias_of(void* q) { if (p && owned) (void)hipFree(p); p = q; cap = 0; owned = false; }
    int reserve(size_t bytes)
    {
        if (!owned) { p = nullptr; cap = 0; owned = true; }
        if (bytes <= cap) return SPX_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();   // reported here; do not leave it for the next call's launch check to find
            return fail(SPX_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        }
        cap = bytes;
        return SPX_OK;
    }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; cap = 0; owned = true; }
    double* d() const { return (double*)p; }
};

// pinned (page-locked, device-visible) host memory: uploads from it are true asynchronous DMA without a staging copy
// inside the runtime, and a kernel can write a handful of result words straight into it
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SPX_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        if (bytes < 4096) bytes = 4096;
        // coherent (fine-grained) whatever HIP_HOST_COHERENT says: kernels read hyper rows out of these buffers and store results
        // and completion flags into them that the host watches while the launch is still running (spx_gp_logprob)
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocCoherent);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            return fail(SPX_ERR_HIP, "hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        }
        cap = bytes;
        return SPX_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

enum Stage {
    ST_SCALE = 0, ST_COV_SELF, ST_CHOL_DIAG, ST_CHOL_PANEL, ST_TRINV, ST_GAMMA_ALPHA,
    ST_COV_CROSS, ST_CROSS_MEAN, ST_PREDICT_GEMM, ST_EI_FINALIZE, ST_MEAN_ARGMAX,
    ST_FACTOR_TOTAL, ST_EI_RUN_TOTAL, ST_COUNT
};
struct spx_handle {
    int device = 0;
    bool inited = false;
    hipStream_t stream = nullptr;    // main stream (also the only one the factorization uses)
    hipStream_t stream2 = nullptr;   // optional producer stream (option "streams" = 2): K(X*,X) of the next
                                     // work item is generated (VALU) while the GEMM of the current one runs (MFMA)
    hipEvent_t ev_sync[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;   // whole-stage timers (factor / ei_run)
    hipEvent_t ev_fac = nullptr;                   // spx_ei_step: the whole factorisation (alpha included) is done (stream)
    hipEvent_t ev_obs = nullptr, ev_p0 = nullptr;  // spx_ei_step: observations scaled (stream) / first K(X*,X) ready (stream2)

    int64_t N = 0, M = 0, index_base = 0;
    int D = 0, Dp = 0, Np = 0, H = 0;
    bool have_obs = false, have_cand = false, have_hyp = false, have_time = false;
    bool factored = false, ran = false, ran_moments = false;
    bool ran_time = false;           // the last pass was SPX_FLAG_TIME_ONLY: predicted durations are valid, EI results are not
    int nmodels = 1;  // 1 = objective GP only, 2 = + log-duration GP
    double best = 0.0;
    int not_pd_draw = -1, not_pd_pivot = -1;
    int64_t kst_budget = 512ll << 20;   // K(X*,X) staging buffer per stream (bytes)
    int nstreams = 1;
    int cov_kind = 0;                   // SPX_COVAR_* (option "covar"); SE = ARDSE kernels on unit length scales
    int lean_lazy = -1;                 // option "lean_lazy": 0 / 1 / -1 = by batch size
    int gemm_variant = 0;               // predict-GEMM variant of THIS handle (option "gemm_waves"); 0 = production
    int64_t fant_budget = 0;            // bytes the per-fantasy partial means may take (an eighth of free memory, <= 2 GB) ...
    int fant_budget_S = -1;             // ... as found when the number of fantasies last changed
    int lean_one = -1;                  // option "lean_one": the log-likelihood call as ONE launch (scaling, right-hand side and the
                                        // reduction inside k_lean_flow) 1 / 0 / -1 = default (on)
    int lean_poll = -1;                 // option "lean_poll": the fused call's results are awaited by polling their pinned flags
                                        // (1, default) instead of hipStreamSynchronize (0)
    int lean_zc = -1;                   // option "lean_zc": the fused launch reads the hyper rows from the pinned staging buffer
                                        // itself (1, default) instead of behind a host-to-device copy (0)
    double* fused_lp = nullptr;         // (spx_gp_logprob -> do_factor: pinned destinations of the fused form's results)
    int* fused_info = nullptr;
    bool fused_ran = false;             // the last do_factor took the fused form
    const void* info_clean_ptr = nullptr;   // the not-PD flags at this address ...
    size_t info_clean_bytes = 0;            // ... in a buffer of this size are all zero (left so by the fused launch)
    int cov_flat = -1;                  // option "cov_flat": k_cov_flat for multi-round K(X*,X) launches 1 / 0 / -1 = default (on)
    int gemm_partial = -1;              // option "gemm_partial": skip the padding of N in the EI pass 1 / 0 / -1 = default (on)
    bool last_skip_pad = false;         // the last EI pass did
    struct spx_multi* multi = nullptr;  // non-null: this handle fronts several per-GPU handles (spx_multi.hip)
    struct spx_comm* comm = nullptr;    // non-null: one-process-per-GPU communicator attached (spx_comm_attach)

    std::vector<double> hyp_host, thyp_host;
    std::vector<double> up_raw, up_tab;     // host staging of the per-call hyper upload (do_factor)

    DevBuf comp, vals, ldur, cand, hyp, htab;
    DevBuf Xs, X2s, s1, Lm, WT, Dinv, gamma, alpha, info, lp;
    DevBuf Cs[2], s2[2], Kst[2], part_ss[2], part_bg[2], time_m[2], ei_draw, ei_mean, mom_m, mom_v, mom_t;
    DevBuf am_val, am_idx, am_out_val, am_out_idx, scratch;
    // pending-experiment fantasies (spx_set_fantasies): S right-hand sides per draw
    int S = 0;
    DevBuf fantT, gammaS, bests, part_bgS[2];
    DevBuf alphaS;                 // [H][S][Np] = W^T Gamma_s, built on the first spx_ei_grad_batch after spx_set_fantasies
    bool alphaS_valid = false;
    DevBuf pt_x, pt_k, pt_dk, pt_t, pt_z, pt_out, pt_kt, pt_dkt, pt_u;   // spx_ei_grad_batch work vectors
    DevBuf rec_send, rec_recv, rec_out;   // {best mean EI, global index} records of the multi-GPU all-gather
    // 2-D partition with a communicator attached (spx_set_partition): the M_total-vector of EI sums / means
    int part_ph = 1;
    int64_t part_M = 0;            // > 0: the collective is the all-reduce(SUM) of EI sums
    int part_H = 0;
    bool ran_2d = false;
    DevBuf ei_sum_full;
    DevBuf sobol_dirs, sobol_out;                                   // spx_sobol_grid
    DevBuf rhs;                                                     // spx_gp_logprob: [H][64][Np] right-hand-side rows
    DevBuf diagL;                                                   // spx_gp_logprob (tile-major path): diag(L), [H][Np]
    int lean_np = 0;                                                // padded size of the last lean factorisation (a multiple of 64, not of 128)
    bool lean_tiled = false;                                        // the last lean factorisation used tile-major storage
    int lean_ps = -1;                                               // option "lean_ps": 0 / 1 / -1 = default (on)
    int lean_merge = -1;                                            // option "lean_merge": scaling + right-hand side as one launch (k_lean_prologue) 1 / 0 / -1 = default (on)
    DevBuf ps_flags;                                                // k_lean_step_ps: progress of every diagonal block, [H][nblk]
    int lean_flow = -1;                                             // option "lean_flow": whole factorisation in one launch (k_lean_flow)
    int ei_fused = -1;                                              // option "ei_fused": N <= 128 without fantasies: the EI pass of a chunk as ONE kernel (k_ei_fused128) 1 / 0 / -1 = default (on)
    int64_t flow_fallbacks = 0;                                     // k_lean_flow hand-off time-outs that sent this handle back to one launch per block column
    bool flow_demoted = false;                                      // ... and it is there now (until flow_rearm_after clean factorisations, or option lean_flow)
    int flow_clean = 0;                                             // clean factorisations since the last time-out
    int flow_rearm_after = 16;                                      // option "flow_rearm_after" (0 = never)
    int64_t flow_rearms = 0;                                        // times the handle went back to k_lean_flow
    int flow_spin_limit = 0;                                        // option "flow_spin_limit": polls before a hand-off gives up (0 = the kernel's default)
    int ranks_seen = 1;                                             // records in the last all-gather's table (spx_comm_exchange)
    bool last_fused = false;                                        // the last EI pass used k_ei_fused128
    int step_overlap = -1;                                          // option "step_overlap": spx_ei_step starts the candidate side beside the factorisation 1 / 0 / -1 = default (on)
    int n_cu = 256;                                                 // compute units of the device (ensure_init)
    int ei_flow = -1;                                               // option "ei_flow": spx_factor through k_lean_flow 1 / 0 / -1 = default (on)
    bool factor_tiled = false;                                      // the EI path's factor is tile-major (k_lean_flow made it)
    int lean_flow_cov = -1;                                         // option "lean_flow_cov": K(X,X) built inside k_lean_flow 1 / 0 / -1 = default (on)
    int lean_flow_yield = -1;                                       // option "lean_flow_yield": 1 / 0 / -1 = default (on)
    int lean_flow_cu = -1;                                          // option "lean_flow_cu": one workgroup per CU 1 / 0 / -1 = by size
    DevBuf flow_flags;                                              // k_lean_flow: [H][nblk + 1][nblk] tile flags + [H][nblk] diagonal progress + the ticket and done counters
    size_t flow_flags_n = 0;                                        // ints the flags were zeroed for
    int flow_gen = 0;                                               // generation of the last call (flags are compared, not cleared)
    PinBuf pin_stage;                                               // staging of the callers' small host buffers (stage_h2d / stage_d2h in spx_api.hip)
    size_t stage_off = 0;
    int stage_copies = -1;              // option "stage_copies"
    PinBuf pin_up, pin_res;                                         // hyper-parameter upload staging; log-likelihood results
    bool handoff_timeout = false;                                   // finish_factor saw info < 0
    bool flow_used = false;                                         // the last factorisation ran k_lean_flow

    double best_val = 0.0;
    int64_t best_idx = -1;

    // timing
    bool timing = false;
    struct Ev { hipEvent_t a, b; int stage; };
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    double st_ms[ST_COUNT] = {0};
    int64_t st_n[ST_COUNT] = {0};
};


// ---- several GPUs behind one handle (spx_multi.hip) --------------------------------------------
struct spx_multi;
void spx_multi_destroy(spx_multi* m);
int spx_multi_set_option(spx_multi* m, const char* name, int64_t value);
int spx_multi_set_observations(spx_multi* m, const double* comp, const double* vals, int64_t N, int32_t D);
int spx_multi_set_candidates(spx_multi* m, const double* cand, int64_t M, int32_t D, int64_t index_base);
int spx_multi_set_hypers(spx_multi* m, const double* hypers, int32_t H);
int spx_multi_set_time_model(spx_multi* m, const double* log_durs, const double* time_hypers);
int spx_multi_factor(spx_multi* m);
int spx_multi_set_fantasies(spx_multi* m, const double* fant, const double* bests, int32_t S);
int spx_multi_ei_run(spx_multi* m, int32_t flags);
int spx_multi_get_best(spx_multi* m, int64_t* best_idx, double* best_val);
int spx_multi_get_ei_mean(spx_multi* m, double* out);
int spx_multi_get_ei_draws(spx_multi* m, double* out);
int spx_multi_get_moments(spx_multi* m, int32_t draw, double* func_m, double* func_v);
int spx_multi_get_time_mean(spx_multi* m, int32_t draw, double* out);
int spx_multi_get_factor(spx_multi* m, int32_t draw, double* K, double* L, double* alpha);
int spx_multi_get_factor_rows(spx_multi* m, int32_t draw, int64_t row0, int64_t nrows, double* L_rows, double* gamma);
int spx_multi_get_cross_cov(spx_multi* m, int32_t draw, int64_t c0, int64_t nc, double* out);
int spx_multi_gp_logprob(spx_multi* m, double* out);
int spx_multi_ei_grad_batch(spx_multi* m, const double* points, int32_t P, double* neg_ei, double* grad);
int spx_multi_sobol_grid(spx_multi* m, const uint32_t* dirs, int32_t dim_max, int32_t dim, int64_t n,
                         int64_t skip, double* grid_out, int32_t as_candidates, double* kernel_ms);
int spx_multi_not_pd_info(spx_multi* m, int32_t* draw, int32_t* pivot);
int spx_multi_get_timings(spx_multi* m, double* ms, int64_t* launches, int n);
int spx_multi_stat(spx_multi* m, const char* name, int64_t* value);
int spx_multi_info(spx_multi* m, int32_t* n_dev, int32_t* transport, int32_t* device_ids, int32_t cap);
// one process per GPU (spx_comm_attach): exchange this handle's record with the other ranks / drop the communicator
int spx_comm_exchange(spx_handle* h);
void spx_comm_release(spx_handle* h);
// single-GPU pieces the multi layer needs
int spx_ensure_init(spx_handle* h);
