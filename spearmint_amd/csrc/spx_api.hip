// libspx host side: handle, device buffers, kernel orchestration, C ABI (include/spx.h).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include <chrono>

#include "spx_internal.h"
#include "np_sum.h"

#define SPX_VERSION 200

static const char* kStageNames[ST_COUNT] = {
    "scale_rows", "cov_self", "chol_diag", "chol_panel", "trinv", "gamma_alpha",
    "cov_cross", "cross_mean", "predict_gemm", "ei_finalize", "mean_argmax",
    "factor_total", "ei_run_total"};


std::string& spx_err_slot()
{
    static thread_local std::string g_err;
    return g_err;
}

std::string& spx_attr_err_slot()
{
    static thread_local std::string g_attr;
    return g_attr;
}

void spx_note_attr_error(const char* kernel, size_t lds_bytes, hipError_t e)
{
    char buf[384];
    snprintf(buf, sizeof buf, "hipFuncSetAttribute(%s, MaxDynamicSharedMemorySize = %zu) failed: %s", kernel, lds_bytes,
             hipGetErrorString(e));
    (void)hipGetLastError();
    if (spx_attr_err_slot().empty()) spx_attr_err_slot() = buf;   // the first one is the cause
}

int spx_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    spx_err_slot() = buf;
    // a hipFuncSetAttribute refusal noted during THIS call (SPX_LDS_ATTR) and not yet reported by a LAUNCHCHK must not
    // outlive the call: the next launch check on this thread, for another handle, would name a kernel it never launched
    if (!spx_attr_err_slot().empty()) {
        spx_err_slot() += " [also: " + spx_attr_err_slot() + "]";
        spx_attr_err_slot().clear();
    }
    return code;
}

static int ensure_init(spx_handle* h)
{
    HIPCHK(hipSetDevice(h->device));
    if (!h->inited) {
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        for (int i = 0; i < 6; ++i) HIPCHK(hipEventCreateWithFlags(&h->ev_sync[i], hipEventDisableTiming));
        HIPCHK(hipEventCreate(&h->ev_t0));
        HIPCHK(hipEventCreate(&h->ev_t1));
        HIPCHK(hipEventCreateWithFlags(&h->ev_obs, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_p0, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_fac, hipEventDisableTiming));
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && ncu > 0) h->n_cu = ncu;
        h->inited = true;
    }
    return SPX_OK;
}

static int ev_begin(spx_handle* h, int stage, hipStream_t strm)
{
    if (!h->timing) return -1;
    if (h->ev_used == h->ev_pool.size()) {
        spx_handle::Ev e;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return -1;
        h->ev_pool.push_back(e);
    }
    spx_handle::Ev& e = h->ev_pool[h->ev_used];
    e.stage = stage;
    (void)hipEventRecord(e.a, strm);
    return (int)h->ev_used++;
}
static void ev_end(spx_handle* h, int id, hipStream_t strm)
{
    if (id >= 0) (void)hipEventRecord(h->ev_pool[id].b, strm);
}
static void ev_collect(spx_handle* h)
{
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b) == hipSuccess) {
            h->st_ms[h->ev_pool[i].stage] += ms;
            h->st_n[h->ev_pool[i].stage] += 1;
        }
    }
    h->ev_used = 0;
}
#define TIMED_S(stage, strm, stmt)              \
    do {                                        \
        int ev_ = ev_begin(h, stage, strm);     \
        stmt;                                   \
        ev_end(h, ev_, strm);                   \
    } while (0)
#define TIMED(stage, stmt) TIMED_S(stage, h->stream, stmt)

int spx_ensure_init(spx_handle* h) { return ensure_init(h); }

static int padded_dim(int D)
{
    if (D <= 4) return 4;
    if (D <= 8) return 8;
    if (D <= 16) return 16;
    return (int)round_up(D, 32);
}

// kernel-level correlation kind of a handle: SE runs as ARDSE on unit length scales (do_factor substitutes them)
static inline int dev_kind(const spx_handle* h) { return h->cov_kind == SPX_COVAR_SE ? SPX_COV_ARDSE : h->cov_kind; }

extern "C" {

int spx_version(void) { return SPX_VERSION; }
const char* spx_last_error(void) { return spx_err_slot().c_str(); }

int spx_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(SPX_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int spx_create(int device_id, spx_handle** out)
{
    if (!out || device_id < 0) return fail(SPX_ERR_ARG, "spx_create: bad arguments");
    spx_handle* h = new spx_handle();
    h->device = device_id;
    *out = h;
    return SPX_OK;
}

void spx_destroy(spx_handle* h)
{
    if (!h) return;
    if (h->multi) { spx_multi_destroy(h->multi); delete h; return; }
    spx_comm_release(h);
    if (h->inited) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->stream2);
        DevBuf* bufs[] = {&h->comp, &h->vals, &h->ldur, &h->cand, &h->hyp, &h->htab, &h->Xs, &h->X2s,
                          &h->s1, &h->Lm, &h->WT, &h->Dinv, &h->gamma, &h->alpha, &h->info, &h->lp,
                          &h->Cs[0], &h->s2[0], &h->Kst[0], &h->part_ss[0], &h->part_bg[0], &h->time_m[0],
                          &h->Cs[1], &h->s2[1], &h->Kst[1], &h->part_ss[1], &h->part_bg[1], &h->time_m[1],
                          &h->fantT, &h->gammaS, &h->bests, &h->part_bgS[0], &h->part_bgS[1],
                          &h->pt_x, &h->pt_k, &h->pt_dk, &h->pt_t, &h->pt_z, &h->pt_out, &h->pt_kt, &h->pt_dkt,
                          &h->ei_draw, &h->ei_mean, &h->mom_m, &h->mom_v, &h->mom_t, &h->am_val, &h->am_idx,
                          &h->am_out_val, &h->am_out_idx, &h->scratch, &h->sobol_dirs, &h->sobol_out, &h->rhs, &h->diagL, &h->ps_flags, &h->flow_flags,
                          &h->alphaS, &h->pt_u, &h->rec_send, &h->rec_recv, &h->rec_out, &h->ei_sum_full};
        for (DevBuf* b : bufs) b->release();
        h->pin_up.release();
        h->pin_res.release();
        h->pin_stage.release();
        for (auto& e : h->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (int i = 0; i < 6; ++i) (void)hipEventDestroy(h->ev_sync[i]);
        (void)hipEventDestroy(h->ev_t0);
        (void)hipEventDestroy(h->ev_t1);
        (void)hipEventDestroy(h->ev_obs);
        (void)hipEventDestroy(h->ev_p0);
        (void)hipEventDestroy(h->ev_fac);
        (void)hipStreamDestroy(h->stream);
        (void)hipStreamDestroy(h->stream2);
    }
    delete h;
}

int spx_set_option(spx_handle* h, const char* name, int64_t value)
{
    if (!h || !name) return fail(SPX_ERR_ARG, "spx_set_option: null");
    if (h->multi) return spx_multi_set_option(h->multi, name, value);
    if (!strcmp(name, "kstar_budget_bytes")) {
        h->kst_budget = value > 0 ? value : (512ll << 20);
        return SPX_OK;
    }
    if (!strcmp(name, "gemm_waves")) {   // predict-GEMM variant of this handle (predict_kernels.hip)
        if (!predict_gemm_variant_ok((int)value))
            return fail(SPX_ERR_ARG, "spx_set_option: gemm_waves=%lld is not a variant of this build", (long long)value);
        h->gemm_variant = (int)value;
        return SPX_OK;
    }
    if (!strcmp(name, "covar")) {   // SPX_COVAR_*: the reference's covar= (gp.py:87-132)
        if (value < SPX_COVAR_MATERN52 || value > SPX_COVAR_SE)
            return fail(SPX_ERR_ARG, "spx_set_option: covar=%lld is not one of SPX_COVAR_*", (long long)value);
        if (h->cov_kind != (int)value) { h->factored = false; h->ran = false; h->ran_time = false; h->S = 0; }
        h->cov_kind = (int)value;
        return SPX_OK;
    }
    if (!strcmp(name, "lean_lazy")) {   // log-likelihood path: trailing updates two steps at a time (1), one (0), by size (-1, default)
        h->lean_lazy = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "step_overlap")) {   // spx_ei_step: candidate scaling + first K(X*,X) on the second stream beside the factorisation (1, default) or behind it (0)
        h->step_overlap = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "ei_fused")) {       // N <= 128, no fantasies: K* -> beta -> EI of a chunk in one kernel (1, default) or the general three-stage path (0)
        h->ei_fused = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "ei_flow")) {        // spx_factor through k_lean_flow (1, default), the left-looking launches (0)
        h->ei_flow = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_flow_cov")) {  // k_lean_flow builds K(X,X) tile by tile itself (1, default) or reads k_cov's (0)
        h->lean_flow_cov = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_flow_yield")) {   // k_lean_flow, two workgroups per CU: yield to a neighbour's diagonal block (1, default) or not (0)
        h->lean_flow_yield = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_flow_cu")) {   // k_lean_flow: one workgroup per CU (1), two (0), by size (-1, default)
        h->lean_flow_cu = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_flow")) {   // log-likelihood path: the whole factorisation as ONE data-flow launch (1, default), one launch per block column (0)
        h->lean_flow = value < 0 ? -1 : (value != 0);
        h->flow_demoted = false;        // asking again re-arms a handle that a hand-off time-out had sent to the launches
        h->flow_clean = 0;
        return SPX_OK;
    }
    if (!strcmp(name, "flow_rearm_after")) {   // clean factorisations after a hand-off time-out before k_lean_flow is tried again (default 16; 0 = never)
        h->flow_rearm_after = value < 0 ? 16 : (int)std::min<int64_t>(value, 1 << 30);
        return SPX_OK;
    }
    if (!strcmp(name, "flow_spin_limit")) {    // polls a k_lean_flow workgroup waits for a hand-off before it gives up (0 = default, 2^20); tests set 1 to see a time-out
        h->flow_spin_limit = value <= 0 ? 0 : (int)std::min<int64_t>(value, 1 << 30);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_ps")) {     // log-likelihood path: panel solve pipelined inside the step launch (1, default), separate launch (0)
        h->lean_ps = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_merge")) {     // log-likelihood path: observation scaling and the right-hand-side rows in one launch (1, default) or two (0)
        h->lean_merge = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_one")) {       // spx_gp_logprob as one launch (1, default) or prologue + k_lean_flow + reduction (0): same bits
        h->lean_one = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "lean_poll")) { h->lean_poll = value < 0 ? -1 : (value != 0); return SPX_OK; }   // (spx_internal.h)
    if (!strcmp(name, "lean_zc")) { h->lean_zc = value < 0 ? -1 : (value != 0); return SPX_OK; }
    if (!strcmp(name, "stage_copies")) {   // callers' small host buffers through the handle's page-locked staging buffer (1, default) or handed to the runtime as they are (0)
        h->stage_copies = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "cov_flat")) {       // K(X*,X) launches of several residency rounds: equal contiguous shares (k_cov_flat; 1, default) or the 3-D grid (0)
        h->cov_flat = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "gemm_partial")) {   // N not a multiple of 128: skip the padding's K steps / row tiles / K* rows (1, default) or compute them (0)
        h->gemm_partial = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (!strcmp(name, "streams")) {  // 1 = everything on one stream (default), 2 = alternate EI work items
        h->nstreams = value == 2 ? 2 : 1;
        return SPX_OK;
    }
    if (!strcmp(name, "timing")) {  // per-launch HIP events on the handle's stream; (re)starts the accumulators
        h->timing = value != 0;
        memset(h->st_ms, 0, sizeof h->st_ms);
        memset(h->st_n, 0, sizeof h->st_n);
        return SPX_OK;
    }
    return fail(SPX_ERR_ARG, "spx_set_option: unknown option '%s'", name);
}

// Copies between a CALLER's host buffer and the device.  The runtime's own path for pageable host memory (a pool of staging
// chunks) stalls for 13-28 ms every few dozen small copies on this stack (profiles/r06_set_obs_spikes.log: one
// spx_set_observations in five, median 0.03 ms) -- at Spearmint's operating sizes that is a whole next().  Buffers up to
// SPX_STAGE_MAX go through the handle's own page-locked staging buffer instead (one host memcpy; larger ones are pinned in place by
// the runtime, which does not show the stall).  stage_begin() at the start of an entry point (the stream is idle: every entry
// point that copies synchronises before it returns); a host-to-device copy is queued on `s`, the caller synchronises as before.
#define SPX_STAGE_MAX (8u << 20)
static void stage_begin(spx_handle* h) { h->stage_off = 0; }
static int stage_h2d(spx_handle* h, void* dst, const void* src, size_t bytes, hipStream_t s)
{
    if (h->stage_copies == 0 || bytes > SPX_STAGE_MAX || h->stage_off + bytes > 4 * (size_t)SPX_STAGE_MAX) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return SPX_OK;
    }
    if (h->pin_stage.cap < h->stage_off + bytes) {
        if (h->stage_off) {           // (growing would move what earlier copies of this call still read)
            HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
            return SPX_OK;
        }
        int rc = h->pin_stage.reserve(bytes > (1u << 20) ? bytes : (1u << 20));
        if (rc) return rc;
    }
    char* st = (char*)h->pin_stage.p + h->stage_off;
    memcpy(st, src, bytes);
    h->stage_off += (bytes + 255) & ~(size_t)255;
    HIPCHK(hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, s));
    return SPX_OK;
}
// device -> caller's buffer, complete on return (the stream is synchronised)
static int stage_d2h(spx_handle* h, void* dst, const void* src, size_t bytes, hipStream_t s)
{
    if (h->stage_copies == 0 || bytes > SPX_STAGE_MAX) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return SPX_OK;
    }
    HIPCHK(hipStreamSynchronize(s));     // (nothing queued may still read the staging buffer)
    int rc = h->pin_stage.reserve(bytes > (1u << 20) ? bytes : (1u << 20));
    if (rc) return rc;
    h->stage_off = 0;
    HIPCHK(hipMemcpyAsync(h->pin_stage.p, src, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(dst, h->pin_stage.p, bytes);
    return SPX_OK;
}

int spx_set_observations(spx_handle* h, const double* comp, const double* vals, int64_t N, int32_t D)
{
    if (h && h->multi) return spx_multi_set_observations(h->multi, comp, vals, N, D);
    if (!h || !comp || !vals || N < 1 || D < 1 || N > (1 << 20))
        return fail(SPX_ERR_ARG, "spx_set_observations: bad arguments (N=%lld, D=%d)", (long long)N, D);
    int rc = ensure_init(h);
    if (rc) return rc;
    if (h->have_cand && D != h->D) h->have_cand = false;
    if (h->have_hyp && D != h->D) { h->have_hyp = false; h->have_time = false; }
    h->N = N; h->D = D; h->Dp = padded_dim(D); h->Np = (int)round_up(N, SPX_PADN);
    if ((rc = h->comp.reserve((size_t)N * D * 8))) return rc;
    if ((rc = h->vals.reserve((size_t)N * 8))) return rc;
    stage_begin(h);
    if ((rc = stage_h2d(h, h->comp.p, comp, (size_t)N * D * 8, h->stream))) return rc;
    if ((rc = stage_h2d(h, h->vals.p, vals, (size_t)N * 8, h->stream))) return rc;
    double b = vals[0];
    for (int64_t i = 1; i < N; ++i) {  // np.min semantics: NaN propagates
        if (vals[i] != vals[i]) { b = vals[i]; break; }
        if (vals[i] < b) b = vals[i];
    }
    h->best = b;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_obs = true; h->have_time = false; h->factored = false; h->ran = false; h->ran_time = false; h->S = 0;
    return SPX_OK;
}

int spx_set_candidates(spx_handle* h, const double* cand, int64_t M, int32_t D, int64_t index_base)
{
    if (h && h->multi) return spx_multi_set_candidates(h->multi, cand, M, D, index_base);
    if (!h || !cand || M < 1 || D < 1)
        return fail(SPX_ERR_ARG, "spx_set_candidates: bad arguments (M=%lld, D=%d)", (long long)M, D);
    if (h->have_obs && D != h->D)
        return fail(SPX_ERR_ARG, "spx_set_candidates: D=%d but observations have D=%d", D, h->D);
    int rc = ensure_init(h);
    if (rc) return rc;
    if (!h->have_obs) { h->D = D; h->Dp = padded_dim(D); }
    h->M = M; h->index_base = index_base;
    if ((rc = h->cand.reserve((size_t)M * D * 8))) return rc;
    stage_begin(h);
    if ((rc = stage_h2d(h, h->cand.p, cand, (size_t)M * D * 8, h->stream))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_cand = true; h->ran = false; h->ran_time = false;
    return SPX_OK;
}

int spx_set_hypers(spx_handle* h, const double* hypers, int32_t H)
{
    if (h && h->multi) return spx_multi_set_hypers(h->multi, hypers, H);
    if (!h || !hypers || H < 1) return fail(SPX_ERR_ARG, "spx_set_hypers: bad arguments (H=%d)", H);
    if (H > NP_PAIRWISE_MAX_N)   // the mean over draws walks numpy's pairwise tree with fixed-depth stacks (np_sum.h)
        return fail(SPX_ERR_ARG, "spx_set_hypers: at most %d hyper-parameter draws (got %d)", NP_PAIRWISE_MAX_N, H);
    if (!h->have_obs) return fail(SPX_ERR_ARG, "spx_set_hypers: call spx_set_observations first");
    h->H = H;
    h->hyp_host.assign(hypers, hypers + (size_t)H * (3 + h->D));
    h->have_hyp = true; h->have_time = false; h->factored = false; h->ran = false; h->ran_time = false; h->S = 0;
    return SPX_OK;
}

int spx_set_time_model(spx_handle* h, const double* log_durs, const double* time_hypers)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_set_time_model: null handle");
    if (h->multi) return spx_multi_set_time_model(h->multi, log_durs, time_hypers);
    if (!log_durs || !time_hypers) { h->have_time = false; h->factored = false; return SPX_OK; }
    if (!h->have_obs || !h->have_hyp)
        return fail(SPX_ERR_ARG, "spx_set_time_model: set observations and hypers first");
    int rc = ensure_init(h);
    if (rc) return rc;
    if ((rc = h->ldur.reserve((size_t)h->N * 8))) return rc;
    stage_begin(h);
    if ((rc = stage_h2d(h, h->ldur.p, log_durs, (size_t)h->N * 8, h->stream))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->thyp_host.assign(time_hypers, time_hypers + (size_t)h->H * (3 + h->D));
    h->have_time = true; h->factored = false; h->ran = false; h->ran_time = false;
    return SPX_OK;
}

// ---------------------------------------------------------------------------
static int finish_factor(spx_handle* h, const std::vector<int>& info, bool tolerate_not_pd, bool lean);

// A hand-off of k_lean_flow timed out (its polls are bounded: an error, never a hang; not seen on a healthy device -- a
// preempted or badly oversubscribed GPU could produce one).  The call is repeated with one launch per block column -- the
// same arithmetic, the same bits -- and the handle STAYS there for `flow_rearm_after` clean factorisations (default 16), then
// tries the data-flow launch again (finish_factor): one bad moment does not cost a long-lived chooser process 30 % of every
// later factorisation.  spx_set_option("lean_flow", 1) re-arms at once.  Counters: spx_get_stat("flow_fallbacks" /
// "flow_rearms" / "flow_enabled").
static void note_flow_timeout(spx_handle* h)
{
    h->flow_fallbacks += 1;
    h->flow_demoted = true;
    h->flow_clean = -1;             // (the repeat of THIS call is not one of the clean factorisations that are counted)
    h->handoff_timeout = false;
}
// ... and the repeated call succeeded: SPX_OK, with a WARNING left in spx_last_error() (text starts with "warning:")
static void warn_flow_fallback(spx_handle* h)
{
    char buf[320];
    snprintf(buf, sizeof buf, "warning: a hand-off inside the data-flow factorisation timed out (%lld so far on this handle); "
             "this call was repeated with one launch per block column (same results) and the handle stays there for %d "
             "clean factorisations", (long long)h->flow_fallbacks, h->flow_rearm_after);
    spx_err_slot() = buf;
}

// defer_sync (log-likelihood path): return with the work queued -- the caller adds its own kernel, copies `info` back
// together with its result, synchronises ONCE and calls finish_factor (one host round trip per call instead of two)
static int do_factor(spx_handle* h, bool tolerate_not_pd, bool lean = false, bool defer_sync = false)
{
    if (!h->have_obs || !h->have_hyp)
        return fail(SPX_ERR_ARG, "spx_factor: observations and hypers must be set first");
    int rc = ensure_init(h);
    if (rc) return rc;
    // lean: objective GP only, Cholesky + forward solve, no W = L^-1 (log-likelihood path)
    const int nm = (h->have_time && !lean) ? 2 : 1;
    if (!lean) h->nmodels = nm;
    const int H = h->H, nh = nm * H, D = h->D, Dp = h->Dp;
    const int64_t N = h->N;
    // The EI path pads the observations to the predict GEMM's 128-row tiles.  The log-likelihood path (tile-major, up to
    // 32 draws) only needs whole 64 x 64 blocks: N <= 64 is ONE diagonal block instead of two, N = 129 .. 192 three
    // instead of four -- the padding block is an identity that costs a full link of the chain of diagonal blocks
    // (~17 us of a 64 us call at N <= 64).  Same bits: padding rows never touch the others.
    const int Np = (lean && nh <= 32) ? (int)round_up(N, SPX_NB) : h->Np;
    if (lean) h->lean_np = Np;
    const int nblk = Np / SPX_NB;
    const int hs = 3 + D;

    // host-side hyper tables: raw rows (for ls) and [mean, noise, amp2, amp2*(1+1e-6)]
    // (members, not locals: the log-likelihood path returns before it synchronises, and the upload must not outlive its source)
    std::vector<double>& raw = h->up_raw;
    std::vector<double>& tab = h->up_tab;
    raw.assign((size_t)nh * hs, 0.0);
    tab.assign((size_t)nh * SPX_HT, 0.0);
    for (int m = 0; m < nm; ++m) {
        const std::vector<double>& src = m ? h->thyp_host : h->hyp_host;
        for (int i = 0; i < H; ++i) {
            const double* r = &src[(size_t)i * hs];
            memcpy(&raw[((size_t)m * H + i) * hs], r, sizeof(double) * hs);
            if (h->cov_kind == SPX_COVAR_SE)   // gp.SE ignores its length scales (gp.py:88)
                for (int d = 0; d < D; ++d) raw[((size_t)m * H + i) * hs + 3 + d] = 1.0;
            double* t = &tab[((size_t)m * H + i) * SPX_HT];
            t[0] = r[0]; t[1] = r[1]; t[2] = r[2];
            t[3] = r[2] * (1 + 1e-6);  // self.amp2*(1+1e-6), GPEIChooser.py:199
        }
    }
    // raw rows and the table travel in ONE upload: htab is a view of the tail of the hyp buffer
    raw.insert(raw.end(), tab.begin(), tab.end());
    if ((rc = h->hyp.reserve(raw.size() * 8))) return rc;
    h->htab.alias_of(h->hyp.d() + (size_t)nh * hs);
    const size_t nn = (size_t)nh * Np * Np;
    if ((rc = h->Xs.reserve((size_t)nh * Np * Dp * 8))) return rc;
    if ((rc = h->X2s.reserve((size_t)nh * Np * Dp * 8))) return rc;
    if ((rc = h->s1.reserve((size_t)nh * Np * 8))) return rc;
    if ((rc = h->Lm.reserve(nn * 8))) return rc;
    if (!lean && (rc = h->WT.reserve(nn * 8))) return rc;
    if ((rc = h->Dinv.reserve((size_t)nh * nblk * SPX_NB * SPX_NB * 8))) return rc;
    if ((rc = h->gamma.reserve((size_t)nh * Np * 8))) return rc;
    if ((rc = h->alpha.reserve((size_t)nh * Np * 8))) return rc;
    if ((rc = h->info.reserve((size_t)nh * sizeof(int)))) return rc;

    hipStream_t s = h->stream;
    h->ev_used = 0;
    hipEvent_t t0 = h->ev_t0, t1 = h->ev_t1;
    // through pinned memory: a pageable source makes the runtime stage the copy and wait for it (the stream is idle here:
    // every entry point that queues work on it synchronises before it returns, so the staging buffer is free)
    if ((rc = h->pin_up.reserve(raw.size() * 8))) return rc;
    memcpy(h->pin_up.p, raw.data(), raw.size() * 8);
    // (the fused log-likelihood launch reads the rows out of the pinned buffer itself -- option lean_zc: one stream operation
    // fewer in front of it; decided below, where `fused` is known: the copy is queued unless that form runs)
    // (measured, profiles/r06_lean_one_ab.log: one launch instead of three is -11 us of 46 at N <= 64 with the polled
    // completion and the zero-copy rows, -12 of 91 at N = 256, -2 ... -4 % up to N = 1024; at N = 2048 with several draws the
    // items' own scaling costs more than the two launches did (+3 %), and with hundreds of items the reads of host memory do)
    const bool fused_early = lean && nh <= 32 && h->lean_merge != 0 && h->lean_flow != 0 && !h->flow_demoted && h->lean_flow_cov != 0
                             && h->lean_one != 0 && h->fused_lp != nullptr && Dp <= 64
                             && (h->lean_one > 0 || nblk <= 16 || nh <= 2);
    int fused_items = (nblk + 1) / 2;
    for (int i = 0; i < nblk; ++i) fused_items += (i + 2) / 2;
    const bool zero_copy = fused_early && h->lean_zc != 0 && !h->timing && (h->lean_zc > 0 || (int64_t)nh * fused_items <= 256);
    if (!zero_copy) HIPCHK(hipMemcpyAsync(h->hyp.p, h->pin_up.p, raw.size() * 8, hipMemcpyHostToDevice, s));
    if (h->timing || !lean) HIPCHK(hipEventRecord(t0, s));
    // the log-likelihood path zeroes info (and the hand-off flags) in its right-hand-side kernel: two stream operations
    // fewer per call
    const bool zero_in_kernel = lean && nh <= 32;
    // (otherwise k_scale_rows, the first kernel below, clears them)
    // (W^T's zeros above the diagonal blocks are written by k_trinv itself)

    const double* ls = h->hyp.d() + 3;
    // The log-likelihood call is a chain of small dependent launches (~4 us each whatever they do): when nothing between
    // them needs x / ls (k_lean_flow builds K(X,X) itself: the default), the scaling of the observations and the
    // right-hand-side rows are ONE launch, further down where the right-hand side used to be written (option lean_merge).
    const bool merged_prologue = lean && nh <= 32 && h->lean_merge != 0 && h->lean_flow != 0 && !h->flow_demoted && h->lean_flow_cov != 0;
    // ... and since round 6 NO launch of their own (option lean_one): k_lean_flow's items scale the rows they need into LDS,
    // generate the right-hand-side rows, and the last item of a draw reduces the log-likelihood into pinned host memory --
    // the call is the upload of the hyper rows and ONE launch.  (Dp <= 64: a block's scaled rows fit the kernel's LDS tile.)
    const bool fused = fused_early;
    h->fused_ran = fused;
    // the not-PD flags: zeroed once per allocation; the fused launch leaves them zero.  (Address AND size: a buffer that grew may
    // come back at the address the smaller one had -- its new words are not the zeros the old ones were.)
    if (fused && (h->info_clean_ptr != h->info.p || h->info_clean_bytes != h->info.cap)) {
        HIPCHK(hipMemsetAsync(h->info.p, 0, h->info.cap, s));
        h->info_clean_ptr = h->info.p;
        h->info_clean_bytes = h->info.cap;
    }
    if (!fused) h->info_clean_ptr = nullptr;
    // x / ls and, in the same launch, the second operand pre-multiplied by 2 (gp.py:50; exact)
    if (!merged_prologue)
        TIMED(ST_SCALE, launch_scale_rows(s, h->comp.d(), N, Np, D, Dp, ls, hs, nh, 1.0, h->Xs.d(), h->s1.d(), h->X2s.d(),
                                          zero_in_kernel ? nullptr : (int*)h->info.p, nh));
    // (spx_ei_step: the candidate side of the EI pass needs x / ls and the hyper table, not the factor -- it starts here,
    // on the second stream, beside the factorisation)
    if (defer_sync && !lean) HIPCHK(hipEventRecord(h->ev_obs, s));
    // The factorisation, blocked with 64x64 tiles; three generations, the same factor bit for bit (every tile receives its
    // update steps in the order 0, 1, 2, ..., through the same MFMA chains, and the diagonal blocks share diag_block):
    //   flow  k_lean_flow: ONE data-flow launch for all block columns of all draws, tile-major storage (the default);
    //   rl    the log-likelihood path's one-step-deep launches per block column, tile-major (k_lean_step_ps, or
    //         k_lean_step [+ k_lean_step2] + k_lean_trsm), up to 32 draws -- the fallback of `flow` there;
    //   else  the batched left-looking launches of the EI path, row-major (k_chol_diag + k_chol_panel) -- the fallback of
    //         `flow` for spx_factor, and the log-likelihood path beyond 32 draws.
    const int rl = (lean && nh <= 32) ? 1 : 0;   // beyond ~40 draws the one-step launches are work-bound and lose
    // The EI path (spx_factor) takes the data-flow launch too (option ei_flow, default on; measured against the left-looking
    // launches: factor stage 0.28 -> 0.19 ms at N = 256 x 10 draws, 6.3 -> 3.8 ms at 2048 x 20, 2.3 -> 1.5 ms at
    // 1024 x 40, 0.85 -> 0.61 ms at 512 x 60): no right-hand-side rows, the diagonal blocks of L kept for spx_get_factor,
    // W = L^-1 from the tile-major factor (k_trinv<true>); every EI result stays what it was.
    const int eflow = (!lean && h->ei_flow != 0) ? 1 : 0;
    const int flow = ((rl || eflow) && h->lean_flow != 0 && !h->flow_demoted) ? 1 : 0;
    const bool tiled = rl || flow;
    // (how busy the launch will be: draws x block columns^1.5 -- the residency rule below was read off scripts/dev/flow_modes.py)
    const double flow_load = (double)nh * nblk * sqrt((double)nblk);
    const bool flow_alone = h->lean_flow_cu >= 0 ? h->lean_flow_cu != 0 : flow_load <= 800.0;
    // k_lean_flow builds the tiles of K(X,X) itself, where they are consumed: -1 ... -8 % per call at every size (no k_cov
    // launch, no round trip of the matrix through memory; option lean_flow_cov, scripts/dev/lean_option_ab.py)
    const bool cov_in_flow = flow && h->lean_flow_cov != 0;
    if (!cov_in_flow)
        TIMED(ST_COV_SELF, launch_cov_self(s, h->Xs.d(), h->s1.d(), h->X2s.d(), h->htab.d(), h->Lm.d(), (int)N, Np, Dp, nh, tiled, dev_kind(h)));
    // Trailing updates two block columns at a time (k_lean_step2) halve the traffic of the trailing matrices but
    // put a second MFMA step in front of every other diagonal block; that pays once the lower triangles of the
    // batch no longer fit the 256 MB Infinity Cache (measured: N=2048 from ~20 draws, N=4096 from 6; -3 ... -16 %),
    // and costs 5-10 % below that.  Same factor either way, bit for bit.
    const int lazy = h->lean_lazy >= 0 ? h->lean_lazy : ((double)nh * Np * Np * 4.0 > 300e6 ? 1 : 0);
    // Panel solve inside the step launch, pipelined behind the diagonal block's pivots (k_lean_step_ps): option lean_ps
    // (measured against a launch of its own per panel solve, scripts/dev/lean_option_ab.py: -1 ... -8 % per call from N = 256
    // up -- 2048: -6.5 % at 4-12 draws, -1 % at one; 1000: -3 ... -13 %; 4096: -4 ... -5 %)
    const int want_ps = h->lean_ps >= 0 ? h->lean_ps : 1;
    const int ps = (rl && !lazy && want_ps && !flow) ? 1 : 0;   // (only without the data-flow launch, below)
    if (ps && (rc = h->ps_flags.reserve((size_t)nh * nblk * sizeof(int)))) return rc;   // zeroed by k_lean_rhs_init
    // The whole factorisation as ONE data-flow launch (k_lean_flow; option lean_flow, default on): against one launch per
    // block column -27 ... -36 % per call at N = 2048 (1-32 draws), -25 ... -34 % at N = 1000, -20 % at N = 256, -6 ... -10 %
    // at N = 64 (profiles/r03_flow_ab.log); the same factor bit for bit
    int* lflags = nullptr; int* dflags = nullptr; int* cu_busy = nullptr; unsigned* tickets = nullptr;
    if (flow) {
        const size_t nfl = (size_t)nh * (nblk + 1) * nblk + (size_t)nh * nblk + 2 + 4096;   // (+ one word per CU: cu_busy)
        if (nfl > h->flow_flags_n || h->flow_gen >= (1 << 27)) {
            if ((rc = h->flow_flags.reserve(nfl * sizeof(int)))) return rc;
            HIPCHK(hipMemsetAsync(h->flow_flags.p, 0, h->flow_flags.cap, h->stream));
            h->flow_flags_n = h->flow_flags.cap / sizeof(int);
            h->flow_gen = 0;
        }
        h->flow_gen += 1;
        tickets = (unsigned*)h->flow_flags.p;          // first (counter, done count): their place does not move with the batch size
        cu_busy = (int*)h->flow_flags.p + 2;           // [xcc][se, sh, cu]: workgroups inside a diagonal block, back to 0 after every launch
        lflags = cu_busy + 4096;
        dflags = lflags + (size_t)nh * (nblk + 1) * nblk;
    }
    h->flow_used = flow != 0;
    // Residency (option lean_flow_cu: 1 / 0 / -1 = by size; scripts/dev/flow_modes.py).  A diagonal block's dependent MFMA
    // chain runs a third slower beside a neighbour whose products keep the matrix pipes busy, and the whole call follows
    // that chain.  Small launches (draws x block columns^1.5 <= 800: N = 2048 up to 4 draws, N = 1000 up to 12) get ONE
    // workgroup per CU (the launch asks for 96 KB of LDS).  Larger ones need the places: two per CU, and a workgroup yields
    // while its neighbour is the next link of a draw's chain -- from the end of its history through its diagonal block
    // (option lean_flow_yield; a word per CU, found by XCC_ID / HW_ID).  Against two per CU without yielding: N = 2048:
    // -11 % at 4 draws, -13 % at 6, -10 % at 8, -2 % at 12; N = 4096: -16 % at 2 draws; N = 1000: -7 % at 20 draws.
    // lean: the right-hand side vals - mean rides through the factorisation as an extra row block,
    // so y = L^-1 (vals - mean) is ready when the last column is
    double* rhs = nullptr;
    if (lean) {
        if ((rc = h->rhs.reserve((size_t)nh * SPX_NB * Np * 8))) return rc;
        rhs = h->rhs.d();
        if (rl) {
            if ((rc = h->diagL.reserve((size_t)nh * Np * 8))) return rc;
            if (fused) {}            // (nothing to launch)
            else if (merged_prologue)     // (implies flow and cov_in_flow: nothing before this point read x / ls)
                TIMED(ST_SCALE, launch_lean_prologue(s, h->comp.d(), N, Np, D, Dp, ls, hs, nh, h->Xs.d(), h->s1.d(), h->X2s.d(),
                                                     h->vals.d(), h->htab.d(), rhs, (int*)h->info.p, ps ? (int*)h->ps_flags.p : nullptr));
            else
                TIMED(ST_GAMMA_ALPHA, launch_lean_rhs_init(s, h->vals.d(), h->htab.d(), rhs, (int)N, Np, nh, (int*)h->info.p,
                                                           ps ? (int*)h->ps_flags.p : nullptr));
        } else {
            TIMED(ST_GAMMA_ALPHA, launch_rhs_init(s, h->vals.d(), h->htab.d(), rhs, (int)N, Np, nh));
        }
        h->lean_tiled = rl != 0;
    }
    h->factor_tiled = !lean && flow;
    FlowFused ff{h->comp.d(), zero_copy ? (const double*)h->pin_up.p : h->hyp.d(), h->vals.d(), D, hs, h->fused_lp, h->fused_info};
    const double* htab_dev = zero_copy ? (const double*)h->pin_up.p + (size_t)nh * hs : h->htab.d();
    if (flow)
        TIMED(ST_CHOL_DIAG, launch_lean_flow(s, h->Lm.d(), h->Dinv.d(), (int*)h->info.p, rhs, lean ? h->diagL.d() : nullptr, lflags, dflags, tickets, Np, nh, h->flow_gen, flow_alone,
                                             cov_in_flow ? h->Xs.d() : nullptr, h->X2s.d(), h->s1.d(), htab_dev, (int)N, Dp, dev_kind(h),
                                             h->lean_flow_yield != 0 ? cu_busy : nullptr, h->flow_spin_limit, fused ? &ff : nullptr));
    for (int k = 0; k < (flow ? 0 : nblk); ++k) {
        if (ps) {
            TIMED(ST_CHOL_DIAG, launch_lean_step_ps(s, h->Lm.d(), h->Dinv.d(), (int*)h->info.p, rhs, h->diagL.d(), (int*)h->ps_flags.p, Np, k, nh));
        } else if (rl) {
            TIMED(ST_CHOL_DIAG, launch_lean_step(s, h->Lm.d(), h->Dinv.d(), (int*)h->info.p, rhs, h->diagL.d(), Np, k, nh, lazy));
            TIMED(ST_CHOL_PANEL, launch_lean_trsm(s, h->Lm.d(), h->Dinv.d(), rhs, Np, k, nh));
        } else {
            TIMED(ST_CHOL_DIAG, launch_chol_diag(s, h->Lm.d(), h->Dinv.d(), (int*)h->info.p, Np, k, nh, 0));
            if (k + 1 < nblk || rhs) TIMED(ST_CHOL_PANEL, launch_chol_panel(s, h->Lm.d(), h->Dinv.d(), Np, k, nh, rhs));
        }
    }
    if (!lean) {
        TIMED(ST_TRINV, launch_trinv(s, h->Lm.d(), h->Dinv.d(), h->WT.d(), Np, nh, flow != 0));
        TIMED(ST_GAMMA_ALPHA, launch_gamma(s, h->WT.d(), h->vals.d(), h->htab.d(), h->gamma.d(), (int)N, Np, H));
        if (nm == 2)
            TIMED(ST_GAMMA_ALPHA, launch_gamma(s, h->WT.d() + (size_t)H * Np * Np, h->ldur.d(),
                                                h->htab.d() + (size_t)H * SPX_HT,
                                                h->gamma.d() + (size_t)H * Np, (int)N, Np, H));
        TIMED(ST_GAMMA_ALPHA, launch_alpha(s, h->WT.d(), h->gamma.d(), h->alpha.d(), Np, nh));
    }
    if (h->timing || !lean) HIPCHK(hipEventRecord(t1, s));
    if (defer_sync && !lean) HIPCHK(hipEventRecord(h->ev_fac, s));   // (spx_ei_step with two streams: alpha is ready)
    if (defer_sync) return SPX_OK;
    std::vector<int> info(nh);
    if ((rc = stage_d2h(h, info.data(), h->info.p, (size_t)nh * sizeof(int), s))) return rc;
    return finish_factor(h, info, tolerate_not_pd, lean);
}

static int finish_factor(spx_handle* h, const std::vector<int>& info, bool tolerate_not_pd, bool lean)
{
    const int nh = (int)info.size(), H = h->H;
    LAUNCHCHK();
    float ms = 0.f;
    if (h->timing) {
        (void)hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1);
        ev_collect(h);
        h->st_ms[ST_FACTOR_TOTAL] += ms; h->st_n[ST_FACTOR_TOTAL] += 1;
    }
    h->not_pd_draw = h->not_pd_pivot = -1;
    for (int i = 0; i < nh; ++i)
        if (info[i] < 0) { // k_lean_flow / k_lean_step_ps: a workgroup gave up waiting for a tile or a diagonal block (bounded spin)
            h->handoff_timeout = true;
            return fail(SPX_ERR_HIP, "log-likelihood factorisation: in-launch hand-off timed out (draw %d)", i);
        }
    if (h->flow_demoted && h->flow_rearm_after > 0 && ++h->flow_clean >= h->flow_rearm_after) {
        h->flow_demoted = false;     // the next factorisation is the data-flow launch again
        h->flow_clean = 0;
        h->flow_rearms += 1;
    }
    for (int i = 0; i < nh; ++i)
        if (info[i]) { h->not_pd_draw = i; h->not_pd_pivot = info[i] - 1; break; }
    h->factored = !lean;
    h->ran = false; h->ran_time = false;
    h->S = 0;
    if (h->not_pd_draw >= 0 && !tolerate_not_pd) {
        h->factored = false;
        return fail(SPX_ERR_NOT_PD, "%d-th leading minor of the array is not positive definite (draw %d%s)",
                    h->not_pd_pivot + 1, h->not_pd_draw % H, h->not_pd_draw >= H ? ", time model" : "");
    }
    return SPX_OK;
}

int spx_factor(spx_handle* h)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_factor: null handle");
    if (h->multi) return spx_multi_factor(h->multi);
    h->handoff_timeout = false;
    int rc = do_factor(h, false);
    if (rc == SPX_ERR_HIP && h->handoff_timeout && h->flow_used) {   // as in spx_gp_logprob: never seen; bounded, then the launches
        note_flow_timeout(h);
        rc = do_factor(h, false);
        if (!rc) warn_flow_fallback(h);
    }
    return rc;
}

int spx_not_pd_info(spx_handle* h, int32_t* draw, int32_t* pivot)
{
    if (!h) return fail(SPX_ERR_ARG, "null handle");
    if (h->multi) return spx_multi_not_pd_info(h->multi, draw, pivot);
    if (draw) *draw = h->not_pd_draw;
    if (pivot) *pivot = h->not_pd_pivot;
    return SPX_OK;
}

int spx_set_fantasies(spx_handle* h, const double* fant, const double* bests, int32_t S)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_set_fantasies: null handle");
    if (h->multi) return spx_multi_set_fantasies(h->multi, fant, bests, S);
    if (!fant || !bests || S <= 0) { h->S = 0; return SPX_OK; }   // clear
    if (!h->factored) return fail(SPX_ERR_ARG, "spx_set_fantasies: call spx_factor first");
    if (S > 4096) return fail(SPX_ERR_ARG, "spx_set_fantasies: at most 4096 fantasies (got %d)", S);
    int rc = ensure_init(h);
    if (rc) return rc;
    const int H = h->H, Np = h->Np;
    const int64_t n = h->N;
    // host transpose to [H][S][n] so that every fantasy column is a contiguous right-hand side
    std::vector<double> ft((size_t)H * S * n);
    for (int d = 0; d < H; ++d)
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < S; ++c)
                ft[((size_t)d * S + c) * n + i] = fant[((size_t)d * n + i) * S + c];
    if ((rc = h->fantT.reserve(ft.size() * 8))) return rc;
    if ((rc = h->gammaS.reserve((size_t)H * S * Np * 8))) return rc;
    if ((rc = h->bests.reserve((size_t)H * S * 8))) return rc;
    hipStream_t s = h->stream;
    stage_begin(h);
    if ((rc = stage_h2d(h, h->fantT.p, ft.data(), ft.size() * 8, s))) return rc;
    if ((rc = stage_h2d(h, h->bests.p, bests, (size_t)H * S * 8, s))) return rc;
    for (int d = 0; d < H; ++d)   // Gamma_d = W_d (F_d - mean_d), one launch per draw, S columns each
        launch_gamma_multi(s, h->WT.d() + (size_t)d * Np * Np, h->fantT.d() + (size_t)d * S * n,
                           h->htab.d() + (size_t)d * SPX_HT, h->gammaS.d() + (size_t)d * S * Np, (int)n, Np, S);
    HIPCHK(hipStreamSynchronize(s));
    LAUNCHCHK();
    h->S = S;
    h->alphaS_valid = false;
    h->ran = false; h->ran_time = false;
    return SPX_OK;
}

// candidate-chunk / draw-group plan for the K(X*,X) staging buffer
static void plan_chunks(const spx_handle* h, int64_t* Mc, int* Hb)
{
    const int64_t Mp = round_up(h->M, SPX_BN);
    int64_t mc_budget = h->kst_budget / (8ll * h->Np) / SPX_BN * SPX_BN;   // what the staging buffer holds of one draw
    if (mc_budget < SPX_BN) mc_budget = SPX_BN;
    int64_t mc = mc_budget < Mp ? mc_budget : Mp;
    // equal-sized chunks (no tiny, inefficient last launch) ...
    const int64_t nchunks = (Mp + mc - 1) / mc;
    mc = round_up((Mp + nchunks - 1) / nchunks, SPX_BN);
    // ... of a whole number of candidate tiles per XCD when there are several: the predict GEMM deals the
    // 128-candidate tiles of a launch round-robin to the 8 XCDs, and a launch with 245 tiles runs as long as one
    // with 248 (measured at C3: 7 chunks of 224 tiles 268 ms per step, 6 of 261 or 5 of 313 tiles 273 ms)
    if (nchunks > 1) {
        const int64_t up = round_up(mc, 8 * SPX_BN);
        if (up <= mc_budget) mc = up;
        else if (mc >= 16 * SPX_BN) mc = mc / (8 * SPX_BN) * (8 * SPX_BN);
    }
    int64_t hb = h->kst_budget / (8ll * h->Np * mc);
    if (hb < 1) hb = 1;
    if (hb > h->H) hb = h->H;
    *Mc = mc;
    *Hb = (int)hb;
}

// factor_pending (spx_ei_step): the factorisation is queued on the stream but not yet checked -- its not-PD flags travel
// home with the winner, and ONE synchronisation ends the step
static int ei_run_impl(spx_handle* h, int32_t flags, bool factor_pending);

// what an EI pass refuses because of its flags alone (nmodels: 2 when a time model is / will be factored)
static int check_run_flags(const spx_handle* h, int32_t flags, int nmodels)
{
    const bool per_sec = (flags & SPX_FLAG_PER_SEC) != 0;
    const bool keep_mom = (flags & SPX_FLAG_KEEP_MOMENTS) != 0;
    const bool time_only = (flags & SPX_FLAG_TIME_ONLY) != 0;
    if (time_only && !(per_sec && keep_mom))
        return fail(SPX_ERR_ARG, "spx_ei_run: SPX_FLAG_TIME_ONLY needs SPX_FLAG_PER_SEC | SPX_FLAG_KEEP_MOMENTS");
    if (time_only && h->comm) return fail(SPX_ERR_ARG, "spx_ei_run: SPX_FLAG_TIME_ONLY has no winner to exchange (communicator attached)");
    if (per_sec && nmodels != 2)
        return fail(SPX_ERR_ARG, "spx_ei_run: SPX_FLAG_PER_SEC needs spx_set_time_model before spx_factor");
    return SPX_OK;
}

int spx_ei_run(spx_handle* h, int32_t flags)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_ei_run: null handle");
    if (h->multi) return spx_multi_ei_run(h->multi, flags & ~SPX_FLAG_TIME_ONLY);   // (several GPUs: the full pass; the durations are a by-product)
    return ei_run_impl(h, flags, false);
}

// spx_factor + spx_ei_run as ONE call with ONE host round trip (what a chooser's next() and bench.py's step are): at small
// N the two synchronisations and the device-to-host copies behind them were a quarter of the step (N = 128, 20 000
// candidates, 10 draws: ~85 of 310 us).  Same kernels, same results; a covariance that is not positive definite is
// reported exactly as spx_factor reports it (SPX_ERR_NOT_PD, spx_not_pd_info), after the one synchronisation.
int spx_ei_step(spx_handle* h, int32_t flags)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_ei_step: null handle");
    if (h->multi) {
        int rc = spx_multi_factor(h->multi);
        return rc ? rc : spx_multi_ei_run(h->multi, flags & ~SPX_FLAG_TIME_ONLY);
    }
    if (h->timing || (flags & SPX_FLAG_TIMING) || !h->have_cand) {   // stage timers bracket each call: keep the two-call form
        int rc = spx_factor(h);
        return rc ? rc : ei_run_impl(h, flags, false);
    }
    // the pass's argument checks BEFORE anything is queued (ei_run_impl repeats them for spx_ei_run)
    int rc = check_run_flags(h, flags, h->have_time ? 2 : 1);
    if (rc) return rc;
    h->handoff_timeout = false;
    rc = do_factor(h, false, false, true);
    if (rc) return rc;
    h->S = 0;                      // a new factorisation drops the fantasies (finish_factor), here before the run
    rc = ei_run_impl(h, flags, true);
    if (rc == SPX_ERR_HIP && h->handoff_timeout && h->flow_used) {   // as in spx_factor: bounded, then the launches
        note_flow_timeout(h);
        rc = do_factor(h, false);
        if (!rc) rc = ei_run_impl(h, flags, false);
        if (!rc) warn_flow_fallback(h);
        return rc;
    }
    if (rc) {
        // any other error exit of a pending step (argument checks that depend on the plan, a failed reservation, a launch
        // error): the factorisation may still be queued -- every entry point returns with the streams idle (the pinned
        // upload staging relies on it) and without a factor it did not check
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->stream2);
        (void)hipGetLastError();
        if (rc != SPX_ERR_NOT_PD) h->factored = false;
    }
    return rc;
}

static int ei_run_impl(spx_handle* h, int32_t flags, bool factor_pending)
{
    if (!h->factored && !factor_pending) return fail(SPX_ERR_ARG, "spx_ei_run: call spx_factor first");
    if (!h->have_cand) return fail(SPX_ERR_ARG, "spx_ei_run: no candidates set");
    const bool per_sec = (flags & SPX_FLAG_PER_SEC) != 0;
    const bool keep_mom = (flags & SPX_FLAG_KEEP_MOMENTS) != 0;
    const bool time_only = (flags & SPX_FLAG_TIME_ONLY) != 0;
    int rc = check_run_flags(h, flags, h->nmodels);
    if (rc) return rc;
    if ((rc = ensure_init(h))) return rc;
    const int H = h->H, D = h->D, Dp = h->Dp, Np = h->Np, hs = 3 + D;
    const int64_t N = h->N, M = h->M, Mp = round_up(M, SPX_BN);
    const int nrb = Np / SPX_BM;
    int64_t Mc; int Hb;
    plan_chunks(h, &Mc, &Hb);
    const int S = h->S;
    if (S > 0) {
        // the per-fantasy partial means are [nrb][2][S][Mc]: keep them under 2 GB (of 288: at C3 size with 100 fantasies the
        // plan's own 28 672-candidate chunks fit; the 256 MB of earlier rounds cut them to 9 856 -- 420 launch pairs of a
        // few dozen workgroups instead of 140)
        // (... of what the device has free: a smaller or fuller GPU takes smaller chunks instead of an allocation failure)
        // (queried when the number of fantasies changes, not per pass: the chunk plan -- buffer sizes, launch counts -- then
        // stays what it was from pass to pass and across ranks that share a device)
        if (h->fant_budget_S != S) {
            size_t mem_free = 0, mem_total = 0;
            int64_t b = 2048ll << 20;
            if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) {
                const int64_t have = (int64_t)(mem_free / 8) + (int64_t)(h->part_bgS[0].cap + h->part_bgS[1].cap + h->scratch.cap) / 2;
                b = std::max<int64_t>(64ll << 20, std::min<int64_t>(b, have));
            } else (void)hipGetLastError();
            h->fant_budget = b;
            h->fant_budget_S = S;
        }
        const int64_t fant_budget = h->fant_budget;
        int64_t cap = fant_budget / ((int64_t)nrb * 2 * S * 8) / SPX_BN * SPX_BN;
        if (cap < SPX_BN) cap = SPX_BN;
        if (Mc > cap) {
            const int64_t nchunks = (Mp + cap - 1) / cap;
            Mc = round_up((Mp + nchunks - 1) / nchunks, SPX_BN);
        }
        // ... and as many draws per launch as that leaves room for (small problems: all of them -- ten launch pairs of a
        // 20 000-candidate pass become one)
        int64_t hb = fant_budget / ((int64_t)nrb * 2 * S * 8 * Mc);
        if (hb > 65535 / S) hb = 65535 / S;       // (grid.y of the per-fantasy EI kernel)
        if (hb < 1) hb = 1;
        if (hb < Hb) Hb = (int)hb;
    }

    // N <= 128 without fantasies: the whole EI pass of a chunk -- K(X,X*), beta = W K*, the moments, EI -- is one kernel
    // with K* and beta in registers (fused_kernels.hip; option ei_fused); same bits as the three-stage path below
    const bool fused = Np == SPX_PADN && S == 0 && h->ei_fused != 0;
    h->last_fused = fused;
    const int ns = fused ? 1 : h->nstreams;
    for (int b = 0; b < 2; ++b) {
        // scaled candidates (all draws) and predicted durations are double-buffered by chunk parity,
        // the K(X*,X) staging buffer and the partial sums by stream
        if ((rc = h->Cs[b].reserve((size_t)H * Mc * Dp * 8))) return rc;
        if ((rc = h->s2[b].reserve((size_t)H * Mc * 8))) return rc;
        if (per_sec && (rc = h->time_m[b].reserve((size_t)H * Mc * 8))) return rc;
        if (b < ns && !fused && !time_only) {
            if ((rc = h->Kst[b].reserve((size_t)Hb * Np * Mc * 8))) return rc;
            if (S > 0 && (rc = h->part_bgS[b].reserve((size_t)nrb * 2 * Hb * S * Mc * 8))) return rc;
        }
    }
    // column sums of beta^2 and beta*gamma per row block for ALL draws of a chunk: written by the
    // GEMM launches and read by one EI-finalize launch per chunk, all on the consumer stream
    if (!fused && !time_only && (rc = h->part_ss[0].reserve((size_t)nrb * H * Mc * 8))) return rc;
    if (!fused && !time_only && (rc = h->part_bg[0].reserve((size_t)nrb * H * Mc * 8))) return rc;
    if (S > 0 && (rc = h->scratch.reserve((size_t)Hb * S * Mc * 8))) return rc;   // EI per (draw of the group, fantasy, candidate)
    if ((rc = h->ei_draw.reserve((size_t)H * Mp * 8))) return rc;
    if ((rc = h->ei_mean.reserve((size_t)Mp * 8))) return rc;
    if (keep_mom) {
        if ((rc = h->mom_m.reserve((size_t)H * Mp * 8))) return rc;
        if ((rc = h->mom_v.reserve((size_t)H * Mp * 8))) return rc;
        if (per_sec && (rc = h->mom_t.reserve((size_t)H * Mp * 8))) return rc;
    }
    const int nab = argmax_blocks(M);
    if ((rc = h->am_val.reserve((size_t)nab * 8))) return rc;
    if ((rc = h->am_idx.reserve((size_t)nab * 8))) return rc;
    if ((rc = h->am_out_val.reserve(8))) return rc;
    if ((rc = h->am_out_idx.reserve(8))) return rc;

    hipStream_t s = h->stream;
    if (flags & SPX_FLAG_TIMING) h->timing = true;
    h->ev_used = 0;
    hipEvent_t t0 = h->ev_t0, t1 = h->ev_t1;
    HIPCHK(hipEventRecord(t0, s));
    // N is padded to the predict GEMM's 128-row tiles with an identity: the production GEMM skips what the padding would
    // multiply (K steps and row tiles from tile ceil(N / 16) on) and K(X*,X) then leaves those rows unwritten.  Same bits.
    const int gemm_nlive = (!fused && h->gemm_partial != 0) ? predict_gemm_padding_plan(h->gemm_variant, (int)N, Np) : 0;
    const bool skip_pad = gemm_nlive > 0;
    const int cov_live_rows = 16 * gemm_nlive;
    h->last_skip_pad = skip_pad;

    const double* ls = h->hyp.d() + 3;
    const size_t nn = (size_t)Np * Np;
    // Streams: G (consumer: predict GEMM + EI finalize) and P (producer: candidate scaling and
    // K(X*,X)).  With one stream P == G and everything is in order.  With two, the K(X*,X)
    // staging buffer is double-buffered per work item and item i+1 is produced while item i is
    // consumed: the producer kernels are VALU/store-bound, the GEMM is MFMA-bound, and at 152 /
    // 178 VGPRs one producer wave fits next to the two GEMM waves of a SIMD.
    // events: [0..1] K* of buffer b ready, [2..3] buffer b consumed, [4..5] chunk parity consumed
    hipStream_t G = h->stream, P = (ns == 2) ? h->stream2 : h->stream;
    // A step (factor_pending): the factorisation is still running on G -- a chain of small launches that leaves most of the
    // chip idle -- and the first producer work of the pass (candidate scaling, K(X*,X) of the first group of draws) depends
    // only on what the factorisation's FIRST kernel wrote.  It goes to the second stream, beside the factorisation
    // (C2: K(X*,X) of all ten draws, 0.10 ms, hidden behind a 0.17 ms factorisation).  A time model's predicted durations
    // need alpha: they stay behind the factorisation on G, with the chunk's other-parity scaling buffers to themselves.
    const bool overlap = factor_pending && ns == 1 && h->step_overlap != 0;
    hipStream_t P0 = overlap ? h->stream2 : P;
    if (overlap) HIPCHK(hipStreamWaitEvent(P0, h->ev_obs, 0));
    // Two streams (option streams = 2) in a step: the producer stream has no other order against the factorisation that is
    // still queued on G.  Its scaling and K(X*,X) launches read x / ls, the row norms and the hyper table (written by the
    // factorisation's first kernel: ev_obs); a time model's predicted durations read alpha of the duration GP -- the END of
    // the factorisation (ev_fac).
    if (factor_pending && ns == 2) HIPCHK(hipStreamWaitEvent(P, per_sec ? h->ev_fac : h->ev_obs, 0));
    int item = 0, chunk = 0;
    for (int64_t c0 = 0; c0 < Mp; c0 += Mc, ++chunk) {
        const int mc = (int)std::min<int64_t>(Mc, Mp - c0);       // multiple of 128
        const int64_t nreal = std::min<int64_t>(mc, M - c0);      // real candidates in the chunk
        const double* xc = h->cand.d() + (size_t)c0 * D;
        const int par = chunk & 1;
        double* Cs = h->Cs[par].d();
        double* s2 = h->s2[par].d();
        double* tm = per_sec ? h->time_m[par].d() : nullptr;
        // the chunk-level buffers of this parity were last read (by EI finalize on G) two chunks ago
        if (ns == 2 && chunk >= 2) HIPCHK(hipStreamWaitEvent(P, h->ev_sync[4 + par], 0));
        if (per_sec) {
            // log-duration GP: predicted duration of every candidate of the chunk, all draws in one launch
            const bool side = overlap && chunk == 0;         // (a step's first chunk: the objective's scaling runs on P0 meanwhile)
            hipStream_t T = side ? G : P;
            double* CsT = side ? h->Cs[par ^ 1].d() : Cs;
            double* s2T = side ? h->s2[par ^ 1].d() : s2;
            TIMED_S(ST_SCALE, T, launch_scale_rows(T, xc, nreal, mc, D, Dp, ls + (size_t)H * hs, hs, H, 2.0, CsT, s2T));
            TIMED_S(ST_CROSS_MEAN, T, launch_cross_mean(T, h->Xs.d() + (size_t)H * Np * Dp, h->s1.d() + (size_t)H * Np, CsT, s2T,
                                                        h->htab.d() + (size_t)H * SPX_HT, h->alpha.d() + (size_t)H * Np, tm,
                                                        (int)N, Np, mc, Dp, H, dev_kind(h)));
            if (keep_mom)   // predicted durations [H][mc] -> [H][Mp] for spx_get_time_mean
                HIPCHK(hipMemcpy2DAsync(h->mom_t.d() + c0, (size_t)Mp * 8, tm, (size_t)mc * 8,
                                        (size_t)std::min<int64_t>(mc, Mp - c0) * 8, (size_t)H, hipMemcpyDeviceToDevice, T));
        }
        if (time_only) continue;      // (the predicted durations of the chunk are in mom_t: nothing else was asked for)
        // 2 * cand / ls and |cand / ls|^2 for every draw (one launch per chunk)
        hipStream_t Pc = (chunk == 0) ? P0 : P;       // (the first chunk's producer work of a step: beside the factorisation)
        TIMED_S(ST_SCALE, Pc, launch_scale_rows(Pc, xc, nreal, mc, D, Dp, ls, hs, H, 2.0, Cs, s2));
        if (fused && Pc != G) {
            HIPCHK(hipEventRecord(h->ev_p0, Pc));
            HIPCHK(hipStreamWaitEvent(G, h->ev_p0, 0));
        }
        if (fused)
            TIMED_S(ST_PREDICT_GEMM, G, launch_ei_fused128(G, dev_kind(h), h->WT.d(), h->gamma.d(), h->Xs.d(), h->s1.d(), Cs, s2,
                                                           h->htab.d(), tm, h->best, h->ei_draw.d(), keep_mom ? h->mom_m.d() : nullptr,
                                                           keep_mom ? h->mom_v.d() : nullptr, (int)N, mc, Dp, H, c0, M, Mp, h->n_cu));
        for (int h0 = 0; h0 < (fused ? 0 : H); h0 += Hb, ++item) {
            const int nhb = std::min(Hb, H - h0);
            const int k = (ns == 2) ? (item & 1) : 0;
            if (ns == 2 && item >= 2) HIPCHK(hipStreamWaitEvent(P, h->ev_sync[2 + k], 0));   // buffer k consumed
            hipStream_t Pi = (item == 0) ? Pc : P;
            TIMED_S(ST_COV_CROSS, Pi, launch_cov_cross(Pi, h->Xs.d() + (size_t)h0 * Np * Dp, h->s1.d() + (size_t)h0 * Np,
                                                      Cs + (size_t)h0 * mc * Dp, s2 + (size_t)h0 * mc,
                                                      h->htab.d() + (size_t)h0 * SPX_HT, h->Kst[k].d(), (int)N, Np, mc, Dp, nhb, dev_kind(h),
                                                      cov_live_rows, h->cov_flat != 0));
            if (ns == 2) {
                HIPCHK(hipEventRecord(h->ev_sync[k], P));
                HIPCHK(hipStreamWaitEvent(G, h->ev_sync[k], 0));
            } else if (Pi != G) {
                HIPCHK(hipEventRecord(h->ev_p0, Pi));
                HIPCHK(hipStreamWaitEvent(G, h->ev_p0, 0));
            }
            // with fantasies the launch's partial sums are consumed right away and sit at draws 0 .. nhb-1
            TIMED_S(ST_PREDICT_GEMM, G, launch_predict_gemm(G, h->gemm_variant, h->WT.d() + (size_t)h0 * nn, h->Kst[k].d(),
                                                            h->gamma.d() + (size_t)h0 * Np, h->part_ss[0].d(),
                                                            h->part_bg[0].d(), Np, mc, nhb, S > 0 ? nhb : H, S > 0 ? 0 : h0,
                                                            S > 0 ? h->gammaS.d() + (size_t)h0 * S * Np : nullptr, S,
                                                            S > 0 ? h->part_bgS[k].d() : nullptr, gemm_nlive));
            if (S > 0)
                TIMED_S(ST_EI_FINALIZE, G, launch_ei_finalize_fant(G, h->part_ss[0].d(), h->part_bgS[k].d(),
                                                                   h->htab.d() + (size_t)h0 * SPX_HT,
                                                                   h->bests.d() + (size_t)h0 * S,
                                                                   per_sec ? tm + (size_t)h0 * mc : nullptr,
                                                                   h->ei_draw.d(), nrb, mc, nhb, S, c0, M, Mp, h0,
                                                                   h->scratch.d()));
            if (ns == 2) HIPCHK(hipEventRecord(h->ev_sync[2 + k], G));
        }
        if (S == 0 && !fused)   // every draw of the chunk in one launch
            TIMED_S(ST_EI_FINALIZE, G, launch_ei_finalize(G, h->part_ss[0].d(), h->part_bg[0].d(), h->htab.d(), tm, h->best,
                                                          h->ei_draw.d(), keep_mom ? h->mom_m.d() : nullptr,
                                                          keep_mom ? h->mom_v.d() : nullptr, nrb, mc, H, c0, M, Mp, 0));
        if (ns == 2) HIPCHK(hipEventRecord(h->ev_sync[4 + par], G));
    }
    if (ns == 2 && per_sec && keep_mom) {   // the duration copies ran on P
        HIPCHK(hipEventRecord(h->ev_sync[0], P));
        HIPCHK(hipStreamWaitEvent(G, h->ev_sync[0], 0));
    }
    // the winner (and, in a step, the factorisation's flags) go home through pinned memory with the last kernel's own stores
    const int n_info = factor_pending ? h->nmodels * H : 0;
    if ((rc = h->pin_res.reserve(16 + (size_t)n_info * sizeof(int)))) return rc;
    double* mirror = (double*)h->pin_res.p;
    TIMED(ST_MEAN_ARGMAX, {
        launch_mean_argmax(s, h->ei_draw.d(), h->ei_mean.d(), M, Mp, H, h->am_val.d(), (int64_t*)h->am_idx.p, h->am_out_val.d(),
                           (int64_t*)h->am_out_idx.p, mirror, (const int*)h->info.p, n_info);
    });
    HIPCHK(hipEventRecord(t1, s));
    HIPCHK(hipStreamSynchronize(s));
    LAUNCHCHK();
    h->best_val = mirror[0];
    h->best_idx = ((const int64_t*)mirror)[1];
    if (factor_pending) {
        std::vector<int> info((const int*)(mirror + 2), (const int*)(mirror + 2) + n_info);
        if ((rc = finish_factor(h, info, false, false))) { h->ran = false; h->ran_time = false; return rc; }
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t0, t1);
    if (h->timing) {
        ev_collect(h);
        h->st_ms[ST_EI_RUN_TOTAL] += ms; h->st_n[ST_EI_RUN_TOTAL] += 1;
    }
    h->ran = !time_only;             // (a time-only pass has no EI results: spx_get_best & co. refuse)
    h->ran_time = time_only;
    h->ran_2d = false;
    h->ran_moments = keep_mom && S == 0 && !time_only;
    if (h->comm) return spx_comm_exchange(h);   // one process per GPU: the winner over all ranks
    return SPX_OK;
}

int spx_get_best(spx_handle* h, int64_t* best_idx, double* best_val)
{
    if (h && h->multi) return spx_multi_get_best(h->multi, best_idx, best_val);
    if (!h || !h->ran) return fail(SPX_ERR_ARG, "spx_get_best: no results (call spx_ei_run)");
    if (best_idx) *best_idx = h->best_idx + h->index_base;
    if (best_val) *best_val = h->best_val;
    return SPX_OK;
}

int spx_get_ei_mean(spx_handle* h, double* out)
{
    if (h && h->multi) return spx_multi_get_ei_mean(h->multi, out);
    if (!h || !out || !h->ran) return fail(SPX_ERR_ARG, "spx_get_ei_mean: no results / null output");
    int rc = ensure_init(h);
    if (rc) return rc;
    // after the all-reduce of a partitioned run: the GLOBAL mean (all draws of all ranks) of this handle's candidates
    const double* src = h->ran_2d ? h->ei_sum_full.d() + h->index_base : h->ei_mean.d();
    return stage_d2h(h, out, src, (size_t)h->M * 8, h->stream);
}

int spx_get_ei_draws(spx_handle* h, double* out)
{
    if (h && h->multi) return spx_multi_get_ei_draws(h->multi, out);
    if (!h || !out || !h->ran) return fail(SPX_ERR_ARG, "spx_get_ei_draws: no results / null output");
    int rc = ensure_init(h);
    if (rc) return rc;
    const int64_t M = h->M, Mp = round_up(M, SPX_BN);
    const int H = h->H;
    std::vector<double> tmp((size_t)H * Mp);
    if ((rc = stage_d2h(h, tmp.data(), h->ei_draw.p, tmp.size() * 8, h->stream))) return rc;
    for (int64_t c = 0; c < M; ++c)
        for (int d = 0; d < H; ++d) out[(size_t)c * H + d] = tmp[(size_t)d * Mp + c];
    return SPX_OK;
}

int spx_get_moments(spx_handle* h, int32_t draw, double* func_m, double* func_v)
{
    if (h && h->multi) return spx_multi_get_moments(h->multi, draw, func_m, func_v);
    if (!h || !h->ran || !h->ran_moments)
        return fail(SPX_ERR_ARG, "spx_get_moments: run spx_ei_run with SPX_FLAG_KEEP_MOMENTS first");
    if (draw < 0 || draw >= h->H) return fail(SPX_ERR_ARG, "spx_get_moments: draw out of range");
    int rc = ensure_init(h);
    if (rc) return rc;
    const int64_t Mp = round_up(h->M, SPX_BN);
    if (func_m) HIPCHK(hipMemcpy(func_m, h->mom_m.d() + (size_t)draw * Mp, (size_t)h->M * 8, hipMemcpyDeviceToHost));
    if (func_v) HIPCHK(hipMemcpy(func_v, h->mom_v.d() + (size_t)draw * Mp, (size_t)h->M * 8, hipMemcpyDeviceToHost));
    return SPX_OK;
}

int spx_get_time_mean(spx_handle* h, int32_t draw, double* out)
{
    if (h && h->multi) return spx_multi_get_time_mean(h->multi, draw, out);
    if (!h || !out || !((h->ran && h->ran_moments) || h->ran_time) || h->nmodels != 2)
        return fail(SPX_ERR_ARG, "spx_get_time_mean: run spx_ei_run with SPX_FLAG_PER_SEC | SPX_FLAG_KEEP_MOMENTS first");
    if (draw < 0 || draw >= h->H) return fail(SPX_ERR_ARG, "spx_get_time_mean: draw out of range");
    int rc = ensure_init(h);
    if (rc) return rc;
    const int64_t Mp = round_up(h->M, SPX_BN);
    HIPCHK(hipMemcpy(out, h->mom_t.d() + (size_t)draw * Mp, (size_t)h->M * 8, hipMemcpyDeviceToHost));
    return SPX_OK;
}

int spx_get_factor(spx_handle* h, int32_t draw, double* K, double* L, double* alpha)
{
    if (h && h->multi) return spx_multi_get_factor(h->multi, draw, K, L, alpha);
    if (!h || !h->factored) return fail(SPX_ERR_ARG, "spx_get_factor: call spx_factor first");
    const int nh = h->nmodels * h->H;
    if (draw < 0 || draw >= nh) return fail(SPX_ERR_ARG, "spx_get_factor: draw out of range");
    int rc = ensure_init(h);
    if (rc) return rc;
    const int Np = h->Np, Dp = h->Dp;
    const int64_t N = h->N;
    const size_t nn = (size_t)Np * Np;
    if (K) {
        if ((rc = h->scratch.reserve(nn * 8))) return rc;
        launch_cov_self(h->stream, h->Xs.d() + (size_t)draw * Np * Dp, h->s1.d() + (size_t)draw * Np,
                        h->X2s.d() + (size_t)draw * Np * Dp, h->htab.d() + (size_t)draw * SPX_HT,
                        h->scratch.d(), (int)N, Np, Dp, 1, false, dev_kind(h));
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy2D(K, (size_t)N * 8, h->scratch.p, (size_t)Np * 8, (size_t)N * 8, (size_t)N, hipMemcpyDeviceToHost));
    }
    if (L && h->factor_tiled) {
        // tile-major factor (k_lean_flow): tile (I, J) at (I nblk + J) * 4096 doubles; below the diagonal in accumulator
        // order -- value q = 4 nt + r of thread t at ((q >> 1) * 256 + t) * 2 + (q & 1) is the element
        // (16 (t >> 6) + ((t & 63) >> 4) + 4 r, 16 nt + (t & 15)) -- the diagonal tiles row-major (diag_block's Lkk)
        std::vector<double> T(nn);
        HIPCHK(hipMemcpy(T.data(), h->Lm.d() + (size_t)draw * nn, nn * 8, hipMemcpyDeviceToHost));
        const int nblk = Np / SPX_NB;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < N; ++j) {
                double v = 0.0;
                if (j <= i) {
                    const int I = (int)(i >> 6), J = (int)(j >> 6), ri = (int)(i & 63), cj = (int)(j & 63);
                    const double* tile = T.data() + ((size_t)I * nblk + J) * 4096;
                    if (I == J) v = tile[ri * 64 + cj];
                    else {
                        const int t = (ri >> 4) * 64 + (ri & 3) * 16 + (cj & 15), q = (cj >> 4) * 4 + ((ri & 15) >> 2);
                        v = tile[((q >> 1) * 256 + t) * 2 + (q & 1)];
                    }
                }
                L[i * N + j] = v;
            }
    } else if (L) {
        HIPCHK(hipMemcpy2D(L, (size_t)N * 8, h->Lm.d() + (size_t)draw * nn, (size_t)Np * 8, (size_t)N * 8, (size_t)N,
                           hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    }
    if (alpha) HIPCHK(hipMemcpy(alpha, h->alpha.d() + (size_t)draw * Np, (size_t)N * 8, hipMemcpyDeviceToHost));
    return SPX_OK;
}

// Rows [row0, row0 + nrows) of L (each N long, zeros above the diagonal) and gamma = L^-1 (vals - mean) of one draw.  What
// the pending branch needs of the factorisation of cov([comp; pend]) (GPEIChooser.py:219-249): with the P pending points
// last, the bottom P rows are [ (L_A^-1 B)^T | L_S ] -- B = cov(comp, pend), L_S the factor of the Schur complement -- so the
// posterior of the pending points is pend_m = L21 gamma[:N] + mean, pend_K = L_S L_S^T - noise I: a P x (N + P) block and one
// vector per draw instead of the whole N x N factor (1.6 MB instead of 671 MB at N = 2048, 20 draws, P = 4).
int spx_get_factor_rows(spx_handle* h, int32_t draw, int64_t row0, int64_t nrows, double* L_rows, double* gamma)
{
    if (h && h->multi) return spx_multi_get_factor_rows(h->multi, draw, row0, nrows, L_rows, gamma);
    if (!h || !h->factored) return fail(SPX_ERR_ARG, "spx_get_factor_rows: call spx_factor first");
    const int nh = h->nmodels * h->H;
    const int64_t N = h->N;
    if (draw < 0 || draw >= nh) return fail(SPX_ERR_ARG, "spx_get_factor_rows: draw out of range");
    if (row0 < 0 || nrows < 0 || row0 + nrows > N) return fail(SPX_ERR_ARG, "spx_get_factor_rows: rows out of range");
    int rc = ensure_init(h);
    if (rc) return rc;
    const int Np = h->Np;
    const size_t nn = (size_t)Np * Np;
    if (L_rows && nrows > 0 && h->factor_tiled) {
        // (tile-major factor: the tiles (I, 0 .. I) of a block row are contiguous; same element map as spx_get_factor)
        const int nblk = Np / SPX_NB;
        std::vector<double> T;
        for (int I = (int)(row0 >> 6); I <= (int)((row0 + nrows - 1) >> 6); ++I) {
            T.resize((size_t)(I + 1) * 4096);
            HIPCHK(hipMemcpy(T.data(), h->Lm.d() + (size_t)draw * nn + (size_t)I * nblk * 4096, T.size() * 8, hipMemcpyDeviceToHost));
            for (int64_t i = std::max<int64_t>(row0, (int64_t)I * 64); i < std::min<int64_t>(row0 + nrows, (int64_t)(I + 1) * 64); ++i) {
                double* out = L_rows + (size_t)(i - row0) * N;
                const int ri = (int)(i & 63);
                for (int64_t j = 0; j < N; ++j) {
                    double v = 0.0;
                    if (j <= i) {
                        const int J = (int)(j >> 6), cj = (int)(j & 63);
                        const double* tile = T.data() + (size_t)J * 4096;
                        if (I == J) v = tile[ri * 64 + cj];
                        else {
                            const int t = (ri >> 4) * 64 + (ri & 3) * 16 + (cj & 15), q = (cj >> 4) * 4 + ((ri & 15) >> 2);
                            v = tile[((q >> 1) * 256 + t) * 2 + (q & 1)];
                        }
                    }
                    out[j] = v;
                }
            }
        }
    } else if (L_rows && nrows > 0) {
        HIPCHK(hipMemcpy2D(L_rows, (size_t)N * 8, h->Lm.d() + (size_t)draw * nn + (size_t)row0 * Np, (size_t)Np * 8, (size_t)N * 8,
                           (size_t)nrows, hipMemcpyDeviceToHost));
        for (int64_t i = row0; i < row0 + nrows; ++i)
            for (int64_t j = i + 1; j < N; ++j) L_rows[(size_t)(i - row0) * N + j] = 0.0;
    }
    if (gamma) HIPCHK(hipMemcpy(gamma, h->gamma.d() + (size_t)draw * Np, (size_t)N * 8, hipMemcpyDeviceToHost));
    return SPX_OK;
}

int spx_get_cross_cov(spx_handle* h, int32_t draw, int64_t c0, int64_t nc, double* out)
{
    if (h && h->multi) return spx_multi_get_cross_cov(h->multi, draw, c0, nc, out);
    if (!h || !out || !h->factored || !h->have_cand)
        return fail(SPX_ERR_ARG, "spx_get_cross_cov: need spx_factor and candidates");
    const int nh = h->nmodels * h->H;
    if (draw < 0 || draw >= nh || c0 < 0 || nc < 1 || c0 + nc > h->M)
        return fail(SPX_ERR_ARG, "spx_get_cross_cov: range error");
    int rc = ensure_init(h);
    if (rc) return rc;
    const int Np = h->Np, Dp = h->Dp, D = h->D, hs = 3 + D;
    const int mc = (int)round_up(nc, SPX_BN);
    DevBuf cs, s2, kst;
    if ((rc = cs.reserve((size_t)mc * Dp * 8)) || (rc = s2.reserve((size_t)mc * 8)) ||
        (rc = kst.reserve((size_t)Np * mc * 8))) {
        cs.release(); s2.release(); kst.release();
        return rc;
    }
    hipStream_t s = h->stream;
    launch_scale_rows(s, h->cand.d() + (size_t)c0 * D, nc, mc, D, Dp, h->hyp.d() + 3 + (size_t)draw * hs, hs, 1, 2.0,
                      cs.d(), s2.d());
    launch_cov_cross(s, h->Xs.d() + (size_t)draw * Np * Dp, h->s1.d() + (size_t)draw * Np, cs.d(), s2.d(),
                     h->htab.d() + (size_t)draw * SPX_HT, kst.d(), (int)h->N, Np, mc, Dp, 1, dev_kind(h));
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess)
        e = hipMemcpy2D(out, (size_t)nc * 8, kst.p, (size_t)mc * 8, (size_t)nc * 8, (size_t)h->N, hipMemcpyDeviceToHost);
    cs.release(); s2.release(); kst.release();
    if (e != hipSuccess) return fail(SPX_ERR_HIP, "spx_get_cross_cov: %s", hipGetErrorString(e));
    return SPX_OK;
}

static int gp_logprob_once(spx_handle* h, double* out);

int spx_gp_logprob(spx_handle* h, double* out)
{
    if (!h || !out) return fail(SPX_ERR_ARG, "spx_gp_logprob: null");
    if (h->multi) return spx_multi_gp_logprob(h->multi, out);
    h->handoff_timeout = false;
    int rc = gp_logprob_once(h, out);
    if (rc == SPX_ERR_HIP && h->handoff_timeout && h->flow_used) {
        // never seen on a healthy device; if the one-launch data flow ever stalls (its spins are bounded), the call is
        // repeated with one launch per block column -- same kernels' arithmetic, same bits -- and the handle stays there
        note_flow_timeout(h);
        rc = gp_logprob_once(h, out);
        if (!rc) warn_flow_fallback(h);
    }
    return rc;
}

#define SPX_POLL_SENTINEL 0x7fffffff   // no not-PD flag has this value (pivots <= 2^20, time-outs < 0)
static int gp_logprob_once(spx_handle* h, double* out)
{
    // the kernels write the H values and the H not-PD flags straight into pinned host memory: no device-to-host copies
    // (two of them, to pageable memory, were 40 us of a 95 us call at N = 64)
    int rc;
    if ((rc = ensure_init(h))) return rc;
    if ((rc = h->pin_res.reserve((size_t)h->H * 12))) return rc;
    double* lp_host = (double*)h->pin_res.p;
    int* info_host = (int*)(lp_host + h->H);
    h->fused_lp = lp_host; h->fused_info = info_host;
    const bool polled = h->lean_poll != 0 && !h->timing;
    if (polled) for (int k = 0; k < h->H; ++k) info_host[k] = SPX_POLL_SENTINEL;   // (coherent pinned memory: in place before the launch is queued)
    rc = do_factor(h, true, true, true);   // K(X,X), Cholesky, forward solve -- no inverse; queued, not yet synchronised
    h->fused_lp = nullptr; h->fused_info = nullptr;
    if (rc) { h->info_clean_ptr = nullptr; return rc; }
    if ((rc = h->lp.reserve((size_t)h->H * 8))) return rc;
    std::vector<int> info(h->H);
    if (h->lean_tiled) {
        if (!h->fused_ran)   // (the fused launch has reduced the log-likelihood itself)
            launch_lean_logprob(h->stream, h->diagL.d(), h->rhs.d(), (const int*)h->info.p, lp_host, info_host, (int)h->N, h->lean_np, h->H);
        // The fused launch stores every draw's value, then -- released at system scope -- its flag into the pinned buffer: the
        // host watches the flags instead of waiting for the stream to drain (the end-of-kernel release and the completion
        // signal are ~7 us of a 42 us call).  What is still running then -- other workgroups' last instructions, the ticket
        // reset -- is ordered in front of whatever this stream is given next.  Bounded: after 20 ms (a long call at N = 8192
        // takes 10) the stream is synchronised the ordinary way.
        // (the device-side flags count as clean again only once this call is known to have ended well)
        const void* const clean_ptr = h->info_clean_ptr;
        if (h->fused_ran) h->info_clean_ptr = nullptr;
        bool seen = false;
        if (h->fused_ran && polled) {
            volatile int* fl = (volatile int*)info_host;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spin = 0;; ++spin) {
                int k = 0;
                while (k < h->H && __atomic_load_n(&fl[k], __ATOMIC_ACQUIRE) != SPX_POLL_SENTINEL) ++k;
                if (k == h->H) { seen = true; break; }
                if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
            }
        }
        if (!seen) HIPCHK(hipStreamSynchronize(h->stream));
        LAUNCHCHK();
        if (h->fused_ran && polled)      // (a launch that never ran -- refused, or a device error -- leaves its flags untouched)
            for (int k = 0; k < h->H; ++k)
                if (info_host[k] == SPX_POLL_SENTINEL)
                    return fail(SPX_ERR_HIP, "spx_gp_logprob: the launch finished without delivering draw %d", k);
        // a hand-off that timed out: a workgroup that gives up later than the draw's reducing item leaves -1 behind it -- the
        // device-side flags are then zeroed again before the next one-launch call
        if (h->fused_ran) {
            bool timed_out = false;
            for (int k = 0; k < h->H; ++k) timed_out |= info_host[k] < 0;
            if (!timed_out) h->info_clean_ptr = clean_ptr;
        }
        memcpy(out, lp_host, (size_t)h->H * 8);
        memcpy(info.data(), info_host, (size_t)h->H * sizeof(int));
    } else {
        launch_logprob(h->stream, h->Lm.d(), h->rhs.d(), (size_t)SPX_NB * h->lean_np, (const int*)h->info.p, h->lp.d(), h->lean_np, h->H);
        HIPCHK(hipMemcpyAsync(out, h->lp.p, (size_t)h->H * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(info.data(), h->info.p, (size_t)h->H * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return finish_factor(h, info, true, true);
}

static int run_grid(spx_handle* h, const double* comp, const double* vals, const double* log_durs,
                    int64_t N, int32_t D, const double* cand, int64_t M, const double* hypers,
                    const double* time_hypers, int32_t H, int32_t flags, double* ei_mean_out,
                    double* ei_draw_out, int64_t* best_idx, double* best_val)
{
    int rc;
    if ((rc = spx_set_observations(h, comp, vals, N, D))) return rc;
    if ((rc = spx_set_candidates(h, cand, M, D, 0))) return rc;
    if ((rc = spx_set_hypers(h, hypers, H))) return rc;
    if (log_durs && (rc = spx_set_time_model(h, log_durs, time_hypers))) return rc;
    if ((rc = spx_ei_step(h, flags))) return rc;
    if ((rc = spx_get_best(h, best_idx, best_val))) return rc;
    if (ei_mean_out && (rc = spx_get_ei_mean(h, ei_mean_out))) return rc;
    if (ei_draw_out && (rc = spx_get_ei_draws(h, ei_draw_out))) return rc;
    return SPX_OK;
}

int spx_ei_grid(spx_handle* h, const double* comp, const double* vals, int64_t N, int32_t D,
                const double* cand, int64_t M, const double* hypers, int32_t H, int32_t flags,
                double* ei_mean_out, double* ei_draw_out, int64_t* best_idx, double* best_val)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_ei_grid: null handle");
    return run_grid(h, comp, vals, nullptr, N, D, cand, M, hypers, nullptr, H, flags & ~SPX_FLAG_PER_SEC,
                    ei_mean_out, ei_draw_out, best_idx, best_val);
}

int spx_ei_per_sec_grid(spx_handle* h, const double* comp, const double* vals, const double* log_durs,
                        int64_t N, int32_t D, const double* cand, int64_t M, const double* hypers,
                        const double* time_hypers, int32_t H, int32_t flags, double* ei_mean_out,
                        double* ei_draw_out, int64_t* best_idx, double* best_val)
{
    if (!h || !log_durs || !time_hypers) return fail(SPX_ERR_ARG, "spx_ei_per_sec_grid: null argument");
    return run_grid(h, comp, vals, log_durs, N, D, cand, M, hypers, time_hypers, H, flags | SPX_FLAG_PER_SEC,
                    ei_mean_out, ei_draw_out, best_idx, best_val);
}

int spx_sobol_grid(spx_handle* h, const uint32_t* dirs, int32_t dim_max, int32_t dim, int64_t n,
                   int64_t skip, double* grid_out, int32_t as_candidates, double* kernel_ms)
{
    if (h && h->multi) return spx_multi_sobol_grid(h->multi, dirs, dim_max, dim, n, skip, grid_out, as_candidates, kernel_ms);
    if (!h || !dirs || dim < 1 || dim > dim_max || n < 1)
        return fail(SPX_ERR_ARG, "spx_sobol_grid: bad arguments (dim=%d of %d, n=%lld)", dim, dim_max, (long long)n);
    if (skip + n - 2 >= (1ll << 30) || skip < -(1ll << 40))
        return fail(SPX_ERR_ARG, "spx_sobol_grid: skip + n - 2 = %lld does not fit 30 direction columns "
                    "(the reference stops with 'Too many calls')", (long long)(skip + n - 2));
    if (as_candidates && h->have_obs && dim != h->D)
        return fail(SPX_ERR_ARG, "spx_sobol_grid: dim=%d but observations have D=%d", dim, h->D);
    int rc = ensure_init(h);
    if (rc) return rc;
    hipStream_t s = h->stream;
    const size_t tbytes = (size_t)dim * 30 * sizeof(uint32_t);
    if ((rc = h->sobol_dirs.reserve(tbytes))) return rc;
    DevBuf& dst = as_candidates ? h->cand : h->sobol_out;
    if ((rc = dst.reserve((size_t)n * dim * 8))) return rc;
    HIPCHK(hipMemcpyAsync(h->sobol_dirs.p, dirs, tbytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(h->ev_t0, s));
    launch_sobol_grid(s, (const uint32_t*)h->sobol_dirs.p, dim, n, skip, dst.d());
    HIPCHK(hipEventRecord(h->ev_t1, s));
    if (grid_out) HIPCHK(hipMemcpyAsync(grid_out, dst.p, (size_t)n * dim * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    LAUNCHCHK();
    if (kernel_ms) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
        *kernel_ms = ms;
    }
    if (as_candidates) {
        if (!h->have_obs) { h->D = dim; h->Dp = padded_dim(dim); }
        h->M = n; h->index_base = 0;
        h->have_cand = true; h->ran = false; h->ran_time = false;
    }
    return SPX_OK;
}

int spx_ei_grad_batch(spx_handle* h, const double* points, int32_t P, double* neg_ei, double* grad)
{
    if (!h || !points || !neg_ei || !grad || P < 1) return fail(SPX_ERR_ARG, "spx_ei_grad_batch: bad argument");
    if (h->multi) return spx_multi_ei_grad_batch(h->multi, points, P, neg_ei, grad);
    if (!h->factored) return fail(SPX_ERR_ARG, "spx_ei_grad_batch: call spx_factor (or spx_ei_grid) first");
    const int S = h->S;
    const bool per_sec = (h->nmodels == 2);   // a time model was factored: EI per second (GPEIperSecChooser.py:349-434)
    if (S > 0 && per_sec)
        return fail(SPX_ERR_ARG, "spx_ei_grad_batch: fantasies with a time model are not defined "
                    "(the reference's per-second refinement ignores pending jobs)");
    int rc = ensure_init(h);
    if (rc) return rc;
    const int H = h->H, D = h->D, Dp = h->Dp, Np = h->Np;
    const int64_t N = h->N;
    const size_t vec = (size_t)H * P * Np * 8;
    if ((rc = h->pt_x.reserve((size_t)P * D * 8))) return rc;
    if ((rc = h->pt_k.reserve(vec))) return rc;
    if ((rc = h->pt_dk.reserve(vec))) return rc;
    if ((rc = h->pt_t.reserve(vec))) return rc;
    if ((rc = h->pt_z.reserve(vec))) return rc;
    if ((rc = h->pt_out.reserve((size_t)H * P * (1 + D) * 8))) return rc;
    if (per_sec) {
        if ((rc = h->pt_kt.reserve(vec))) return rc;
        if ((rc = h->pt_dkt.reserve(vec))) return rc;
    }
    hipStream_t s = h->stream;
    if (S > 0) {
        if ((rc = h->pt_u.reserve(vec))) return rc;
        if (!h->alphaS_valid) {   // alpha_s = W^T Gamma_s = K^-1 (fant_s - mean) for every fantasy column (:501-502)
            if ((rc = h->alphaS.reserve((size_t)H * S * Np * 8))) return rc;
            launch_trimvT_multi(s, h->WT.d(), h->gammaS.d(), h->alphaS.d(), Np, H, S);
            h->alphaS_valid = true;
        }
    }
    stage_begin(h);
    if ((rc = stage_h2d(h, h->pt_x.p, points, (size_t)P * D * 8, s))) return rc;
    launch_point_cov(s, h->Xs.d(), h->s1.d(), h->hyp.d(), h->htab.d(), h->pt_x.d(), h->pt_k.d(), h->pt_dk.d(),
                     (int)N, Np, D, Dp, H, P, dev_kind(h));
    launch_trimv_multi(s, h->WT.d(), h->pt_k.d(), h->pt_t.d(), Np, H, P);        // t = W k
    launch_trimvT_multi(s, h->WT.d(), h->pt_t.d(), h->pt_z.d(), Np, H, P);       // z = W^T t = K^-1 k
    if (per_sec)   // k and dk/dr2 of the log-duration GP (table rows H..2H-1)
        launch_point_cov(s, h->Xs.d() + (size_t)H * Np * Dp, h->s1.d() + (size_t)H * Np,
                         h->hyp.d() + (size_t)H * (3 + D), h->htab.d() + (size_t)H * SPX_HT, h->pt_x.d(),
                         h->pt_kt.d(), h->pt_dkt.d(), (int)N, Np, D, Dp, H, P, dev_kind(h));
    launch_point_finish(s, h->Xs.d(), h->hyp.d(), h->htab.d(), h->alpha.d(), h->pt_k.d(), h->pt_dk.d(),
                        h->pt_t.d(), h->pt_z.d(), h->pt_x.d(), h->best, h->pt_out.d(), (int)N, Np, D, Dp, H, P,
                        per_sec ? h->pt_kt.d() : nullptr, per_sec ? h->pt_dkt.d() : nullptr, S,
                        S > 0 ? h->gammaS.d() : nullptr, S > 0 ? h->alphaS.d() : nullptr,
                        S > 0 ? h->bests.d() : nullptr, S > 0 ? h->pt_u.d() : nullptr);
    std::vector<double> out((size_t)H * P * (1 + D));
    if ((rc = stage_d2h(h, out.data(), h->pt_out.p, out.size() * 8, s))) return rc;
    LAUNCHCHK();
    // sum over draws in draw order, as grad_optimize_ei_over_hypers does (:368-380)
    for (int p = 0; p < P; ++p) {
        double f = 0.0;
        double* g = grad + (size_t)p * D;
        for (int d = 0; d < D; ++d) g[d] = 0.0;
        for (int i = 0; i < H; ++i) {
            const double* o = &out[((size_t)i * P + p) * (1 + D)];
            f += -o[0];
            for (int d = 0; d < D; ++d) g[d] = g[d] + o[1 + d];
        }
        neg_ei[p] = f;
    }
    return SPX_OK;
}

int spx_ei_grad(spx_handle* h, const double* point, double* neg_ei_sum, double* grad)
{
    return spx_ei_grad_batch(h, point, 1, neg_ei_sum, grad);
}

int spx_get_stat(spx_handle* h, const char* name, int64_t* value)
{
    if (!h || !name || !value) return fail(SPX_ERR_ARG, "spx_get_stat: null");
    if (h->multi) {
        if (!strcmp(name, "ranks_seen") || !strcmp(name, "flow_fallbacks") || !strcmp(name, "flow_rearms") || !strcmp(name, "obs_dims"))
            return spx_multi_stat(h->multi, name, value);
        return fail(SPX_ERR_ARG, "spx_get_stat: ask the per-device handles (single-GPU handles only)");
    }
    if (!strcmp(name, "flow_fallbacks")) *value = h->flow_fallbacks;          // hand-off time-outs of k_lean_flow so far
    else if (!strcmp(name, "flow_enabled")) *value = (h->lean_flow != 0 && !h->flow_demoted);   // 0 while a time-out keeps the handle on one launch per block column
    else if (!strcmp(name, "flow_rearms")) *value = h->flow_rearms;           // times the handle went back to k_lean_flow after a fallback
    else if (!strcmp(name, "ranks_seen")) *value = h->comm ? h->ranks_seen : 1;   // size of the communicator the last exchange ran on (its table has one record per rank)
    else if (!strcmp(name, "n_cu")) *value = h->n_cu;
    else if (!strcmp(name, "obs_dims")) *value = h->have_obs ? h->D : 0;   // D of the resident observations (0: none)
    else if (!strcmp(name, "last_step_fused")) *value = h->last_fused ? 1 : 0; // the last EI pass ran k_ei_fused128
    else if (!strcmp(name, "last_step_skipped_padding")) *value = h->last_skip_pad ? 1 : 0;   // ... skipped the padding of N (k_predict_gemm_tail)
    else if (!strcmp(name, "hip_runtime_version") || !strcmp(name, "hip_driver_version")) {
        // what the process runs on: a measurement names it (bench.py's line), two boxes of one pool differed in round 5
        int v = 0;
        HIPCHK(name[4] == 'r' ? hipRuntimeGetVersion(&v) : hipDriverGetVersion(&v));
        *value = v;
    } else if (!strcmp(name, "clock_khz") || !strcmp(name, "mem_clock_khz") || !strcmp(name, "wall_clock_khz")
               || !strcmp(name, "l2_bytes") || !strcmp(name, "mem_bus_bits")) {
        int rc = ensure_init(h);
        if (rc) return rc;
        int v = 0;
        const hipDeviceAttribute_t a = !strcmp(name, "clock_khz") ? hipDeviceAttributeClockRate
            : !strcmp(name, "mem_clock_khz") ? hipDeviceAttributeMemoryClockRate
            : !strcmp(name, "wall_clock_khz") ? hipDeviceAttributeWallClockRate
            : !strcmp(name, "l2_bytes") ? hipDeviceAttributeL2CacheSize : hipDeviceAttributeMemoryBusWidth;
        HIPCHK(hipDeviceGetAttribute(&v, a, h->device));
        *value = v;
    }
    else return fail(SPX_ERR_ARG, "spx_get_stat: unknown statistic '%s'", name);
    return SPX_OK;
}

int spx_get_timings(spx_handle* h, double* ms, int64_t* launches, int n)
{
    if (!h) return fail(SPX_ERR_ARG, "null handle");
    if (h->multi) return spx_multi_get_timings(h->multi, ms, launches, n);
    for (int i = 0; i < n && i < ST_COUNT; ++i) {
        if (ms) ms[i] = h->st_ms[i];
        if (launches) launches[i] = h->st_n[i];
    }
    return ST_COUNT;
}

const char* spx_timing_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

}  // extern "C"
