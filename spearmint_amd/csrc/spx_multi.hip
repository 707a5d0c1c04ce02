// Several GPUs behind ONE spx_handle, for the unmodified single-process Spearmint driver
// (include/spx.h: spx_create_multi).
//
// Decomposition (the reference has one cross-candidate step, np.argmax(np.mean(overall_ei, axis=1)),
// GPEIChooser.py:153): candidate rows are sharded contiguously over the devices, observations and
// hyper-parameter draws are replicated and every device factors all draws itself (21 ms at C3, cheaper
// than shipping 20 x 33.5 MB factors).  Per-candidate arithmetic does not depend on the shard, so the
// EI bits equal the one-GPU run.  One host thread per device drives its per-GPU engine (spx_api.hip).
//
// The single collective: every device packs its {best mean EI (fp64), global index (int64)} into a
// 16-byte record and the devices exchange them with ONE ncclAllGather (RCCL over xGMI; n x 16 bytes,
// latency-bound); then every device runs the same reduction -- first NaN wins, else the larger
// value, ties to the lower index (numpy's argmax rule; contiguous shards keep "lower index"
// meaningful) -- so all of them hold the winner.  RCCL has no MAXLOC and a MAX all-reduce on a packed
// key would lose mantissa bits; the all-gather of pairs is the exact one-collective form.
//
// librccl is bound at run time (dlopen) the first time a multi-device handle is created: the
// single-GPU path does not pay for mapping a 570 MB library, and a box without RCCL still runs it.
// Device ids that repeat (several engines on ONE GPU -- how the 1-GPU test box exercises this file)
// cannot form an RCCL communicator; the records then travel through host memory ("host" transport),
// everything else being identical.
#include <dlfcn.h>
#include <stdlib.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "spx_internal.h"

// ---- RCCL, bound at run time ----------------------------------------------------------------------
// The handful of types and constants of the (stable) NCCL / RCCL C API that this file uses, declared here so
// that libspx builds on a box without the RCCL headers (the single-GPU path never loads the library at all).
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;        // enum in nccl.h; 0 = ncclSuccess
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclChar = 0, ncclFloat64 = 8;
static const ncclRedOp_t ncclSum = 0;

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    int version = 0;                 // ncclGetVersion's code: 22707 = 2.27.7
};
// What the hand-declared types above are known to match: ncclUniqueId of 128 bytes, ncclChar = 0, ncclFloat64 = 8,
// ncclSum = 0, (sendbuf, recvbuf, count, type, [op,] comm, stream) -- the NCCL 2 API since the version code took the form
// X * 10000 + Y * 100 + Z (2.9).  Tested on this pool: RCCL 2.27.7.
#define SPX_RCCL_MIN_VERSION 21000
#define SPX_RCCL_MAX_VERSION 30000

static int load_rccl(RcclApi* r)
{
    static std::mutex mu;
    static RcclApi api;
    std::lock_guard<std::mutex> lk(mu);
    if (!api.lib) {
        const char* env = getenv("SPX_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* lib = nullptr;
        for (const char* nm : names)
            if (nm && *nm && (lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return fail(SPX_ERR_HIP, "spx_create_multi: cannot load librccl (%s)", dlerror());
        RcclApi a;
        a.lib = lib;
        *(void**)&a.CommInitAll = dlsym(lib, "ncclCommInitAll");
        *(void**)&a.GetUniqueId = dlsym(lib, "ncclGetUniqueId");
        *(void**)&a.CommInitRank = dlsym(lib, "ncclCommInitRank");
        *(void**)&a.CommDestroy = dlsym(lib, "ncclCommDestroy");
        *(void**)&a.AllGather = dlsym(lib, "ncclAllGather");
        *(void**)&a.AllReduce = dlsym(lib, "ncclAllReduce");
        *(void**)&a.GroupStart = dlsym(lib, "ncclGroupStart");
        *(void**)&a.GroupEnd = dlsym(lib, "ncclGroupEnd");
        *(void**)&a.GetErrorString = dlsym(lib, "ncclGetErrorString");
        *(void**)&a.GetVersion = dlsym(lib, "ncclGetVersion");
        // the version first: nothing else of a library we do not know is called, no struct is passed to it
        if (!a.GetVersion || a.GetVersion(&a.version) != ncclSuccess) {
            dlclose(lib);
            return fail(SPX_ERR_HIP, "spx_create_multi: the loaded librccl has no usable ncclGetVersion");
        }
        const char* any = getenv("SPX_RCCL_ANY_VERSION");
        if ((a.version < SPX_RCCL_MIN_VERSION || a.version >= SPX_RCCL_MAX_VERSION) && !(any && *any && *any != '0')) {
            const int v = a.version;
            dlclose(lib);
            return fail(SPX_ERR_HIP, "spx_create_multi: librccl reports version code %d; this build binds the NCCL 2 API "
                        "by hand and accepts %d <= code < %d (tested: 22707 = RCCL 2.27.7); set SPX_RCCL_ANY_VERSION=1 to "
                        "try anyway", v, SPX_RCCL_MIN_VERSION, SPX_RCCL_MAX_VERSION);
        }
        if (!a.CommInitAll || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.AllReduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
            dlclose(lib);
            return fail(SPX_ERR_HIP, "spx_create_multi: librccl lacks an expected symbol");
        }
        api = a;
    }
    *r = api;
    return SPX_OK;
}

// ---- one persistent host thread per device -----------------------------------------------------------
class Workers {
public:
    explicit Workers(int n) : n_(n), rc_(n, 0), err_(n)
    {
        for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // run fn(i) on worker i for every i; returns the first non-zero status (error text -> spx_last_error)
    int run(const std::function<int(int)>& fn)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            pending_ = n_;
            ++gen_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        for (int i = 0; i < n_; ++i)
            if (rc_[i]) {
                spx_err_slot() = err_[i] + " (device slot " + std::to_string(i) + ")";
                return rc_[i];
            }
        return SPX_OK;
    }

private:
    void loop(int i)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(int)>* fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            int rc;
            try {
                rc = (*fn)(i);
            } catch (const std::exception& ex) {   // e.g. std::bad_alloc: an error code, not a dead process
                rc = fail(SPX_ERR_HIP, "exception on a device thread: %s", ex.what());
            } catch (...) {
                rc = fail(SPX_ERR_HIP, "unknown exception on a device thread");
            }
            std::string e = rc ? spx_err_slot() : std::string();
            {
                std::lock_guard<std::mutex> lk(mu_);
                rc_[i] = rc;
                err_[i] = e;
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<int(int)>* fn_ = nullptr;
    uint64_t gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;
    std::vector<int> rc_;
    std::vector<std::string> err_;
};

struct SpxRecord { double val; int64_t idx; };   // idx < 0: this device scored no candidates

struct spx_multi {
    int n = 0;
    std::vector<int> devs;
    std::vector<spx_handle*> kids;
    int transport = SPX_TRANSPORT_HOST;
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    Workers* pool = nullptr;
    // candidate shards: kid i owns rows [lo[i], hi[i]) of the caller's candidate array
    int64_t M = 0, index_base = 0;
    int active = 0;
    std::vector<int64_t> lo, hi;
    int64_t N = 0;
    int D = 0, H = 0;
    // replicated hyper draws; spx_gp_logprob shards the draws over the devices and leaves the kids
    // with different subsets, so the full set is re-broadcast before the next factorisation
    std::vector<double> hyp_host, thyp_host, ldur_host;
    bool have_hyp = false, have_time = false, hyp_dirty = false;
    std::vector<int64_t> lp_lo;   // draw ranges of the last sharded spx_gp_logprob
    int lp_parts = 0;             // devices that took part in it
    bool last_was_logprob = false;
    bool ran = false;
    SpxRecord best{0.0, -1};
    // Optional 2-D partition (SURVEY.md 8(e) "hypers x candidates"; spx_set_partition): n = ph x pc devices, device i
    // evaluates the draws of hyper shard i % ph for the candidates of shard i / ph, and the single collective is ONE
    // all-reduce(SUM) of the zero-padded M-vector of per-candidate EI sums (exchange_sums).  ph = 1: candidates only.
    int ph = 1;
    std::vector<char> on;             // device i holds candidates
    std::vector<int64_t> hlo, hhi;    // device i's draws [hlo, hhi)
    bool ran2d = false;
};

static inline int kid_rh(const spx_multi* m, int i) { return i % m->ph; }
static inline int kid_rc(const spx_multi* m, int i) { return i / m->ph; }

static void shard(int64_t total, int parts, int r, int64_t* lo, int64_t* hi)
{
    const int64_t base = total / parts, extra = total % parts;
    *lo = r * base + (r < extra ? r : extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
}

// ---- the record kernels ------------------------------------------------------------------------------
__global__ void k_make_record(const double* __restrict__ val, const int64_t* __restrict__ idx, int64_t base,
                              int active, SpxRecord* __restrict__ rec)
{
    SpxRecord r;
    r.val = active ? *val : 0.0;
    r.idx = active ? (*idx + base) : -1;
    *rec = r;
}

__device__ __forceinline__ bool rec_better(const SpxRecord& a, const SpxRecord& b)
{
    if (b.idx < 0) return a.idx >= 0;
    if (a.idx < 0) return false;
    const bool an = (a.val != a.val), bn = (b.val != b.val);
    if (an || bn) {
        if (an && bn) return a.idx < b.idx;
        return an;
    }
    if (a.val > b.val) return true;
    if (a.val < b.val) return false;
    return a.idx < b.idx;
}

// the identical final reduction every device runs on the gathered table
__global__ void k_pick_record(const SpxRecord* __restrict__ table, int n, SpxRecord* __restrict__ out)
{
    SpxRecord best{0.0, -1};
    for (int i = 0; i < n; ++i)
        if (rec_better(table[i], best)) best = table[i];
    *out = best;
}

#define NCCLCHK(m, name, call)                                                                   \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess)                                                                   \
            return fail(SPX_ERR_HIP, "%s failed: %s", name, (m)->rccl.GetErrorString(r_));       \
    } while (0)

static int exchange_best(spx_multi* m)
{
    const int n = m->n;
    for (int i = 0; i < n; ++i) {
        spx_handle* k = m->kids[i];
        int rc = spx_ensure_init(k);
        if (rc) return rc;
        if ((rc = k->rec_send.reserve(sizeof(SpxRecord)))) return rc;
        if ((rc = k->rec_recv.reserve(sizeof(SpxRecord) * n))) return rc;
        if ((rc = k->rec_out.reserve(sizeof(SpxRecord)))) return rc;
        const int act = m->on[i] ? 1 : 0;
        hipLaunchKernelGGL(k_make_record, dim3(1), dim3(1), 0, k->stream, (const double*)k->am_out_val.p,
                           (const int64_t*)k->am_out_idx.p, k->index_base, act, (SpxRecord*)k->rec_send.p);
    }
    if (m->transport == SPX_TRANSPORT_RCCL) {
        NCCLCHK(m, "ncclGroupStart", m->rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            spx_handle* k = m->kids[i];
            ncclResult_t r = m->rccl.AllGather(k->rec_send.p, k->rec_recv.p, sizeof(SpxRecord), ncclChar, m->comms[i], k->stream);
            if (r != ncclSuccess) {
                (void)m->rccl.GroupEnd();      // the group must not stay open on this thread: the next exchange would nest inside it
                return fail(SPX_ERR_HIP, "ncclAllGather failed: %s", m->rccl.GetErrorString(r));
            }
        }
        NCCLCHK(m, "ncclGroupEnd (the group of ncclAllGather / ncclAllReduce calls)", m->rccl.GroupEnd());
    } else {
        std::vector<SpxRecord> table(n);
        for (int i = 0; i < n; ++i) {
            spx_handle* k = m->kids[i];
            HIPCHK(hipSetDevice(k->device));
            HIPCHK(hipMemcpyAsync(&table[i], k->rec_send.p, sizeof(SpxRecord), hipMemcpyDeviceToHost, k->stream));
            HIPCHK(hipStreamSynchronize(k->stream));
        }
        for (int i = 0; i < n; ++i) {
            spx_handle* k = m->kids[i];
            HIPCHK(hipSetDevice(k->device));
            HIPCHK(hipMemcpyAsync(k->rec_recv.p, table.data(), sizeof(SpxRecord) * n, hipMemcpyHostToDevice, k->stream));
            HIPCHK(hipStreamSynchronize(k->stream));   // `table` is pageable host memory
        }
    }
    for (int i = 0; i < n; ++i) {
        spx_handle* k = m->kids[i];
        HIPCHK(hipSetDevice(k->device));
        hipLaunchKernelGGL(k_pick_record, dim3(1), dim3(1), 0, k->stream, (const SpxRecord*)k->rec_recv.p, n,
                           (SpxRecord*)k->rec_out.p);
    }
    std::vector<SpxRecord> outs(n);
    for (int i = 0; i < n; ++i) {
        spx_handle* k = m->kids[i];
        HIPCHK(hipSetDevice(k->device));
        HIPCHK(hipMemcpyAsync(&outs[i], k->rec_out.p, sizeof(SpxRecord), hipMemcpyDeviceToHost, k->stream));
        HIPCHK(hipStreamSynchronize(k->stream));
        LAUNCHCHK();
    }
    for (int i = 1; i < n; ++i)   // every device must hold the same winner
        if (outs[i].idx != outs[0].idx || memcmp(&outs[i].val, &outs[0].val, 8))
            return fail(SPX_ERR_HIP, "multi-GPU argmax: device slots 0 and %d disagree (%lld vs %lld)", i,
                        (long long)outs[0].idx, (long long)outs[i].idx);
    m->best = outs[0];
    return SPX_OK;
}

// The collective of the 2-D partition: every device scatters sum_{its draws} EI[c, h] of its candidates into a
// zero-padded M-vector (k_sum_over_draws: numpy's summation order over the local draws), ONE all-reduce(SUM) of the
// M doubles completes the sums (ncclAllReduce in one group over the devices' streams; through host memory, summed in
// device order, when the device ids repeat), and every device divides by the total number of draws and takes numpy's
// argmax over all M candidates -- no per-draw EI leaves the devices, no host-side reduction.
static int exchange_sums(spx_multi* m)
{
    const int n = m->n;
    const int64_t M = m->M;
    for (int i = 0; i < n; ++i) {
        spx_handle* k = m->kids[i];
        int rc = spx_ensure_init(k);
        if (rc) return rc;
        const int nab = argmax_blocks(M);
        if ((rc = k->ei_sum_full.reserve((size_t)M * 8))) return rc;
        if ((rc = k->am_val.reserve((size_t)nab * 8))) return rc;
        if ((rc = k->am_idx.reserve((size_t)nab * 8))) return rc;
        if ((rc = k->am_out_val.reserve(8))) return rc;
        if ((rc = k->am_out_idx.reserve(8))) return rc;
        HIPCHK(hipMemsetAsync(k->ei_sum_full.p, 0, (size_t)M * 8, k->stream));
        if (m->on[i])
            launch_sum_over_draws(k->stream, k->ei_draw.d(), k->ei_sum_full.d() + m->lo[i], k->M, round_up(k->M, SPX_BN), k->H);
    }
    if (m->transport == SPX_TRANSPORT_RCCL) {
        NCCLCHK(m, "ncclGroupStart", m->rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            spx_handle* k = m->kids[i];
            ncclResult_t r = m->rccl.AllReduce(k->ei_sum_full.p, k->ei_sum_full.p, (size_t)M, ncclFloat64, ncclSum, m->comms[i], k->stream);
            if (r != ncclSuccess) {
                (void)m->rccl.GroupEnd();      // (as above)
                return fail(SPX_ERR_HIP, "ncclAllReduce failed: %s", m->rccl.GetErrorString(r));
            }
        }
        NCCLCHK(m, "ncclGroupEnd (the group of ncclAllGather / ncclAllReduce calls)", m->rccl.GroupEnd());
    } else {
        std::vector<double> acc((size_t)M), tmp((size_t)M);
        for (int i = 0; i < n; ++i) {
            spx_handle* k = m->kids[i];
            HIPCHK(hipSetDevice(k->device));
            HIPCHK(hipMemcpyAsync(i ? tmp.data() : acc.data(), k->ei_sum_full.p, (size_t)M * 8, hipMemcpyDeviceToHost, k->stream));
            HIPCHK(hipStreamSynchronize(k->stream));
            if (i) for (int64_t c = 0; c < M; ++c) acc[c] += tmp[c];
        }
        for (int i = 0; i < n; ++i) {
            spx_handle* k = m->kids[i];
            HIPCHK(hipSetDevice(k->device));
            HIPCHK(hipMemcpyAsync(k->ei_sum_full.p, acc.data(), (size_t)M * 8, hipMemcpyHostToDevice, k->stream));
            HIPCHK(hipStreamSynchronize(k->stream));   // `acc` is pageable host memory
        }
    }
    std::vector<SpxRecord> outs(n);
    for (int i = 0; i < n; ++i) {
        spx_handle* k = m->kids[i];
        HIPCHK(hipSetDevice(k->device));
        launch_div_scalar(k->stream, k->ei_sum_full.d(), M, (double)m->H);
        launch_argmax(k->stream, k->ei_sum_full.d(), M, k->am_val.d(), (int64_t*)k->am_idx.p, k->am_out_val.d(),
                      (int64_t*)k->am_out_idx.p);
        HIPCHK(hipMemcpyAsync(&outs[i].val, k->am_out_val.p, 8, hipMemcpyDeviceToHost, k->stream));
        HIPCHK(hipMemcpyAsync(&outs[i].idx, k->am_out_idx.p, 8, hipMemcpyDeviceToHost, k->stream));
    }
    for (int i = 0; i < n; ++i) {
        spx_handle* k = m->kids[i];
        HIPCHK(hipSetDevice(k->device));
        HIPCHK(hipStreamSynchronize(k->stream));
        LAUNCHCHK();
    }
    for (int i = 1; i < n; ++i)   // every device must hold the same winner
        if (outs[i].idx != outs[0].idx || memcmp(&outs[i].val, &outs[0].val, 8))
            return fail(SPX_ERR_HIP, "multi-GPU argmax (2-D partition): device slots 0 and %d disagree (%lld vs %lld)", i,
                        (long long)outs[0].idx, (long long)outs[i].idx);
    m->best.val = outs[0].val;
    m->best.idx = outs[0].idx + m->index_base;
    return SPX_OK;
}

// candidate shard of device i for M candidates: pc = n / ph shards, device i takes shard i / ph
static void plan_cands(const spx_multi* m, int64_t M, std::vector<int64_t>& lo, std::vector<int64_t>& hi,
                       std::vector<char>& on, int* active)
{
    const int pc = m->n / m->ph;
    const int actc = (int)std::min<int64_t>(pc, M);
    int act = 0;
    lo.assign(m->n, M); hi.assign(m->n, M); on.assign(m->n, 0);
    for (int i = 0; i < m->n; ++i) {
        const int rc = kid_rc(m, i);
        if (rc < actc) { shard(M, actc, rc, &lo[i], &hi[i]); on[i] = 1; ++act; }
    }
    *active = act;
}

// draw shard of device i: all H draws (ph = 1) or shard i % ph of ph
static int plan_draws(spx_multi* m, int H)
{
    if (m->ph > 1 && H < m->ph)
        return fail(SPX_ERR_ARG, "2-D partition: %d hyper shards need at least that many draws (H=%d)", m->ph, H);
    m->hlo.assign(m->n, 0); m->hhi.assign(m->n, H);
    if (m->ph > 1)
        for (int i = 0; i < m->n; ++i) shard(H, m->ph, kid_rh(m, i), &m->hlo[i], &m->hhi[i]);
    return SPX_OK;
}

// (re-)broadcast the hyper draws -- each device its shard -- e.g. after a sharded spx_gp_logprob
static int push_hypers(spx_multi* m)
{
    const int hs = 3 + m->D;
    return m->pool->run([m, hs](int i) {
        const int64_t a = m->hlo[i], b = m->hhi[i];
        int r = spx_set_hypers(m->kids[i], m->hyp_host.data() + (size_t)a * hs, (int)(b - a));
        if (!r && m->have_time)
            r = spx_set_time_model(m->kids[i], m->ldur_host.data(), m->thyp_host.data() + (size_t)a * hs);
        return r;
    });
}

static int sync_hypers(spx_multi* m)
{
    if (!m->hyp_dirty) return SPX_OK;
    if (!m->have_hyp) return fail(SPX_ERR_ARG, "spx_factor: observations and hypers must be set first");
    int rc = push_hypers(m);
    if (!rc) m->hyp_dirty = false;
    return rc;
}

// ---- one process per GPU: a communicator attached to a single-GPU handle -------------------------------
// spx_comm_attach makes spx_ei_run end with the same exchange as the multi-device handle, across
// processes: k_make_record -> ONE ncclAllGather of the 16-byte records on the handle's stream ->
// k_pick_record; spx_get_best then returns the global winner on every rank.  With a partition set
// (spx_set_partition) the collective is instead ONE ncclAllReduce(SUM) of the M_total-vector of EI sums.
struct spx_comm {
    RcclApi rccl;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
};

static int comm_exchange_sums(spx_handle* k)
{
    spx_comm* c = k->comm;
    const int64_t Mt = k->part_M;
    if (k->index_base < 0 || k->index_base + k->M > Mt)
        return fail(SPX_ERR_ARG, "spx_ei_run: candidate rows [%lld, %lld) do not fit the partition's M_total=%lld",
                    (long long)k->index_base, (long long)(k->index_base + k->M), (long long)Mt);
    int rc;
    const int nab = argmax_blocks(Mt);
    if ((rc = k->ei_sum_full.reserve((size_t)Mt * 8))) return rc;
    if ((rc = k->am_val.reserve((size_t)nab * 8))) return rc;
    if ((rc = k->am_idx.reserve((size_t)nab * 8))) return rc;
    HIPCHK(hipMemsetAsync(k->ei_sum_full.p, 0, (size_t)Mt * 8, k->stream));
    launch_sum_over_draws(k->stream, k->ei_draw.d(), k->ei_sum_full.d() + k->index_base, k->M, round_up(k->M, SPX_BN), k->H);
    ncclResult_t r = c->rccl.AllReduce(k->ei_sum_full.p, k->ei_sum_full.p, (size_t)Mt, ncclFloat64, ncclSum, c->comm, k->stream);
    if (r != ncclSuccess) return fail(SPX_ERR_HIP, "ncclAllReduce failed: %s", c->rccl.GetErrorString(r));
    launch_div_scalar(k->stream, k->ei_sum_full.d(), Mt, (double)k->part_H);
    launch_argmax(k->stream, k->ei_sum_full.d(), Mt, k->am_val.d(), (int64_t*)k->am_idx.p, k->am_out_val.d(),
                  (int64_t*)k->am_out_idx.p);
    HIPCHK(hipMemcpyAsync(&k->best_val, k->am_out_val.p, 8, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(hipMemcpyAsync(&k->best_idx, k->am_out_idx.p, 8, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(hipStreamSynchronize(k->stream));
    LAUNCHCHK();
    k->best_idx -= k->index_base;   // spx_get_best adds the base back: the index is global
    k->ran_2d = true;
    return SPX_OK;
}

int spx_comm_exchange(spx_handle* k)
{
    spx_comm* c = k->comm;
    int rc = spx_ensure_init(k);
    if (rc) return rc;
    if (k->part_M > 0) return comm_exchange_sums(k);
    k->ran_2d = false;
    if ((rc = k->rec_send.reserve(sizeof(SpxRecord)))) return rc;
    if ((rc = k->rec_recv.reserve(sizeof(SpxRecord) * c->nranks))) return rc;
    if ((rc = k->rec_out.reserve(sizeof(SpxRecord)))) return rc;
    hipLaunchKernelGGL(k_make_record, dim3(1), dim3(1), 0, k->stream, (const double*)k->am_out_val.p,
                       (const int64_t*)k->am_out_idx.p, k->index_base, 1, (SpxRecord*)k->rec_send.p);
    ncclResult_t r = c->rccl.AllGather(k->rec_send.p, k->rec_recv.p, sizeof(SpxRecord), ncclChar, c->comm, k->stream);
    if (r != ncclSuccess) return fail(SPX_ERR_HIP, "ncclAllGather failed: %s", c->rccl.GetErrorString(r));
    hipLaunchKernelGGL(k_pick_record, dim3(1), dim3(1), 0, k->stream, (const SpxRecord*)k->rec_recv.p, c->nranks,
                       (SpxRecord*)k->rec_out.p);
    SpxRecord out;
    HIPCHK(hipMemcpyAsync(&out, k->rec_out.p, sizeof out, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(hipStreamSynchronize(k->stream));
    LAUNCHCHK();
    k->best_idx = out.idx - k->index_base;   // spx_get_best adds the base back
    k->best_val = out.val;
    k->ranks_seen = c->nranks;               // the table k_pick_record reduced had this many records
    return SPX_OK;
}

void spx_comm_release(spx_handle* k)
{
    if (!k->comm) return;
    if (k->comm->comm) (void)k->comm->rccl.CommDestroy(k->comm->comm);
    delete k->comm;
    k->comm = nullptr;
}

static int multi_set_partition(spx_multi* m, int32_t hyper_shards);

extern "C" {

int spx_rccl_version(int32_t* version_code)
{
    RcclApi api;
    int rc = load_rccl(&api);
    if (rc) return rc;
    if (version_code) *version_code = api.version;
    return SPX_OK;
}

int spx_create_multi(const int* device_ids, int32_t n_dev, spx_handle** out)
{
    return spx_create_multi_transport(device_ids, n_dev, SPX_TRANSPORT_NONE, out);
}

int spx_create_multi_transport(const int* device_ids, int32_t n_dev, int32_t transport, spx_handle** out)
{
    if (transport != SPX_TRANSPORT_NONE && transport != SPX_TRANSPORT_RCCL && transport != SPX_TRANSPORT_HOST)
        return fail(SPX_ERR_ARG, "spx_create_multi_transport: transport must be SPX_TRANSPORT_NONE (by the device list), _RCCL or _HOST");
    if (!device_ids || !out || n_dev < 1 || n_dev > 64)
        return fail(SPX_ERR_ARG, "spx_create_multi: bad arguments (n_dev=%d)", n_dev);
    for (int i = 0; i < n_dev; ++i)
        if (device_ids[i] < 0) return fail(SPX_ERR_ARG, "spx_create_multi: negative device id");
    bool distinct = true;
    for (int i = 0; i < n_dev; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j]) distinct = false;
    spx_multi* m = new spx_multi();
    m->n = n_dev;
    m->devs.assign(device_ids, device_ids + n_dev);
    // transport asked for explicitly (spx_create_multi_transport): HOST = records through host memory even on distinct
    // GPUs; RCCL = the RCCL code path even for repeated device ids (a real librccl refuses such a communicator; the
    // thread-rendezvous stand-in of the tests, tests/c/fake_rccl.hip through SPX_RCCL_LIB, accepts it -- how a 1-GPU box
    // runs the group section with n > 1)
    const bool want_host = transport == SPX_TRANSPORT_HOST, want_rccl = transport == SPX_TRANSPORT_RCCL;
    m->transport = ((distinct && !want_host) || want_rccl) ? SPX_TRANSPORT_RCCL : SPX_TRANSPORT_HOST;
    if (m->transport == SPX_TRANSPORT_RCCL) {
        int rc = load_rccl(&m->rccl);
        if (rc) { delete m; return rc; }
        m->comms.resize(n_dev);
        ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), n_dev, device_ids);
        if (r != ncclSuccess) {
            rc = fail(SPX_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n_dev, m->rccl.GetErrorString(r));
            delete m;
            return rc;
        }
    }
    for (int i = 0; i < n_dev; ++i) {
        spx_handle* k = nullptr;
        int rc = spx_create(device_ids[i], &k);
        if (rc) { spx_multi_destroy(m); return rc; }
        m->kids.push_back(k);
    }
    m->lo.assign(n_dev, 0);
    m->hi.assign(n_dev, 0);
    m->on.assign(n_dev, 0);
    m->hlo.assign(n_dev, 0);
    m->hhi.assign(n_dev, 0);
    m->pool = new Workers(n_dev);
    spx_handle* front = new spx_handle();
    front->device = device_ids[0];
    front->multi = m;
    *out = front;
    return SPX_OK;
}

int spx_multi_query(spx_handle* h, int32_t* n_dev, int32_t* transport, int32_t* device_ids, int32_t cap)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_multi_query: null handle");
    if (!h->multi) {
        if (n_dev) *n_dev = 1;
        if (transport) *transport = SPX_TRANSPORT_NONE;
        if (device_ids && cap > 0) device_ids[0] = h->device;
        return SPX_OK;
    }
    return spx_multi_info(h->multi, n_dev, transport, device_ids, cap);
}

int spx_comm_unique_id(char* id_out)
{
    if (!id_out) return fail(SPX_ERR_ARG, "spx_comm_unique_id: null");
    RcclApi api;
    int rc = load_rccl(&api);
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(SPX_ERR_HIP, "ncclGetUniqueId failed: %s", api.GetErrorString(r));
    static_assert(sizeof(ncclUniqueId) == SPX_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof id);
    return SPX_OK;
}

int spx_comm_attach(spx_handle* h, const char* id, int32_t nranks, int32_t rank)
{
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(SPX_ERR_ARG, "spx_comm_attach: bad arguments (nranks=%d, rank=%d)", nranks, rank);
    if (h->multi) return fail(SPX_ERR_ARG, "spx_comm_attach: a multi-device handle has its own communicator");
    int rc = spx_ensure_init(h);   // hipSetDevice(h->device): the communicator binds to the current device
    if (rc) return rc;
    spx_comm_release(h);
    spx_comm* c = new spx_comm();
    if ((rc = load_rccl(&c->rccl))) { delete c; return rc; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = c->rccl.CommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        rc = fail(SPX_ERR_HIP, "ncclCommInitRank(%d of %d) failed: %s", rank, nranks, c->rccl.GetErrorString(r));
        delete c;
        return rc;
    }
    c->nranks = nranks;
    c->rank = rank;
    h->comm = c;
    h->ran = false;
    return SPX_OK;
}

int spx_set_partition(spx_handle* h, int32_t hyper_shards, int64_t M_total, int32_t H_total)
{
    if (!h) return fail(SPX_ERR_ARG, "spx_set_partition: null handle");
    if (hyper_shards < 1) return fail(SPX_ERR_ARG, "spx_set_partition: hyper_shards=%d", hyper_shards);
    if (h->multi) return multi_set_partition(h->multi, hyper_shards);
    if (M_total < 0 || (M_total > 0 && H_total < 1))
        return fail(SPX_ERR_ARG, "spx_set_partition: bad totals (M_total=%lld, H_total=%d)", (long long)M_total, H_total);
    if (h->comm && M_total > 0 && h->comm->nranks % hyper_shards)
        return fail(SPX_ERR_ARG, "spx_set_partition: %d hyper shards do not divide %d ranks", hyper_shards, h->comm->nranks);
    h->part_ph = hyper_shards;
    h->part_M = M_total;
    h->part_H = H_total;
    h->ran = false;
    h->ran_2d = false;
    return SPX_OK;
}

}  // extern "C"

int spx_multi_info(spx_multi* m, int32_t* n_dev, int32_t* transport, int32_t* device_ids, int32_t cap)
{
    if (n_dev) *n_dev = m->n;
    if (transport) *transport = m->transport;
    for (int i = 0; device_ids && i < cap && i < m->n; ++i) device_ids[i] = m->devs[i];
    return SPX_OK;
}

int spx_multi_stat(spx_multi* m, const char* name, int64_t* value)
{
    if (!strcmp(name, "ranks_seen")) { *value = m->n; return SPX_OK; }   // the device slots of the handle (= the size of its communicator)
    if (!strcmp(name, "obs_dims")) { *value = m->N > 0 ? m->D : 0; return SPX_OK; }   // D of the resident observations (0: none)
    if (!strcmp(name, "flow_fallbacks") || !strcmp(name, "flow_rearms")) {   // summed over the devices' handles
        int64_t sum = 0;
        for (spx_handle* k : m->kids) {
            int64_t v = 0;
            int rc = spx_get_stat(k, name, &v);
            if (rc) return rc;
            sum += v;
        }
        *value = sum;
        return SPX_OK;
    }
    return fail(SPX_ERR_ARG, "spx_get_stat: unknown statistic '%s' of a multi-device handle", name);
}

void spx_multi_destroy(spx_multi* m)
{
    if (!m) return;
    delete m->pool;
    m->pool = nullptr;
    for (size_t i = 0; i < m->comms.size(); ++i)
        if (m->comms[i]) (void)m->rccl.CommDestroy(m->comms[i]);
    for (spx_handle* k : m->kids) spx_destroy(k);
    delete m;
}

// the partition decides who holds which draws and candidates: both must be set again afterwards
static int multi_set_partition(spx_multi* m, int32_t hyper_shards)
{
    if (m->n % hyper_shards)
        return fail(SPX_ERR_ARG, "spx_set_partition: %d hyper shards do not divide %d devices", hyper_shards, m->n);
    if (hyper_shards != m->ph) {
        m->ph = hyper_shards;
        m->have_hyp = false; m->have_time = false; m->hyp_dirty = false;
        m->M = 0; m->active = 0; m->on.assign(m->n, 0);
        m->ran = false; m->ran2d = false; m->last_was_logprob = false;
        // the devices' own state must not survive either (a stale factorisation of the full draw set)
        return m->pool->run([m](int i) { m->kids[i]->have_hyp = false; m->kids[i]->have_cand = false;
                                         m->kids[i]->factored = false; m->kids[i]->ran = false; return (int)SPX_OK; });
    }
    return SPX_OK;
}

int spx_multi_set_option(spx_multi* m, const char* name, int64_t value)
{
    int rc = m->pool->run([=](int i) { return spx_set_option(m->kids[i], name, value); });
    // "covar" invalidates the devices' factorisations and results: so it does here
    if (!rc && !strcmp(name, "covar")) { m->ran = false; m->ran2d = false; m->last_was_logprob = false; }
    return rc;
}

int spx_multi_set_observations(spx_multi* m, const double* comp, const double* vals, int64_t N, int32_t D)
{
    int rc = m->pool->run([=](int i) { return spx_set_observations(m->kids[i], comp, vals, N, D); });
    if (rc) return rc;
    if (m->have_hyp && D != m->D) { m->have_hyp = false; m->have_time = false; }
    if (D != m->D) { m->M = 0; m->active = 0; m->on.assign(m->n, 0); }
    m->N = N; m->D = D;
    m->have_time = false; m->ran = false; m->last_was_logprob = false;
    return SPX_OK;
}

int spx_multi_set_candidates(spx_multi* m, const double* cand, int64_t M, int32_t D, int64_t index_base)
{
    if (!cand || M < 1 || D < 1)
        return fail(SPX_ERR_ARG, "spx_set_candidates: bad arguments (M=%lld, D=%d)", (long long)M, D);
    // the new shard bounds are committed only when every device has taken its rows
    std::vector<int64_t> lo, hi;
    std::vector<char> on;
    int act = 0;
    plan_cands(m, M, lo, hi, on, &act);
    const int64_t* lop = lo.data(); const int64_t* hip_ = hi.data(); const char* onp = on.data();
    int rc = m->pool->run([=](int i) {
        if (!onp[i]) return (int)SPX_OK;
        return spx_set_candidates(m->kids[i], cand + (size_t)lop[i] * D, hip_[i] - lop[i], D, index_base + lop[i]);
    });
    if (rc) { m->M = 0; m->active = 0; m->on.assign(m->n, 0); m->ran = false; return rc; }
    m->lo = lo; m->hi = hi; m->on = on;
    m->M = M; m->index_base = index_base; m->active = act; m->ran = false;
    if (!m->D) m->D = D;
    return SPX_OK;
}

int spx_multi_set_hypers(spx_multi* m, const double* hypers, int32_t H)
{
    if (!hypers || H < 1) return fail(SPX_ERR_ARG, "spx_set_hypers: bad arguments (H=%d)", H);
    if (!m->D) return fail(SPX_ERR_ARG, "spx_set_hypers: call spx_set_observations first");
    int rc = plan_draws(m, H);
    if (rc) return rc;
    m->H = H;
    m->hyp_host.assign(hypers, hypers + (size_t)H * (3 + m->D));
    m->have_time = false;
    rc = push_hypers(m);
    if (rc) { m->have_hyp = false; return rc; }
    m->have_hyp = true; m->hyp_dirty = false; m->ran = false; m->last_was_logprob = false;
    return SPX_OK;
}

int spx_multi_set_time_model(spx_multi* m, const double* log_durs, const double* time_hypers)
{
    int rc = sync_hypers(m);
    if (rc) return rc;
    if (!log_durs || !time_hypers) {
        m->have_time = false;
        m->ran = false;
        return m->pool->run([=](int i) { return spx_set_time_model(m->kids[i], nullptr, nullptr); });
    }
    if (!m->have_hyp) return fail(SPX_ERR_ARG, "spx_set_time_model: set observations and hypers first");
    m->ldur_host.assign(log_durs, log_durs + m->N);
    m->thyp_host.assign(time_hypers, time_hypers + (size_t)m->H * (3 + m->D));
    const int hs = 3 + m->D;
    rc = m->pool->run([=](int i) {
        return spx_set_time_model(m->kids[i], m->ldur_host.data(), m->thyp_host.data() + (size_t)m->hlo[i] * hs);
    });
    if (rc) { m->have_time = false; return rc; }
    m->have_time = true;
    m->ran = false;
    return SPX_OK;
}

int spx_multi_factor(spx_multi* m)
{
    int rc = sync_hypers(m);
    if (rc) return rc;
    m->last_was_logprob = false;
    m->ran = false;
    return m->pool->run([=](int i) { return spx_factor(m->kids[i]); });
}

int spx_multi_set_fantasies(spx_multi* m, const double* fant, const double* bests, int32_t S)
{
    m->ran = false;
    if (m->ph > 1 && fant && bests && S > 0)
        return fail(SPX_ERR_ARG, "spx_set_fantasies: not available in the 2-D partition (spx_set_partition)");
    return m->pool->run([=](int i) { return spx_set_fantasies(m->kids[i], fant, bests, S); });
}

int spx_multi_ei_run(spx_multi* m, int32_t flags)
{
    if (m->active < 1) return fail(SPX_ERR_ARG, "spx_ei_run: no candidates set");
    int rc = m->pool->run([=](int i) { return m->on[i] ? spx_ei_run(m->kids[i], flags) : (int)SPX_OK; });
    if (rc) return rc;
    if ((rc = (m->ph > 1) ? exchange_sums(m) : exchange_best(m))) return rc;
    m->ran = true;
    m->ran2d = m->ph > 1;
    return SPX_OK;
}

int spx_multi_get_best(spx_multi* m, int64_t* best_idx, double* best_val)
{
    if (!m->ran) return fail(SPX_ERR_ARG, "spx_get_best: no results (call spx_ei_run)");
    if (best_idx) *best_idx = m->best.idx;
    if (best_val) *best_val = m->best.val;
    return SPX_OK;
}

int spx_multi_get_ei_mean(spx_multi* m, double* out)
{
    if (!out || !m->ran) return fail(SPX_ERR_ARG, "spx_get_ei_mean: no results / null output");
    if (m->ran2d) {   // every device holds the whole vector
        spx_handle* k = m->kids[0];
        HIPCHK(hipSetDevice(k->device));
        HIPCHK(hipMemcpy(out, k->ei_sum_full.p, (size_t)m->M * 8, hipMemcpyDeviceToHost));
        return SPX_OK;
    }
    return m->pool->run([=](int i) { return m->on[i] ? spx_get_ei_mean(m->kids[i], out + m->lo[i]) : (int)SPX_OK; });
}

int spx_multi_get_ei_draws(spx_multi* m, double* out)
{
    if (!out || !m->ran) return fail(SPX_ERR_ARG, "spx_get_ei_draws: no results / null output");
    if (m->ph == 1)
        return m->pool->run([=](int i) {
            return m->on[i] ? spx_get_ei_draws(m->kids[i], out + (size_t)m->lo[i] * m->H) : (int)SPX_OK;
        });
    // 2-D: device i holds the block [lo, hi) x [hlo, hhi) of overall_ei (M x H)
    return m->pool->run([=](int i) {
        if (!m->on[i]) return (int)SPX_OK;
        const int64_t mc = m->hi[i] - m->lo[i], hl = m->hhi[i] - m->hlo[i];
        std::vector<double> tmp((size_t)mc * hl);
        int rc = spx_get_ei_draws(m->kids[i], tmp.data());
        if (rc) return rc;
        for (int64_t c = 0; c < mc; ++c)
            memcpy(out + (size_t)(m->lo[i] + c) * m->H + m->hlo[i], &tmp[(size_t)c * hl], (size_t)hl * 8);
        return (int)SPX_OK;
    });
}

// device i's local index of global draw `draw` of the objective model, or -1
static inline int local_draw(const spx_multi* m, int i, int32_t draw)
{
    return (draw >= m->hlo[i] && draw < m->hhi[i]) ? (int)(draw - m->hlo[i]) : -1;
}

int spx_multi_get_moments(spx_multi* m, int32_t draw, double* func_m, double* func_v)
{
    if (!m->ran) return fail(SPX_ERR_ARG, "spx_get_moments: run spx_ei_run with SPX_FLAG_KEEP_MOMENTS first");
    if (draw < 0 || draw >= m->H) return fail(SPX_ERR_ARG, "spx_get_moments: draw out of range");
    return m->pool->run([=](int i) {
        const int ld = local_draw(m, i, draw);
        return (m->on[i] && ld >= 0) ? spx_get_moments(m->kids[i], ld, func_m ? func_m + m->lo[i] : nullptr,
                                                       func_v ? func_v + m->lo[i] : nullptr)
                                     : (int)SPX_OK;
    });
}

int spx_multi_get_time_mean(spx_multi* m, int32_t draw, double* out)
{
    if (!out || !m->ran) return fail(SPX_ERR_ARG, "spx_get_time_mean: no results / null output");
    if (draw < 0 || draw >= m->H) return fail(SPX_ERR_ARG, "spx_get_time_mean: draw out of range");
    return m->pool->run([=](int i) {
        const int ld = local_draw(m, i, draw);
        return (m->on[i] && ld >= 0) ? spx_get_time_mean(m->kids[i], ld, out + m->lo[i]) : (int)SPX_OK;
    });
}

// draw (objective model: 0..H-1, time model: H..2H-1) -> (device of candidate shard 0 that holds it, local draw)
static int owner_of_draw(const spx_multi* m, int32_t draw, int* kid, int* ld)
{
    const bool tm = draw >= m->H;
    const int32_t d = tm ? draw - m->H : draw;
    if (draw < 0 || d >= m->H) return fail(SPX_ERR_ARG, "draw out of range");
    for (int i = 0; i < m->ph; ++i) {
        const int l = local_draw(m, i, d);
        if (l >= 0) { *kid = i; *ld = l + (tm ? (int)(m->hhi[i] - m->hlo[i]) : 0); return SPX_OK; }
    }
    return fail(SPX_ERR_ARG, "draw out of range");
}

int spx_multi_get_factor_rows(spx_multi* m, int32_t draw, int64_t row0, int64_t nrows, double* L_rows, double* gamma)
{
    int kid = 0, ld = draw;   // replicated (ph = 1): every device holds every draw
    if (m->ph > 1) { int rc = owner_of_draw(m, draw, &kid, &ld); if (rc) return rc; }
    return spx_get_factor_rows(m->kids[kid], ld, row0, nrows, L_rows, gamma);
}

int spx_multi_get_factor(spx_multi* m, int32_t draw, double* K, double* L, double* alpha)
{
    int kid = 0, ld = draw;   // replicated (ph = 1): every device holds every draw
    if (m->ph > 1) { int rc = owner_of_draw(m, draw, &kid, &ld); if (rc) return rc; }
    return spx_get_factor(m->kids[kid], ld, K, L, alpha);
}

int spx_multi_get_cross_cov(spx_multi* m, int32_t draw, int64_t c0, int64_t nc, double* out)
{
    if (!out || c0 < 0 || nc < 1 || c0 + nc > m->M) return fail(SPX_ERR_ARG, "spx_get_cross_cov: range error");
    int kid0 = 0, ld = draw;
    if (m->ph > 1) { int rc = owner_of_draw(m, draw, &kid0, &ld); if (rc) return rc; }
    const int64_t N = m->N;
    for (int i = kid0; i < m->n; i += m->ph) {       // the devices of this draw shard, one per candidate shard
        if (!m->on[i]) continue;
        const int64_t a = std::max(c0, m->lo[i]), b = std::min(c0 + nc, m->hi[i]);
        if (a >= b) continue;
        std::vector<double> tmp((size_t)N * (b - a));
        int rc = spx_get_cross_cov(m->kids[i], ld, a - m->lo[i], b - a, tmp.data());
        if (rc) return rc;
        for (int64_t r = 0; r < N; ++r)
            memcpy(out + (size_t)r * nc + (a - c0), &tmp[(size_t)r * (b - a)], (size_t)(b - a) * 8);
    }
    return SPX_OK;
}

// the draws of a log-likelihood batch are independent: shard them over the devices
int spx_multi_gp_logprob(spx_multi* m, double* out)
{
    if (!out) return fail(SPX_ERR_ARG, "spx_gp_logprob: null");
    if (!m->have_hyp) return fail(SPX_ERR_ARG, "spx_factor: observations and hypers must be set first");
    const int H = m->H, hs = 3 + m->D;
    const int parts = std::min(m->n, H);
    m->lp_lo.assign(m->n + 1, H);
    for (int i = 0; i < parts; ++i) {
        int64_t lo, hi;
        shard(H, parts, i, &lo, &hi);
        m->lp_lo[i] = lo;
    }
    m->lp_parts = parts;
    m->hyp_dirty = true;
    m->last_was_logprob = true;
    m->ran = false;
    return m->pool->run([=](int i) {
        if (i >= parts) return (int)SPX_OK;
        const int64_t lo = m->lp_lo[i], hi = m->lp_lo[i + 1];
        int rc = spx_set_hypers(m->kids[i], m->hyp_host.data() + (size_t)lo * hs, (int)(hi - lo));
        if (!rc) rc = spx_gp_logprob(m->kids[i], out + lo);
        return rc;
    });
}

int spx_multi_ei_grad_batch(spx_multi* m, const double* points, int32_t P, double* neg_ei, double* grad)
{
    if (m->ph > 1) return fail(SPX_ERR_ARG, "spx_ei_grad_batch: not available in the 2-D partition (spx_set_partition)");
    if (m->hyp_dirty) return fail(SPX_ERR_ARG, "spx_ei_grad_batch: call spx_factor (or spx_ei_grid) first");
    const int parts = std::min<int>(m->n, P);
    const int D = m->D;
    return m->pool->run([=](int i) {
        if (i >= parts) return (int)SPX_OK;
        int64_t lo, hi;
        shard(P, parts, i, &lo, &hi);
        return spx_ei_grad_batch(m->kids[i], points + (size_t)lo * D, (int)(hi - lo), neg_ei + lo, grad + (size_t)lo * D);
    });
}

int spx_multi_sobol_grid(spx_multi* m, const uint32_t* dirs, int32_t dim_max, int32_t dim, int64_t n,
                         int64_t skip, double* grid_out, int32_t as_candidates, double* kernel_ms)
{
    if (!as_candidates) return spx_sobol_grid(m->kids[0], dirs, dim_max, dim, n, skip, grid_out, 0, kernel_ms);
    if (n < 1) return fail(SPX_ERR_ARG, "spx_sobol_grid: bad arguments (n=%lld)", (long long)n);
    // the point of seed s has a closed form, so every device generates its own shard in place
    std::vector<int64_t> lo, hi;
    std::vector<char> on;
    int act = 0;
    plan_cands(m, n, lo, hi, on, &act);
    const int64_t* lop = lo.data(); const int64_t* hip_ = hi.data(); const char* onp = on.data();
    std::vector<double> ms(m->n, 0.0);
    double* msp = ms.data();
    int rc = m->pool->run([=](int i) {
        if (!onp[i]) return (int)SPX_OK;
        // (with a 2-D partition several devices hold the same shard: each writes the same rows of grid_out)
        int r = spx_sobol_grid(m->kids[i], dirs, dim_max, dim, hip_[i] - lop[i], skip + lop[i],
                               (grid_out && kid_rh(m, i) == 0) ? grid_out + (size_t)lop[i] * dim : nullptr, 1, msp + i);
        if (!r) m->kids[i]->index_base = lop[i];
        return r;
    });
    if (rc) { m->M = 0; m->active = 0; m->on.assign(m->n, 0); m->ran = false; return rc; }
    if (kernel_ms) {
        *kernel_ms = 0.0;
        for (double v : ms) *kernel_ms = std::max(*kernel_ms, v);
    }
    m->lo = lo; m->hi = hi; m->on = on;
    m->M = n; m->index_base = 0; m->active = act; m->ran = false;
    if (!m->D) m->D = dim;
    return SPX_OK;
}

int spx_multi_not_pd_info(spx_multi* m, int32_t* draw, int32_t* pivot)
{
    int d = -1, p = -1;
    if (m->last_was_logprob) {
        // only the devices that took part in the last batch: an idle one still remembers an older call
        for (int i = 0; i < m->lp_parts && d < 0; ++i) {
            int32_t di = -1, pi = -1;
            spx_not_pd_info(m->kids[i], &di, &pi);
            if (di >= 0) { d = (int)m->lp_lo[i] + di; p = pi; }
        }
    } else {
        // factorisation: the first failing draw in global numbering (objective draws 0..H-1, time model H..2H-1);
        // the devices of candidate shard 0 hold every draw between them
        for (int i = 0; i < m->ph && i < m->n; ++i) {
            int32_t di = -1, pi = -1;
            spx_not_pd_info(m->kids[i], &di, &pi);
            if (di < 0) continue;
            const int hl = (int)(m->hhi[i] - m->hlo[i]);
            const int gd = (m->ph > 1) ? (di >= hl ? m->H + (int)m->hlo[i] + (di - hl) : (int)m->hlo[i] + di) : di;
            if (d < 0 || gd < d) { d = gd; p = pi; }
        }
    }
    if (draw) *draw = d;
    if (pivot) *pivot = p;
    return SPX_OK;
}

// per stage the MAXIMUM over the devices (a straggler GPU shows; launch counts are device 0's)
int spx_multi_get_timings(spx_multi* m, double* ms, int64_t* launches, int n)
{
    const int ns = spx_get_timings(m->kids[0], ms, launches, n);
    if (ms)
        for (int i = 1; i < m->n; ++i) {
            std::vector<double> t((size_t)std::max(n, 1), 0.0);
            spx_get_timings(m->kids[i], t.data(), nullptr, n);
            for (int j = 0; j < n && j < ns; ++j) ms[j] = std::max(ms[j], t[j]);
        }
    return ns;
}
