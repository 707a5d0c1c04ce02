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
#include <rccl/rccl.h>
#include <stdlib.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "spx_internal.h"

// ---- RCCL, bound at run time ----------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

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
        *(void**)&a.GroupStart = dlsym(lib, "ncclGroupStart");
        *(void**)&a.GroupEnd = dlsym(lib, "ncclGroupEnd");
        *(void**)&a.GetErrorString = dlsym(lib, "ncclGetErrorString");
        if (!a.CommInitAll || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
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
            int rc = (*fn)(i);
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
    bool last_was_logprob = false;
    bool ran = false;
    SpxRecord best{0.0, -1};
};

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

#define NCCLCHK(m, call)                                                                         \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess)                                                                   \
            return fail(SPX_ERR_HIP, "%s failed: %s", #call, (m)->rccl.GetErrorString(r_));      \
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
        const int act = i < m->active;
        hipLaunchKernelGGL(k_make_record, dim3(1), dim3(1), 0, k->stream, (const double*)k->am_out_val.p,
                           (const int64_t*)k->am_out_idx.p, k->index_base, act, (SpxRecord*)k->rec_send.p);
    }
    if (m->transport == SPX_TRANSPORT_RCCL) {
        NCCLCHK(m, m->rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            spx_handle* k = m->kids[i];
            NCCLCHK(m, m->rccl.AllGather(k->rec_send.p, k->rec_recv.p, sizeof(SpxRecord), ncclChar, m->comms[i],
                                         k->stream));
        }
        NCCLCHK(m, m->rccl.GroupEnd());
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
        HIPCHK(hipGetLastError());
    }
    for (int i = 1; i < n; ++i)   // every device must hold the same winner
        if (outs[i].idx != outs[0].idx || memcmp(&outs[i].val, &outs[0].val, 8))
            return fail(SPX_ERR_HIP, "multi-GPU argmax: device slots 0 and %d disagree (%lld vs %lld)", i,
                        (long long)outs[0].idx, (long long)outs[i].idx);
    m->best = outs[0];
    return SPX_OK;
}

// re-broadcast the full hyper set after a sharded spx_gp_logprob
static int sync_hypers(spx_multi* m)
{
    if (!m->hyp_dirty) return SPX_OK;
    if (!m->have_hyp) return fail(SPX_ERR_ARG, "spx_factor: observations and hypers must be set first");
    int rc = m->pool->run([m](int i) {
        int r = spx_set_hypers(m->kids[i], m->hyp_host.data(), m->H);
        if (!r && m->have_time) r = spx_set_time_model(m->kids[i], m->ldur_host.data(), m->thyp_host.data());
        return r;
    });
    if (!rc) m->hyp_dirty = false;
    return rc;
}

// ---- one process per GPU: a communicator attached to a single-GPU handle -------------------------------
// spx_comm_attach makes spx_ei_run end with the same exchange as the multi-device handle, across
// processes: k_make_record -> ONE ncclAllGather of the 16-byte records on the handle's stream ->
// k_pick_record; spx_get_best then returns the global winner on every rank.
struct spx_comm {
    RcclApi rccl;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
};

int spx_comm_exchange(spx_handle* k)
{
    spx_comm* c = k->comm;
    int rc = spx_ensure_init(k);
    if (rc) return rc;
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
    HIPCHK(hipGetLastError());
    k->best_idx = out.idx - k->index_base;   // spx_get_best adds the base back
    k->best_val = out.val;
    return SPX_OK;
}

void spx_comm_release(spx_handle* k)
{
    if (!k->comm) return;
    if (k->comm->comm) (void)k->comm->rccl.CommDestroy(k->comm->comm);
    delete k->comm;
    k->comm = nullptr;
}

extern "C" {

int spx_create_multi(const int* device_ids, int32_t n_dev, spx_handle** out)
{
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
    const char* force = getenv("SPX_MULTI_TRANSPORT");
    m->transport = (distinct && !(force && !strcmp(force, "host"))) ? SPX_TRANSPORT_RCCL : SPX_TRANSPORT_HOST;
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

}  // extern "C"

int spx_multi_info(spx_multi* m, int32_t* n_dev, int32_t* transport, int32_t* device_ids, int32_t cap)
{
    if (n_dev) *n_dev = m->n;
    if (transport) *transport = m->transport;
    for (int i = 0; device_ids && i < cap && i < m->n; ++i) device_ids[i] = m->devs[i];
    return SPX_OK;
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

int spx_multi_set_option(spx_multi* m, const char* name, int64_t value)
{
    return m->pool->run([=](int i) { return spx_set_option(m->kids[i], name, value); });
}

int spx_multi_set_observations(spx_multi* m, const double* comp, const double* vals, int64_t N, int32_t D)
{
    int rc = m->pool->run([=](int i) { return spx_set_observations(m->kids[i], comp, vals, N, D); });
    if (rc) return rc;
    if (m->have_hyp && D != m->D) { m->have_hyp = false; m->have_time = false; }
    if (D != m->D) { m->M = 0; m->active = 0; }
    m->N = N; m->D = D;
    m->have_time = false; m->ran = false; m->last_was_logprob = false;
    return SPX_OK;
}

int spx_multi_set_candidates(spx_multi* m, const double* cand, int64_t M, int32_t D, int64_t index_base)
{
    if (!cand || M < 1 || D < 1)
        return fail(SPX_ERR_ARG, "spx_set_candidates: bad arguments (M=%lld, D=%d)", (long long)M, D);
    const int act = (int)std::min<int64_t>(m->n, M);
    for (int i = 0; i < m->n; ++i) {
        if (i < act) shard(M, act, i, &m->lo[i], &m->hi[i]);
        else m->lo[i] = m->hi[i] = M;
    }
    int rc = m->pool->run([=](int i) {
        if (i >= act) return (int)SPX_OK;
        return spx_set_candidates(m->kids[i], cand + (size_t)m->lo[i] * D, m->hi[i] - m->lo[i], D,
                                  index_base + m->lo[i]);
    });
    if (rc) return rc;
    m->M = M; m->index_base = index_base; m->active = act; m->ran = false;
    if (!m->D) m->D = D;
    return SPX_OK;
}

int spx_multi_set_hypers(spx_multi* m, const double* hypers, int32_t H)
{
    int rc = m->pool->run([=](int i) { return spx_set_hypers(m->kids[i], hypers, H); });
    if (rc) return rc;
    m->H = H;
    m->hyp_host.assign(hypers, hypers + (size_t)H * (3 + m->D));
    m->have_hyp = true; m->have_time = false; m->hyp_dirty = false; m->ran = false; m->last_was_logprob = false;
    return SPX_OK;
}

int spx_multi_set_time_model(spx_multi* m, const double* log_durs, const double* time_hypers)
{
    int rc = sync_hypers(m);
    if (rc) return rc;
    rc = m->pool->run([=](int i) { return spx_set_time_model(m->kids[i], log_durs, time_hypers); });
    if (rc) return rc;
    m->have_time = log_durs && time_hypers;
    if (m->have_time) {
        m->ldur_host.assign(log_durs, log_durs + m->N);
        m->thyp_host.assign(time_hypers, time_hypers + (size_t)m->H * (3 + m->D));
    }
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
    return m->pool->run([=](int i) { return spx_set_fantasies(m->kids[i], fant, bests, S); });
}

int spx_multi_ei_run(spx_multi* m, int32_t flags)
{
    if (m->active < 1) return fail(SPX_ERR_ARG, "spx_ei_run: no candidates set");
    int rc = m->pool->run([=](int i) { return i < m->active ? spx_ei_run(m->kids[i], flags) : (int)SPX_OK; });
    if (rc) return rc;
    if ((rc = exchange_best(m))) return rc;
    m->ran = true;
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
    return m->pool->run([=](int i) { return i < m->active ? spx_get_ei_mean(m->kids[i], out + m->lo[i]) : (int)SPX_OK; });
}

int spx_multi_get_ei_draws(spx_multi* m, double* out)
{
    if (!out || !m->ran) return fail(SPX_ERR_ARG, "spx_get_ei_draws: no results / null output");
    return m->pool->run([=](int i) {
        return i < m->active ? spx_get_ei_draws(m->kids[i], out + (size_t)m->lo[i] * m->H) : (int)SPX_OK;
    });
}

int spx_multi_get_moments(spx_multi* m, int32_t draw, double* func_m, double* func_v)
{
    if (!m->ran) return fail(SPX_ERR_ARG, "spx_get_moments: run spx_ei_run with SPX_FLAG_KEEP_MOMENTS first");
    return m->pool->run([=](int i) {
        return i < m->active ? spx_get_moments(m->kids[i], draw, func_m ? func_m + m->lo[i] : nullptr,
                                               func_v ? func_v + m->lo[i] : nullptr)
                             : (int)SPX_OK;
    });
}

int spx_multi_get_time_mean(spx_multi* m, int32_t draw, double* out)
{
    if (!out || !m->ran) return fail(SPX_ERR_ARG, "spx_get_time_mean: no results / null output");
    return m->pool->run([=](int i) { return i < m->active ? spx_get_time_mean(m->kids[i], draw, out + m->lo[i]) : (int)SPX_OK; });
}

int spx_multi_get_factor(spx_multi* m, int32_t draw, double* K, double* L, double* alpha)
{
    return spx_get_factor(m->kids[0], draw, K, L, alpha);   // replicated: every device holds every draw
}

int spx_multi_get_cross_cov(spx_multi* m, int32_t draw, int64_t c0, int64_t nc, double* out)
{
    if (!out || c0 < 0 || nc < 1 || c0 + nc > m->M) return fail(SPX_ERR_ARG, "spx_get_cross_cov: range error");
    const int64_t N = m->N;
    for (int i = 0; i < m->active; ++i) {
        const int64_t a = std::max(c0, m->lo[i]), b = std::min(c0 + nc, m->hi[i]);
        if (a >= b) continue;
        std::vector<double> tmp((size_t)N * (b - a));
        int rc = spx_get_cross_cov(m->kids[i], draw, a - m->lo[i], b - a, tmp.data());
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
    const int act = (int)std::min<int64_t>(m->n, n);
    for (int i = 0; i < m->n; ++i) {
        if (i < act) shard(n, act, i, &m->lo[i], &m->hi[i]);
        else m->lo[i] = m->hi[i] = n;
    }
    std::vector<double> ms(m->n, 0.0);
    double* msp = ms.data();
    int rc = m->pool->run([=](int i) {
        if (i >= act) return (int)SPX_OK;
        int r = spx_sobol_grid(m->kids[i], dirs, dim_max, dim, m->hi[i] - m->lo[i], skip + m->lo[i],
                               grid_out ? grid_out + (size_t)m->lo[i] * dim : nullptr, 1, msp + i);
        if (!r) m->kids[i]->index_base = m->lo[i];
        return r;
    });
    if (rc) return rc;
    if (kernel_ms) {
        *kernel_ms = 0.0;
        for (double v : ms) *kernel_ms = std::max(*kernel_ms, v);
    }
    m->M = n; m->index_base = 0; m->active = act; m->ran = false;
    if (!m->D) m->D = dim;
    return SPX_OK;
}

int spx_multi_not_pd_info(spx_multi* m, int32_t* draw, int32_t* pivot)
{
    int d = -1, p = -1;
    if (m->last_was_logprob) {
        for (int i = 0; i < m->n && d < 0; ++i) {
            int32_t di = -1, pi = -1;
            spx_not_pd_info(m->kids[i], &di, &pi);
            if (di >= 0) { d = (int)m->lp_lo[i] + di; p = pi; }
        }
    } else {
        int32_t di = -1, pi = -1;
        spx_not_pd_info(m->kids[0], &di, &pi);
        d = di; p = pi;
    }
    if (draw) *draw = d;
    if (pivot) *pivot = p;
    return SPX_OK;
}

int spx_multi_get_timings(spx_multi* m, double* ms, int64_t* launches, int n)
{
    return spx_get_timings(m->kids[0], ms, launches, n);
}
