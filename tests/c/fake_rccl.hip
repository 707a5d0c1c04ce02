// TEST INFRASTRUCTURE -- a thread-rendezvous stand-in for librccl, NOT part of the product.
//
// libspx binds RCCL at run time (spearmint_amd/csrc/spx_multi.hip: dlopen, SPX_RCCL_LIB names the file).  A one-GPU test box
// cannot form a real communicator of more than one rank, so the code around the collectives -- the group section of a
// multi-device handle (ncclCommInitAll + ncclGroupStart / ncclAllGather x P / ncclGroupEnd), the record table for P > 1,
// the 2-D partition's ncclAllReduce, spx_comm_attach(nranks = P) -- would never run there.  This library implements the ten
// entry points libspx binds with the NCCL 2 signatures, for P ranks that live in ONE process on ANY devices (all on
// device 0 on the test box):
//
//   * ranks of a communicator rendezvous through a mutex / condition variable: inside an ncclGroup the calling thread
//     holds all P calls; outside one (spx_comm_attach: one host thread per rank) every rank's call blocks until all P
//     have arrived (20 s, then an error -- never a hang);
//   * the data moves on the callers' own streams, ordered by events exactly as a collective orders them: every rank's
//     send buffer is staged (so in-place calls work), every stream waits for all stagings, copies / sums, and nobody's
//     stream goes on before everybody has read;
//   * ncclAllReduce(SUM, float64) adds the ranks' vectors in rank order (((r0 + r1) + r2) + ...).
//
// FAKE_RCCL_FAIL=allgather|allreduce|groupend|initall|initrank makes that entry point return an error (its string says
// "injected"), FAKE_RCCL_VERSION overrides the version code (default 22707), fake_rccl_stats() reports what ran.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <vector>

typedef int ncclResult_t;
enum { kOk = 0, kUnhandled = 1, kSystem = 2, kInternal = 3, kInvalidArg = 4, kInvalidUsage = 5, kInjected = 100, kTimeout = 101 };
enum { kAllGather = 1, kAllReduce = 2 };
#define FAKE_MAX_RANKS 64

struct Op {
    int kind = 0;
    const void* send = nullptr;
    void* recv = nullptr;
    size_t bytes = 0;      // per rank (all-gather) / of the whole vector (all-reduce)
    hipStream_t stream = nullptr;
};

struct World {
    int n = 0;
    int refs = 0;
    std::vector<int> dev;
    std::vector<void*> stage;
    std::vector<size_t> stage_cap;
    std::vector<hipEvent_t> ready, done;
    // rendezvous of calls made outside a group (one host thread per rank)
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Op> slot;
    std::vector<char> have;
    int arrived = 0;
    uint64_t gen = 0;
    ncclResult_t last = kOk;
};

struct ncclComm { World* w; int rank; };
typedef ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;

static std::mutex g_mu;
static std::map<uint64_t, World*> g_by_id;
static uint64_t g_next_id = 1;
static int g_stat_allgather = 0, g_stat_allreduce = 0, g_stat_groups = 0, g_stat_max_ranks = 0, g_stat_collectives = 0;
static thread_local int t_group_depth = 0;
static thread_local std::vector<std::pair<ncclComm*, Op>>* t_pending = nullptr;

static bool inject(const char* what)
{
    const char* e = getenv("FAKE_RCCL_FAIL");
    return e && !strcmp(e, what);
}

struct PtrPack { const double* p[FAKE_MAX_RANKS]; };
__global__ void k_fake_sum(double* __restrict__ out, PtrPack src, int n_ranks, size_t count)
{
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    double s = src.p[0][c];
    for (int r = 1; r < n_ranks; ++r) s = s + src.p[r][c];
    out[c] = s;
}

static World* new_world(int n, const int* devs)
{
    World* w = new World();
    w->n = n;
    w->dev.assign(n, 0);
    for (int i = 0; i < n; ++i) w->dev[i] = devs ? devs[i] : -1;
    w->stage.assign(n, nullptr);
    w->stage_cap.assign(n, 0);
    w->ready.assign(n, nullptr);
    w->done.assign(n, nullptr);
    w->slot.assign(n, Op());
    w->have.assign(n, 0);
    return w;
}

static void free_world(World* w)
{
    for (int i = 0; i < w->n; ++i) {
        if (w->dev[i] >= 0) (void)hipSetDevice(w->dev[i]);
        if (w->stage[i]) (void)hipFree(w->stage[i]);
        if (w->ready[i]) (void)hipEventDestroy(w->ready[i]);
        if (w->done[i]) (void)hipEventDestroy(w->done[i]);
    }
    delete w;
}

#define FHIP(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return kUnhandled; } } while (0)

// all P calls of ONE collective are known: put the work on the callers' streams
static ncclResult_t run_collective(World* w, const std::vector<Op>& ops)
{
    const int n = w->n;
    for (int j = 1; j < n; ++j)
        if (ops[j].kind != ops[0].kind || ops[j].bytes != ops[0].bytes) return kInvalidUsage;
    const size_t bytes = ops[0].bytes;
    for (int j = 0; j < n; ++j) {          // 1. stage every rank's contribution on its own stream
        FHIP(hipSetDevice(w->dev[j]));
        if (!w->ready[j]) {
            FHIP(hipEventCreateWithFlags(&w->ready[j], hipEventDisableTiming));
            FHIP(hipEventCreateWithFlags(&w->done[j], hipEventDisableTiming));
        }
        if (w->stage_cap[j] < bytes) {
            if (w->stage[j]) { FHIP(hipDeviceSynchronize()); FHIP(hipFree(w->stage[j])); w->stage[j] = nullptr; }
            FHIP(hipMalloc(&w->stage[j], bytes));
            w->stage_cap[j] = bytes;
        }
        FHIP(hipMemcpyAsync(w->stage[j], ops[j].send, bytes, hipMemcpyDefault, ops[j].stream));
        FHIP(hipEventRecord(w->ready[j], ops[j].stream));
    }
    for (int i = 0; i < n; ++i) {          // 2. every rank reads all of them
        FHIP(hipSetDevice(w->dev[i]));
        for (int j = 0; j < n; ++j) FHIP(hipStreamWaitEvent(ops[i].stream, w->ready[j], 0));
        if (ops[0].kind == kAllGather) {
            for (int j = 0; j < n; ++j)
                FHIP(hipMemcpyAsync((char*)ops[i].recv + (size_t)j * bytes, w->stage[j], bytes, hipMemcpyDefault, ops[i].stream));
        } else {
            PtrPack pk;
            for (int j = 0; j < n; ++j) pk.p[j] = (const double*)w->stage[j];
            const size_t count = bytes / 8;
            hipLaunchKernelGGL(k_fake_sum, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ops[i].stream, (double*)ops[i].recv, pk, n, count);
            FHIP(hipGetLastError());
        }
        FHIP(hipEventRecord(w->done[i], ops[i].stream));
    }
    for (int j = 0; j < n; ++j) {          // 3. nobody goes on (and restages) before everybody has read
        FHIP(hipSetDevice(w->dev[j]));
        for (int i = 0; i < n; ++i) FHIP(hipStreamWaitEvent(ops[j].stream, w->done[i], 0));
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_stat_collectives;
        if (n > g_stat_max_ranks) g_stat_max_ranks = n;
    }
    return kOk;
}

// a call outside a group: wait for the other ranks' calls of the same collective, the last one to arrive runs it
static ncclResult_t rendezvous(ncclComm* c, const Op& op)
{
    World* w = c->w;
    if (w->dev[c->rank] < 0) { int d = 0; (void)hipGetDevice(&d); w->dev[c->rank] = d; }
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->have[c->rank]) return kInvalidUsage;          // two calls of one rank in one collective
    w->slot[c->rank] = op;
    w->have[c->rank] = 1;
    const uint64_t my_gen = w->gen;
    if (++w->arrived == w->n) {
        w->last = run_collective(w, w->slot);
        w->arrived = 0;
        w->have.assign(w->n, 0);
        ++w->gen;
        w->cv.notify_all();
        return w->last;
    }
    if (!w->cv.wait_for(lk, std::chrono::seconds(20), [&] { return w->gen != my_gen; })) {
        w->have[c->rank] = 0;                              // gave up: leave the slot as it was found
        --w->arrived;
        return kTimeout;
    }
    return w->last;
}

static ncclResult_t submit(ncclComm* c, const Op& op)
{
    if (!c || !c->w) return kInvalidArg;
    if (t_group_depth > 0) {
        if (!t_pending) t_pending = new std::vector<std::pair<ncclComm*, Op>>();
        t_pending->push_back(std::make_pair(c, op));
        return kOk;
    }
    if (c->w->n == 1) {
        if (c->w->dev[0] < 0) { int d = 0; (void)hipGetDevice(&d); c->w->dev[0] = d; }
        std::vector<Op> one(1, op);
        return run_collective(c->w, one);
    }
    return rendezvous(c, op);
}

extern "C" {

ncclResult_t ncclGetVersion(int* v)
{
    const char* e = getenv("FAKE_RCCL_VERSION");
    *v = e ? atoi(e) : 22707;
    return kOk;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case kOk: return "fake rccl: success";
        case kInjected: return "fake rccl: injected failure (FAKE_RCCL_FAIL)";
        case kTimeout: return "fake rccl: rendezvous timed out (a rank never made its call)";
        case kInvalidUsage: return "fake rccl: invalid usage (mismatched calls of one collective)";
        case kInvalidArg: return "fake rccl: invalid argument";
        default: return "fake rccl: HIP error inside the stand-in";
    }
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs)
{
    if (inject("initall")) return kInjected;
    if (!comms || n < 1 || n > FAKE_MAX_RANKS) return kInvalidArg;
    World* w = new_world(n, devs);
    if (!devs) for (int i = 0; i < n; ++i) w->dev[i] = i;
    w->refs = n;
    for (int i = 0; i < n; ++i) comms[i] = new ncclComm{w, i};
    return kOk;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> lk(g_mu);
    const uint64_t v = g_next_id++;
    memcpy(id->internal, "FAKERCCL", 8);
    memcpy(id->internal + 8, &v, 8);
    return kOk;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (inject("initrank")) return kInjected;
    if (!comm || nranks < 1 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks || memcmp(id.internal, "FAKERCCL", 8))
        return kInvalidArg;
    uint64_t v;
    memcpy(&v, id.internal + 8, 8);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_mu);
    World*& w = g_by_id[v];
    if (!w) w = new_world(nranks, nullptr);
    if (w->n != nranks) return kInvalidUsage;
    w->dev[rank] = dev;
    ++w->refs;
    *comm = new ncclComm{w, rank};
    return kOk;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return kOk;
    World* w = c->w;
    bool last;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        last = --w->refs == 0;
        if (last)
            for (auto it = g_by_id.begin(); it != g_by_id.end(); ++it)
                if (it->second == w) { g_by_id.erase(it); break; }
    }
    if (last) free_world(w);
    delete c;
    return kOk;
}

ncclResult_t ncclGroupStart(void)
{
    ++t_group_depth;
    return kOk;
}

ncclResult_t ncclGroupEnd(void)
{
    if (t_group_depth <= 0) return kInvalidUsage;
    if (--t_group_depth > 0) return kOk;
    std::vector<std::pair<ncclComm*, Op>> calls;
    if (t_pending) calls.swap(*t_pending);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_stat_groups;
    }
    if (inject("groupend")) return kInjected;
    // the calls of the group, collective by collective: the k-th call of every rank of a world belongs together
    std::map<World*, std::vector<std::vector<Op>>> per;    // world -> [collective][rank]
    std::map<World*, std::vector<int>> count;
    for (auto& pc : calls) {
        World* w = pc.first->w;
        std::vector<int>& cnt = count[w];
        if (cnt.empty()) cnt.assign(w->n, 0);
        const int k = cnt[pc.first->rank]++;
        std::vector<std::vector<Op>>& v = per[w];
        if ((int)v.size() <= k) v.resize(k + 1, std::vector<Op>(w->n));
        v[k][pc.first->rank] = pc.second;
    }
    for (auto& kv : per) {
        World* w = kv.first;
        for (int r = 0; r < w->n; ++r)
            if (count[w][r] != count[w][0]) return kInvalidUsage;       // a rank of the communicator is missing from the group
        for (auto& ops : kv.second) {
            ncclResult_t rc = run_collective(w, ops);
            if (rc != kOk) return rc;
        }
    }
    return kOk;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t c, hipStream_t s)
{
    if (inject("allgather")) return kInjected;
    static const int width[] = {1, 1, 4, 4, 8, 8, 2, 4, 8};       // ncclInt8 ... ncclFloat64
    if (dtype < 0 || dtype > 8) return kInvalidArg;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_stat_allgather;
    }
    Op op;
    op.kind = kAllGather; op.send = send; op.recv = recv; op.bytes = count * width[dtype]; op.stream = s;
    return submit(c, op);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int redop, ncclComm_t c, hipStream_t s)
{
    if (inject("allreduce")) return kInjected;
    if (dtype != 8 || redop != 0) return kInvalidArg;            // what libspx uses: ncclFloat64, ncclSum
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_stat_allreduce;
    }
    Op op;
    op.kind = kAllReduce; op.send = send; op.recv = recv; op.bytes = count * 8; op.stream = s;
    return submit(c, op);
}

// what ran, for the tests: calls of ncclAllGather / ncclAllReduce, groups closed, collectives executed, most ranks in one
void fake_rccl_stats(int* allgather, int* allreduce, int* groups, int* collectives, int* max_ranks)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (allgather) *allgather = g_stat_allgather;
    if (allreduce) *allreduce = g_stat_allreduce;
    if (groups) *groups = g_stat_groups;
    if (collectives) *collectives = g_stat_collectives;
    if (max_ranks) *max_ranks = g_stat_max_ranks;
}

}  // extern "C"
