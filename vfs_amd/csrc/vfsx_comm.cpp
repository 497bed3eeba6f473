// vfsx_comm.cpp -- the cross-GPU exchange of the row-sharded search behind the C ABI:
// one process per GPU, each rank scans its contiguous row range, the per-rank top-k lists meet in ONE
// RCCL all-gather over xGMI (nq * k * 24 bytes per rank: latency-bound, so one collective per query
// batch, never per query) and every rank merges them by (distance, global row).
//
// No reference counterpart: the reference is one process and one thread (rank.rs:90-92).
//
// RCCL is bound at run time (dlopen of librccl.so.1: in a Python host that already loaded PyTorch's copy --
// same soname -- that copy is reused; a C/C++/Rust host gets /opt/rocm's), so libvfsx.so itself has no link-time
// dependency on it and single-GPU hosts never load it.
#include <dlfcn.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include <rccl/rccl.h> // types and prototypes only; every call goes through the table below

#include "vfsx_internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle)
                break;
        }
        if (!r.handle) {
            r.error = std::string("librccl.so.1 cannot be loaded: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty())
                r.error = std::string("librccl.so.1 has no symbol ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    });
    return r;
}

int fail(int code, const std::string &msg)
{
    vfsx::set_last_error(msg);
    return code;
}

int rccl_ready()
{
    Rccl &r = rccl();
    if (!r.error.empty())
        return fail(VFSX_ERR_NO_DEVICE, "vfsx_comm: " + r.error);
    return VFSX_OK;
}

std::string rccl_msg(const char *what, ncclResult_t e)
{
    return std::string("vfsx_comm: ") + what + ": " + rccl().GetErrorString(e);
}

} // namespace

struct vfsx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    vfsx_hit *d_gather = nullptr; // [world][nq][k]: every rank's ranked lists, rank-major
    size_t gather_hits = 0;
    // host-buffer entry point (vfsx_search_sharded_host): its own stream, query and result buffers
    hipStream_t stream = nullptr;
    float *d_queries = nullptr;
    size_t query_floats = 0;
    vfsx_hit *d_out = nullptr;
    size_t out_hits = 0;
    vfsx_hit *h_out = nullptr; // pinned: the D2H of the merged hits
    size_t h_out_hits = 0;
    uint64_t allocs = 0; // hipMalloc / hipHostMalloc calls made for this communicator (vfsx_comm_alloc_count)
};

namespace {

// Sizes vfsx_comm_init reserves up front, so that calls within them never allocate: hits per rank and batch
// (nq * k; the default is BASELINE config 4's 4096 queries x top-10) and query floats of the host entry point.
constexpr size_t kDefaultHitsPerRank = 4096u * 10u;
constexpr size_t kDefaultQueryFloats = 4096u * 1024u;

size_t env_size(const char *name, size_t dflt)
{
    const char *v = std::getenv(name);
    if (!v || !*v)
        return dflt;
    char *end = nullptr;
    const unsigned long long x = std::strtoull(v, &end, 10);
    return (end && *end == 0) ? (size_t)x : dflt;
}

template <typename T>
int reserve_buf(vfsx_comm *cm, T *&p, size_t &have, size_t want, const char *what, bool pinned = false)
{
    if (want <= have)
        return VFSX_OK;
    if (p)
        (void)(pinned ? hipHostFree(p) : hipFree(p));
    p = nullptr;
    have = 0;
    const hipError_t e = pinned ? hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault)
                                : hipMalloc((void **)&p, want * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(VFSX_ERR_INTERNAL, std::string("vfsx_comm: allocation of the ") + what + ": " + hipGetErrorString(e));
    }
    ++cm->allocs;
    have = want;
    return VFSX_OK;
}

int reserve_all(vfsx_comm *cm, size_t hits_per_rank, size_t query_floats)
{
    if (int rc = reserve_buf(cm, cm->d_gather, cm->gather_hits, hits_per_rank * (size_t)cm->world, "gather buffer"))
        return rc;
    if (int rc = reserve_buf(cm, cm->d_out, cm->out_hits, hits_per_rank, "result buffer"))
        return rc;
    if (int rc = reserve_buf(cm, cm->h_out, cm->h_out_hits, hits_per_rank, "pinned result buffer", true))
        return rc;
    if (query_floats)
        if (int rc = reserve_buf(cm, cm->d_queries, cm->query_floats, query_floats, "query buffer"))
            return rc;
    return VFSX_OK;
}

// ncclCommInitRank is a collective that cannot be cancelled from outside: a peer that died before joining
// leaves every other rank inside it for good.  The watchdog bounds the wait: when the init has not returned
// after VFSX_COMM_INIT_TIMEOUT_S seconds (default 300, 0 = wait for ever) it says so on stderr and ends the
// process with status 3 -- the one thing that frees the GPU and lets a launcher tear the job down.
struct InitWatchdog {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    std::thread th;
    InitWatchdog(int rank, int world)
    {
        const size_t limit = env_size("VFSX_COMM_INIT_TIMEOUT_S", 300);
        if (!limit)
            return;
        th = std::thread([this, rank, world, limit] {
            std::unique_lock<std::mutex> lk(mu);
            if (!cv.wait_for(lk, std::chrono::seconds(limit), [this] { return done; })) {
                std::fprintf(stderr,
                             "vfsx_comm_init: rank %d of %d still waits in ncclCommInitRank after %zu s -- a peer never "
                             "joined; exiting with status 3 (VFSX_COMM_INIT_TIMEOUT_S)\n",
                             rank, world, limit);
                std::fflush(stderr);
                _exit(3);
            }
        });
    }
    ~InitWatchdog()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
        if (th.joinable())
            th.join();
    }
};

// all-gather of this rank's slot of d_gather (in place) + merge into d_hits, all on `s`
int gather_and_merge(vfsx_comm *cm, uint32_t nq, uint32_t k, vfsx_hit *d_hits, hipStream_t s)
{
    const size_t per_rank = (size_t)nq * k;
    const ncclResult_t e = rccl().AllGather(cm->d_gather + (size_t)cm->rank * per_rank, cm->d_gather,
                                            per_rank * sizeof(vfsx_hit), ncclChar, cm->comm, s);
    if (e != ncclSuccess)
        return fail(VFSX_ERR_INTERNAL, rccl_msg("ncclAllGather", e));
    const hipError_t h = vfsx::launch_merge_hits(cm->d_gather, (uint32_t)cm->world, nq, k, d_hits, s);
    if (h != hipSuccess)
        return fail(VFSX_ERR_INTERNAL, std::string("vfsx_comm: merge launch: ") + hipGetErrorString(h));
    return VFSX_OK;
}

} // namespace

extern "C" {

int vfsx_comm_unique_id(void *id_out)
{
    vfsx::set_last_error("");
    if (!id_out)
        return fail(VFSX_ERR_INVALID_VECTOR, "vfsx_comm_unique_id: null buffer");
    if (int rc = rccl_ready())
        return rc;
    static_assert(sizeof(ncclUniqueId) == VFSX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t e = rccl().GetUniqueId(&id);
    if (e != ncclSuccess)
        return fail(VFSX_ERR_INTERNAL, rccl_msg("ncclGetUniqueId", e));
    __builtin_memcpy(id_out, &id, sizeof(id));
    return VFSX_OK;
}

vfsx_comm *vfsx_comm_init(int rank, int world, const void *id, int device)
{
    vfsx::set_last_error("");
    if (!id || world < 1 || rank < 0 || rank >= world) {
        fail(VFSX_ERR_INVALID_VECTOR, "vfsx_comm_init: bad rank / world / id");
        return nullptr;
    }
    if (rccl_ready() != VFSX_OK)
        return nullptr;
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        fail(VFSX_ERR_NO_DEVICE, "vfsx_comm_init: HIP device " + std::to_string(device) + " is not usable");
        return nullptr;
    }
    ncclUniqueId uid;
    __builtin_memcpy(&uid, id, sizeof(uid));
    vfsx_comm *cm = new (std::nothrow) vfsx_comm();
    if (!cm) {
        fail(VFSX_ERR_INTERNAL, "out of host memory");
        return nullptr;
    }
    cm->rank = rank;
    cm->world = world;
    cm->device = device;
    ncclResult_t e;
    {
        InitWatchdog watchdog(rank, world);
        e = rccl().CommInitRank(&cm->comm, world, uid, rank); // collective: returns when all ranks joined
    }
    if (e != ncclSuccess) {
        fail(VFSX_ERR_INTERNAL, rccl_msg("ncclCommInitRank", e));
        delete cm;
        return nullptr;
    }
    // every buffer the exchange needs, now: calls within these sizes never allocate (vfsx_comm_reserve widens them)
    if (reserve_all(cm, env_size("VFSX_COMM_HITS", kDefaultHitsPerRank), env_size("VFSX_COMM_QUERY_FLOATS", kDefaultQueryFloats)) != VFSX_OK ||
        hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking) != hipSuccess) {
        const std::string msg = vfsx_last_error();
        vfsx_comm_destroy(cm);
        fail(VFSX_ERR_INTERNAL, msg.empty() ? std::string("vfsx_comm_init: stream creation failed") : msg);
        return nullptr;
    }
    return cm;
}

int vfsx_comm_reserve(vfsx_comm *cm, uint64_t hits_per_rank, uint64_t query_floats)
{
    vfsx::set_last_error("");
    if (!cm)
        return fail(VFSX_ERR_INVALID_VECTOR, "vfsx_comm_reserve: null communicator");
    if (hipSetDevice(cm->device) != hipSuccess)
        return fail(VFSX_ERR_NO_DEVICE, "vfsx_comm: HIP device not usable");
    return reserve_all(cm, (size_t)hits_per_rank, (size_t)query_floats);
}

uint64_t vfsx_comm_alloc_count(const vfsx_comm *cm) { return cm ? cm->allocs : 0; }

void vfsx_comm_destroy(vfsx_comm *cm)
{
    if (!cm)
        return;
    (void)hipSetDevice(cm->device);
    if (cm->comm)
        (void)rccl().CommDestroy(cm->comm);
    if (cm->d_gather)
        (void)hipFree(cm->d_gather);
    if (cm->d_queries)
        (void)hipFree(cm->d_queries);
    if (cm->d_out)
        (void)hipFree(cm->d_out);
    if (cm->h_out)
        (void)hipHostFree(cm->h_out);
    if (cm->stream)
        (void)hipStreamDestroy(cm->stream);
    delete cm;
}

int vfsx_comm_rank(const vfsx_comm *cm) { return cm ? cm->rank : -1; }
int vfsx_comm_world(const vfsx_comm *cm) { return cm ? cm->world : 0; }

int vfsx_comm_rccl_version(void)
{
    vfsx::set_last_error("");
    if (rccl_ready() != VFSX_OK)
        return 0;
    int v = 0;
    return rccl().GetVersion(&v) == ncclSuccess ? v : 0;
}

int vfsx_comm_merge_hits(vfsx_comm *cm, const vfsx_hit *d_local, uint32_t nq, uint32_t k, vfsx_hit *d_hits, void *stream)
{
    vfsx::set_last_error("");
    if (!cm || !d_local || !d_hits || k == 0)
        return fail(VFSX_ERR_INVALID_VECTOR, "vfsx_comm_merge_hits: bad arguments");
    if (nq == 0)
        return VFSX_OK;
    if (hipSetDevice(cm->device) != hipSuccess)
        return fail(VFSX_ERR_NO_DEVICE, "vfsx_comm: HIP device not usable");
    const size_t per_rank = (size_t)nq * k;
    if (int rc = reserve_buf(cm, cm->d_gather, cm->gather_hits, per_rank * cm->world, "gather buffer"))
        return rc;
    const hipStream_t s = (hipStream_t)stream;
    const hipError_t h = hipMemcpyAsync(cm->d_gather + (size_t)cm->rank * per_rank, d_local, per_rank * sizeof(vfsx_hit),
                                        hipMemcpyDeviceToDevice, s);
    if (h != hipSuccess)
        return fail(VFSX_ERR_INTERNAL, std::string("vfsx_comm: staging copy: ") + hipGetErrorString(h));
    return gather_and_merge(cm, nq, k, d_hits, s);
}

int vfsx_search_sharded(vfsx_corpus *c, vfsx_comm *cm, const float *d_queries, uint32_t nq, uint32_t k, int metric,
                        vfsx_hit *d_hits, void *stream)
{
    vfsx::set_last_error("");
    if (!c || !cm)
        return fail(VFSX_ERR_INVALID_VECTOR, "vfsx_search_sharded: null corpus / communicator");
    if (nq == 0)
        return VFSX_OK;
    if (k == 0 || k > VFSX_MAX_K)
        return fail(VFSX_ERR_INVALID_VECTOR, "k (result_limit) must be in 1.." + std::to_string(VFSX_MAX_K));
    if (hipSetDevice(cm->device) != hipSuccess)
        return fail(VFSX_ERR_NO_DEVICE, "vfsx_comm: HIP device not usable");
    const size_t per_rank = (size_t)nq * k;
    if (!d_hits)
        return fail(VFSX_ERR_INVALID_VECTOR, "vfsx_search_sharded: null output buffer");
    // (a batch beyond the reserved size grows the buffer: the one place this call may allocate, on every rank alike)
    if (int rc = reserve_buf(cm, cm->d_gather, cm->gather_hits, per_rank * cm->world, "gather buffer"))
        return rc;
    // the local top-k goes straight into this rank's slot of the gather buffer (in-place all-gather)
    vfsx_hit *slot = cm->d_gather + (size_t)cm->rank * per_rank;
    const int rc_local = vfsx_search_device(c, d_queries, nq, k, metric, slot, stream);
    if (rc_local == VFSX_OK)
        return gather_and_merge(cm, nq, k, d_hits, (hipStream_t)stream);
    // The local scan failed (bad arguments on this rank only, an allocation that ran dry): the peers are already in,
    // or on their way into, the all-gather, and would wait there for ever.  Join it with an empty list, THEN report.
    const std::string why = vfsx_last_error();
    if (vfsx::launch_empty_hits(slot, nq, k, k, (hipStream_t)stream) == hipSuccess)
        (void)gather_and_merge(cm, nq, k, d_hits, (hipStream_t)stream);
    return fail(rc_local, why + " (this rank joined the all-gather with an empty list)");
}

int vfsx_search_sharded_host(vfsx_corpus *c, vfsx_comm *cm, const float *queries, uint32_t nq, uint32_t k, int metric,
                             uint64_t *out_ids, float *out_dist, uint32_t *out_count)
{
    vfsx::set_last_error("");
    if (!c || !cm)
        return fail(VFSX_ERR_INVALID_VECTOR, "vfsx_search_sharded_host: null corpus / communicator");
    if (nq == 0)
        return VFSX_OK;
    if (!queries || !out_ids || !out_dist || !out_count)
        return fail(VFSX_ERR_INVALID_VECTOR, "null buffer");
    if (hipSetDevice(cm->device) != hipSuccess)
        return fail(VFSX_ERR_NO_DEVICE, "vfsx_comm: HIP device not usable");
    const uint32_t dim = vfsx_corpus_dim(c);
    if (int rc = reserve_buf(cm, cm->d_queries, cm->query_floats, (size_t)nq * dim, "query buffer"))
        return rc;
    if (int rc = reserve_buf(cm, cm->d_out, cm->out_hits, (size_t)nq * k, "result buffer"))
        return rc;
    if (int rc = reserve_buf(cm, cm->h_out, cm->h_out_hits, (size_t)nq * k, "pinned result buffer", true))
        return rc;
    hipError_t h = hipMemcpyAsync(cm->d_queries, queries, (size_t)nq * dim * sizeof(float), hipMemcpyHostToDevice, cm->stream);
    if (h != hipSuccess)
        return fail(VFSX_ERR_INTERNAL, std::string("vfsx_comm: query upload: ") + hipGetErrorString(h));
    if (int rc = vfsx_search_sharded(c, cm, cm->d_queries, nq, k, metric, cm->d_out, cm->stream))
        return rc;
    const vfsx_hit *hits = cm->h_out;
    h = hipMemcpyAsync(cm->h_out, cm->d_out, (size_t)nq * k * sizeof(vfsx_hit), hipMemcpyDeviceToHost, cm->stream);
    if (h == hipSuccess)
        h = hipStreamSynchronize(cm->stream);
    if (h != hipSuccess)
        return fail(VFSX_ERR_INTERNAL, std::string("vfsx_comm: result download: ") + hipGetErrorString(h));
    for (uint32_t q = 0; q < nq; ++q) {
        uint32_t cnt = 0;
        for (uint32_t j = 0; j < k; ++j) {
            const vfsx_hit &x = hits[(size_t)q * k + j];
            if (x.flags & 1u) {
                out_ids[(size_t)q * k + cnt] = x.id;
                out_dist[(size_t)q * k + cnt] = x.dist;
                ++cnt;
            }
        }
        out_count[q] = cnt;
    }
    return VFSX_OK;
}

} // extern "C"
