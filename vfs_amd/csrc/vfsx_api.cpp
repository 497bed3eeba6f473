// vfsx_api.cpp -- the C ABI of include/vfsx.h: corpus residency in HBM, the pinned
// double-buffered ingest pipeline, search orchestration and error mapping.
// There is deliberately no CPU compute path in this file: every search runs the
// HIP kernels of vfsx_kernels.hip or fails.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "vfsx_internal.h"

namespace {

using vfsx::fail;
using vfsx::guarded;

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
                       ? VFSX_ERR_NO_DEVICE
                       : VFSX_ERR_INTERNAL;
        fail(code, std::string(what) + ": " + hipGetErrorString(e));
    }
}
#define HIP_CHECK(x) hip_check((x), #x)

uint64_t env_u64(const char *name, uint64_t dflt)
{
    const char *v = std::getenv(name);
    if (!v || !*v)
        return dflt;
    char *end = nullptr;
    unsigned long long x = std::strtoull(v, &end, 10);
    return (end && *end == 0) ? (uint64_t)x : dflt;
}

int device_count_quiet()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

template <typename T>
struct DeviceBuf {
    T *p = nullptr;
    size_t n = 0; // elements
    void reserve(size_t want)
    {
        if (want <= n)
            return;
        if (p)
            HIP_CHECK(hipFree(p));
        p = nullptr;
        n = 0;
        HIP_CHECK(hipMalloc((void **)&p, want * sizeof(T)));
        n = want;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// ---- segmented device arrays -------------------------------------------------------------------------------
// The corpus, its norms and ids (and the int8 operands of the batch path) grow without ever moving a stored row:
// one contiguous VIRTUAL range per array is reserved up front (hipMemAddressReserve: address space, no memory) and
// physical slabs are created and mapped behind the part already in use as the store grows (hipMemCreate /
// hipMemMap / hipMemSetAccess) -- slabs of ONE size per array, a power of two between min_slab and kSlabBytes fixed
// by the first request (see `slab` below for why).  The kernels keep seeing ONE base pointer and the rb64
// indexing they always had (no slab table, no per-access indirection), growth costs a map call instead of
// "allocate twice the size and copy" -- which capped a store that was not sized up front at a third of the
// card (old block + new block live together) -- and device pointers handed out earlier (a captured HIP graph)
// stay valid.  The reference's feed is an unbounded append (storage_manager.rs:139-157).
// Where the driver refuses virtual memory management (VFSX_VMM=0 forces this), the array is one hipMalloc block
// that grows by reallocation and copy, as in round 2.
constexpr size_t kSlabBytes = 1ull << 30;

struct VmArray {
    char *base = nullptr;
    size_t reserved = 0; // bytes of address space (VMM) -- 0: the hipMalloc fallback
    size_t mapped = 0;   // bytes backed by memory, from base
    int device = 0;
    struct Slab {
        hipMemGenericAllocationHandle_t h;
        size_t off, bytes;
    };
    std::vector<Slab> slabs;
    uint64_t grow_calls = 0, copy_bytes = 0; // (diagnostics: vfsx_corpus_storage_info)

    static bool vmm_wanted()
    {
        static const bool on = [] {
            const char *e = std::getenv("VFSX_VMM");
            return !(e && e[0] == '0');
        }();
        return on;
    }

    // All slabs of one array have the SAME size, a power of two fixed at the first mapping: the first request rounded
    // up, between min_slab and kSlabBytes.  This driver's hipMemSetAccess returns "invalid argument" for a slab whose
    // size differs from its predecessors' in the same range (4, 4, 8 KiB; 2, 2, 4, 6 MiB; 2, 2, 4, 8, 16 MiB all failed
    // at the last one, 64 x 2 MiB, 40 x 64 MiB and 6 x 1 GiB went through: tools/probe/vmm_probe.cpp), and the
    // granularity it reports, 4 KiB, is not one to build on.
    size_t min_slab = 2u << 20;
    size_t slab = 0;

    // reserve address space for at most `max_bytes`; on any failure the array silently becomes a hipMalloc block
    void init(int dev, size_t max_bytes)
    {
        device = dev;
        if (!vmm_wanted() || max_bytes == 0)
            return;
        const size_t want = (max_bytes + kSlabBytes - 1) / kSlabBytes * kSlabBytes;
        void *p = nullptr;
        if (hipMemAddressReserve(&p, want, kSlabBytes, nullptr, 0) != hipSuccess || !p) {
            (void)hipGetLastError();
            return;
        }
        base = (char *)p;
        reserved = want;
    }

    // one more slab behind the mapped part: false (and nothing changed) when the driver refuses
    bool map_slab(size_t bytes)
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        Slab sl{};
        sl.off = mapped;
        sl.bytes = bytes;
        if (hipMemCreate(&sl.h, bytes, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        hipError_t e = hipMemMap(base + mapped, bytes, 0, sl.h, 0);
        if (e == hipSuccess) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(base + mapped, bytes, &acc, 1);
            if (e != hipSuccess)
                (void)hipMemUnmap(base + mapped, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipMemRelease(sl.h);
            return false;
        }
        slabs.push_back(sl);
        mapped += bytes;
        return true;
    }

    // the driver stopped mapping slabs: move what is stored into one hipMalloc block and carry on as the fallback
    void demote(size_t want, size_t used, hipStream_t s)
    {
        const size_t cap = std::max(want, mapped * 2);
        char *nb = nullptr;
        HIP_CHECK(hipMalloc((void **)&nb, cap));
        if (used) {
            hipError_t e = hipMemcpyAsync(nb, base, used, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess)
                e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(nb);
                HIP_CHECK(e);
            }
            copy_bytes += used;
        }
        release();
        base = nb;
        mapped = cap;
    }

    // make [0, want) usable; `used` = bytes that hold data (what a copy has to move), on stream s
    void grow(size_t want, size_t used, hipStream_t s)
    {
        if (want <= mapped)
            return;
        ++grow_calls;
        if (reserved) {
            if (want > reserved)
                fail(VFSX_ERR_MEMTABLE, "corpus exceeds the address range reserved for it (the device's memory size)");
            if (!slab) {
                slab = min_slab;
                while (slab < want && slab < kSlabBytes)
                    slab <<= 1;
            }
            while (mapped < want) {
                if (mapped + slab > reserved || !map_slab(slab)) {
                    demote(want, used, s);
                    return;
                }
            }
            return;
        }
        // fallback: one block, doubled and copied
        const size_t cap = std::max(want, mapped * 2);
        char *nb = nullptr;
        HIP_CHECK(hipMalloc((void **)&nb, cap));
        if (used) {
            hipError_t e = hipMemcpyAsync(nb, base, used, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess)
                e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(nb);
                HIP_CHECK(e);
            }
            copy_bytes += used;
        }
        if (base)
            (void)hipFree(base);
        base = nb;
        mapped = cap;
    }

    void release()
    {
        if (reserved) {
            for (const Slab &sl : slabs) {
                (void)hipMemUnmap(base + sl.off, sl.bytes);
                (void)hipMemRelease(sl.h);
            }
            (void)hipMemAddressFree(base, reserved);
        } else if (base) {
            (void)hipFree(base);
        }
        slabs.clear();
        base = nullptr;
        reserved = mapped = 0;
        slab = 0;
    }
};

} // namespace

struct vfsx_corpus {
    int device = 0;
    uint32_t dim = 0;
    int dtype = VFSX_F32;
    float scale = 127.0f;
    uint32_t epc = 4;      // elements per 16-byte chunk
    uint32_t d_chunks = 0; // chunks per row
    uint64_t capacity = 0; // rows reserved (multiple of 64)
    uint64_t n_rows = 0;
    uint64_t row_base = 0;
    uint64_t id_base = 1;  // implicit ids: id_base + local row (storage_manager.rs:76)
    bool explicit_ids = false;
    std::vector<uint64_t> ids_host; // only when explicit_ids
    uint64_t max_id = 0;            // largest explicit id stored (implicit continuation: max_id + 1)

    // the three arrays that grow with the store (VmArray: address range reserved once, memory mapped as it fills);
    // d_corpus / d_norms / d_ids are their base addresses, which never change after the first mapping (VMM)
    VmArray va_corpus, va_norms, va_ids;
    VmArray va_mirror, va_rn2, va_inv; // the batch path's per-row operands (d_mirror, d_rn2, d_inv_rn), grown the same way
    uint64_t max_rows = 0; // rows the reserved ranges can hold
    uint4 *d_corpus = nullptr;
    float *d_norms = nullptr;
    uint64_t *d_ids = nullptr;
    float *d_lut = nullptr;

    hipStream_t stream = nullptr; // ingest + host-API searches
    // the search workspaces below are shared by every search on this corpus: searches on different streams are
    // chained (the new stream waits for what the previous one had enqueued) instead of left to race
    hipStream_t last_search_stream = nullptr; // compared, never used: it may be a destroyed stream's handle
    bool searched = false;
    // ingest staging (pinned host + device), double buffered
    size_t stage_bytes = 0;
    void *h_stage[2] = {nullptr, nullptr};
    void *d_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};

    // search workspace
    DeviceBuf<uint64_t> lists;
    DeviceBuf<float> q_in, q_pad;
    DeviceBuf<vfsx_hit> hits;
    void *h_call = nullptr, *d_call = nullptr; // vfsx_search's pinned block for small calls (host / device address)
    DeviceBuf<uint64_t> row_idx;
    DeviceBuf<uint8_t> row_out;
    uint32_t max_blocks = 0;
    int unroll_override = 0; // 0 = auto
    uint64_t nq_per_pass_max = 8;
    int nt_override = 2;     // 2 = auto
    int pipe_override = 2;   // 2 = auto
    bool balanced_grid = true; // VFSX_BALANCED_GRID=0: always the full grid (A/B measurements)
    uint32_t i8_wide = 3;      // VFSX_I8_WIDE: 16-wave int8 scan workgroups per CU (0 = off, 1, 2; 3 = by corpus size)
    uint32_t n_cus = 0;

    // small appends batched on the host (vfsx_corpus_set_insert_batch; off by default): single-row inserts cost a
    // memcpy, the rows reach the device together -- when `insert_batch` of them wait, or before anything reads the store
    uint32_t insert_batch = 0;
    uint64_t pend_n = 0;
    int pend_mode = 0;
    bool pend_has_ids = false;
    std::vector<uint8_t> pend_rows;
    std::vector<uint64_t> pend_ids;

    // host-side metadata of rows loaded from a VFS log (VectorMetadata, vector.rs:11-17), by local row
    using RowMeta = vfsx::RowMeta;
    std::vector<RowMeta> meta; // empty, or one entry per row

    // many-query batches on the matrix cores (vfsx_mfma.hip): bf16 mirror + workspaces
    bool mfma_enabled = false;
    uint4 *d_mirror = nullptr;
    uint64_t mirror_rows = 0; // rows the mirror currently covers (0 = stale / absent)
    size_t mirror_bytes = 0;
    uint32_t *d_mirror_stats = nullptr; // device: [0] a row norm outside the path's range exists, [1] R_x (vfsx_mfma.hip)
    float *d_rn2 = nullptr;      // [mirror rows padded] float2 (b, b^2 (1 - gamma)) for Euclidean batches
    float *d_inv_rn = nullptr;   // int8 stores: [same rows] 1 / (scale b); the rb64 corpus is the MFMA operand, no mirror
    size_t rn2_rows = 0;
    DeviceBuf<float> q_norm, q_eps, q_scale, q_dk;
    DeviceBuf<uint32_t> long_state, long_hist; // result_limit > kLongResultK: radix-select state / histograms
    DeviceBuf<uint64_t> long_sel;              // ... and the compacted keys of one chunk per query
    DeviceBuf<uint32_t> fb_idx;  // exact fallback of flagged queries: their indices, compacted copies, hits
    DeviceBuf<float> fb_queries;
    DeviceBuf<vfsx_hit> fb_hits;
    bool mirror_bad = false;
    // f32 stores: which mirror the prefilter multiplies.  bf16 (2 B / element) or int8 with a scale per row (1 B /
    // element, twice the MFMA rate, ~3.5x the quantization residual).  Auto starts on int8 and moves to bf16 for good
    // when the measured residual R_x is large (heavy-tailed rows) or a batch sends too many queries back.
    enum MirrorKind { kMirrorBf16 = 0, kMirrorI8 = 1 };
    int mirror_kind = kMirrorI8;
    int mirror_pref = 0; // 0 auto, 1 bf16, 2 int8 (vfsx_corpus_enable_mfma's argument - 1, or VFSX_MFMA_MIRROR)
    float mirror_rx = 0.0f;
    uint32_t mfma_min_batch = 4;  // 0 = by store type and size (mfma_min_batch_for)
    uint32_t mfma_kp_override = 0; // VFSX_MFMA_KP: first-pass candidates per query (0 = by k)
    DeviceBuf<uint8_t> q_mirror;
    DeviceBuf<int> q_bad, q_flags;
    DeviceBuf<float> cand;
    DeviceBuf<uint64_t> sel;
    int *h_flags = nullptr; // pinned: the per-query flags of the matrix-core path, read once per call
    size_t h_flags_cap = 0;
    uint64_t mfma_queries = 0, mfma_fallbacks = 0;
    // back-off: when most of a batch failed the bound check (near-duplicate-heavy data), the prefilter
    // was wasted work; skip it for the next `mfma_backoff` batches, doubling the pause while it keeps
    // happening (1, 3, 7, ... 63 batches), and probe again afterwards
    uint32_t mfma_backoff = 0, mfma_backoff_next = 0;
    uint32_t mirror_win_queries = 0, mirror_win_flagged = 0; // auto mirror choice: queries / margin failures seen
    uint32_t small_win_queries = 0, small_win_flagged = 0;   // calls of < 4 queries through the path: the same
    uint32_t small_off = 0;                                  // ... and how many such queries still skip it

    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    double prof_ms = 0.0;
    uint64_t prof_launches = 0;

    size_t block_bytes() const { return (size_t)d_chunks * 1024u; }
    size_t corpus_bytes(uint64_t rows) const { return (size_t)(rows / 64u) * block_bytes(); }
};

namespace {

using vfsx::ScanArgs;
using vfsx::SelectArgs;

void use_device(const vfsx_corpus *c) { HIP_CHECK(hipSetDevice(c->device)); }

uint64_t round_up64(uint64_t x) { return (x + 63u) / 64u * 64u; }

// a shard holds at most 2^32 - 66 rows (the row index travels in 32 bits of the composite key, the top values are
// sentinels); its capacity, a whole number of 64-row blocks, therefore at most 2^32 - 64
constexpr uint64_t kMaxCapacityRows = 0xffffffc0ull;
constexpr size_t kPinnedCallBytes = 256u << 10; // vfsx_search: queries + hits of a call up to this size go through pinned memory

void reserve_rows(vfsx_corpus *c, uint64_t want_rows)
{
    if (want_rows > 0xfffffffeull - 64u) // (before the capacity check: the last row-block has room for two rows more)
        fail(VFSX_ERR_MEMTABLE, "corpus shard limited to 2^32-66 rows per device");
    if (want_rows <= c->capacity)
        return;
    if (c->va_corpus.reserved && want_rows > c->max_rows)
        fail(VFSX_ERR_MEMTABLE, "corpus of " + std::to_string(want_rows) + " rows exceeds the device's memory");
    const uint64_t rows = std::max<uint64_t>(round_up64(want_rows), 64);
    const size_t used_rows = (size_t)round_up64(c->n_rows);
    // the corpus grows first; the rows its mapped bytes hold become the capacity the side arrays follow
    c->va_corpus.grow(c->corpus_bytes(rows), c->corpus_bytes(used_rows), c->stream);
    uint64_t cap = std::min<uint64_t>(c->va_corpus.mapped / c->block_bytes() * 64u, kMaxCapacityRows);
    if (c->va_corpus.reserved)
        cap = std::min<uint64_t>(cap, c->max_rows);
    const size_t norms_before = c->va_norms.mapped;
    c->va_norms.grow(cap * sizeof(float), c->n_rows * sizeof(float), c->stream);
    if (c->explicit_ids)
        c->va_ids.grow(cap * sizeof(uint64_t), c->n_rows * sizeof(uint64_t), c->stream);
    c->d_corpus = (uint4 *)c->va_corpus.base;
    c->d_norms = (float *)c->va_norms.base;
    c->d_ids = (uint64_t *)c->va_ids.base;
    // rows past n_rows are read by the scan but masked (row < n_rows): any corpus bytes will do; their norms are zero
    if (c->va_norms.reserved) {
        if (c->va_norms.mapped > norms_before)
            HIP_CHECK(hipMemsetAsync(c->va_norms.base + norms_before, 0, c->va_norms.mapped - norms_before, c->stream));
    } else {
        const size_t keep = c->n_rows * sizeof(float);
        HIP_CHECK(hipMemsetAsync(c->va_norms.base + keep, 0, c->va_norms.mapped - keep, c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->capacity = cap;
}

void ensure_staging(vfsx_corpus *c)
{
    if (c->h_stage[0])
        return;
    size_t bytes = (size_t)env_u64("VFSX_STAGE_MB", 32) << 20;
    size_t row_bytes = (size_t)c->dim * sizeof(float); // widest source row (f32)
    bytes = std::max(bytes, row_bytes);
    for (int i = 0; i < 2; ++i) {
        HIP_CHECK(hipHostMalloc(&c->h_stage[i], bytes, hipHostMallocDefault));
        HIP_CHECK(hipMalloc(&c->d_stage[i], bytes));
        HIP_CHECK(hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming));
    }
    c->stage_bytes = bytes;
}

// host rows -> pinned staging.  One core moves ~12-20 GB/s, a PCIe 5 x16 link ~55 GB/s: split large
// copies over a few threads so the H2D DMA, not this memcpy, sets the ingest rate.
void staging_copy(void *dst, const void *src, size_t bytes)
{
    static const unsigned n_threads = (unsigned)std::min<uint64_t>(8, std::max<uint64_t>(1, env_u64("VFSX_INGEST_THREADS", 4)));
    if (bytes < (8u << 20) || n_threads <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t part = (bytes / n_threads + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) {
        const size_t off = part * t;
        if (off >= bytes)
            break;
        const size_t len = std::min(part, bytes - off);
        pool.emplace_back([=] { std::memcpy((char *)dst + off, (const char *)src + off, len); });
    }
    std::memcpy(dst, src, std::min(part, bytes));
    for (auto &th : pool)
        th.join();
}

// make ids explicit: materialise the implicit run for rows already stored.  Host and device state change only
// after every allocation and copy has succeeded.
void make_ids_explicit(vfsx_corpus *c)
{
    if (c->explicit_ids)
        return;
    std::vector<uint64_t> ids(c->n_rows);
    for (uint64_t r = 0; r < c->n_rows; ++r)
        ids[r] = c->id_base + r;
    c->va_ids.grow(std::max<uint64_t>(c->capacity, 64) * sizeof(uint64_t), 0, c->stream);
    uint64_t *d_ids = (uint64_t *)c->va_ids.base;
    if (c->n_rows)
        HIP_CHECK(hipMemcpy(d_ids, ids.data(), c->n_rows * sizeof(uint64_t), hipMemcpyHostToDevice));
    c->ids_host.swap(ids);
    c->d_ids = d_ids;
    c->max_id = c->n_rows ? c->id_base + c->n_rows - 1 : 0;
    c->explicit_ids = true;
}

// mode 0: f32 rows -> f32 corpus, 1: i8 -> i8, 2: f32 -> i8 (quantize on device)
void append_now(vfsx_corpus *c, const void *rows, const uint64_t *ids, uint64_t n, int mode)
{
    if (n == 0)
        return;
    if (!rows)
        fail(VFSX_ERR_INVALID_VECTOR, "Vector data cannot be empty"); // storage_manager.rs:319-321
    use_device(c);
    if (ids)
        make_ids_explicit(c);
    reserve_rows(c, c->n_rows + n);
    ensure_staging(c);
    const size_t src_elem = (mode == 1) ? 1 : 4;
    const size_t src_row = (size_t)c->dim * src_elem;
    const uint64_t rows_per_batch = std::max<uint64_t>(1, c->stage_bytes / src_row);
    const uint8_t *src = (const uint8_t *)rows;
    uint64_t done = 0;
    int buf = 0;
    while (done < n) {
        const uint64_t take = std::min(rows_per_batch, n - done);
        // the previous kernel that read d_stage[buf] (and the copy out of h_stage[buf]) must be over
        HIP_CHECK(hipEventSynchronize(c->stage_done[buf]));
        staging_copy(c->h_stage[buf], src + done * src_row, take * src_row);
        HIP_CHECK(hipMemcpyAsync(c->d_stage[buf], c->h_stage[buf], take * src_row, hipMemcpyHostToDevice,
                                 c->stream));
        HIP_CHECK(vfsx::launch_ingest(mode, c->d_stage[buf], take, c->dim, c->d_corpus, c->d_chunks,
                                      c->n_rows + done, c->scale, c->stream));
        HIP_CHECK(hipEventRecord(c->stage_done[buf], c->stream));
        done += take;
        buf ^= 1;
    }
    HIP_CHECK(vfsx::launch_row_norms(c->dtype, c->d_corpus, c->d_chunks, c->d_lut, c->n_rows, n, c->d_norms,
                                     c->stream));
    // the new rows' ids: staged here, committed to ids_host / max_id / n_rows only once the device work is known
    // to have succeeded (a failed append leaves the corpus exactly as it was; rows past n_rows are never read)
    std::vector<uint64_t> new_ids;
    uint64_t new_max = c->max_id;
    if (c->explicit_ids) {
        new_ids.resize(n);
        for (uint64_t i = 0; i < n; ++i) {
            // no ids given on a store that has explicit ones: continue after the largest id present
            // (next_id semantics, storage_manager.rs:126-130), never by row position
            new_ids[i] = ids ? ids[i] : new_max + 1;
            new_max = std::max(new_max, new_ids[i]);
        }
        HIP_CHECK(hipMemcpyAsync(c->d_ids + c->n_rows, new_ids.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice,
                                 c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->explicit_ids) {
        c->ids_host.insert(c->ids_host.end(), new_ids.begin(), new_ids.end());
        c->max_id = new_max;
    }
    c->n_rows += n;
}

// rows staged by small appends -> the device, in one append (order kept: everything that appends or reads comes here first)
void commit_pending(vfsx_corpus *c)
{
    if (!c->pend_n)
        return;
    // append_now leaves the corpus as it was when it fails; the rows then STAY staged -- their appends have returned
    // success, so they are not dropped behind the caller's back: the call that needed the commit reports the error,
    // and so does every later one until a commit goes through (or the corpus is destroyed)
    append_now(c, c->pend_rows.data(), c->pend_has_ids ? c->pend_ids.data() : nullptr, c->pend_n, c->pend_mode);
    c->pend_n = 0;
    c->pend_rows.clear(); // (keeps the capacity)
    c->pend_ids.clear();
}

constexpr uint64_t kSmallAppendRows = 16; // appends up to this many rows are staged when batching is on

void append_impl(vfsx_corpus *c, const void *rows, const uint64_t *ids, uint64_t n, int mode)
{
    if (c->insert_batch && n && n <= kSmallAppendRows && rows) {
        if (c->n_rows + c->pend_n + n > 0xfffffffeull - 64u)
            fail(VFSX_ERR_MEMTABLE, "corpus shard limited to 2^32-66 rows per device");
        if (c->pend_n && (mode != c->pend_mode || (ids != nullptr) != c->pend_has_ids))
            commit_pending(c);
        const size_t row_bytes = (size_t)c->dim * (mode == 1 ? 1u : 4u);
        const uint8_t *src = (const uint8_t *)rows;
        c->pend_rows.insert(c->pend_rows.end(), src, src + n * row_bytes);
        if (ids)
            c->pend_ids.insert(c->pend_ids.end(), ids, ids + n);
        c->pend_mode = mode;
        c->pend_has_ids = ids != nullptr;
        c->pend_n += n;
        if (c->pend_n >= c->insert_batch)
            commit_pending(c);
        return;
    }
    commit_pending(c);
    append_now(c, rows, ids, n, mode);
}

bool is_cosine(int metric) { return metric == VFSX_COSINE || metric == VFSX_SIMD_COSINE; }

void prof_begin(vfsx_corpus *c, hipStream_t s, size_t &slot)
{
    if (!c->profiling)
        return;
    if (c->prof_used == c->prof_events.size()) {
        hipEvent_t a, b;
        HIP_CHECK(hipEventCreate(&a));
        HIP_CHECK(hipEventCreate(&b));
        c->prof_events.push_back({a, b});
    }
    slot = c->prof_used++;
    HIP_CHECK(hipEventRecord(c->prof_events[slot].first, s));
}

void prof_end(vfsx_corpus *c, hipStream_t s, size_t slot)
{
    if (!c->profiling)
        return;
    HIP_CHECK(hipEventRecord(c->prof_events[slot].second, s));
}

void prof_collect(vfsx_corpus *c)
{
    for (size_t i = 0; i < c->prof_used; ++i) {
        HIP_CHECK(hipEventSynchronize(c->prof_events[i].second));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
        c->prof_ms += ms;
        c->prof_launches += 1;
    }
    c->prof_used = 0;
}

// Device memory the exact scan needs beyond its inputs and outputs: the per-workgroup top-k lists the scan
// kernel hands to the select kernel, and (only when the dimension is not a whole number of 16-byte chunks, or
// for the integer-domain metric) a re-laid-out copy of the queries.  nullptr = the corpus' own buffers, grown
// with hipMalloc on demand; vfsx_search_device_noalloc passes slices of the library's static device array.
struct ScanWorkspace {
    uint64_t *lists;
    size_t lists_elems;
    float *q_stage;
    size_t q_stage_elems;
};

// One pass of the exact scan (k <= kMaxKPerPass hits per query): enqueue scan + select on `s`, no host
// synchronisation.  Query q's hits go to d_hits[q * (long_k ? long_k : k) ..].
// long_k != 0 (result_limit > kLongResultK; never with ws or the integer-domain metric): the scan writes every row's
// composite key instead of lists and the long_k hits per query come from the radix select (vfsx_kernels.hip, "long
// results"); k must be 1 then (it only sizes the unused list LDS).
void exact_scan_pass(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k, int metric, vfsx_hit *d_hits,
                     hipStream_t s, const ScanWorkspace *ws, uint32_t long_k = 0)
{
    const bool i8dom = metric == VFSX_EUCLIDEAN_I8DOMAIN;
    const uint32_t out_stride = long_k ? long_k : k;
    const bool profiling = c->profiling && !ws; // event creation allocates: never on the no-allocation path

    // queries the kernels cannot read in place are staged: quantized and packed for the integer-domain
    // metric, zero padded to a whole chunk otherwise
    const uint32_t q_stride_pad = c->d_chunks * c->epc;
    const uint32_t n_words = c->d_chunks * 4u;
    const bool staged = i8dom || q_stride_pad != c->dim;
    const uint32_t q_stride = i8dom ? n_words + 4u : (staged ? q_stride_pad : c->dim);
    float *q_stage = nullptr;
    uint32_t stage_cap = nq; // queries the staging buffer holds
    if (staged && ws) {
        q_stage = ws->q_stage;
        stage_cap = (uint32_t)std::min<uint64_t>(nq, ws->q_stage_elems / q_stride);
        if (stage_cap == 0)
            fail(VFSX_ERR_INVALID_VECTOR, "vfsx_search_device_noalloc: dim " + std::to_string(c->dim) +
                                              " exceeds the static workspace (one staged query must fit " +
                                              std::to_string(ws->q_stage_elems * sizeof(float)) + " bytes)");
    } else if (staged) {
        c->q_pad.reserve((size_t)nq * q_stride);
        q_stage = c->q_pad.p;
    }
    auto stage_queries = [&](uint32_t q0, uint32_t qn) { // queries [q0, q0 + qn) -> q_stage[0 .. qn)
        if (i8dom)
            HIP_CHECK(vfsx::launch_query_quantize(d_queries + (size_t)q0 * c->dim, qn, c->dim, c->scale,
                                                  (uint32_t *)q_stage, n_words, q_stride, s));
        else
            HIP_CHECK(vfsx::launch_query_pad(d_queries + (size_t)q0 * c->dim, qn, c->dim, q_stage, q_stride, s));
    };
    if (staged && !ws)
        stage_queries(0, nq);

    const uint32_t n_rowblocks = (uint32_t)((c->n_rows + 63u) / 64u);
    // a corpus that fits the 256 MiB Infinity Cache is re-read from it: keep the default
    // cache policy there, stream everything larger with nontemporal loads
    const bool big = c->corpus_bytes(round_up64(c->n_rows)) > (192ull << 20);
    const int nt = c->nt_override == 2 ? (big ? 1 : 0) : (c->nt_override ? 1 : 0);
    int unroll = c->unroll_override;
    if (unroll != 4 && unroll != 8 && unroll != 16)
        unroll = c->dtype == VFSX_I8 ? 4 : (c->d_chunks >= 64 ? 16 : 8);
    const int pipe = (int)c->pipe_override == 2 ? 0 : (c->pipe_override ? 1 : 0);
    const int scan_variant = unroll * 2 + nt + ((pipe && unroll <= 8) ? 64 : 0);
    // queries per corpus pass: as many as the batch offers (max 8), while the workgroup's
    // LDS (lists + staged queries [+ 32 KiB LUT for int8]) stays within 64 KiB
    int nq_per_pass = 1;
    if (!i8dom) {
        uint32_t cap = (uint32_t)std::min<uint64_t>(8, c->nq_per_pass_max);
        // int8 rows shorter than 512 elements: two passes of 4 queries beat one of 8 (10M x 128: 0.83 vs 1.14 ms,
        // 2M x 256: 0.36 vs 0.50 ms; at 768 elements 8 per pass is 10 % ahead) -- tools/exact_batch.py
        if (c->dtype == VFSX_I8 && c->d_chunks < 32u)
            cap = std::min(cap, 4u);
        const size_t lds_static = c->dtype == VFSX_I8 ? 32768 : 16;
        for (int cand : {8, 4, 2}) {
            if ((uint32_t)cand <= cap && nq >= (uint32_t)cand &&
                vfsx::scan_lds_bytes(k, cand, q_stride) + lds_static <= 65536) {
                nq_per_pass = cand;
                break;
            }
        }
    }
    // single-query int8 scans of big corpora: 16-wave workgroups around one 64 KiB table (vfsx_kernels.hip)
    const bool wide = c->dtype == VFSX_I8 && !i8dom && nq_per_pass == 1 && k <= 64u && c->i8_wide != 0 &&
                      n_rowblocks >= 16u * c->n_cus && !long_k; // (only scan_kernel writes keys)
    // ... one such workgroup per CU, or two (with 8 instead of 4 loads in flight per lane) when every wave still
    // gets >= 16 row-blocks: C3 212 -> 209.5 us (alone, neither the second workgroup nor the deeper unroll gains)
    const uint32_t wide_wgs = c->i8_wide != 3u ? c->i8_wide : (n_rowblocks >= 512u * c->n_cus ? 2u : 1u);
    const int wide_unroll = c->unroll_override ? c->unroll_override : (wide_wgs == 2u ? 8 : 4);
    const uint32_t wg_waves = wide ? vfsx::kWideScanWaves : 4u;
    uint32_t grid_x = std::min<uint32_t>((n_rowblocks + wg_waves - 1u) / wg_waves,
                                         wide ? c->n_cus * wide_wgs : c->max_blocks);
    if (grid_x && !wide) {
        // Even shares when a wave only gets a few row-blocks: with r = ceil(row-blocks / waves), the
        // smallest grid that still covers the corpus gives (almost) every wave exactly r of them, so the
        // waves finish together instead of 1 - frac(r) of them idling through the last round (C2:
        // 15625 row-blocks, 1024 -> 977 workgroups, 455 -> 447 us).  With many row-blocks per wave the
        // imbalance is noise and the full grid is (slightly) better.
        const uint32_t per_wave = (n_rowblocks + 4u * grid_x - 1u) / (4u * grid_x);
        if (per_wave <= 16u && c->balanced_grid)
            grid_x = (n_rowblocks + 4u * per_wave - 1u) / (4u * per_wave);
    }
    // queries per launch: bounded by gridDim.y, by the list workspace (256 MiB of the corpus' own, or the
    // caller's) and by the staging buffer
    uint32_t q_per_launch = std::min<uint32_t>(nq, stage_cap);
    uint64_t *lists = nullptr;
    if (grid_x) {
        const uint64_t per_q = long_k ? (uint64_t)n_rowblocks * 64u * sizeof(uint64_t) // every row's key
                                      : (uint64_t)grid_x * k * sizeof(uint64_t);
        const uint64_t lists_bytes = ws ? ws->lists_elems * sizeof(uint64_t) : (256ull << 20);
        if (ws && per_q > lists_bytes)
            fail(VFSX_ERR_INVALID_VECTOR, "vfsx_search_device_noalloc: " + std::to_string(grid_x) + " workgroup lists of k = " +
                                              std::to_string(k) + " exceed the static workspace");
        q_per_launch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(q_per_launch, lists_bytes / per_q));
        q_per_launch = std::min<uint32_t>(q_per_launch, 32768u);
        if (!ws)
            c->lists.reserve((size_t)q_per_launch * per_q / sizeof(uint64_t));
        lists = ws ? ws->lists : c->lists.p;
        if (long_k) {
            const size_t st_words = ((size_t)q_per_launch * vfsx::long_state_bytes() + 3u) / 4u;
            c->long_state.reserve(st_words);
            c->long_sel.reserve((size_t)q_per_launch * vfsx::kLongChunkHits);
            if (c->long_hist.n < (size_t)q_per_launch * 256u) {
                c->long_hist.reserve((size_t)q_per_launch * 256u);
                HIP_CHECK(hipMemsetAsync(c->long_hist.p, 0, c->long_hist.n * sizeof(uint32_t), s)); // zero between selects
            }
        }
    }
    for (uint32_t q0 = 0; q0 < nq; q0 += q_per_launch) {
        const uint32_t qn = std::min(q_per_launch, nq - q0);
        if (staged && ws)
            stage_queries(q0, qn); // the previous launch's kernels are ahead of it on the stream
        if (grid_x) {
            ScanArgs a{};
            a.corpus = c->d_corpus;
            a.row_norms = c->d_norms;
            a.queries = !staged ? d_queries + (size_t)q0 * c->dim : (ws ? q_stage : q_stage + (size_t)q0 * q_stride);
            a.lut = c->d_lut;
            a.lists = lists;
            a.n_rows = (uint32_t)c->n_rows;
            a.d_chunks = c->d_chunks;
            a.n_rowblocks = n_rowblocks;
            a.q_stride = q_stride;
            a.k = k;
            a.n_lists = grid_x;
            a.write_all = long_k != 0;
            size_t slot = 0;
            if (profiling)
                prof_begin(c, s, slot);
            if (i8dom)
                HIP_CHECK(vfsx::launch_scan_i8dom(a, qn, grid_x, nt != 0, c->scale, s));
            else if (wide)
                HIP_CHECK(vfsx::launch_scan_i8_wide(metric, a, qn, grid_x, nt != 0, wide_unroll, s));
            else
                HIP_CHECK(vfsx::launch_scan(c->dtype, metric, a, qn, grid_x, scan_variant, nq_per_pass, s));
            if (profiling)
                prof_end(c, s, slot);
        }
        if (long_k) {
            vfsx_hit *out = d_hits + (size_t)q0 * out_stride;
            if (!grid_x) { // empty corpus: no hits
                HIP_CHECK(vfsx::launch_empty_hits(out, qn, long_k, out_stride, s));
                continue;
            }
            HIP_CHECK(hipMemsetAsync(c->long_state.p, 0, (size_t)qn * vfsx::long_state_bytes(), s)); // no cursor yet
            for (uint32_t off = 0; off < long_k; off += vfsx::kLongChunkHits) {
                const uint32_t kp = std::min(vfsx::kLongChunkHits, long_k - off);
                HIP_CHECK(vfsx::launch_long_select(lists, n_rowblocks * 64u, qn, kp, c->long_state.p, c->long_hist.p,
                                                   c->long_sel.p, c->explicit_ids ? c->d_ids : nullptr, out, c->row_base,
                                                   c->id_base, out_stride, off, s));
            }
            continue;
        }
        SelectArgs sa{};
        sa.lists = lists;
        sa.ids = c->explicit_ids ? c->d_ids : nullptr;
        sa.out = d_hits + (size_t)q0 * out_stride;
        sa.row_base = c->row_base;
        sa.id_base = c->id_base;
        sa.n_lists = grid_x;
        sa.k = k;
        sa.out_stride = out_stride;
        HIP_CHECK(vfsx::launch_select(sa, qn, s));
    }
}

// the exact scan: enqueue the whole search on `s`; no host synchronisation
void exact_search_device(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k, int metric,
                         vfsx_hit *d_hits, hipStream_t s, const ScanWorkspace *ws = nullptr)
{
    if (metric < VFSX_EUCLIDEAN || metric > VFSX_EUCLIDEAN_I8DOMAIN)
        fail(VFSX_ERR_INVALID_VECTOR, "unknown distance method");
    const bool i8dom = metric == VFSX_EUCLIDEAN_I8DOMAIN;
    if (i8dom && c->dtype != VFSX_I8)
        fail(VFSX_ERR_INVALID_VECTOR, "the integer-domain distance needs a quantized (int8) corpus");
    if (i8dom && c->dim > 32768u)
        fail(VFSX_ERR_INVALID_VECTOR, "the integer-domain distance supports dim <= 32768");
    if (k == 0 || k > VFSX_MAX_K)
        fail(VFSX_ERR_INVALID_VECTOR, "k (result_limit) must be in 1.." + std::to_string(VFSX_MAX_K));
    if (nq == 0)
        return;
    if (!d_queries || !d_hits)
        fail(VFSX_ERR_INVALID_VECTOR, "null query/output buffer");
    use_device(c);
    if (c->prof_used > 4096)
        prof_collect(c);
    constexpr uint32_t P = vfsx::kMaxKPerPass;
    if (k > vfsx::kLongResultK && !ws && !i8dom) { // long results: one scan, then the radix select (any k)
        exact_scan_pass(c, d_queries, nq, 1, metric, d_hits, s, nullptr, k);
        return;
    }
    // the list path: per-wave lists of at most P entries
    if (k > P && ws)
        fail(VFSX_ERR_INVALID_VECTOR, "vfsx_search_device_noalloc: k is limited to " + std::to_string(P) +
                                          " (a longer result needs the key buffer of the radix select)");
    if (k > P)
        fail(VFSX_ERR_INVALID_VECTOR, "the integer-domain extension supports k <= " + std::to_string(P));
    exact_scan_pass(c, d_queries, nq, k, metric, d_hits, s, ws);
}

void free_mirror(vfsx_corpus *c)
{
    c->va_mirror.release();
    c->va_rn2.release();
    c->va_inv.release();
    c->d_rn2 = nullptr;
    c->d_inv_rn = nullptr;
    c->rn2_rows = 0;
    c->d_mirror = nullptr;
    c->mirror_rows = 0;
    c->mirror_bytes = 0;
}

// relative allowance for the f32 errors in the Euclidean batch check (vfsx_mfma.hip): the reference's
// sequential sums of d terms, the row norm b vs ||x||, the query norm a vs ||q||, and the epilogue's roundings
float mfma_gamma(uint32_t k_pad) { return ((float)k_pad + 32.0f) * 0x1p-23f; }

// (re)build what the batch path keeps per row when rows were appended since: f32 stores, the bf16 or int8 mirror;
// int8 stores, only the per-row factors (the prefilter multiplies the rb64 corpus itself)
constexpr float kMirrorI8MaxResidual = 0x1p-6f; // auto: above this R_x the int8 mirror's margin is too wide

bool mirror_is_i8(const vfsx_corpus *c) { return c->dtype == VFSX_F32 && c->mirror_kind == vfsx_corpus::kMirrorI8; }
bool prefilter_is_i8(const vfsx_corpus *c) { return c->dtype == VFSX_I8 || mirror_is_i8(c); }
uint32_t mirror_i8_chunks(const vfsx_corpus *c) { return (c->dim + 15u) / 16u; }

void ensure_mirror(vfsx_corpus *c, hipStream_t s)
{
    if (c->mirror_rows == c->n_rows && c->d_rn2)
        return;
    for (;;) {
        const bool i8 = prefilter_is_i8(c);
        const vfsx::MfmaGeometry g = vfsx::mfma_geometry((uint32_t)c->n_rows, c->dim, i8);
        const size_t n_pad = (size_t)g.n_rowtiles * 256u;
        // int8 operands are sized for the corpus' CAPACITY, so that appended rows only add their own part below
        const vfsx::MfmaGeometry gc = vfsx::mfma_geometry((uint32_t)std::max<uint64_t>(c->capacity, c->n_rows), c->dim, i8);
        const size_t cap_pad = i8 ? (size_t)gc.n_rowtiles * 256u : n_pad;
        const size_t want = c->dtype == VFSX_I8 ? 0
                            : (mirror_is_i8(c) ? (cap_pad / 64u) * mirror_i8_chunks(c) * 1024u : g.mirror_bytes);
        // appended rows on an int8 operand that was complete and good before: only rows [mirror_rows, n_rows) -- the
        // operands are segmented arrays like the corpus (VmArray), so growing them keeps what is already built
        uint32_t row0 = 0;
        if (i8 && c->d_rn2 && c->mirror_rows > 0 && c->mirror_rows < c->n_rows && !c->mirror_bad && c->mirror_bytes != 0)
            row0 = (uint32_t)c->mirror_rows;
        if (c->mirror_bytes == 0 && (c->va_mirror.base || c->va_rn2.base))
            free_mirror(c); // the mirror's kind changed: start over
        if (!c->va_rn2.base && !c->va_rn2.reserved) { // first use: address ranges for the device's worth of rows
            const size_t rows_max = ((size_t)c->max_rows + 255u) / 256u * 256u;
            if (c->va_corpus.reserved) {
                c->va_mirror.min_slab = 64u << 20;
                c->va_mirror.init(c->device, c->dtype == VFSX_I8 ? 0 : rows_max * ((size_t)g.k_pad + 16u) * 2u);
                c->va_rn2.init(c->device, rows_max * 2u * sizeof(float));
                c->va_inv.init(c->device, rows_max * sizeof(float));
            }
        }
        const size_t mirror_before = c->va_mirror.mapped, inv_before = c->va_inv.mapped;
        if (want > c->va_mirror.mapped)
            c->va_mirror.grow(want, row0 ? mirror_before : 0, s);
        c->d_mirror = (uint4 *)c->va_mirror.base;
        c->mirror_bytes = want ? c->va_mirror.mapped : 1; // (int8 stores keep no mirror: any non-zero value = "built")
        if (!c->d_mirror_stats)
            HIP_CHECK(hipMalloc((void **)&c->d_mirror_stats, 2 * sizeof(uint32_t)));
        uint32_t st0[2] = {0, 0};
        if (row0)
            std::memcpy(&st0[1], &c->mirror_rx, sizeof(float)); // the maximum residual carries over
        HIP_CHECK(hipMemcpyAsync(c->d_mirror_stats, st0, sizeof(st0), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s)); // (st0 lives on this stack frame)
        if (cap_pad > c->rn2_rows) { // Euclidean batches: (b, b^2 (1 - gamma)) per row; int8 operands: the row factor
            c->va_rn2.grow(cap_pad * 2u * sizeof(float), row0 ? c->rn2_rows * 2u * sizeof(float) : 0, s);
            c->va_inv.grow(cap_pad * sizeof(float), row0 ? c->rn2_rows * sizeof(float) : 0, s);
            c->rn2_rows = std::min(c->va_rn2.mapped / (2u * sizeof(float)), c->va_inv.mapped / sizeof(float));
        }
        c->d_rn2 = (float *)c->va_rn2.base;
        c->d_inv_rn = (float *)c->va_inv.base;
        if (row0 && mirror_is_i8(c)) { // what growth added starts at zero (rows beyond n_rows are read and masked)
            if (c->va_mirror.mapped > mirror_before)
                HIP_CHECK(hipMemsetAsync(c->va_mirror.base + mirror_before, 0, c->va_mirror.mapped - mirror_before, s));
            if (c->va_inv.mapped > inv_before)
                HIP_CHECK(hipMemsetAsync(c->va_inv.base + inv_before, 0, c->va_inv.mapped - inv_before, s));
        }
        if (mirror_is_i8(c)) {
            if (!row0) {
                HIP_CHECK(hipMemsetAsync(c->d_mirror, 0, c->va_mirror.mapped, s)); // rows beyond n_rows are read (and masked)
                HIP_CHECK(hipMemsetAsync(c->d_inv_rn, 0, c->va_inv.mapped, s));
            }
            HIP_CHECK(vfsx::launch_mfma_mirror_i8(c->d_corpus, c->d_norms, c->d_mirror, c->d_inv_rn, (uint32_t)c->n_rows,
                                                  c->d_chunks, mirror_i8_chunks(c), g.k_pad, c->d_mirror_stats, s, row0));
        } else if (c->dtype == VFSX_F32) {
            HIP_CHECK(vfsx::launch_mfma_mirror(c->dtype, c->d_corpus, c->d_norms, c->d_lut, c->d_mirror,
                                               (uint32_t)c->n_rows, c->dim, c->d_chunks, g, c->d_mirror_stats, s));
        }
        HIP_CHECK(vfsx::launch_mfma_rownorm_pairs(c->d_norms, (uint32_t)c->n_rows, (uint32_t)n_pad, mfma_gamma(g.k_pad),
                                                  c->d_rn2, c->dtype == VFSX_I8 ? c->d_inv_rn : nullptr, c->scale,
                                                  c->d_mirror_stats, s, row0));
        uint32_t st[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(st, c->d_mirror_stats, sizeof(st), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        c->mirror_bad = st[0] != 0;
        std::memcpy(&c->mirror_rx, &st[1], sizeof(float));
        if (mirror_is_i8(c) && c->mirror_pref == 0 && !c->mirror_bad && c->mirror_rx > kMirrorI8MaxResidual) {
            c->mirror_kind = vfsx_corpus::kMirrorBf16; // heavy-tailed rows: the bf16 mirror's bound is ~3.5x tighter
            c->mirror_bytes = 0;                       // (forces the reallocation above)
            continue;
        }
        break;
    }
    c->mirror_rows = c->n_rows;
}

// cosine batches on the matrix cores: prefilter -> select -> exact re-score + safety check,
// then the exact scan for every query the check could not clear (see vfsx_mfma.hip).
// Synchronises `s` once to read the per-query flags.
void mfma_search_device(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k, int metric,
                        vfsx_hit *d_hits, hipStream_t s)
{
    use_device(c);
    ensure_mirror(c, s);
    const bool i8 = prefilter_is_i8(c);
    // (query tiles per prefilter launch: a call of more queries than one round holds runs several launches of
    // q_round; the geometry is the first one's, the others are at least as large)
    const vfsx::MfmaGeometry g = vfsx::mfma_geometry((uint32_t)c->n_rows, c->dim, i8, c->n_cus, nq);
    const size_t qm_elem = i8 ? 1u : 2u; // query mirror: int8 or bf16
    // the int8 prefilter's row operand: the int8 store itself, or the int8 mirror of an f32 store
    const uint4 *a_rows = mirror_is_i8(c) ? c->d_mirror : c->d_corpus;
    const uint32_t a_chunks = mirror_is_i8(c) ? mirror_i8_chunks(c) : c->d_chunks;
    const uint64_t a_bytes = mirror_is_i8(c) ? c->mirror_bytes : c->corpus_bytes(c->capacity);
    // first-pass candidates per query; queries with more rows above their safety margin are decided again from
    // 256 candidates in the second device pass below (C4, k = 10: 64 -> 68.3 ms per batch, 128 -> 70.6 ms, no
    // query left for the exact scan either way)
    uint32_t kp = k <= 16 ? 64u : (k <= 32 ? 128u : 256u);
    if (c->mfma_kp_override == 64u || c->mfma_kp_override == 128u || c->mfma_kp_override == 256u)
        kp = std::max<uint32_t>(c->mfma_kp_override, k <= 64u ? 64u : 256u); // experiments: VFSX_MFMA_KP
    const int euclid = is_cosine(metric) ? 0 : 1;

    const uint32_t q_stride_pad = c->d_chunks * c->epc;
    const bool need_pad = (q_stride_pad != c->dim);
    const float *q_ptr = d_queries;
    uint32_t q_stride = c->dim;
    if (need_pad) {
        c->q_pad.reserve((size_t)nq * q_stride_pad);
        HIP_CHECK(vfsx::launch_query_pad(d_queries, nq, c->dim, c->q_pad.p, q_stride_pad, s));
        q_ptr = c->q_pad.p;
        q_stride = q_stride_pad;
    }
    // queries per round: candidate workspace (3 floats per list and query) capped at 1 GiB
    const uint64_t per_q = (uint64_t)g.n_lists * 3u * sizeof(float);
    uint32_t q_round = (uint32_t)std::max<uint64_t>(256, std::min<uint64_t>(nq, (1ull << 30) / per_q) / 256u * 256u);
    q_round = std::min<uint32_t>(q_round, (nq + 255u) / 256u * 256u);
    c->cand.reserve((size_t)q_round * 3u * g.n_lists);
    c->sel.reserve((size_t)q_round * 256u); // kp in the first pass, 256 in the second
    c->q_mirror.reserve((size_t)q_round * g.k_pad * qm_elem);
    c->q_bad.reserve(nq);
    c->q_norm.reserve(nq);
    c->q_eps.reserve(nq);
    c->q_dk.reserve(nq);
    if (i8)
        c->q_scale.reserve(nq);
    c->q_flags.reserve(nq);
    if (nq > c->h_flags_cap) { // pinned, so that the read-back is one small asynchronous copy (no pageable staging)
        if (c->h_flags)
            (void)hipHostFree(c->h_flags);
        c->h_flags = nullptr;
        c->h_flags_cap = 0;
        const size_t cap = std::max<size_t>(256, nq);
        HIP_CHECK(hipHostMalloc((void **)&c->h_flags, cap * sizeof(int), hipHostMallocDefault));
        c->h_flags_cap = cap;
    }
    std::memset(c->h_flags, 0, (size_t)nq * sizeof(int));
    // Calls of a few queries (one round): the flags are read right after the first pass and the second pass is only
    // launched when one of them is set -- almost never, so the usual call saves two launches (~10 us of 210 at C2
    // size); a flagged call pays one more stream synchronisation.
    const bool early_flags = nq <= 8u && kp < 256u;
    bool flags_known = false;

    for (uint32_t q0 = 0; q0 < nq; q0 += q_round) {
        const uint32_t qn = std::min(q_round, nq - q0);
        if (i8)
            HIP_CHECK(vfsx::launch_mfma_query_prep_i8(d_queries + (size_t)q0 * c->dim, qn, c->dim, c->q_mirror.p, g.KS,
                                                      c->q_bad.p + q0, c->q_norm.p + q0, c->q_eps.p + q0,
                                                      c->q_scale.p + q0, mirror_is_i8(c) ? c->d_mirror_stats : nullptr,
                                                      s));
        else
            HIP_CHECK(vfsx::launch_mfma_query_prep(d_queries + (size_t)q0 * c->dim, qn, c->dim, c->q_mirror.p, g.KS,
                                                   c->q_bad.p + q0, c->q_norm.p + q0, c->q_eps.p + q0,
                                                   c->d_mirror_stats, s));
        size_t slot = 0;
        prof_begin(c, s, slot);
        if (i8)
            HIP_CHECK(vfsx::launch_mfma_prefilter_i8(a_rows, a_chunks, a_bytes, c->q_mirror.p, c->cand.p, (uint32_t)c->n_rows, g, qn, euclid,
                                                     c->d_rn2, c->d_inv_rn, c->q_norm.p + q0, c->q_eps.p + q0,
                                                     c->q_scale.p + q0, s));
        else
            HIP_CHECK(vfsx::launch_mfma_prefilter(c->d_mirror, c->q_mirror.p, c->cand.p, (uint32_t)c->n_rows, g, qn,
                                                  euclid, c->d_rn2, c->q_norm.p + q0, c->q_eps.p + q0, s));
        prof_end(c, s, slot);
        HIP_CHECK(vfsx::launch_mfma_select(c->cand.p, c->sel.p, qn, g.n_lists, kp, nullptr, s));
        vfsx::MfmaRescoreArgs a{};
        a.corpus = c->d_corpus;
        a.row_norms = c->d_norms;
        a.queries = q_ptr + (size_t)q0 * q_stride;
        a.lut = c->d_lut;
        a.sel = c->sel.p;
        a.q_bad = c->q_bad.p + q0;
        a.ids = c->explicit_ids ? c->d_ids : nullptr;
        a.out = d_hits + (size_t)q0 * k;
        a.flags = c->q_flags.p + q0;
        a.n_rows = (uint32_t)c->n_rows;
        a.d_chunks = c->d_chunks;
        a.q_stride = q_stride;
        a.k = k;
        a.kp = kp;
        a.q_eps = c->q_eps.p + q0;
        a.q_scale = i8 ? c->q_scale.p + q0 : nullptr;
        a.row_base = c->row_base;
        a.id_base = c->id_base;
        a.strip_tiles = g.strip_code(); // (tiles per strip | kStripOwnLists: how a candidate's list decodes)
        a.euclid = euclid;
        a.q_norm = c->q_norm.p + q0;
        a.gamma = mfma_gamma(g.k_pad);
        a.dk_out = c->q_dk.p + q0;
        HIP_CHECK(vfsx::launch_mfma_rescore(c->dtype, a, qn, s));
        // second pass, entirely on the device: a query that failed because more than kp rows sit above its
        // safety margin gets the widest candidate set the re-score kernel takes (256) and is decided
        // again; queries that passed exit at once.  (A list with three entries above the margin stays
        // flagged: only the exact scan can look inside it.)
        bool second_pass = kp < 256u;
        if (early_flags) {
            HIP_CHECK(hipMemcpyAsync(c->h_flags, c->q_flags.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            second_pass = false;
            for (uint32_t q = 0; q < nq; ++q)
                second_pass = second_pass || c->h_flags[q] != 0;
            flags_known = !second_pass;
        }
        if (second_pass) {
            static const bool wide_select = std::getenv("VFSX_MFMA_WIDE") && std::getenv("VFSX_MFMA_WIDE")[0] == 's';
            if (wide_select) // (A/B: round 2's sorted 256-deep select)
                HIP_CHECK(vfsx::launch_mfma_select(c->cand.p, c->sel.p, qn, g.n_lists, 256u, a.flags, s));
            else // every candidate above the first pass' margin, unsorted (mfma_filter_kernel)
                HIP_CHECK(vfsx::launch_mfma_filter(c->cand.p, c->sel.p, qn, g.n_lists, a.flags, a.q_bad, c->q_dk.p + q0,
                                                   a.q_eps, a.q_scale, a.q_norm, a.gamma, euclid, s));
            a.dk_out = nullptr;
            a.kp = 256u;
            a.only_flagged = 1;
            HIP_CHECK(vfsx::launch_mfma_rescore(c->dtype, a, qn, s));
        }
    }
    if (!flags_known) {
        HIP_CHECK(hipMemcpyAsync(c->h_flags, c->q_flags.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    c->mfma_queries += nq;
    {
        uint32_t flagged = 0;
        for (uint32_t q = 0; q < nq; ++q)
            flagged += c->h_flags[q] ? 1u : 0u;
        if (nq < 4u) { // see mfma_min_batch_for
            c->small_win_queries += nq;
            c->small_win_flagged += flagged;
            if (c->small_win_queries >= 64u) {
                if (c->small_win_flagged * 4u > c->small_win_queries)
                    c->small_off = 8192u;
                c->small_win_queries = c->small_win_flagged = 0;
            }
        }
        bool to_bf16 = false;
        if (mirror_is_i8(c) && c->mirror_pref == 0) {
            // auto: does the int8 mirror's wider margin send too many of this data's queries back?  (Queries the
            // path never took -- zero / out-of-range norms -- are flagged too and say nothing about the mirror.)
            // Judged over windows of >= 64 queries, so that one flagged single-query call decides nothing.
            uint32_t by_margin = 0;
            if (flagged) {
                std::vector<int> bad(nq);
                HIP_CHECK(hipMemcpyAsync(bad.data(), c->q_bad.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                for (uint32_t q = 0; q < nq; ++q)
                    by_margin += (c->h_flags[q] && !bad[q]) ? 1u : 0u;
            }
            c->mirror_win_queries += nq;
            c->mirror_win_flagged += by_margin;
            if (c->mirror_win_queries >= 64u) {
                to_bf16 = (uint64_t)c->mirror_win_flagged * 8u > c->mirror_win_queries;
                c->mirror_win_queries = c->mirror_win_flagged = 0;
            }
        }
        if (to_bf16) { // bf16 mirror from the next batch on (rebuilt by ensure_mirror), for good
            c->mirror_kind = vfsx_corpus::kMirrorBf16;
            c->mirror_rows = 0;
            c->mirror_bytes = 0;
        } else if (flagged * 2u > nq) {
            c->mfma_backoff_next = std::min<uint32_t>(c->mfma_backoff_next * 2u + 1u, 63u);
            c->mfma_backoff = c->mfma_backoff_next;
        } else {
            c->mfma_backoff_next = 0;
        }
    }
    // Exact scan for the queries the check flagged (not part of the profiled kernel: the event brackets
    // on this path time the prefilter GEMM only).  They are compacted into one batch first, so that they
    // share corpus passes (8 queries per pass) wherever they sat in the caller's batch.
    std::vector<uint32_t> flagged;
    for (uint32_t q = 0; q < nq; ++q)
        if (c->h_flags[q])
            flagged.push_back(q);
    if (!flagged.empty()) {
        const bool prof_saved = c->profiling;
        c->profiling = false;
        const uint32_t nf = (uint32_t)flagged.size();
        if (nf == nq) {
            exact_search_device(c, d_queries, nq, k, metric, d_hits, s);
        } else {
            c->fb_idx.reserve(nf);
            c->fb_queries.reserve((size_t)nf * c->dim);
            c->fb_hits.reserve((size_t)nf * k);
            HIP_CHECK(hipMemcpyAsync(c->fb_idx.p, flagged.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            HIP_CHECK(vfsx::launch_move_records(d_queries, c->fb_queries.p, c->fb_idx.p, nf, c->dim, 0, s));
            exact_search_device(c, c->fb_queries.p, nf, k, metric, c->fb_hits.p, s);
            HIP_CHECK(vfsx::launch_move_records(c->fb_hits.p, d_hits, c->fb_idx.p, nf,
                                                (uint32_t)(k * sizeof(vfsx_hit) / sizeof(uint32_t)), 1, s));
            HIP_CHECK(hipStreamSynchronize(s)); // `flagged` (pageable host memory) must outlive the H2D copy
        }
        c->mfma_fallbacks += nf;
        c->profiling = prof_saved;
    }
}

// A search on a stream other than the previous search's: both use the corpus' workspaces, so the new one must
// start after everything the previous one enqueued.  The previous stream is the CALLER's and may have been
// destroyed since (a stream per request) -- and HIP does not survive hipEventRecord / hipStreamWaitEvent on a dead
// handle (it faults in the runtime), so that handle is never touched again: the switch waits for the whole device
// instead.  Costs nothing while the caller stays on one stream; a host that alternates streams on ONE corpus handle
// pays a device-wide wait per switch (the contract is one search at a time per handle, include/vfsx.h).
void chain_search_stream(vfsx_corpus *c, hipStream_t s)
{
    if (c->searched && c->last_search_stream != s) {
        c->last_search_stream = s; // first: a failure below must not leave the old handle behind
        use_device(c);
        HIP_CHECK(hipDeviceSynchronize());
    }
    c->last_search_stream = s;
    c->searched = true;
}

// Smallest batch the matrix-core path takes (profiles/r02/batch_crossover.txt).  From 4 queries on it wins on every
// shape measured.  Below that the exact scan shares one pass among the queries -- but on an f32 store the prefilter
// reads the MIRROR, 1 B (int8) or 2 B (bf16) per element where the scan reads 4: once the bytes saved outweigh the
// path's fixed ~120 us (launch chain, re-score gather, flag read-back) even a single query is faster through it
// (1M x 768: 0.201 vs 0.449 ms; 12.5M x 768: 1.51 vs 5.47 ms) -- from 1.75e8 elements on with the int8 mirror, 3e8 with
// the bf16 one (profiles/r03/single_query_crossover.txt: round 3's shorter chain moved both down from 2.5e8 / 4e8).
// VFSX_MFMA_MIN_BATCH overrides.
// Only for k <= 32 (at k = 100 on random 768-d rows most single queries fail the check: 1M x 768 2150 q/s exact,
// 870 through the path), and only while such calls mostly pass: a window of 64 of them with more than a quarter
// flagged keeps the next 8192 on the exact scan (then the path is probed again).
uint32_t mfma_min_batch_for(const vfsx_corpus *c, uint32_t k)
{
    if (c->mfma_min_batch)
        return c->mfma_min_batch;
    if (c->dtype == VFSX_F32 && k <= 32u && c->small_off == 0u) {
        const uint64_t elems = c->n_rows * (uint64_t)c->dim;
        if (elems >= (mirror_is_i8(c) ? 175000000ull : 300000000ull))
            return 1;
    }
    return 4;
}

void search_device_impl(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k, int metric,
                        vfsx_hit *d_hits, hipStream_t s)
{
    if (metric < VFSX_EUCLIDEAN || metric > VFSX_EUCLIDEAN_I8DOMAIN)
        fail(VFSX_ERR_INVALID_VECTOR, "unknown distance method");
    if (k == 0 || k > VFSX_MAX_K)
        fail(VFSX_ERR_INVALID_VECTOR, "k (result_limit) must be in 1.." + std::to_string(VFSX_MAX_K));
    if (nq == 0)
        return;
    if (!d_queries || !d_hits)
        fail(VFSX_ERR_INVALID_VECTOR, "null query/output buffer");
    chain_search_stream(c, s);
    if (nq < 4u && c->small_off > 0u)
        c->small_off -= std::min(c->small_off, nq);
    // the reference's four metrics; the integer-domain extension has its own kernel
    const bool metric_ok = metric != VFSX_EUCLIDEAN_I8DOMAIN;
    const bool mfma_ok = c->mfma_enabled && metric_ok && nq >= mfma_min_batch_for(c, k) && k <= vfsx::kMfmaMaxK && c->dim <= vfsx::kMfmaMaxDim &&
                         c->n_rows >= 16384 && c->n_rows <= vfsx::kMfmaMaxRows && !(c->mirror_rows == c->n_rows && c->mirror_bad);
    if (mfma_ok && c->mfma_backoff > 0) {
        --c->mfma_backoff; // see vfsx_corpus::mfma_backoff
        exact_search_device(c, d_queries, nq, k, metric, d_hits, s);
        return;
    }
    if (mfma_ok) {
        use_device(c);
        ensure_mirror(c, s);
        if (!c->mirror_bad) {
            mfma_search_device(c, d_queries, nq, k, metric, d_hits, s);
            return;
        }
    }
    exact_search_device(c, d_queries, nq, k, metric, d_hits, s);
}

// ---- the entry point that allocates nothing (vfsx_search_device_noalloc) ----------------------------------
// Its workspace is this file-scope device array instead of hipMalloc'ed buffers: every GPU the library runs on
// holds one copy (9 MiB) until the process ends.  In host code the NAME `scratch` is only a stand-in for the
// symbol: it is never passed to a kernel or a HIP call.  Its device address is looked up once per device
// with hipGetSymbolAddress (that device current) and kept in g_scratch_ptr[device]; kernels and memsets get
// that pointer.
constexpr size_t kScratchListsBytes = 8u << 20; // per-workgroup top-k lists: <= 1024 lists x k <= 1024 x 8 B
constexpr size_t kScratchQueryBytes = 1u << 20; // staged (padded / quantized) queries of one launch
static __device__ __align__(256) unsigned char scratch[kScratchListsBytes + kScratchQueryBytes];

constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[kMaxScratchDevices] = {};

// device address of this device's copy of `scratch` (looked up on first use; vfsx_init does it up front)
void *scratch_for_device(int device)
{
    if (device < 0 || device >= kMaxScratchDevices)
        fail(VFSX_ERR_NO_DEVICE, "HIP device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        // the symbol's address is per device and is looked up with that device current; the caller's current
        // device is put back afterwards (vfsx_init(n) must not leave device n - 1 selected on the caller's thread)
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
        HIP_CHECK(hipSetDevice(device));
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (prev >= 0 && prev != device)
            (void)hipSetDevice(prev);
        HIP_CHECK(e);
        if (!p)
            fail(VFSX_ERR_INTERNAL, "hipGetSymbolAddress returned a null address for the static workspace");
        g_scratch_ptr[device] = p;
    }
    return g_scratch_ptr[device];
}

vfsx_corpus *checked(vfsx_corpus *c)
{
    if (!c)
        fail(VFSX_ERR_INVALID_VECTOR, "null corpus handle");
    return c;
}

// checked + whatever small appends staged: every entry point that reads the store starts here
vfsx_corpus *ready(vfsx_corpus *c)
{
    checked(c);
    commit_pending(c);
    return c;
}

void fetch_rows(vfsx_corpus *c, const uint64_t *rows, uint64_t n, int as_f32, void *out)
{
    if (n == 0)
        return;
    if (!rows || !out)
        fail(VFSX_ERR_INVALID_VECTOR, "null buffer");
    for (uint64_t i = 0; i < n; ++i)
        if (rows[i] >= c->n_rows)
            fail(VFSX_ERR_INVALID_VECTOR, "row index out of range");
    use_device(c);
    const size_t elem = (c->dtype == VFSX_I8 && !as_f32) ? 1 : 4;
    c->row_idx.reserve(n);
    c->row_out.reserve((size_t)n * c->dim * elem);
    HIP_CHECK(hipMemcpyAsync(c->row_idx.p, rows, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(vfsx::launch_get_rows(c->dtype, c->d_corpus, c->dim, c->d_chunks, c->d_lut, c->row_idx.p, n,
                                    as_f32, c->row_out.p, c->stream));
    HIP_CHECK(hipMemcpyAsync(out, c->row_out.p, (size_t)n * c->dim * elem, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

} // namespace

extern "C" {

int vfsx_device_count(void) { return device_count_quiet(); }

int vfsx_init(int n_devices)
{
    return guarded([&] {
        if (n_devices < 0)
            fail(VFSX_ERR_INVALID_VECTOR, "negative device count");
        const int have = device_count_quiet();
        if (have < std::max(n_devices, 1))
            fail(VFSX_ERR_NO_DEVICE, "vfsx_init: " + std::to_string(n_devices) + " HIP device(s) wanted, " +
                                         std::to_string(have) + " visible (vfsx has no CPU fallback)");
        for (int d = 0; d < std::max(n_devices, 1); ++d) {
            hipDeviceProp_t prop;
            HIP_CHECK(hipGetDeviceProperties(&prop, d));
            if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !std::getenv("VFSX_ALLOW_ANY_ARCH"))
                fail(VFSX_ERR_NO_DEVICE, std::string("vfsx is built for gfx950 only, device ") + std::to_string(d) +
                                             " is " + prop.gcnArchName);
            (void)scratch_for_device(d); // vfsx_search_device_noalloc then does no set-up work on its first call
        }
    });
}

const char *vfsx_version(void) { return "vfsx 0.1 gfx950 rb64"; }

vfsx_corpus *vfsx_corpus_create(uint32_t dim, int dtype, float scale, uint64_t capacity_rows, int device)
{
    vfsx_corpus *c = nullptr;
    int rc = guarded([&] {
        if (dim == 0)
            fail(VFSX_ERR_INVALID_VECTOR, "Vector data cannot be empty"); // storage_manager.rs:319
        if (dtype != VFSX_F32 && dtype != VFSX_I8)
            fail(VFSX_ERR_INVALID_VECTOR, "unknown corpus dtype");
        if (dtype == VFSX_I8 && !(scale == scale) )
            fail(VFSX_ERR_INVALID_VECTOR, "scale_factor must not be NaN");
        int ndev = device_count_quiet();
        if (ndev <= 0)
            fail(VFSX_ERR_NO_DEVICE, "no HIP device visible: vfsx has no CPU fallback");
        if (device < 0 || device >= ndev)
            fail(VFSX_ERR_NO_DEVICE, "HIP device index out of range");
        HIP_CHECK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !std::getenv("VFSX_ALLOW_ANY_ARCH"))
            fail(VFSX_ERR_NO_DEVICE, std::string("vfsx is built for gfx950 only, device is ") + prop.gcnArchName);
        c = new vfsx_corpus();
        c->device = device;
        c->dim = dim;
        c->dtype = dtype;
        c->scale = dtype == VFSX_I8 ? scale : 1.0f;
        c->epc = vfsx::elems_per_chunk(dtype);
        c->d_chunks = (dim + c->epc - 1) / c->epc;
        c->max_blocks = (uint32_t)prop.multiProcessorCount * (uint32_t)env_u64("VFSX_BLOCKS_PER_CU", 4);
        // scan variant (tuned on MI355X): deep unroll for long f32 rows,
        // 4 for int8 (the dequant keeps the VALU/LDS busier); chosen per launch in
        // search_device_impl together with the nontemporal hint.
        c->unroll_override = (int)env_u64("VFSX_SCAN_UNROLL", 0);
        c->nt_override = (int)env_u64("VFSX_SCAN_NT", 2);
        c->nq_per_pass_max = env_u64("VFSX_SCAN_NQ", 8);
        c->pipe_override = (int)env_u64("VFSX_SCAN_PIPE", 2);
        c->balanced_grid = env_u64("VFSX_BALANCED_GRID", 1) != 0;
        c->n_cus = (uint32_t)prop.multiProcessorCount;
        c->i8_wide = (uint32_t)std::min<uint64_t>(env_u64("VFSX_I8_WIDE", 3), 3);
        c->mfma_min_batch = (uint32_t)env_u64("VFSX_MFMA_MIN_BATCH", 0); // 0: the measured crossovers (mfma_min_batch_for)
        c->mfma_kp_override = (uint32_t)env_u64("VFSX_MFMA_KP", 0);
        HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_CHECK(hipMalloc((void **)&c->d_lut, 256 * sizeof(float)));
        if (dtype == VFSX_I8)
            HIP_CHECK(vfsx::launch_lut(c->scale, c->d_lut, c->stream));
        else
            HIP_CHECK(hipMemsetAsync(c->d_lut, 0, 256 * sizeof(float), c->stream));
        // address ranges for the arrays that grow with the store: as many rows as the device's memory could hold
        {
            const uint64_t by_mem = std::max<uint64_t>(64, (uint64_t)prop.totalGlobalMem / c->block_bytes() * 64u);
            c->max_rows = std::min<uint64_t>(by_mem, kMaxCapacityRows);
            c->va_corpus.min_slab = 64u << 20; // row data: 64 MiB .. 1 GiB per slab; side arrays: 2 MiB .. 1 GiB
            c->va_corpus.init(device, c->corpus_bytes(c->max_rows));
            if (c->va_corpus.reserved) {
                c->va_norms.init(device, c->max_rows * sizeof(float));
                c->va_ids.init(device, c->max_rows * sizeof(uint64_t));
                if (!c->va_norms.reserved || !c->va_ids.reserved) { // all three or none
                    c->va_corpus.release();
                    c->va_norms.release();
                    c->va_ids.release();
                }
            }
        }
        reserve_rows(c, std::max<uint64_t>(capacity_rows, 64));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    });
    if (rc != VFSX_OK) {
        if (c)
            vfsx_corpus_destroy(c);
        return nullptr;
    }
    return c;
}

void vfsx_corpus_destroy(vfsx_corpus *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    for (auto &pe : c->prof_events) {
        (void)hipEventDestroy(pe.first);
        (void)hipEventDestroy(pe.second);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->h_stage[i])
            (void)hipHostFree(c->h_stage[i]);
        if (c->d_stage[i])
            (void)hipFree(c->d_stage[i]);
        if (c->stage_done[i])
            (void)hipEventDestroy(c->stage_done[i]);
    }
    free_mirror(c);
    if (c->d_mirror_stats)
        (void)hipFree(c->d_mirror_stats);
    c->q_mirror.release();
    c->q_bad.release();
    c->q_norm.release();
    c->q_eps.release();
    c->q_dk.release();
    c->q_scale.release();
    c->long_state.release();
    c->long_hist.release();
    c->long_sel.release();
    c->fb_idx.release();
    c->fb_queries.release();
    c->fb_hits.release();
    c->q_flags.release();
    c->cand.release();
    c->sel.release();
    c->lists.release();
    c->q_in.release();
    c->q_pad.release();
    c->hits.release();
    if (c->h_flags)
        (void)hipHostFree(c->h_flags);
    c->h_flags = nullptr;
    if (c->h_call)
        (void)hipHostFree(c->h_call);
    c->h_call = c->d_call = nullptr;
    c->row_idx.release();
    c->row_out.release();
    c->va_corpus.release();
    c->va_norms.release();
    c->va_ids.release();
    if (c->d_lut)
        (void)hipFree(c->d_lut);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

int vfsx_corpus_append(vfsx_corpus *c, const void *rows, const uint64_t *ids, uint64_t n)
{
    return guarded([&] {
        checked(c);
        append_impl(c, rows, ids, n, c->dtype == VFSX_I8 ? 1 : 0);
    });
}

int vfsx_corpus_append_quantize(vfsx_corpus *c, const float *rows, const uint64_t *ids, uint64_t n)
{
    return guarded([&] {
        checked(c);
        if (c->dtype != VFSX_I8) // VFSVector::quantize on a non-quantizing store is never called
            fail(VFSX_ERR_INVALID_VECTOR, "append_quantize needs a VFSX_I8 corpus");
        append_impl(c, rows, ids, n, 2);
    });
}

uint64_t vfsx_corpus_rows(const vfsx_corpus *c) { return c ? c->n_rows + c->pend_n : 0; }
int vfsx_corpus_set_insert_batch(vfsx_corpus *c, uint32_t max_pending_rows)
{
    return guarded([&] {
        checked(c);
        c->insert_batch = max_pending_rows;
        if (max_pending_rows == 0)
            commit_pending(c);
    });
}

int vfsx_corpus_commit(vfsx_corpus *c)
{
    return guarded([&] { ready(c); });
}

uint32_t vfsx_corpus_dim(const vfsx_corpus *c) { return c ? c->dim : 0; }
int vfsx_corpus_dtype(const vfsx_corpus *c) { return c ? c->dtype : -1; }

int vfsx_corpus_set_row_base(vfsx_corpus *c, uint64_t row_base)
{
    return guarded([&] {
        checked(c);
        c->row_base = row_base;
    });
}

int vfsx_corpus_get_rows_f32(vfsx_corpus *c, const uint64_t *rows, uint64_t n, float *out)
{
    return guarded([&] { fetch_rows(ready(c), rows, n, 1, out); });
}

int vfsx_corpus_get_rows_raw(vfsx_corpus *c, const uint64_t *rows, uint64_t n, void *out)
{
    return guarded([&] { fetch_rows(ready(c), rows, n, 0, out); });
}

int vfsx_corpus_get_ids(vfsx_corpus *c, const uint64_t *rows, uint64_t n, uint64_t *out)
{
    return guarded([&] {
        ready(c);
        if (n && (!rows || !out))
            fail(VFSX_ERR_INVALID_VECTOR, "null buffer");
        for (uint64_t i = 0; i < n; ++i) {
            if (rows[i] >= c->n_rows)
                fail(VFSX_ERR_INVALID_VECTOR, "row index out of range");
            out[i] = c->explicit_ids ? c->ids_host[rows[i]] : c->id_base + rows[i];
        }
    });
}

int vfsx_corpus_fill_synthetic(vfsx_corpus *c, uint64_t seed, uint64_t row0, uint64_t n, uint64_t id0)
{
    return guarded([&] {
        ready(c);
        // an empty corpus, or one that holds implicit ids which id0 continues (appending in pieces: the growth tests)
        if (c->n_rows != 0 && (c->explicit_ids || id0 != c->id_base + c->n_rows))
            fail(VFSX_ERR_MEMTABLE, "fill_synthetic needs an empty corpus, or ids that continue the implicit run");
        if (n == 0)
            return;
        use_device(c);
        const uint64_t at = c->n_rows;
        reserve_rows(c, at + n);
        // keep each launch's grid below 2^31 blocks
        const uint64_t rows_per_launch = std::max<uint64_t>(64, ((1ull << 36) / ((uint64_t)c->d_chunks * 64u)) / 64u * 64u);
        for (uint64_t r = 0; r < n; r += rows_per_launch) {
            const uint64_t take = std::min(rows_per_launch, n - r);
            HIP_CHECK(vfsx::launch_fill_synthetic(c->dtype, c->d_corpus, c->dim, c->d_chunks, at + r, take, seed,
                                                  row0 + r, c->scale, c->stream));
        }
        HIP_CHECK(vfsx::launch_row_norms(c->dtype, c->d_corpus, c->d_chunks, c->d_lut, at, n, c->d_norms, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (at == 0)
            c->id_base = id0;
        c->n_rows = at + n;
    });
}

int vfsx_corpus_storage_info(vfsx_corpus *c, uint64_t *capacity_rows, uint64_t *segments, uint64_t *copied_bytes,
                             uint64_t *device_bytes)
{
    return guarded([&] {
        ready(c);
        const VmArray *all[] = {&c->va_corpus, &c->va_norms, &c->va_ids, &c->va_mirror, &c->va_rn2, &c->va_inv};
        uint64_t copied = 0, bytes = 0;
        for (const VmArray *a : all) {
            copied += a->copy_bytes;
            bytes += a->mapped;
        }
        if (capacity_rows)
            *capacity_rows = c->capacity;
        if (segments) // slabs behind the row data (0: the driver gave no virtual memory management; one block, grown by copy)
            *segments = c->va_corpus.reserved ? c->va_corpus.slabs.size() : 0;
        if (copied_bytes)
            *copied_bytes = copied;
        if (device_bytes)
            *device_bytes = bytes;
    });
}

int vfsx_search(vfsx_corpus *c, const float *queries, uint32_t nq, uint32_t k, int metric, uint64_t *out_ids,
                float *out_dist, uint32_t *out_count)
{
    return guarded([&] {
        ready(c);
        if (nq == 0)
            return;
        if (!queries || !out_ids || !out_dist || !out_count)
            fail(VFSX_ERR_INVALID_VECTOR, "null buffer");
        if (k == 0 || k > VFSX_MAX_K)
            fail(VFSX_ERR_INVALID_VECTOR, "k (result_limit) must be in 1.." + std::to_string(VFSX_MAX_K));
        use_device(c);
        c->q_in.reserve((size_t)nq * c->dim);
        const size_t q_bytes = (size_t)nq * c->dim * sizeof(float), h_bytes = (size_t)nq * k * sizeof(vfsx_hit);
        std::vector<vfsx_hit> h_big;
        const vfsx_hit *h = nullptr;
        if (q_bytes + h_bytes <= kPinnedCallBytes) {
            // Small calls (a query or a few, tens of hits: what Ranker::search hands over) through ONE pinned block:
            // the query is copied out of it without HIP's pageable staging, and the kernels write the hits straight
            // into it (host memory the device can address), so there is no copy back at all -- one stream
            // synchronisation is the whole host side.  C1 (100k x 128): 36.6 -> see DESIGN 6 "Host buffers".
            if (!c->h_call) {
                HIP_CHECK(hipHostMalloc(&c->h_call, kPinnedCallBytes, hipHostMallocDefault));
                HIP_CHECK(hipHostGetDevicePointer(&c->d_call, c->h_call, 0));
            }
            // hits first (24-byte records from offset 0), the queries behind them
            std::memcpy((char *)c->h_call + h_bytes, queries, q_bytes);
            HIP_CHECK(hipMemcpyAsync(c->q_in.p, (char *)c->h_call + h_bytes, q_bytes, hipMemcpyHostToDevice, c->stream));
            search_device_impl(c, c->q_in.p, nq, k, metric, (vfsx_hit *)c->d_call, c->stream);
            HIP_CHECK(hipStreamSynchronize(c->stream));
            h = (const vfsx_hit *)c->h_call;
        } else {
            c->hits.reserve((size_t)nq * k);
            HIP_CHECK(hipMemcpyAsync(c->q_in.p, queries, q_bytes, hipMemcpyHostToDevice, c->stream));
            search_device_impl(c, c->q_in.p, nq, k, metric, c->hits.p, c->stream);
            h_big.resize((size_t)nq * k);
            HIP_CHECK(hipMemcpyAsync(h_big.data(), c->hits.p, h_bytes, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            h = h_big.data();
        }
        for (uint32_t q = 0; q < nq; ++q) {
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < k; ++j) {
                const vfsx_hit &x = h[(size_t)q * k + j];
                if (x.flags & 1u) {
                    out_ids[(size_t)q * k + cnt] = x.id;
                    out_dist[(size_t)q * k + cnt] = x.dist;
                    ++cnt;
                }
            }
            out_count[q] = cnt;
        }
    });
}

int vfsx_search_device(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k, int metric,
                       vfsx_hit *d_hits, void *stream)
{
    return guarded([&] { search_device_impl(ready(c), d_queries, nq, k, metric, d_hits, (hipStream_t)stream); });
}

int vfsx_search_device_noalloc(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k, int metric,
                               vfsx_hit *d_hits, void *stream)
{
    return guarded([&] {
        checked(c);
        if (c->pend_n) // committing them copies, launches and may map memory: not in the call that allocates nothing
            fail(VFSX_ERR_MEMTABLE, "vfsx_search_device_noalloc: " + std::to_string(c->pend_n) +
                                        " appended row(s) wait in the insert batch; call vfsx_corpus_commit first");
        ScanWorkspace ws{};
        unsigned char *base = (unsigned char *)scratch_for_device(c->device);
        ws.lists = (uint64_t *)base;
        ws.lists_elems = kScratchListsBytes / sizeof(uint64_t);
        ws.q_stage = (float *)(base + kScratchListsBytes);
        ws.q_stage_elems = kScratchQueryBytes / sizeof(float);
        // always the exact scan: the matrix-core batch path grows its workspaces with hipMalloc
        exact_search_device(c, d_queries, nq, k, metric, d_hits, (hipStream_t)stream, &ws);
    });
}

int vfsx_merge_hits_device(const vfsx_hit *d_in, uint32_t n_lists, uint32_t nq, uint32_t k, vfsx_hit *d_out,
                           int device, void *stream)
{
    return guarded([&] {
        if (!d_in || !d_out || n_lists == 0 || k == 0)
            fail(VFSX_ERR_INVALID_VECTOR, "bad merge arguments");
        if (nq == 0)
            return;
        if (device_count_quiet() <= 0)
            fail(VFSX_ERR_NO_DEVICE, "no HIP device visible: vfsx has no CPU fallback");
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(vfsx::launch_merge_hits(d_in, n_lists, nq, k, d_out, (hipStream_t)stream));
    });
}

int vfsx_corpus_enable_mfma(vfsx_corpus *c, int enable)
{
    return guarded([&] {
        ready(c);
        use_device(c);
        if (enable < 0 || enable > 3)
            fail(VFSX_ERR_INVALID_VECTOR, "vfsx_corpus_enable_mfma: enable must be 0 (off), 1 (auto), 2 (bf16 mirror) or 3 (int8 mirror)");
        c->mfma_enabled = enable != 0;
        c->mfma_backoff = c->mfma_backoff_next = 0;
        c->mirror_win_queries = c->mirror_win_flagged = 0;
        c->small_win_queries = c->small_win_flagged = c->small_off = 0;
        if (!enable) {
            free_mirror(c);
            return;
        }
        int pref = enable - 1; // 0 auto, 1 bf16, 2 int8
        if (pref == 0) {
            const char *e = std::getenv("VFSX_MFMA_MIRROR");
            if (e && std::strcmp(e, "bf16") == 0)
                pref = 1;
            else if (e && std::strcmp(e, "int8") == 0)
                pref = 2;
        }
        const int kind = pref == 1 ? vfsx_corpus::kMirrorBf16 : vfsx_corpus::kMirrorI8;
        if (pref != c->mirror_pref || (pref != 0 && kind != c->mirror_kind)) {
            c->mirror_pref = pref;
            c->mirror_kind = kind; // (auto starts over on the int8 mirror)
            c->mirror_rows = 0;
            c->mirror_bytes = 0;
        }
        if (c->n_rows && c->n_rows <= vfsx::kMfmaMaxRows)
            ensure_mirror(c, c->stream);
    });
}

int vfsx_corpus_mfma_mirror(vfsx_corpus *c, int *kind, float *residual)
{
    return guarded([&] {
        checked(c);
        if (kind)
            *kind = !c->mfma_enabled ? 0 : (c->dtype == VFSX_I8 ? 3 : (mirror_is_i8(c) ? 2 : 1));
        if (residual)
            *residual = c->mfma_enabled && c->dtype == VFSX_F32 && c->mirror_rows ? c->mirror_rx : 0.0f;
    });
}

int vfsx_corpus_mfma_stats(vfsx_corpus *c, uint64_t *queries, uint64_t *fallbacks, int reset)
{
    return guarded([&] {
        checked(c);
        if (queries)
            *queries = c->mfma_queries;
        if (fallbacks)
            *fallbacks = c->mfma_fallbacks;
        if (reset)
            c->mfma_queries = c->mfma_fallbacks = 0;
    });
}

int vfsx_corpus_set_profiling(vfsx_corpus *c, int enabled)
{
    return guarded([&] {
        checked(c);
        use_device(c);
        if (!enabled)
            prof_collect(c);
        c->profiling = enabled != 0;
    });
}

int vfsx_corpus_scan_time_ms(vfsx_corpus *c, double *total_ms, uint64_t *launches, int reset)
{
    return guarded([&] {
        checked(c);
        use_device(c);
        prof_collect(c);
        if (total_ms)
            *total_ms = c->prof_ms;
        if (launches)
            *launches = c->prof_launches;
        if (reset) {
            c->prof_ms = 0.0;
            c->prof_launches = 0;
        }
    });
}

} // extern "C"

// ---- F1: ingest from the reference's on-disk log (serializer.rs:25-153) ---------------------
// The file walker and record decoder are host-only code in vfsx_host.cpp (built by g++, fuzzed under
// ASan/UBSan); here the decoded records are staged in batches and appended to the device-resident corpus.
using vfsx::LogRecord;
using vfsx::MappedFile;
using vfsx::json_escape;

extern "C" {

int vfsx_corpus_load_log(vfsx_corpus *c, const char *path, uint64_t *n_loaded)
{
    return vfsx_corpus_load_log_range(c, path, 0, ~0ull, n_loaded);
}

int vfsx_corpus_load_log_range(vfsx_corpus *c, const char *path, uint64_t first_record, uint64_t max_records,
                               uint64_t *n_loaded)
{
    return guarded([&] {
        ready(c);
        if (!path)
            fail(VFSX_ERR_INVALID_VECTOR, "null path");
        MappedFile f(path);
        const size_t elem = c->dtype == VFSX_I8 ? 1 : 4;
        const uint64_t batch_rows = std::max<uint64_t>(1, (8u << 20) / ((uint64_t)c->dim * elem));
        std::vector<uint8_t> rows;
        std::vector<uint64_t> ids;
        std::vector<vfsx_corpus::RowMeta> metas;
        rows.reserve((size_t)batch_rows * c->dim * elem);
        uint64_t total = 0;
        auto flush = [&] {
            if (ids.empty())
                return;
            if (c->meta.size() < c->n_rows)
                c->meta.resize(c->n_rows); // rows appended earlier without metadata
            append_impl(c, rows.data(), ids.data(), ids.size(), c->dtype == VFSX_I8 ? 1 : 0);
            for (auto &m : metas)
                c->meta.push_back(std::move(m));
            total += ids.size();
            rows.clear();
            ids.clear();
            metas.clear();
        };
        vfsx::walk_log(f.p, f.n, [&](const LogRecord &r) {
            if (r.n != c->dim) // tcp.rs:307-312
                fail(VFSX_ERR_INVALID_VECTOR, "Vector dimension mismatch. Expected " + std::to_string(c->dim) +
                                                  ", got " + std::to_string(r.n));
            const size_t off = rows.size();
            rows.resize(off + (size_t)c->dim * elem);
            if (c->dtype == VFSX_I8) {
                if (r.variant != 1) // VFSVector::dequantize on Dense (vector.rs:196-198) -> load_batch fails
                    fail(VFSX_ERR_INVALID_VECTOR, "No se puede descuantizar un vector que ya está descuantizado");
                if (r.scale != c->scale)
                    fail(VFSX_ERR_INVALID_VECTOR, "record scale_factor differs from the corpus scale");
                std::memcpy(rows.data() + off, r.values, c->dim);
            } else if (r.variant == 0) {
                std::memcpy(rows.data() + off, r.values, (size_t)c->dim * 4);
            } else { // Quantized record into a dense corpus: as_f32_vec (vector.rs:284-288)
                float *dst = (float *)(rows.data() + off);
                for (uint32_t j = 0; j < c->dim; ++j)
                    dst[j] = (float)(int8_t)r.values[j] / r.scale;
            }
            ids.push_back(r.id);
            metas.push_back(r.meta);
            if (ids.size() >= batch_rows)
                flush();
        }, first_record, max_records);
        flush();
        if (n_loaded)
            *n_loaded = total;
    });
}

int vfsx_corpus_get_metadata_json(vfsx_corpus *c, uint64_t row, char *buf, uint64_t cap, uint64_t *needed)
{
    return guarded([&] {
        ready(c);
        if (row >= c->n_rows)
            fail(VFSX_ERR_INVALID_VECTOR, "row index out of range");
        std::string s = "{\"name\":";
        static const vfsx_corpus::RowMeta empty;
        const vfsx_corpus::RowMeta &m = row < c->meta.size() ? c->meta[row] : empty;
        json_escape(s, m.name);
        s += ",\"tags\":[";
        for (size_t i = 0; i < m.tags.size(); ++i) {
            if (i)
                s += ',';
            json_escape(s, m.tags[i]);
        }
        s += "],\"created_at\":";
        json_escape(s, m.created_at);
        s += "}";
        if (needed)
            *needed = s.size() + 1;
        if (buf && cap) {
            const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
            std::memcpy(buf, s.data(), n);
            buf[n] = 0;
        }
    });
}

} // extern "C"
