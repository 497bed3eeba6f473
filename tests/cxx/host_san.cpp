// host_san.cpp -- TEST INFRASTRUCTURE: the library's HOST code (vfsx_api.cpp, vfsx_comm.cpp, vfsx_host.cpp and the
// launchers' host halves) under AddressSanitizer + UndefinedBehaviorSanitizer while it drives a real GPU.
// The device code is not instrumented (GPU sanitizers do not run on this pool); the library's sources are compiled
// into this one executable by `make hostsan` (vfs_amd/csrc/Makefile) with the sanitizers on the host side only.
// Exercises through the C ABI what the Python suite exercises, with an eye on the host-side bookkeeping that moves
// memory: staged one-row inserts and their commits, growth across storage segments, id bookkeeping, row read-back,
// the log parser on the reference's own records, the batch path's workspaces, the long-result path, the entry point
// that allocates nothing, a world-1 communicator.  Self-checking (exact path vs batch path, read-back vs input);
// exit status 0 = every check passed and no sanitizer report (the sanitizers abort the process on a finding).
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vfsx.h"

static int failures = 0, checks = 0;
#define CHECK(cond, ...)                                                                                           \
    do {                                                                                                           \
        ++checks;                                                                                                  \
        if (!(cond)) {                                                                                             \
            ++failures;                                                                                            \
            std::printf("FAIL %s:%d  ", __FILE__, __LINE__);                                                       \
            std::printf(__VA_ARGS__);                                                                              \
            std::printf("  [last error: %s]\n", vfsx_last_error());                                                \
        }                                                                                                          \
    } while (0)
#define OK(call) CHECK((call) == VFSX_OK, "%s", #call)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) & 0xffffff) / 8388608.0f - 1.0f; // [-1, 1)
}

static std::vector<float> random_rows(uint64_t n, uint32_t d)
{
    std::vector<float> v(n * d);
    for (float &x : v)
        x = frand();
    return v;
}

struct Hits {
    std::vector<uint64_t> ids;
    std::vector<float> dist;
    std::vector<uint32_t> count;
};

static Hits search(vfsx_corpus *c, const std::vector<float> &q, uint32_t nq, uint32_t k, int metric)
{
    Hits h;
    h.ids.assign((size_t)nq * k, 0);
    h.dist.assign((size_t)nq * k, 0.0f);
    h.count.assign(nq, 0);
    OK(vfsx_search(c, q.data(), nq, k, metric, h.ids.data(), h.dist.data(), h.count.data()));
    return h;
}

static bool same(const Hits &a, const Hits &b)
{
    return a.ids == b.ids && a.count == b.count &&
           std::memcmp(a.dist.data(), b.dist.data(), a.dist.size() * sizeof(float)) == 0;
}

static void staged_inserts()
{
    const uint32_t d = 100;
    vfsx_corpus *c = vfsx_corpus_create(d, VFSX_F32, 0.0f, 0, 0);
    CHECK(c != nullptr, "create");
    if (!c)
        return;
    OK(vfsx_corpus_set_insert_batch(c, 64));
    const std::vector<float> rows = random_rows(1000, d);
    for (uint64_t i = 0; i < 1000; ++i)
        OK(vfsx_corpus_append(c, rows.data() + i * d, nullptr, 1));
    CHECK(vfsx_corpus_rows(c) == 1000, "rows %llu", (unsigned long long)vfsx_corpus_rows(c));
    // a row still staged is found by the next search (the search commits)
    std::vector<float> q(rows.end() - d, rows.end());
    Hits h = search(c, q, 1, 5, VFSX_EUCLIDEAN);
    CHECK(h.count[0] == 5 && h.ids[0] == 1000 && h.dist[0] == 0.0f, "last staged row not first: id %llu dist %g",
          (unsigned long long)h.ids[0], h.dist[0]);
    // explicit ids after implicit ones, then implicit again (mode switches commit what is staged)
    const uint64_t ids[3] = {5000, 5001, 7};
    OK(vfsx_corpus_append(c, rows.data(), ids, 3));
    OK(vfsx_corpus_append(c, rows.data() + 3 * d, nullptr, 2)); // continue at max(id) + 1 = 5002
    uint64_t want_rows[5] = {1000, 1001, 1002, 1003, 1004}, got_ids[5] = {0};
    OK(vfsx_corpus_get_ids(c, want_rows, 5, got_ids));
    CHECK(got_ids[0] == 5000 && got_ids[2] == 7 && got_ids[3] == 5002 && got_ids[4] == 5003, "ids %llu %llu %llu %llu",
          (unsigned long long)got_ids[0], (unsigned long long)got_ids[2], (unsigned long long)got_ids[3],
          (unsigned long long)got_ids[4]);
    std::vector<float> back(5 * d);
    OK(vfsx_corpus_get_rows_f32(c, want_rows, 5, back.data()));
    CHECK(std::memcmp(back.data(), rows.data(), 5 * d * sizeof(float)) == 0, "read-back differs");
    OK(vfsx_corpus_commit(c));
    OK(vfsx_corpus_set_insert_batch(c, 0));
    // bad calls leave an error and no damage
    CHECK(vfsx_corpus_append(c, nullptr, nullptr, 1) == VFSX_ERR_INVALID_VECTOR, "null rows accepted");
    uint64_t bad_row = 1u << 30;
    CHECK(vfsx_corpus_get_rows_f32(c, &bad_row, 1, back.data()) == VFSX_ERR_INVALID_VECTOR, "row out of range accepted");
    CHECK(vfsx_search(c, q.data(), 1, 0, VFSX_COSINE, h.ids.data(), h.dist.data(), h.count.data()) != VFSX_OK, "k = 0 accepted");
    CHECK(vfsx_search(c, q.data(), 1, 5, 17, h.ids.data(), h.dist.data(), h.count.data()) != VFSX_OK, "metric 17 accepted");
    vfsx_corpus_destroy(c);
}

static void growth_and_batches()
{
    // 260 000 x 100 f32 = 104 MB: more than one 64 MiB segment, appended in uneven pieces
    const uint32_t d = 100;
    vfsx_corpus *c = vfsx_corpus_create(d, VFSX_F32, 0.0f, 0, 0);
    CHECK(c != nullptr, "create");
    if (!c)
        return;
    const std::vector<float> rows = random_rows(260000, d);
    uint64_t done = 0;
    for (uint64_t piece : {1ull, 63ull, 4096ull, 100000ull, 155840ull}) {
        OK(vfsx_corpus_append(c, rows.data() + done * d, nullptr, piece));
        done += piece;
    }
    CHECK(done == 260000 && vfsx_corpus_rows(c) == done, "rows");
    uint64_t cap = 0, seg = 0, copied = 0, bytes = 0;
    OK(vfsx_corpus_storage_info(c, &cap, &seg, &copied, &bytes));
    std::printf("storage: capacity %llu rows, %llu segments, %llu bytes copied, %llu bytes mapped\n",
                (unsigned long long)cap, (unsigned long long)seg, (unsigned long long)copied, (unsigned long long)bytes);
    CHECK(cap >= done && bytes >= done * d * 4, "storage info");
    // exact path: single queries, a multi-query pass, a long result (radix select), both metrics
    const std::vector<float> q = random_rows(300, d);
    const Hits e1 = search(c, q, 1, 10, VFSX_COSINE);
    const Hits e8 = search(c, q, 8, 10, VFSX_EUCLIDEAN);
    const Hits elong = search(c, q, 2, 300, VFSX_COSINE);
    CHECK(e1.count[0] == 10 && e8.count[7] == 10 && elong.count[1] == 300, "hit counts");
    for (uint32_t i = 1; i < 300; ++i)
        if (!(elong.dist[i - 1] <= elong.dist[i])) {
            CHECK(false, "long result not ascending at %u", i);
            break;
        }
    const Hits e300c = search(c, q, 300, 10, VFSX_COSINE), e64e = search(c, q, 64, 10, VFSX_EUCLIDEAN);
    // the batch path must return the same bytes: 64 queries (small-batch prefilter) and 300 (256-query tiles)
    OK(vfsx_corpus_enable_mfma(c, 1));
    const Hits m300c = search(c, q, 300, 10, VFSX_COSINE), m64e = search(c, q, 64, 10, VFSX_EUCLIDEAN);
    CHECK(same(e300c, m300c), "batch path (300 q, cosine) differs from the exact scan");
    CHECK(same(e64e, m64e), "batch path (64 q, euclidean) differs from the exact scan");
    // appended rows extend the operand; one of them must win its own query
    OK(vfsx_corpus_append(c, q.data(), nullptr, 4));
    const Hits m4 = search(c, q, 4, 3, VFSX_COSINE);
    CHECK(m4.ids[0] == 260001 && m4.ids[3] == 260002, "appended rows not found through the batch path: %llu %llu",
          (unsigned long long)m4.ids[0], (unsigned long long)m4.ids[3]);
    uint64_t mq = 0, mf = 0;
    OK(vfsx_corpus_mfma_stats(c, &mq, &mf, 0));
    int kind = -1;
    float resid = -1.0f;
    OK(vfsx_corpus_mfma_mirror(c, &kind, &resid));
    std::printf("batch path: %llu queries, %llu re-run by the exact scan, mirror kind %d residual %g\n",
                (unsigned long long)mq, (unsigned long long)mf, kind, resid);
    CHECK(mq >= 364, "batch path not taken");
    OK(vfsx_corpus_enable_mfma(c, 0));

    // device-buffer entry points: the allocating one and the one that allocates nothing must agree byte for byte
    float *d_q = nullptr;
    vfsx_hit *d_a = nullptr, *d_b = nullptr;
    CHECK(hipMalloc((void **)&d_q, 5 * d * sizeof(float)) == hipSuccess, "hipMalloc");
    CHECK(hipMalloc((void **)&d_a, 5 * 10 * sizeof(vfsx_hit)) == hipSuccess, "hipMalloc");
    CHECK(hipMalloc((void **)&d_b, 5 * 10 * sizeof(vfsx_hit)) == hipSuccess, "hipMalloc");
    CHECK(hipMemcpy(d_q, q.data(), 5 * d * sizeof(float), hipMemcpyHostToDevice) == hipSuccess, "H2D");
    OK(vfsx_search_device(c, d_q, 5, 10, VFSX_EUCLIDEAN, d_a, nullptr));
    OK(vfsx_search_device_noalloc(c, d_q, 5, 10, VFSX_EUCLIDEAN, d_b, nullptr));
    CHECK(hipDeviceSynchronize() == hipSuccess, "sync");
    std::vector<vfsx_hit> ha(50), hb(50);
    CHECK(hipMemcpy(ha.data(), d_a, 50 * sizeof(vfsx_hit), hipMemcpyDeviceToHost) == hipSuccess, "D2H");
    CHECK(hipMemcpy(hb.data(), d_b, 50 * sizeof(vfsx_hit), hipMemcpyDeviceToHost) == hipSuccess, "D2H");
    CHECK(std::memcmp(ha.data(), hb.data(), 50 * sizeof(vfsx_hit)) == 0, "noalloc differs");

    // world-1 communicator (RCCL bound at run time): the sharded entry points reduce to the local ranking
    if (vfsx_comm_rccl_version() > 0) {
        unsigned char id[VFSX_COMM_ID_BYTES];
        OK(vfsx_comm_unique_id(id));
        vfsx_comm *cm = vfsx_comm_init(0, 1, id, 0);
        CHECK(cm != nullptr, "comm init");
        if (cm) {
            OK(vfsx_comm_reserve(cm, 8 * 10, 8 * d));
            const uint64_t allocs = vfsx_comm_alloc_count(cm);
            Hits s;
            s.ids.assign(80, 0), s.dist.assign(80, 0.0f), s.count.assign(8, 0);
            OK(vfsx_search_sharded_host(c, cm, q.data(), 8, 10, VFSX_EUCLIDEAN, s.ids.data(), s.dist.data(), s.count.data()));
            Hits e8b = search(c, q, 8, 10, VFSX_EUCLIDEAN);
            CHECK(same(s, e8b), "sharded (world 1) differs from the local ranking");
            OK(vfsx_search_sharded(c, cm, d_q, 5, 10, VFSX_EUCLIDEAN, d_b, nullptr));
            CHECK(hipDeviceSynchronize() == hipSuccess, "sync");
            CHECK(vfsx_comm_alloc_count(cm) == allocs, "the communicator allocated after set-up");
            CHECK(vfsx_comm_rank(cm) == 0 && vfsx_comm_world(cm) == 1, "rank / world");
            vfsx_comm_destroy(cm);
        }
    } else {
        std::printf("RCCL not found: communicator part skipped (%s)\n", vfsx_last_error());
    }
    (void)hipFree(d_q);
    (void)hipFree(d_a);
    (void)hipFree(d_b);
    vfsx_corpus_destroy(c);
}

static void quantized_store()
{
    const uint32_t d = 72;
    vfsx_corpus *c = vfsx_corpus_create(d, VFSX_I8, 127.0f, 0, 0);
    CHECK(c != nullptr, "create");
    if (!c)
        return;
    OK(vfsx_corpus_set_insert_batch(c, 32));
    const std::vector<float> rows = random_rows(70000, d);
    OK(vfsx_corpus_append_quantize(c, rows.data(), nullptr, 69990)); // one big append ...
    for (uint64_t i = 69990; i < 70000; ++i)                         // ... then one row at a time (staged)
        OK(vfsx_corpus_append_quantize(c, rows.data() + i * d, nullptr, 1));
    std::vector<int8_t> want(70000ull * d), got(3 * d);
    OK(vfsx_quantize_rows_host(rows.data(), 70000ull * d, 127.0f, want.data()));
    const uint64_t probe[3] = {0, 69989, 69999};
    OK(vfsx_corpus_get_rows_raw(c, probe, 3, got.data()));
    for (int i = 0; i < 3; ++i)
        CHECK(std::memcmp(got.data() + i * d, want.data() + probe[i] * d, d) == 0,
              "stored bytes of row %llu differ from the host quantizer's", (unsigned long long)probe[i]);
    const std::vector<float> q = random_rows(130, d);
    const Hits e = search(c, q, 130, 10, VFSX_COSINE), ei = search(c, q, 2, 10, VFSX_EUCLIDEAN_I8DOMAIN);
    CHECK(ei.count[1] == 10, "integer-domain mode");
    OK(vfsx_corpus_enable_mfma(c, 1));
    const Hits m = search(c, q, 130, 10, VFSX_COSINE);
    CHECK(same(e, m), "int8 batch path differs from the exact scan");
    vfsx_corpus_destroy(c);
}

static void log_file(const char *path)
{
    vfsx_log_info info;
    std::memset(&info, 0, sizeof(info));
    OK(vfsx_log_inspect(path, &info));
    std::printf("log: %llu records, dim %u, dtype %d, ids %llu..%llu\n", (unsigned long long)info.records, info.dim,
                info.dtype, (unsigned long long)info.first_id, (unsigned long long)info.last_id);
    CHECK(info.records == 8 && info.dim > 0, "the reference's first 8 records");
    vfsx_corpus *c = vfsx_corpus_create(info.dim, info.dtype, info.scale, 0, 0);
    CHECK(c != nullptr, "create");
    if (!c)
        return;
    uint64_t n = 0;
    OK(vfsx_corpus_load_log(c, path, &n));
    CHECK(n == info.records, "loaded %llu", (unsigned long long)n);
    uint64_t needed = 0;
    char small[4];
    (void)vfsx_corpus_get_metadata_json(c, 0, small, sizeof(small), &needed); // too small: reports what it needs
    CHECK(needed > sizeof(small), "metadata size");
    std::string meta(needed + 1, '\0');
    OK(vfsx_corpus_get_metadata_json(c, 0, &meta[0], meta.size(), &needed));
    CHECK(meta.find("name") != std::string::npos, "metadata json: %s", meta.c_str());
    uint64_t n2 = 0;
    OK(vfsx_corpus_load_log_range(c, path, 6, 100, &n2)); // a shard's range past the end: the 2 that exist
    CHECK(n2 == 2 && vfsx_corpus_rows(c) == 10, "range load %llu", (unsigned long long)n2);
    CHECK(vfsx_log_inspect("/nonexistent/vectors.dat", &info) == VFSX_ERR_IO, "missing file");
    vfsx_corpus_destroy(c);
}

int main(int argc, char **argv)
{
    if (vfsx_init(1) != VFSX_OK) {
        std::printf("no usable GPU: %s\n", vfsx_last_error());
        return 2;
    }
    std::printf("%s\n", vfsx_version());
    staged_inserts();
    growth_and_batches();
    quantized_store();
    if (argc > 1)
        log_file(argv[1]);
    std::printf("%d checks, %d failures\n", checks, failures);
    std::fflush(stdout);
    // no exit handlers: ROCm's sanitizer runtime unloads its device-allocator hooks before the HIP runtime's own
    // finalizer frees its memory, and that order trips a CHECK inside the sanitizer (not a finding about this code)
    _exit(failures ? 1 : 0);
}
