// ranker_parity.cpp -- TEST INFRASTRUCTURE: parity of the C++ host mirror (vfs_amd/cxx/vfsx_ranker.hpp,
// the stand-in for the reference's Rust `Ranker`, rank.rs:61-105) against the CPU oracle
// (oracle/vfs_oracle.c), linked directly.  Run on the GPU box by tests/test_gpu_cxx.py.
//
// Reads like the reference's own flow: build a store, register vectors, `Ranker::new(...)`,
// `ranker.search(&query, batch, Some(limit), &mut manager)`, compare (id, distance) pairs.
// Bar: ids identical, distances bit-exact (f32 and int8 stores).
#include <cstdio>
#include <cstring>
#include <vector>

#include "vfsx_ranker.hpp"

extern "C" {
// oracle/vfs_oracle.c
void vfso_fill_synthetic_f32(uint64_t seed, uint64_t row0, uint64_t n, uint32_t d, float *out);
void vfso_quantize(const float *v, uint64_t n, int has_scale, float scale, int8_t *out);
uint64_t vfso_topk_scan(const float *rows_f32, const int8_t *rows_i8, float scale, const uint64_t *ids, uint64_t n_rows,
                        uint32_t d, const float *query, int metric, uint64_t limit, uint64_t *out_ids, float *out_dist,
                        uint64_t *out_pos);
uint64_t vfso_exact_search(const float *rows, const uint64_t *ids, uint64_t n_rows, uint32_t d, const float *query,
                           int metric, uint64_t batch, uint64_t limit, uint64_t *out_ids, float *out_dist);
}

static int failures = 0, cases = 0;

static void compare(const char *what, const std::vector<std::pair<uint64_t, float>> &got, const uint64_t *oi,
                    const float *od, uint64_t m)
{
    ++cases;
    bool ok = got.size() == m;
    for (uint64_t i = 0; ok && i < m; ++i)
        ok = got[i].first == oi[i] && std::memcmp(&got[i].second, &od[i], 4) == 0;
    if (!ok) {
        ++failures;
        std::printf("FAIL %s: %zu hits vs %llu\n", what, got.size(), (unsigned long long)m);
    }
}

int main()
{
    using vfsx::DistanceMethod;
    try {
        const DistanceMethod methods[] = {DistanceMethod::Euclidean, DistanceMethod::Cosine,
                                          DistanceMethod::SimdEuclidean, DistanceMethod::SimdCosine};
        for (uint32_t d : {16u, 100u, 768u}) {
            const uint64_t n = d == 768 ? 8000 : 20000;
            std::vector<float> rows(n * d), queries(3 * d);
            vfso_fill_synthetic_f32(0x5EED0001, 0, n, d, rows.data());
            vfso_fill_synthetic_f32(0x5EED0002, 0, 3, d, queries.data());
            // exact duplicates far apart: ties must come back in scan order (rank.rs:170, stable sort)
            std::memcpy(&rows[(n - 5) * d], &rows[17 * d], d * sizeof(float));
            std::vector<int8_t> qrows(n * d);
            vfso_quantize(rows.data(), n * d, 1, 127.0f, qrows.data());

            vfsx::Corpus dense(d), quant(d, /*quantize=*/true);
            dense.register_vectors(rows.data(), n);
            quant.register_vectors(rows.data(), n); // quantized on the device (vector.rs:94-108)

            for (int qi = 0; qi < 3; ++qi) {
                const std::vector<float> q(queries.begin() + qi * d, queries.begin() + (qi + 1) * d);
                for (DistanceMethod m : methods)
                    for (size_t limit : {1u, 5u, 10u, 100u}) {
                        std::vector<uint64_t> oi(limit);
                        std::vector<float> od(limit);
                        vfsx::Ranker ranker(vfsx::SearchType::Exact, m, limit);
                        uint64_t cnt = vfso_topk_scan(rows.data(), nullptr, 127.0f, nullptr, n, d, q.data(), (int)m,
                                                      limit, oi.data(), od.data(), nullptr);
                        compare("dense", ranker.search(q, 7, limit, dense), oi.data(), od.data(), cnt);
                        cnt = vfso_topk_scan(nullptr, qrows.data(), 127.0f, nullptr, n, d, q.data(), (int)m, limit,
                                             oi.data(), od.data(), nullptr);
                        compare("quantized", ranker.search(q, 1000, limit, quant), oi.data(), od.data(), cnt);
                    }
                // the labelled integer-domain extension on the quantized store
                std::vector<uint64_t> oi(10);
                std::vector<float> od(10);
                vfsx::Ranker fast(vfsx::SearchType::Exact, DistanceMethod::EuclideanI8Domain, 10);
                uint64_t cnt = vfso_topk_scan(nullptr, qrows.data(), 127.0f, nullptr, n, d, q.data(), 4, 10, oi.data(),
                                              od.data(), nullptr);
                compare("i8-domain", fast.search(q, 5, 10, quant), oi.data(), od.data(), cnt);
            }
        }
        // the literal per-batch loop of rank.rs:119-186 on a small store: any batch size, default limit 5
        {
            const uint32_t d = 8;
            const uint64_t n = 300;
            std::vector<float> rows(n * d), q(d);
            vfso_fill_synthetic_f32(7, 0, n, d, rows.data());
            vfso_fill_synthetic_f32(8, 0, 1, d, q.data());
            vfsx::Corpus store(d);
            store.register_vectors(rows.data(), n);
            vfsx::Ranker ranker(vfsx::SearchType::Exact, DistanceMethod::Cosine);
            for (size_t batch : {1u, 3u, 64u, 1000u}) {
                uint64_t oi[5];
                float od[5];
                uint64_t cnt = vfso_exact_search(rows.data(), nullptr, n, d, q.data(), 1, batch, 5, oi, od);
                compare("literal loop", ranker.search(q, batch, std::nullopt, store), oi, od, cnt);
            }
        }
        // long results (result_limit > 128: radix select) and the matrix-core batch path under every operand an f32
        // store can have, through the C++ host mirror
        {
            const uint32_t d = 48;
            const uint64_t n = 20000;
            std::vector<float> rows(n * d), queries(6 * d);
            vfso_fill_synthetic_f32(21, 0, n, d, rows.data());
            vfso_fill_synthetic_f32(22, 0, 6, d, queries.data());
            vfsx::Corpus store(d);
            store.register_vectors(rows.data(), n);
            const std::vector<float> q0(queries.begin(), queries.begin() + d);
            for (size_t limit : {129u, 1000u, 5000u}) {
                std::vector<uint64_t> oi(limit);
                std::vector<float> od(limit);
                vfsx::Ranker ranker(vfsx::SearchType::Exact, DistanceMethod::Euclidean, limit);
                const uint64_t cnt = vfso_topk_scan(rows.data(), nullptr, 127.0f, nullptr, n, d, q0.data(), 0, limit,
                                                    oi.data(), od.data(), nullptr);
                compare("long result", ranker.search(q0, 64, limit, store), oi.data(), od.data(), cnt);
            }
            for (int mirror : {1, 2, 3}) { // auto, bf16, int8
                store.enable_mfma(true, mirror);
                const auto kind = store.mfma_mirror();
                if (kind.first != (mirror == 2 ? 1 : 2)) {
                    std::printf("FAIL mfma_mirror: asked %d, got kind %d\n", mirror, kind.first);
                    ++failures;
                }
                std::vector<uint64_t> ids(6 * 10);
                std::vector<float> dist(6 * 10);
                std::vector<uint32_t> count(6);
                if (vfsx_search(store.handle(), queries.data(), 6, 10, VFSX_COSINE, ids.data(), dist.data(), count.data()) != VFSX_OK) {
                    std::printf("FAIL batch search: %s\n", vfsx_last_error());
                    ++failures;
                    continue;
                }
                for (int qi = 0; qi < 6; ++qi) {
                    uint64_t oi[10];
                    float od[10];
                    const uint64_t cnt = vfso_topk_scan(rows.data(), nullptr, 127.0f, nullptr, n, d, queries.data() + qi * d,
                                                        1, 10, oi, od, nullptr);
                    std::vector<std::pair<uint64_t, float>> got;
                    for (uint32_t j = 0; j < count[qi]; ++j)
                        got.emplace_back(ids[qi * 10 + j], dist[qi * 10 + j]);
                    compare("mfma batch", got, oi, od, cnt);
                }
            }
            store.enable_mfma(false);
        }
        std::printf("%s %d cases, %d failures\n", failures ? "FAIL" : "ok", cases, failures);
        return failures ? 1 : 0;
    } catch (const vfsx::Error &e) {
        std::printf("vfsx error %d: %s\n", e.code, e.what());
        return e.code == VFSX_ERR_NO_DEVICE ? 77 : 10;
    }
}
