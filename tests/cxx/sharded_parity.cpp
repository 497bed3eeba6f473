// sharded_parity.cpp -- TEST INFRASTRUCTURE: a C++ host runs SURVEY.md 8(e) end to end through the C ABI, with
// no Python and no torch.distributed anywhere, against the CPU oracle (linked directly):
//   (1) two shards held on one GPU: vfsx_search_device per shard -> [shard][query][k] lists ->
//       vfsx_merge_hits_device == the oracle's ranking of the whole corpus (ties across the cut included);
//   (2) a world-1 RCCL communicator (vfsx_comm_*): vfsx_search_sharded (device buffers), Ranker::search_sharded
//       (host buffers) and vfsx_comm_merge_hits run the real ncclAllGather + merge and equal the oracle.
// A box has one GPU, so the communicator has one rank; ranks > 1 differ only in the all-gather's peer count.
// Run on the GPU box by tests/test_gpu_cxx.py.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "vfsx_ranker.hpp"

extern "C" {
void vfso_fill_synthetic_f32(uint64_t seed, uint64_t row0, uint64_t n, uint32_t d, float *out);
uint64_t vfso_topk_scan(const float *rows_f32, const int8_t *rows_i8, float scale, const uint64_t *ids, uint64_t n_rows,
                        uint32_t d, const float *query, int metric, uint64_t limit, uint64_t *out_ids, float *out_dist,
                        uint64_t *out_pos);
}

static int failures = 0, cases = 0;
#define HIP_OK(x)                                                                                                     \
    do {                                                                                                              \
        hipError_t e_ = (x);                                                                                          \
        if (e_ != hipSuccess) {                                                                                       \
            std::printf("HIP failure %s: %s\n", #x, hipGetErrorString(e_));                                           \
            return 11;                                                                                                \
        }                                                                                                             \
    } while (0)

static void compare_hits(const char *what, const vfsx_hit *got, const uint64_t *oi, const float *od, uint64_t m, uint32_t k)
{
    ++cases;
    bool ok = true;
    for (uint32_t j = 0; ok && j < k; ++j) {
        if (j < m)
            ok = (got[j].flags & 1u) && got[j].id == oi[j] && std::memcmp(&got[j].dist, &od[j], 4) == 0;
        else
            ok = !(got[j].flags & 1u);
    }
    if (!ok) {
        ++failures;
        std::printf("FAIL %s\n", what);
    }
}

int main()
{
    try {
        const uint32_t d = 96, nq = 4, k = 10;
        const uint64_t n = 30000, cut = 17011;
        std::vector<float> rows(n * d), queries(nq * d);
        vfso_fill_synthetic_f32(0x5EED0001, 0, n, d, rows.data());
        vfso_fill_synthetic_f32(0x5EED0002, 0, nq, d, queries.data());
        // the same vector on both sides of the cut: the merge must keep scan order among equal distances
        std::memcpy(&rows[(cut + 5) * d], &rows[123 * d], d * sizeof(float));
        std::memcpy(&queries[0], &rows[123 * d], d * sizeof(float));

        float *d_q = nullptr;
        vfsx_hit *d_lists = nullptr, *d_out = nullptr;
        HIP_OK(hipMalloc((void **)&d_q, nq * d * sizeof(float)));
        HIP_OK(hipMalloc((void **)&d_lists, 2 * nq * k * sizeof(vfsx_hit)));
        HIP_OK(hipMalloc((void **)&d_out, nq * k * sizeof(vfsx_hit)));
        HIP_OK(hipMemcpy(d_q, queries.data(), nq * d * sizeof(float), hipMemcpyHostToDevice));
        std::vector<vfsx_hit> out(nq * k);
        std::vector<uint64_t> oi(k);
        std::vector<float> od(k);

        vfsx::Corpus whole(d), lo(d), hi(d);
        whole.register_vectors(rows.data(), n);
        lo.register_vectors(rows.data(), cut);
        std::vector<uint64_t> hi_ids(n - cut);
        for (uint64_t i = 0; i < n - cut; ++i)
            hi_ids[i] = cut + i + 1; // ids 1..N in scan order (storage_manager.rs:76)
        hi.register_vectors(rows.data() + cut * d, n - cut, hi_ids.data());
        hi.set_row_base(cut);

        for (int metric : {0, 1}) {
            // (1) two shards, one GPU
            vfsx::check(vfsx_search_device(lo.handle(), d_q, nq, k, metric, d_lists, nullptr));
            vfsx::check(vfsx_search_device(hi.handle(), d_q, nq, k, metric, d_lists + nq * k, nullptr));
            vfsx::check(vfsx_merge_hits_device(d_lists, 2, nq, k, d_out, 0, nullptr));
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipMemcpy(out.data(), d_out, out.size() * sizeof(vfsx_hit), hipMemcpyDeviceToHost));
            for (uint32_t q = 0; q < nq; ++q) {
                const uint64_t m = vfso_topk_scan(rows.data(), nullptr, 127.0f, nullptr, n, d, &queries[q * d], metric, k,
                                                  oi.data(), od.data(), nullptr);
                compare_hits("two shards on one GPU", &out[q * k], oi.data(), od.data(), m, k);
            }
        }

        // (2) the RCCL communicator, one rank
        const vfsx::Comm::Id id = vfsx::Comm::unique_id();
        vfsx::Comm comm(0, 1, id, 0);
        if (comm.rank() != 0 || comm.world() != 1 || vfsx_comm_rccl_version() <= 0) {
            std::printf("FAIL communicator shape\n");
            ++failures;
        }
        for (int metric : {0, 1}) {
            HIP_OK(hipMemset(d_out, 0, nq * k * sizeof(vfsx_hit)));
            vfsx::check(vfsx_search_sharded(whole.handle(), comm.handle(), d_q, nq, k, metric, d_out, nullptr));
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipMemcpy(out.data(), d_out, out.size() * sizeof(vfsx_hit), hipMemcpyDeviceToHost));
            for (uint32_t q = 0; q < nq; ++q) {
                const uint64_t m = vfso_topk_scan(rows.data(), nullptr, 127.0f, nullptr, n, d, &queries[q * d], metric, k,
                                                  oi.data(), od.data(), nullptr);
                compare_hits("vfsx_search_sharded", &out[q * k], oi.data(), od.data(), m, k);
                // Ranker-level call (host buffers)
                vfsx::Ranker ranker(vfsx::SearchType::Exact, (vfsx::DistanceMethod)metric);
                const std::vector<float> qv(queries.begin() + q * d, queries.begin() + (q + 1) * d);
                const auto got = ranker.search_sharded(qv, 5, k, whole, comm);
                ++cases;
                bool ok = got.size() == m;
                for (uint64_t j = 0; ok && j < m; ++j)
                    ok = got[j].first == oi[j] && std::memcmp(&got[j].second, &od[j], 4) == 0;
                if (!ok) {
                    ++failures;
                    std::printf("FAIL Ranker::search_sharded\n");
                }
            }
            // the exchange alone: this rank's lists in, the (one-rank) global ranking out
            vfsx::check(vfsx_search_device(whole.handle(), d_q, nq, k, metric, d_lists, nullptr));
            vfsx::check(vfsx_comm_merge_hits(comm.handle(), d_lists, nq, k, d_out, nullptr));
            HIP_OK(hipDeviceSynchronize());
            std::vector<vfsx_hit> a(nq * k);
            HIP_OK(hipMemcpy(a.data(), d_lists, a.size() * sizeof(vfsx_hit), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(out.data(), d_out, out.size() * sizeof(vfsx_hit), hipMemcpyDeviceToHost));
            ++cases;
            if (std::memcmp(a.data(), out.data(), a.size() * sizeof(vfsx_hit)) != 0) {
                ++failures;
                std::printf("FAIL vfsx_comm_merge_hits\n");
            }
        }
        (void)hipFree(d_q);
        (void)hipFree(d_lists);
        (void)hipFree(d_out);
        std::printf("%s %d cases, %d failures\n", failures ? "FAIL" : "ok", cases, failures);
        return failures ? 1 : 0;
    } catch (const vfsx::Error &e) {
        std::printf("vfsx error %d: %s\n", e.code, e.what());
        return e.code == VFSX_ERR_NO_DEVICE ? 77 : 10;
    }
}
