// Self-test of the C++ Ranker mirror through the C ABI (run on the GPU box by
// tests/test_gpu_cxx.py).  Mirrors tests.sh:73-104: a 4-d quantizing store, the vector
// [1,2,3,4], an exact Euclidean search -> id 1 at sqrt(14) (it saturates to [1,1,1,1]).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "vfsx_ranker.hpp"

int main()
{
    try {
        vfsx::Corpus store(4, /*quantize=*/true);
        const float v[4] = {1.f, 2.f, 3.f, 4.f};
        store.register_vectors(v, 1);
        vfsx::Ranker ranker(vfsx::SearchType::Exact, vfsx::DistanceMethod::Euclidean, 6);
        auto hits = ranker.search({1.f, 2.f, 3.f, 4.f}, /*top_k as batch (tcp.rs:403)*/ 2, 6, store);
        const float want = std::sqrt(14.0f);
        if (hits.size() != 1 || hits[0].first != 1 || std::memcmp(&hits[0].second, &want, 4) != 0) {
            std::printf("FAIL: %zu hits, id %llu dist %.9g\n", hits.size(),
                        hits.empty() ? 0ull : (unsigned long long)hits[0].first, hits.empty() ? 0.0 : hits[0].second);
            return 1;
        }
        if (!ranker.search({1.f, 2.f, 3.f, 4.f}, 0, 6, store).empty())
            return 2;
        try {
            ranker.search({1.f, 2.f, 3.f}, 2, 6, store);
            return 3;
        } catch (const vfsx::Error &e) {
            if (e.code != VFSX_ERR_INVALID_VECTOR)
                return 4;
        }
        std::printf("ok id=%llu dist=%.9g\n", (unsigned long long)hits[0].first, hits[0].second);
        return 0;
    } catch (const vfsx::Error &e) {
        std::printf("vfsx error %d: %s\n", e.code, e.what());
        return e.code == VFSX_ERR_NO_DEVICE ? 77 : 10;
    }
}
