// Latency of the host-buffer entry point (vfsx_search: the shape of the reference's Ranker::search, host vectors in,
// (id, distance) pairs out) against the device-buffer one, single queries back to back.  Build:
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probe/host_latency.cpp -o tools/probe/host_latency \
//       -Lvfs_amd/lib -lvfsx -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../../vfs_amd/lib' -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vfsx.h"

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void run(uint64_t n, uint32_t d, int dtype, int metric, int iters)
{
    vfsx_corpus *c = vfsx_corpus_create(d, dtype, 127.0f, n, 0);
    if (!c || vfsx_corpus_fill_synthetic(c, 1, 0, n, 1) != VFSX_OK) {
        std::printf("setup failed: %s\n", vfsx_last_error());
        return;
    }
    std::vector<float> q((size_t)64 * d);
    for (size_t i = 0; i < q.size(); ++i)
        q[i] = (float)((i * 2654435761u) % 2000) / 1000.0f - 1.0f;
    const uint32_t k = 10;
    uint64_t ids[10];
    float dist[10];
    uint32_t cnt = 0;
    for (int i = 0; i < 50; ++i)
        vfsx_search(c, q.data() + (size_t)(i % 64) * d, 1, k, metric, ids, dist, &cnt);
    std::vector<double> t(iters);
    for (int i = 0; i < iters; ++i) {
        const double t0 = now_us();
        vfsx_search(c, q.data() + (size_t)(i % 64) * d, 1, k, metric, ids, dist, &cnt);
        t[i] = now_us() - t0;
    }
    std::sort(t.begin(), t.end());
    const double host_med = t[iters / 2], host_p99 = t[iters * 99 / 100];
    // device buffers, one sync per query
    float *d_q = nullptr;
    vfsx_hit *d_h = nullptr;
    (void)hipMalloc((void **)&d_q, q.size() * sizeof(float));
    (void)hipMalloc((void **)&d_h, k * sizeof(vfsx_hit));
    (void)hipMemcpy(d_q, q.data(), q.size() * sizeof(float), hipMemcpyHostToDevice);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    for (int i = 0; i < 50; ++i)
        vfsx_search_device(c, d_q + (size_t)(i % 64) * d, 1, k, metric, d_h, s);
    (void)hipStreamSynchronize(s);
    for (int i = 0; i < iters; ++i) {
        const double t0 = now_us();
        vfsx_search_device(c, d_q + (size_t)(i % 64) * d, 1, k, metric, d_h, s);
        (void)hipStreamSynchronize(s);
        t[i] = now_us() - t0;
    }
    std::sort(t.begin(), t.end());
    // device buffers, launches pipelined (one synchronisation at the end): what bench.py's `value` measures, without
    // the Python / ctypes overhead per call
    (void)hipStreamSynchronize(s);
    const double p0 = now_us();
    for (int i = 0; i < iters; ++i)
        vfsx_search_device(c, d_q + (size_t)(i % 64) * d, 1, k, metric, d_h, s);
    const double p_issue = now_us() - p0;
    (void)hipStreamSynchronize(s);
    const double p_all = now_us() - p0;
    std::printf("          pipelined: %.2f us per query (%.0f q/s), host side %.2f us per call\n", p_all / iters,
                iters / p_all * 1e6, p_issue / iters);
    std::printf("%9llu x %4u %s metric %d: host buffers median %.1f us (p99 %.1f), device buffers + sync median %.1f us (p99 %.1f); "
                "first hit id %llu\n",
                (unsigned long long)n, d, dtype ? "int8" : "f32", metric, host_med, host_p99, t[iters / 2],
                t[iters * 99 / 100], (unsigned long long)ids[0]);
    (void)hipFree(d_q);
    (void)hipFree(d_h);
    (void)hipStreamDestroy(s);
    vfsx_corpus_destroy(c);
}

int main()
{
    if (vfsx_init(1) != VFSX_OK) {
        std::printf("%s\n", vfsx_last_error());
        return 2;
    }
    run(100000, 128, VFSX_F32, VFSX_SIMD_EUCLIDEAN, 4000); // C1
    run(100000, 64, VFSX_F32, VFSX_SIMD_EUCLIDEAN, 4000);
    run(1000000, 768, VFSX_F32, VFSX_COSINE, 1500);        // C2
    run(10000000, 128, VFSX_I8, VFSX_EUCLIDEAN, 1500);     // C3
    return 0;
}
