// Test infrastructure (GPU): the wave-wide primitives of vfs_amd/csrc/vfsx_device.h against a host model --
// the lane exchanges xor_lane32<1..32>, the 64-key bitonic sort, and the register top-k lists' bulk paths
// (offer_dense / merge_list) against their one-at-a-time inserts, for k = 1 .. 128, with duplicates of the
// sentinel, dense and sparse offers.  Prints "ok <cases>" or the first mismatch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "vfsx_device.h"

using namespace vfsx;

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);                         \
            return 2;                                                                                                  \
        }                                                                                                              \
    } while (0)

__global__ void xor_kernel(const uint32_t *in, uint32_t *out)
{
    const uint32_t v = in[threadIdx.x];
    out[0 * 64 + threadIdx.x] = xor_lane32<1>(v);
    out[1 * 64 + threadIdx.x] = xor_lane32<2>(v);
    out[2 * 64 + threadIdx.x] = xor_lane32<4>(v);
    out[3 * 64 + threadIdx.x] = xor_lane32<8>(v);
    out[4 * 64 + threadIdx.x] = xor_lane32<16>(v);
    out[5 * 64 + threadIdx.x] = xor_lane32<32>(v);
}

__global__ void sort_kernel(const uint64_t *in, uint64_t *out)
{
    out[blockIdx.x * 64 + threadIdx.x] = wave_sort_desc(in[blockIdx.x * 64 + threadIdx.x]);
}

// one wave per case: `n_vec` vectors of 64 keys go into a list of k; mode 0 = offer (inserts), 1 = offer_dense,
// 2 = the vectors are first folded into side lists of up to 4 vectors each (offer), published with spill() and
// merged with merge_list -- what the workgroup folds and the select do
template <typename TopK>
__global__ void topk_kernel(const uint64_t *in, uint64_t *out, uint32_t n_vec, uint32_t k, int mode)
{
    __shared__ uint64_t lds[128];
    const uint64_t *src = in + (size_t)blockIdx.x * n_vec * 64u;
    TopK top;
    top.init(k, nullptr);
    if (mode == 2) {
        for (uint32_t v0 = 0; v0 < n_vec; v0 += 4) {
            TopK side;
            side.init(k, nullptr);
            for (uint32_t v = v0; v < n_vec && v < v0 + 4; ++v)
                side.offer(src[v * 64u + threadIdx.x]);
            __syncthreads();
            side.spill(lds);
            __syncthreads();
            top.merge_list(lds);
        }
    } else {
        for (uint32_t v = 0; v < n_vec; ++v) {
            const uint64_t c = src[v * 64u + threadIdx.x];
            if (mode == 0)
                top.offer(c);
            else
                top.offer_dense(c);
        }
    }
    __syncthreads();
    top.spill(lds);
    __syncthreads();
    out[(size_t)blockIdx.x * 128u + threadIdx.x] = lds[threadIdx.x];
    if (topk_list_stride(k) > 64u)
        out[(size_t)blockIdx.x * 128u + 64u + threadIdx.x] = lds[64u + threadIdx.x];
}

int main()
{
    std::mt19937_64 rng(12345);
    // --- lane exchanges
    {
        std::vector<uint32_t> h(64), o(6 * 64);
        for (auto &x : h)
            x = (uint32_t)rng();
        uint32_t *d_in, *d_out;
        CK(hipMalloc(&d_in, 64 * 4));
        CK(hipMalloc(&d_out, 6 * 64 * 4));
        CK(hipMemcpy(d_in, h.data(), 64 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(xor_kernel, dim3(1), dim3(64), 0, 0, d_in, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), d_out, 6 * 64 * 4, hipMemcpyDeviceToHost));
        for (int s = 0; s < 6; ++s)
            for (int l = 0; l < 64; ++l)
                if (o[s * 64 + l] != h[l ^ (1 << s)]) {
                    std::printf("xor_lane32<%d> lane %d: got %08x want %08x\n", 1 << s, l, o[s * 64 + l], h[l ^ (1 << s)]);
                    return 1;
                }
        (void)hipFree(d_in);
        (void)hipFree(d_out);
    }
    size_t cases = 6;
    // --- the sort: random keys, few distinct keys, sentinels
    {
        const int n = 256;
        std::vector<uint64_t> h(n * 64), o(n * 64);
        for (int b = 0; b < n; ++b)
            for (int l = 0; l < 64; ++l) {
                uint64_t x = rng();
                if (b % 4 == 1)
                    x &= 7u; // many duplicates
                if (b % 4 == 2 && (rng() & 3u) == 0)
                    x = kSentinel;
                if (b % 4 == 3)
                    x = (x << 32) | (uint32_t)l; // composite-like
                h[b * 64 + l] = x;
            }
        uint64_t *d_in, *d_out;
        CK(hipMalloc(&d_in, h.size() * 8));
        CK(hipMalloc(&d_out, h.size() * 8));
        CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(sort_kernel, dim3(n), dim3(64), 0, 0, d_in, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
        for (int b = 0; b < n; ++b) {
            std::vector<uint64_t> w(h.begin() + b * 64, h.begin() + b * 64 + 64);
            std::sort(w.begin(), w.end(), std::greater<uint64_t>());
            for (int l = 0; l < 64; ++l)
                if (o[b * 64 + l] != w[l]) {
                    std::printf("wave_sort_desc block %d lane %d: got %016llx want %016llx\n", b, l,
                                (unsigned long long)o[b * 64 + l], (unsigned long long)w[l]);
                    return 1;
                }
        }
        cases += n;
        (void)hipFree(d_in);
        (void)hipFree(d_out);
    }
    // --- the lists: host model = the k smallest of everything offered (keys unique except the sentinel)
    for (int two = 0; two < 2; ++two) {
        const uint32_t ks1[] = {1, 2, 7, 10, 31, 32, 33, 63, 64};
        const uint32_t ks2[] = {1, 10, 64, 65, 66, 100, 127, 128};
        const uint32_t *ks = two ? ks2 : ks1;
        const size_t nk = two ? sizeof(ks2) / 4 : sizeof(ks1) / 4;
        for (size_t ki = 0; ki < nk; ++ki)
            for (uint32_t n_vec : {1u, 2u, 3u, 9u, 40u})
                for (int mode = 0; mode < 3; ++mode) {
                    const uint32_t k = ks[ki];
                    const int n = 24;
                    std::vector<uint64_t> h((size_t)n * n_vec * 64), o((size_t)n * 128);
                    for (int b = 0; b < n; ++b)
                        for (uint32_t i = 0; i < n_vec * 64; ++i) {
                            // unique composite keys; case-dependent share of sentinels and of a low-entropy key part
                            uint64_t key = (b % 3 == 1) ? (rng() & 15u) : (rng() >> 40);
                            uint64_t x = (key << 32) | i;
                            if (b % 3 == 2 && (rng() % 3u) != 0)
                                x = kSentinel;
                            if (b == 5)
                                x = ((uint64_t)(n_vec * 64 - i) << 32) | i; // strictly decreasing: every key is a new best
                            h[((size_t)b * n_vec * 64) + i] = x;
                        }
                    uint64_t *d_in, *d_out;
                    CK(hipMalloc(&d_in, h.size() * 8));
                    CK(hipMalloc(&d_out, o.size() * 8));
                    CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
                    CK(hipMemset(d_out, 0, o.size() * 8));
                    if (two)
                        hipLaunchKernelGGL((topk_kernel<WaveTopKReg2>), dim3(n), dim3(64), 0, 0, d_in, d_out, n_vec, k, mode);
                    else
                        hipLaunchKernelGGL((topk_kernel<WaveTopKReg>), dim3(n), dim3(64), 0, 0, d_in, d_out, n_vec, k, mode);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost));
                    for (int b = 0; b < n; ++b) {
                        std::vector<uint64_t> w(h.begin() + (size_t)b * n_vec * 64, h.begin() + (size_t)(b + 1) * n_vec * 64);
                        std::sort(w.begin(), w.end());
                        const uint32_t stride = two && k > 64 ? 128u : 64u;
                        for (uint32_t i = 0; i < stride; ++i) {
                            const uint64_t want = (i < k && i < w.size()) ? w[i] : kSentinel;
                            if (o[(size_t)b * 128 + i] != want) {
                                std::printf("%s k=%u n_vec=%u mode=%d case %d entry %u: got %016llx want %016llx\n",
                                            two ? "WaveTopKReg2" : "WaveTopKReg", k, n_vec, mode, b, i,
                                            (unsigned long long)o[(size_t)b * 128 + i], (unsigned long long)want);
                                return 1;
                            }
                        }
                    }
                    cases += n;
                    (void)hipFree(d_in);
                    (void)hipFree(d_out);
                }
    }
    std::printf("ok %zu\n", cases);
    return 0;
}
