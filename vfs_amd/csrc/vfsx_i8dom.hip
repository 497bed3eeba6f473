// vfsx_i8dom.hip -- the labelled *integer-domain* Euclidean mode for int8 corpora
// (VFSX_EUCLIDEAN_I8DOMAIN; SURVEY.md 8a "A6 decision", definition (ii)).
//
// NOT what the reference's Ranker computes (that is definition (i): f32 query against
// dequantized rows, vfsx_kernels.hip).  Here the query is quantized with the
// reference's own rule -- VFSVector::as_i8_vec(Some(scale)) = Vector::quantize
// (vector.rs:293-302, 94-108) -- and
//       S    = sum_j (qi_j - x_j)^2          exactly, in integers
//       dist = sqrt(S as f32) / scale         two correctly rounded f32 operations
// S is order independent, so the row sum runs on v_dot4_i32_i8 (4 multiply-adds per
// lane-op): S = sum qi^2 + sum x^2 - 2 sum qi*x.  At ~0.5 lane-ops per element the
// kernel is HBM-bound again (the faithful int8 kernel is VALU-bound at ~5 ops/element).
// Ranking is the same composite (distance key, scan position) order as everywhere else;
// integer S produces many exact ties, which that order resolves as the reference's
// stable sort would (rank.rs:170-173).
#include <hip/hip_runtime.h>

#include "vfsx_device.h"
#include "vfsx_internal.h"

#pragma clang fp contract(off)

namespace vfsx {
namespace {

__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
}

// one workgroup per query: qw[qi][j/4] = packed quantized query (zero padded to the
// chunk), qw[qi][d_chunks*4] = sum qi^2
__global__ __launch_bounds__(256) void query_quantize_kernel(const float *__restrict__ q_in, uint32_t dim,
                                                             float scale, uint32_t *__restrict__ qw,
                                                             uint32_t n_words, uint32_t qw_stride)
{
    __shared__ int part[256];
    const uint32_t qi = blockIdx.x;
    int qq = 0;
    for (uint32_t w = threadIdx.x; w < n_words; w += 256) {
        uint32_t word = 0;
        for (uint32_t b = 0; b < 4; ++b) {
            const uint32_t j = w * 4 + b;
            int v = 0;
            if (j < dim) { // vector.rs:106 with Some(scale): (val * scale).round() as i8
                float r = roundf(f_mul(q_in[(size_t)qi * dim + j], scale));
                v = r != r ? 0 : (r >= 127.0f ? 127 : (r <= -128.0f ? -128 : (int)r));
            }
            qq += v * v;
            word |= ((uint32_t)v & 0xffu) << (8 * b);
        }
        qw[(size_t)qi * qw_stride + w] = word;
    }
    part[threadIdx.x] = qq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        qw[(size_t)qi * qw_stride + n_words] = (uint32_t)part[0];
}

// grid = (n_lists, nq); block = 4 waves; same work split and list output as scan_kernel.
template <typename TopK, int UNROLL, bool NT>
__global__ __launch_bounds__(kBlock) void scan_i8dom_kernel(const uint4 *__restrict__ corpus,
                                                            const uint32_t *__restrict__ qwords,
                                                            uint64_t *__restrict__ lists, uint32_t n_rows,
                                                            uint32_t d_chunks, uint32_t n_rowblocks,
                                                            uint32_t qw_stride, uint32_t k, uint32_t n_lists,
                                                            float scale)
{
    extern __shared__ __align__(16) uint64_t dyn_lds[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t list_stride = topk_list_stride(k);
    const uint32_t *__restrict__ qw = qwords + (size_t)blockIdx.y * qw_stride; // wave-uniform: s_load
    const int QQ = (int)qw[d_chunks * 4u];

    TopK top;
    top.init(k, dyn_lds + (size_t)wave * list_stride);
    const uint32_t n_waves = gridDim.x * kWavesPerBlock;

    for (uint32_t blk = blockIdx.x * kWavesPerBlock + wave; blk < n_rowblocks; blk += n_waves) {
        const uint4 *__restrict__ p = corpus + (size_t)blk * d_chunks * 64u + lane;
        int dot = 0, xx = 0;
        auto fold = [&](const uint4 v, uint32_t c) {
            const uint4 qv = *(const uint4 *)(qw + c * 4u);
            dot = dot4(v.x, qv.x, dot);
            dot = dot4(v.y, qv.y, dot);
            dot = dot4(v.z, qv.z, dot);
            dot = dot4(v.w, qv.w, dot);
            xx = dot4(v.x, v.x, xx);
            xx = dot4(v.y, v.y, xx);
            xx = dot4(v.z, v.z, xx);
            xx = dot4(v.w, v.w, xx);
        };
        uint32_t c = 0;
        for (; c + UNROLL <= d_chunks; c += UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                v[u] = load_chunk<NT>(p + (size_t)(c + u) * 64u);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                fold(v[u], c + u);
        }
        for (; c < d_chunks; ++c)
            fold(load_chunk<NT>(p + (size_t)c * 64u), c);

        const uint32_t row = blk * 64u + lane;
        uint64_t comp = kSentinel;
        if (row < n_rows) {
            const uint32_t S = (uint32_t)(QQ + xx - 2 * dot); // < 2^31 for dim <= 32768
            const float dist = f_div(f_sqrt((float)S), scale);
            comp = ((uint64_t)dist_key(dist) << 32) | row;
        }
        top.offer_dense(comp);
    }

    // fold the 4 waves' lists into wave 0's, one list per workgroup to HBM ([q][j][list])
    __syncthreads();
    uint64_t *mine = dyn_lds + (size_t)wave * list_stride;
    top.spill(mine);
    __syncthreads();
    for (uint32_t s = 1; s < (uint32_t)kWavesPerBlock; s <<= 1) { // merge tree (as in scan_kernel)
        if ((wave & (2u * s - 1u)) == 0u) {
            top.merge_list(dyn_lds + (size_t)(wave + s) * list_stride);
            top.spill(mine);
        }
        __syncthreads();
    }
    if (wave == 0)
        top.store(lists + (size_t)blockIdx.y * n_lists * k + blockIdx.x, n_lists);
}

template <typename TopK>
hipError_t launch_t(const ScanArgs &a, uint32_t nq, uint32_t grid_x, bool nt, float scale, size_t lds, hipStream_t s)
{
    const dim3 grid(grid_x, nq), block(kBlock);
    const uint32_t *qw = (const uint32_t *)a.queries;
    if (nt)
        hipLaunchKernelGGL((scan_i8dom_kernel<TopK, 8, true>), grid, block, lds, s, a.corpus, qw, a.lists, a.n_rows,
                           a.d_chunks, a.n_rowblocks, a.q_stride, a.k, a.n_lists, scale);
    else
        hipLaunchKernelGGL((scan_i8dom_kernel<TopK, 8, false>), grid, block, lds, s, a.corpus, qw, a.lists, a.n_rows,
                           a.d_chunks, a.n_rowblocks, a.q_stride, a.k, a.n_lists, scale);
    return hipGetLastError();
}

} // namespace

hipError_t launch_query_quantize(const float *d_queries, uint32_t nq, uint32_t dim, float scale, uint32_t *d_qw,
                                 uint32_t n_words, uint32_t qw_stride, hipStream_t s)
{
    hipLaunchKernelGGL(query_quantize_kernel, dim3(nq), dim3(256), 0, s, d_queries, dim, scale, d_qw, n_words,
                       qw_stride);
    return hipGetLastError();
}

// a.queries = packed quantized queries (launch_query_quantize), a.q_stride = their stride in dwords
hipError_t launch_scan_i8dom(const ScanArgs &a, uint32_t nq, uint32_t grid_x, bool nt, float scale, hipStream_t s)
{
    const size_t lds = scan_lds_bytes(a.k, 1, 0);
    return a.k <= kMaxRegisterK ? launch_t<WaveTopKReg>(a, nq, grid_x, nt, scale, lds, s)
                                : launch_t<WaveTopKLds>(a, nq, grid_x, nt, scale, lds, s);
}

} // namespace vfsx
