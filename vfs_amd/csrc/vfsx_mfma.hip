// vfsx_mfma.hip -- many-query batches (BASELINE config C4) as a dense contraction
// Q . X^T on the CDNA4 matrix cores, with a fused per-lane top-3 epilogue, followed by
// an EXACT sequential-order re-score of the surviving candidates and a rigorous check
// that no row outside them can belong to the reference's top-k.
//
// The reference has no multi-query API (rank.rs:90 takes one &VFSVector); the batch is
// defined as "A3 + A4 applied independently per query" (batch note), so
// the observable result of this path must equal the single-query scan's bit for bit.
// Pipeline (Cosine; the Euclidean variant differs only in the prefilter epilogue and the check, see
// mfma_prefilter256_kernel MODE 1 and mfma_rescore_kernel):
//
//   mirror   X~ = bf16(x / ||x||) in MFMA fragment order          (built once per corpus)
//   prep     Q~ = bf16(q / ||q||) in the same order               (per batch)
//   prefilter  S~ = X~ . Q~^T on v_mfma_f32_32x32x16_bf16; each lane keeps the best three
//              scores of its (query, row-subset) over a strip of 4096 rows, the row's
//              position packed into the 10 low mantissa bits            -> cand[q][3][list]
//   select   per query: the kp best approximate scores over all lists  -> sel[q][kp]
//   rescore  per query: exact sequential f32 distance (the scan's arithmetic) of the kp
//            rows, exact top-k by (distance, row), and the safety check:
//              every row NOT re-scored has  s~ < c_k - eps_q,  c_k = 1 - (k-th exact distance)
//            with eps_q bounding |s~ - cos_reference| for this query against EVERY row (the error
//            model at the end of this file: the mirrors' measured bf16 rounding residuals, not a
//            typical-case figure).  A query that fails it is flagged and re-run by the exact scan.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "vfsx_device.h"
#include "vfsx_internal.h"

#pragma clang fp contract(off)

namespace vfsx {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t kMaxStripTiles = 16;  // 256-row tiles per strip (runtime): up to 4096 rows share one top-3 per lane
constexpr uint32_t kListDepth = 3;       // scores kept per candidate list
constexpr uint32_t kIdxBits = 10;        // 4 bits tile-in-strip, 2 bits mi, 4 bits accumulator register
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1u;

__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f)
{
    uint32_t u = __float_as_uint(f); // finite inputs only (non-finite rows/queries are flagged, not mirrored)
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// The batch path's f32 error terms (mfma_f32_slack, gamma) are RELATIVE f32 bounds: they hold while no product or
// sum the reference forms is subnormal or overflows.  Norms inside [2^-30, 2^30] guarantee that for everything
// that matters (a term that still underflows is below 2^-89 of the result's scale).  A row outside the range
// (or with a zero / non-finite norm) disables the path for the corpus, a query outside it goes to the exact scan.
__device__ __forceinline__ bool norm_in_mfma_range(float n) { return n >= 0x1p-30f && n <= 0x1p30f; }

// ---- mirror: rb64 corpus -> normalised bf16 in fragment order ----------------------------
// unit (g, s) = 1 KiB: lane l holds row 32g + (l & 31), k = 16s + 8(l >> 5) .. +7 -- exactly
// the A/B operand of v_mfma_f32_32x32x16_bf16, so staging is a linear copy and every
// fragment read is a conflict-free lane-linear ds_read_b128.
// stats[0]: set when a row's norm is outside the path's range; stats[1]: see mfma_residual_kernel.
template <int DTYPE>
__global__ void mfma_mirror_kernel(const uint4 *__restrict__ corpus, const float *__restrict__ row_norms,
                                   const float *__restrict__ lut, uint4 *__restrict__ mirror, uint32_t n_rows,
                                   uint32_t dim, uint32_t d_chunks, uint32_t KS, uint64_t total_units,
                                   uint32_t *__restrict__ stats)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t unit = t >> 6;
    if (unit >= total_units)
        return;
    const uint32_t lane = (uint32_t)(t & 63u);
    const uint32_t g = (uint32_t)(unit / KS), s = (uint32_t)(unit % KS);
    const uint32_t row = g * 32u + (lane & 31u);
    const uint32_t k0 = s * 16u + (lane >> 5) * 8u;
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < n_rows) {
        const float n2 = row_norms[row];
        const bool ok = norm_in_mfma_range(n2);
        if (!ok) {
            if (s == 0 && lane < 32u)
                atomicOr(&stats[0], 1u);
        } else {
            const uint8_t *base = (const uint8_t *)(corpus + (size_t)(row >> 6) * d_chunks * 64u + (row & 63u));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = k0 + j;
                float x = 0.0f;
                if (k < dim) {
                    if (DTYPE == VFSX_I8)
                        x = lut[base[(size_t)(k >> 4) * 1024u + (k & 15u)]];
                    else
                        x = *(const float *)(base + (size_t)(k >> 2) * 1024u + (k & 3u) * 4u);
                }
                h[j] = f32_to_bf16_rne(x / n2);
            }
        }
    }
    mirror[unit * 64u + lane] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16),
                                           h[6] | (h[7] << 16));
}

// What the bf16 rounding did to one mirrored element: y = fl(x / n) is what was rounded, y - bf16(y) is
// exactly representable in f32 (the low bits of y), so the residual is measured, not estimated.
__device__ __forceinline__ float bf16_residual(float y)
{
    return y - __uint_as_float(f32_to_bf16_rne(y) << 16);
}
// upper bound of || x/n - mirror(x) ||_2 from the f32 sum of the squared residuals of `K` elements: the sum
// and the sqrt lose at most (K/2 + 2) 2^-24 of it, and y itself is within 2^-24 of x/n per element
__device__ __forceinline__ float residual_norm_bound(float sum_sq, uint32_t K)
{
    return __builtin_sqrtf(sum_sq) * (1.0f + ((float)K + 8.0f) * 0x1p-24f) + 0x1p-23f;
}

// ---- mirror residual: R_x = max over rows of || x/b - x~ ||_2 (error model at the end of the file) ------
// One thread per row, walking its chunks like row_norms_kernel; the wave's maximum goes to stats[1] by an
// integer atomicMax on the float's bits (non-negative floats order like their bit patterns).
template <int DTYPE>
__global__ void mfma_residual_kernel(const uint4 *__restrict__ corpus, const float *__restrict__ row_norms,
                                     const float *__restrict__ lut, uint32_t n_rows, uint32_t d_chunks, uint32_t K,
                                     uint32_t *__restrict__ stats)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    float r = 0.0f;
    if (row < n_rows) {
        const float n2 = row_norms[row];
        if (norm_in_mfma_range(n2)) {
            const uint4 *p = corpus + (size_t)(row >> 6) * d_chunks * 64u + (row & 63u);
            float acc = 0.0f;
            for (uint32_t c = 0; c < d_chunks; ++c) {
                const uint4 v = p[(size_t)c * 64u];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                if (DTYPE == VFSX_I8) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float d = bf16_residual(lut[(w[e >> 2] >> (8 * (e & 3))) & 0xffu] / n2);
                        acc += d * d;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = bf16_residual(__uint_as_float(w[e]) / n2);
                        acc += d * d;
                    }
                }
            }
            r = residual_norm_bound(acc, K);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        r = fmaxf(r, __shfl_xor(r, o, 64));
    if ((threadIdx.x & 63u) == 0u && r > 0.0f)
        atomicMax(&stats[1], __float_as_uint(r));
}

// ---- int8 mirror of an f32 store: rb64 int8 rows with a scale per row ------------------------------------
// x^ = x / b (b = the f32 row norm), s_x = max_j |x^_j| / 127, x~_j = round(x^_j / s_x): the row becomes an int8 row in
// the rb64 layout of an int8 store (16 k per 16-byte chunk), so that the int8 prefilter reads it exactly as it reads
// an int8 corpus, at half the bytes per k of the bf16 mirror and twice the MFMA rate.  inv_rn[row] = s_x turns the
// integer dot product into the cosine up to the query's scale.  The quantization residual r_x = || x^ - s_x x~ ||_2
// is measured per row like the bf16 one; the corpus maximum goes to stats[1] (error model at the end of the file).
// One thread per row (lane l of a wave = row 64 b + l: every load and store is a coalesced 1 KiB).
__global__ void mfma_mirror_i8_kernel(const uint4 *__restrict__ corpus, const float *__restrict__ row_norms,
                                      uint4 *__restrict__ mirror, float *__restrict__ inv_rn, uint32_t n_rows,
                                      uint32_t d_chunks, uint32_t d_chunks8, uint32_t K, uint32_t *__restrict__ stats,
                                      uint32_t row0)
{
    const uint32_t row = row0 + blockIdx.x * blockDim.x + threadIdx.x; // rows [row0, n_rows): an append only adds these
    float r = 0.0f;
    if (row < n_rows) {
        const float b = row_norms[row];
        const uint4 *p = corpus + (size_t)(row >> 6) * d_chunks * 64u + (row & 63u);
        uint4 *out = mirror + (size_t)(row >> 6) * d_chunks8 * 64u + (row & 63u);
        float mx = 0.0f;
        for (uint32_t c = 0; c < d_chunks; ++c) {
            const uint4 v = p[(size_t)c * 64u];
            mx = fmaxf(fmaxf(mx, fabsf(__uint_as_float(v.x))),
                       fmaxf(fabsf(__uint_as_float(v.y)), fmaxf(fabsf(__uint_as_float(v.z)), fabsf(__uint_as_float(v.w)))));
        }
        // division by b > 0 is monotone, so max_j |fl(x_j / b)| = fl(max_j |x_j| / b)
        const float sx = norm_in_mfma_range(b) ? (mx / b) / 127.0f : 0.0f;
        if (!(sx > 0.0f) || !norm_in_mfma_range(sx)) {
            atomicOr(&stats[0], 1u); // zero / out-of-range row: the corpus does not take the batch path
            inv_rn[row] = 0.0f;
        } else {
            float acc = 0.0f;
            for (uint32_t c8 = 0; c8 < d_chunks8; ++c8) {
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t c = c8 * 4u + u;
                    if (c < d_chunks) { // (a store's chunks beyond dim hold zeros: they quantize to 0 exactly)
                        const uint4 v = p[(size_t)c * 64u];
                        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = __uint_as_float(x[e]) / b;
                            const float t = fminf(fmaxf(rintf(y / sx), -127.0f), 127.0f);
                            const float d = y - sx * t;
                            acc += d * d;
                            w[u] |= ((uint32_t)(int)t & 0xffu) << (8 * e);
                        }
                    }
                }
                out[(size_t)c8 * 64u] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            inv_rn[row] = sx;
            r = residual_norm_bound(acc, K) + 0x1p-22f; // + the f32 roundings inside d (<= 3 * 2^-24 ||x^||)
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        r = fmaxf(r, __shfl_xor(r, o, 64));
    if ((threadIdx.x & 63u) == 0u && r > 0.0f)
        atomicMax(&stats[1], __float_as_uint(r));
}

// ---- Euclidean batches: per-row (b, b^2 (1 - gamma)) with b = the f32 row norm the mirror used ----
// inv (int8 stores, may be null): 1 / (scale b) = 1 / ||stored bytes||, the factor that turns the integer dot product
// into the cosine up to the query's own scale (the store holds x = byte / scale, vector.rs:138); stats[0] is raised
// for a norm outside the path's range (the bf16 mirror kernel does the same).
__global__ void mfma_rownorm_pairs_kernel(const float *__restrict__ row_norms, uint32_t n_rows, uint32_t n_pad,
                                          float gamma, float2 *__restrict__ out, float *__restrict__ inv,
                                          float i8_scale, uint32_t *__restrict__ stats, uint32_t row0)
{
    const uint32_t r = row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad)
        return;
    float b = 0.0f, bsq = 0.0f, iv = 0.0f;
    if (r < n_rows) {
        b = row_norms[r];
        if (inv)
            iv = 1.0f / (i8_scale * b);
        if (norm_in_mfma_range(b) && (!inv || norm_in_mfma_range(fabsf(iv)))) {
            bsq = (b * b) * (1.0f - gamma);
        } else {
            iv = 0.0f;
            atomicOr(&stats[0], 1u);
            b = 0.0f;
        }
    }
    out[r] = make_float2(b, bsq);
    if (inv)
        inv[r] = iv;
}

// ---- query prep: one wave per query -------------------------------------------------------
// Writes the query's mirror row, its norm a, and eps_q >= |s~ - q.x / (a b)| for every row x of the corpus
// (error model at the end of the file): (R_x + r_q)(1 + 2^-7) + f32 slack, with r_q this query's own measured
// rounding residual and R_x = stats[1] the mirror's.
__global__ void mfma_query_prep_kernel(const float *__restrict__ q_in, uint32_t nq, uint32_t dim,
                                       uint16_t *__restrict__ q_mirror, uint32_t KS, int *__restrict__ q_bad,
                                       float *__restrict__ q_norm, float *__restrict__ q_eps,
                                       const uint32_t *__restrict__ stats, float f32_slack)
{
    const uint32_t q = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const float *src = q_in + (size_t)q * dim;
    float s = 0.0f;
    for (uint32_t j = lane; j < dim; j += 64)
        s += src[j] * src[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_xor(s, o, 64);
    const float n1 = __builtin_sqrtf(s);
    const bool ok = norm_in_mfma_range(n1);
    float res = 0.0f;
    // every k of the padded row is written: zeros beyond dim, and a zero row for a query that takes the exact scan
    // (rows of queries beyond nq in the last 32-query unit stay as they are: a column of the product only reaches
    // its own query's scores, and those are never stored)
    for (uint32_t k = lane; k < KS * 16u; k += 64) {
        const uint32_t unit = (q >> 5) * KS + (k >> 4);
        const uint32_t l2 = (q & 31u) + 32u * ((k & 15u) >> 3);
        uint16_t m = 0;
        if (ok && k < dim) {
            const float y = src[k] / n1;
            m = (uint16_t)f32_to_bf16_rne(y);
            const float d = bf16_residual(y);
            res += d * d;
        }
        q_mirror[((size_t)unit * 64u + l2) * 8u + (k & 7u)] = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        res += __shfl_xor(res, o, 64);
    if (lane == 0) {
        q_bad[q] = ok ? 0 : 1;
        q_norm[q] = ok ? n1 : 0.0f; // the scalar the mirror row was divided by (Euclidean epilogue + check)
        const float r_q = residual_norm_bound(res, KS * 16u);
        const float r_x = __uint_as_float(stats[1]);
        q_eps[q] = (r_x + r_q) * (1.0f + 0x1p-7f) + f32_slack;
    }
}

// ---- query prep, int8 stores: one wave per query -------------------------------------------------------
// q^ = q / a is quantized to int8 with its own scale: s_q = max |q^_j| / 127, q~_j = round(q^_j / s_q).  The mirror
// row is int8 in the fragment order of v_mfma_i32_32x32x32_i8 (unit = 32 queries x 32 k, lane l: query l & 31,
// 16 consecutive k from 16 (l >> 5)), zero beyond dim.  The rows are exact integers, so the whole score error is the
// query's: |s_q S / ||b|| - cos| <= r_q = || q^ - s_q q~ ||_2 (Cauchy-Schwarz), measured here with its own f32
// roundings added; eps_q = r_q (1 + 2^-7) + f32 slack as on the bf16 path.
__global__ void mfma_query_prep_i8_kernel(const float *__restrict__ q_in, uint32_t nq, uint32_t dim,
                                          int8_t *__restrict__ q_mirror, uint32_t KS, int *__restrict__ q_bad,
                                          float *__restrict__ q_norm, float *__restrict__ q_eps,
                                          float *__restrict__ q_scale, const uint32_t *__restrict__ stats,
                                          float f32_slack)
{
    const uint32_t q = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const float *src = q_in + (size_t)q * dim;
    float s = 0.0f, mx = 0.0f;
    for (uint32_t j = lane; j < dim; j += 64) {
        s += src[j] * src[j];
        mx = fmaxf(mx, fabsf(src[j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    const float n1 = __builtin_sqrtf(s);
    const bool ok = norm_in_mfma_range(n1) && mx > 0.0f;
    const float sq = ok ? (mx / n1) / 127.0f : 0.0f; // the largest component maps to +-127 (up to rounding: clamped)
    float res = 0.0f;
    for (uint32_t k = lane; k < KS * 32u; k += 64) { // the whole padded row, as in mfma_query_prep_kernel
        const uint32_t unit = (q >> 5) * KS + (k >> 5);
        const uint32_t l2 = (q & 31u) + 32u * ((k & 31u) >> 4);
        int8_t m = 0;
        if (ok && k < dim) {
            const float y = src[k] / n1;
            float t = rintf(y / sq);
            t = fminf(fmaxf(t, -127.0f), 127.0f);
            m = (int8_t)(int)t;
            const float d = y - sq * t; // two roundings of <= 2^-24 |y| each: covered below
            res += d * d;
        }
        q_mirror[((size_t)unit * 64u + l2) * 16u + (k & 15u)] = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        res += __shfl_xor(res, o, 64);
    if (lane == 0) {
        q_bad[q] = ok ? 0 : 1;
        q_norm[q] = ok ? n1 : 0.0f;
        q_scale[q] = sq;
        // + 2^-22: the f32 roundings inside d (<= 3 * 2^-24 ||q^||) and of q^ itself
        const float r_q = residual_norm_bound(res, KS * 32u) + 0x1p-22f;
        // int8 store: the rows are exact (stats == nullptr, R_x = 0).  int8 mirror of an f32 store: the rows carry
        // R_x = stats[1] too, and |s_x s_q S - x^.q^| <= r_x ||s_q q~|| + ||x^|| r_q <= r_x + r_q + r_x r_q (+ roundings)
        const float r_x = stats ? __uint_as_float(stats[1]) : 0.0f;
        q_eps[q] = ((r_x + r_q) * (1.0f + 0x1p-7f) + r_x * r_q) + f32_slack;
    }
}

// ---- prefilter GEMM ---------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 frag_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v)
{
    union {
        uint4 u;
        bf16x8 b;
    } c;
    c.u = v;
    return c.b;
}

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// one 32 x 32 MFMA on a 16-byte fragment pair: 16 bf16 k (f32 accumulators) or 32 int8 k (exact i32 accumulators)
__device__ __forceinline__ f32x16 mfma_32x32(uint4 a, uint4 b, f32x16 c)
{
#ifdef VFSX_EXP_M16 // timing experiment (WRONG results): the same flops, bytes and registers on the 16x16x32 shape
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c1, 0, 0, 0);
    c[0] = c0[0], c[1] = c0[1], c[2] = c0[2], c[3] = c0[3], c[4] = c1[0], c[5] = c1[1], c[6] = c1[2], c[7] = c1[3];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ i32x16 mfma_32x32(uint4 a, uint4 b, i32x16 c)
{
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}

// ---- prefilter GEMM, 256 rows x 256 queries per workgroup ------------------------------------------
// 8 waves as 2 (M) x 4 (N), 128 rows x 64 queries per wave (4 x 2 MFMA tiles, 128 accumulator registers),
// 6 fragment reads per 8 MFMAs: 128 flop per byte moved L2 -> LDS.  (Round 1 measured a 128 x 128 tile,
// 64 flop/B, at 0.72 PFLOP/s and a 256 x 128 tile, 85 flop/B, at 0.9: both sat at ~11 TB/s of LDS fill,
// not at the matrix cores; those kernels and a 16x16x32-MFMA twin of this one, which measured the same as
// this one, were removed in round 2.)  K runs in steps of 16*KU through an NSTAGE-deep ring of LDS slots
// (KU=4: 2 x 64 KiB) filled by buffer_load ... lds: the stage for K-step it+NSTAGE-1 is issued during K-step
// `it`, a counted s_waitcnt vmcnt leaves NSTAGE-2 stages in flight across the (raw) barrier.
//
// Two compiler facts are load-bearing.  hipcc's waitcnt pass orders every ds_read of a __shared__ object
// behind the pending LDS-DMAs it can attribute to that object and, unable to count them in order, does so
// with s_waitcnt vmcnt(0) -- draining the ring once per unrolled round and serialising the freshly issued
// stage with the MFMAs.  So (1) every DMA's LDS destination is laundered through an empty asm (the compiler
// cannot tell which object it writes), and (2) there is one __shared__ object per ring slot and the K loop is
// unrolled NSTAGE times so that every slot index is a compile-time constant.  The ring's own protocol (counted
// vmcnt + s_barrier before a slot is read, barrier before it is refilled) is what makes the reads safe.
//
// Candidate lists: list = strip * 4 + wave_m * 2 + (lane >> 5); row position in the 10 low
// mantissa bits = tile-in-strip (4 bits) | mi (2 bits) | accumulator register (4 bits).
//
// MODE 0 (Cosine): the candidate score is s~ itself.  MODE 1 (Euclidean): the same contraction, but
// the epilogue turns s~ into an UPPER bound of  V = 2 q.x - ||x||^2  (D^2 = ||q||^2 - V, so larger V =
// closer):  key = 2a * (b * (s~ + eps_q)) - b^2 (1 - gamma), a / b = the norms the two mirrors were
// divided by, eps_q >= |s~ - q.x / (a b)| per query (q_eps, written by mfma_query_prep_kernel), gamma
// covering the f32 error of b^2 vs ||x||^2; the three roundings here are covered by the surplus f32 slack
// inside eps_q (error model at the end of the file).  rn2[row] = (b, b^2 (1 - gamma)).
//
// I8 (int8 stores): the same kernel on v_mfma_i32_32x32x32_i8 -- twice the k per instruction at the same cycles --
// with NO mirror of the corpus at all.  A 16-byte rb64 chunk is 16 consecutive k of one row, which is exactly one
// lane's share of that MFMA's A operand (lane l: row l & 31, k-half l >> 5), so the stage reads the rb64 corpus
// itself: per-lane source offset (l >> 5) * 1024 + (l & 31) * 16 inside a (row-block, chunk pair), everything else
// in SGPRs as before.  The query operand is int8 in the same fragment order (mfma_query_prep_i8_kernel), ZERO
// beyond dim: K is padded to the ring round (256) by reading whatever follows the row's last chunk (the next
// row-block, or 0 past the allocation: the buffer resource is bounded) times a zero query byte.  The dot product
// S = sum b_j q~_j is exact in int32; the score is S / (scale ||x||), one v_cvt and one multiply in the epilogue
// (Euclidean: two FMAs on the same per-row / per-query constants as the bf16 epilogue).
#ifndef VFSX_PF_KU // K-step (in 16-byte fragment k-slices) and ring depth of mfma_prefilter256_kernel
#define VFSX_PF_KU 4
#define VFSX_PF_NSTAGE 2
#endif

#ifdef VFSX_EXP_STAMPS // diagnostic build only: where a wave's cycles go inside a K-step (s_memtime sums over all waves)
__device__ unsigned long long exp_stamps[8];
#define VFSX_STAMP(x) const uint64_t x = __builtin_amdgcn_s_memtime()
#else
#define VFSX_STAMP(x)
#endif

template <int KU, int NSTAGE, int MODE, bool I8>
__global__ __launch_bounds__(512) void mfma_prefilter256_kernel(const uint4 *__restrict__ A,
                                                                const uint4 *__restrict__ B,
                                                                float *__restrict__ cand, uint32_t n_rows,
                                                                uint32_t n_rowtiles, uint32_t KS, uint32_t n_qtiles,
                                                                uint32_t nq, uint32_t n_lists, uint32_t strip_tiles,
                                                                const float2 *__restrict__ rn2,
                                                                const float *__restrict__ q_norm,
                                                                const float *__restrict__ q_eps,
                                                                uint32_t a_chunks, uint64_t a_bytes,
                                                                const float *__restrict__ inv_rn,
                                                                const float *__restrict__ q_scale, uint32_t qpx,
                                                                uint32_t n_strips)
{
    constexpr uint32_t TROWS = 256u, TQ = 256u;
    constexpr uint32_t A_UNITS = 8u * KU, UNITS = 16u * KU;
    constexpr uint32_t G = 2u * KU; // global_load_lds instructions per wave and stage
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    static_assert((size_t)NSTAGE * UNITS * 1024u <= 160u * 1024u, "ring exceeds the LDS");
    __shared__ __align__(16) uint4 lds_s0[UNITS * 64];
    __shared__ __align__(16) uint4 lds_s1[UNITS * 64];
    __shared__ __align__(16) uint4 lds_s2[NSTAGE > 2 ? UNITS * 64 : 4];
    __shared__ __align__(16) uint4 lds_s3[NSTAGE > 3 ? UNITS * 64 : 4];
    auto slot = [&](int j) -> uint4 * { return j == 0 ? lds_s0 : (j == 1 ? lds_s1 : (j == 2 ? lds_s2 : lds_s3)); };

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // SGPR
    const uint32_t wave_m = wave >> 2, wave_n = wave & 3u;

    // XCD-aware mapping (speed only).  Workgroups with equal id % 8 share an XCD and its 4 MiB L2.  The query tiles
    // are cut into G groups of qpx (G divides 8); XCD x works on group x % G -- its qpx tiles of Q~ stay L2-resident
    // -- and on every (8 / G)-th row strip, qpx workgroups per strip side by side: a row tile is fetched into the
    // XCD's L2 once and served from there to the other qpx - 1, and the XCDs that need the same strip walk the strips
    // at the same pace, so that one HBM read serves them through the Infinity Cache.  The host picks qpx (pf_grid)
    // and sizes the grid to 8 * qpx * ceil(n_strips / (8 / G)); surplus workgroups leave here, as a whole.
    const uint32_t id = blockIdx.x, xcd = id & 7u, within = id >> 3;
    const uint32_t n_groups = n_qtiles / qpx, lanes = 8u / n_groups;
    const uint32_t qtile = (xcd % n_groups) * qpx + within % qpx;
    const uint32_t strip = (within / qpx) * lanes + xcd / n_groups;
    if (strip >= n_strips)
        return;
    const uint32_t tile0 = strip * strip_tiles;
    const uint32_t ntiles = min(strip_tiles, n_rowtiles - tile0);
    const uint32_t KT = KS / KU;
    const uint32_t n_it = ntiles * KT;

    // waves 0-3 stage the A (row) units, waves 4-7 the B (query) units: G pieces (1 KiB each) per wave
    // and stage, as buffer_load_dwordx4 ... lds: the resource (SGPRs) points at the tile's first unit,
    // the piece offset travels in an SGPR and the only VGPR operand is the constant lane * 16 -- no
    // per-piece 64-bit vector address arithmetic.  Pieces are issued two per 8-MFMA group (see kstep)
    // instead of as a burst behind the barrier.
    const bool a_wave = wave < 4u;
    const bool rb64_src = I8 && a_wave; // this wave stages straight out of the rb64 corpus
    const uint32_t lane16 = rb64_src ? (lane >> 5) * 1024u + (lane & 31u) * 16u : lane * 16u;
    // All bookkeeping below is wave-uniform and incremental (SGPRs, a handful of s_add per K-step):
    // per-piece constants, the next stage's K offset and tile base, the current K-step's position.
    uint32_t pc_soff[G], pc_lds[G];
#pragma unroll
    for (uint32_t u = 0; u < G; ++u) {
        const uint32_t unit = (wave & 3u) * G + u; // 0 .. 8 KU - 1
        const uint32_t grp = unit / KU, ks = unit % KU;
        // rb64: 32-row group grp = row-block grp >> 1 of the tile, its rows 32 (grp & 1) ..; k-slice ks = chunk pair
        pc_soff[u] = rb64_src ? (grp >> 1) * a_chunks * 1024u + (grp & 1u) * 512u + ks * 2048u : (grp * KS + ks) * 1024u;
        pc_lds[u] = ((a_wave ? 0u : A_UNITS) + unit) * 1024u;
    }
    const uint32_t q_tile_bytes = 8u * KS * 1024u; // one 256-query tile of the query mirror
    const uint32_t tile_bytes = rb64_src ? 4u * a_chunks * 1024u : q_tile_bytes; // one 256-row tile of the rows
    const uint32_t koff_step = rb64_src ? KU * 2048u : KU * 1024u;               // one K-step inside a row / unit row
    const uint32_t koff_end = koff_step * (KS / KU);
    uintptr_t nx_base = a_wave ? (uintptr_t)A + (uintptr_t)tile0 * tile_bytes : (uintptr_t)B + (uintptr_t)qtile * q_tile_bytes;
    const uintptr_t a_end = (uintptr_t)A + a_bytes;
    uint32_t nx_idx = 0, nx_koff = 0; // next stage to issue: index, byte offset of its K-slice inside a unit row
    auto next_rsrc = [&]() -> __amdgpu_buffer_rsrc_t {
        // rb64 source: bounded by the corpus allocation (the K padding and the last tile's missing row-blocks
        // read past it and must get zeros, not a fault)
        uint32_t bound = 0xffffffffu;
        if (rb64_src) {
            const uintptr_t left = nx_base < a_end ? a_end - nx_base : 0;
            bound = left < 0xffffffffu ? (uint32_t)left : 0xffffffffu;
        }
        return __builtin_amdgcn_make_buffer_rsrc((void *)nx_base, 0, bound, 0x00020000);
    };
    auto advance = [&]() { // past the last stage: stay (the redundant re-issue keeps the vmcnt arithmetic constant)
        if (nx_idx + 1u < n_it) {
            ++nx_idx;
            nx_koff += koff_step;
            if (nx_koff == koff_end) {
                nx_koff = 0;
                if (a_wave)
                    nx_base += tile_bytes;
            }
        }
    };
    auto stage_piece = [&](__amdgpu_buffer_rsrc_t rsrc, uint32_t koff, uint32_t slot_base, uint32_t u) {
        // LDS destination laundered (see the kernel's header comment)
        uint32_t a = slot_base + pc_lds[u];
        asm volatile("" : "+s"(a));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(uintptr_t)a, 16,
                                                 (int)lane16, (int)(pc_soff[u] + koff), 0, 0);
    };
    auto lds_addr = [&](uint4 *p) -> uint32_t {
        return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)p;
    };
    auto stage = [&](uint4 *dst) {
        const __amdgpu_buffer_rsrc_t rsrc = next_rsrc();
#pragma unroll
        for (uint32_t u = 0; u < G; ++u)
            stage_piece(rsrc, nx_koff, lds_addr(dst), u);
        advance();
    };

    using AccT = typename std::conditional<I8, i32x16, f32x16>::type;
    // int8: the first k-slice of every tile multiplies into C = 0 and the epilogue leaves the accumulators as they
    // are (c3b +4 %, c4i8 +0.5 %); bf16: the epilogue zeroes them as it reads them -- there the zeroing hides under
    // the partner wave's MFMAs and the C = 0 form measured 0.8 % slower (profiles/r03/c4_analysis.md)
    constexpr bool ZERO_C = I8;
    AccT acc[4][2];
    if (!ZERO_C) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[mi][ni][r] = 0;
    }
    const float ninf = __uint_as_float(0xff800000u);
    float m1[2] = {ninf, ninf}, m2[2] = {ninf, ninf}, m3[2] = {ninf, ninf};
    uint32_t keep_mask = ~kIdxMask;
    asm volatile("" : "+v"(keep_mask)); // keep it in a VGPR (see the epilogue)
    uint32_t cur_kt = 0, cur_ts = 0;    // bf16 (flat loop): position of the running K-step inside the strip
    // MODE 1, per query column: bf16: cq = 2a, ce = eps_q;  int8: cq = 2a s_q (times S ||x|| inv_rn), ce = 2a eps_q (times ||x||)
    float cq[2] = {0.0f, 0.0f}, ce[2] = {0.0f, 0.0f};
    if (MODE == 1) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const uint32_t q = qtile * TQ + wave_n * 64u + (uint32_t)ni * 32u + (lane & 31u);
            if (q < nq) {
                const float a2 = 2.0f * q_norm[q];
                cq[ni] = I8 ? a2 * q_scale[q] : a2;
                ce[ni] = I8 ? a2 * q_eps[q] : q_eps[q];
            }
        }
    }

    auto fold_tile = [&](auto ragged_c, uint32_t ts, uint32_t tile_row0) __attribute__((always_inline)) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                // accumulator registers 4 rq .. 4 rq + 3 are 4 consecutive rows
                const uint32_t row4 = tile_row0 + wave_m * 128u + (uint32_t)mi * 32u + 8u * (uint32_t)rq + 4u * (lane >> 5);
                float bn[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
                if (MODE == 1) { // 32-byte aligned: row4 % 4 == 0, rn2 padded to the tile
                    const float4 p0 = *(const float4 *)(rn2 + row4);
                    const float4 p1 = *(const float4 *)(rn2 + row4 + 2u);
                    bn[0] = p0.x, bs[0] = p0.y, bn[1] = p0.z, bs[1] = p0.w;
                    bn[2] = p1.x, bs[2] = p1.y, bn[3] = p1.z, bs[3] = p1.w;
                }
                float rf[4] = {0.f, 0.f, 0.f, 0.f}; // int8: S * inv_rn = the cosine up to the query's scale s_q
                if (I8) {                           // (padded to the tile, 0 beyond n_rows)
                    const float4 p = *(const float4 *)(inv_rn + row4);
                    rf[0] = p.x, rf[1] = p.y, rf[2] = p.z, rf[3] = p.w;
                    if (MODE == 1) { // q.x = a ||x|| cos: the row's factor becomes ||x|| inv_rn
#pragma unroll
                        for (int rl = 0; rl < 4; ++rl)
                            rf[rl] *= bn[rl];
                    }
                }
#pragma unroll
                for (int rl = 0; rl < 4; ++rl) {
                    const int r = rq * 4 + rl;
                    // the index is wave-uniform: keep it in an SGPR so that mask (VGPR) + index is ONE
                    // v_and_or_b32 per element
                    const uint32_t idx =
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)((ts << 6) | ((uint32_t)mi << 4) | (uint32_t)r));
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
#ifdef VFSX_EXP_NOFOLD // experiment (wrong results): what the kernel costs without its epilogue
                        if (r != 0) {
                            acc[mi][ni][r] = 0;
                            continue;
                        }
#endif
                        float v = (float)acc[mi][ni][r];
                        if (I8 && MODE == 1) // upper bound of V = 2 q.x - ||x||^2: 2a ||x|| (s_q S inv_rn + eps_q) - ||x||^2 (1 - gamma)
                            v = __builtin_fmaf(v * rf[rl], cq[ni], __builtin_fmaf(bn[rl], ce[ni], -bs[rl]));
                        else if (I8) // S inv_rn: the cosine up to the query's scale s_q (mfma_rescore_kernel applies it)
                            v = v * rf[rl];
                        else if (MODE == 1)
                            v = cq[ni] * (bn[rl] * (v + ce[ni])) - bs[rl];
                        float sp = __uint_as_float((__float_as_uint(v) & keep_mask) | idx);
                        if (RAGGED && row4 + (uint32_t)rl >= n_rows)
                            sp = ninf;
                        // sorted-triple insertion (m1 >= m2 >= m3): two v_med3_f32 and one v_max_f32; the
                        // max as asm because fmaxf on bit-assembled floats costs two canonicalising v_max more
                        m3[ni] = __builtin_amdgcn_fmed3f(m2[ni], m3[ni], sp);
                        m2[ni] = __builtin_amdgcn_fmed3f(m1[ni], m2[ni], sp);
                        asm("v_max_f32 %0, %1, %2" : "=v"(m1[ni]) : "v"(m1[ni]), "v"(sp));
                        if (!ZERO_C) // (int8: not reset, the next tile's first k-slice multiplies into C = 0)
                            acc[mi][ni][r] = 0;
                    }
                }
            }
    };
    // tile finished: fold its 128 scores per lane into the running top-3 of each query column.
    // C/D layout: column (query) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    auto fold_one_tile = [&](uint32_t ts) __attribute__((always_inline)) {
        const uint32_t tile_row0 = (tile0 + ts) * TROWS;
        if (tile_row0 + TROWS > n_rows)
            fold_tile(std::true_type{}, ts, tile_row0); // last, ragged tile: rows >= n_rows masked
        else
            fold_tile(std::false_type{}, ts, tile_row0);
    };

#ifdef VFSX_EXP_STAMPS
    uint64_t st_vm = 0, st_bar = 0, st_body = 0, st_fold = 0, st_n = 0;
    const uint64_t st_begin = __builtin_amdgcn_s_memtime(), st_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
    // Fragment registers: two sets, k-slice ks uses set ks & 1 (KU is even, so every K-step starts on set 0).
    // PREFETCH (NSTAGE >= 3): the fragments of k-slice ks + 1 are read while the MFMAs of k-slice ks run, ACROSS the
    // barrier too -- the last k-slice of K-step `it` reads the first fragments of K-step it + 1.  That needs stage
    // it + 1 complete and visible one barrier early, so barrier(it) waits for the stages up to it + 1 (all but the
    // youngest NSTAGE - 3 in flight), which a ring of >= 3 slots affords.  Without it every barrier is followed by a
    // bubble in which all 8 waves wait for their first ds_reads (48 KiB through the LDS at once) and no MFMA issues:
    // stamped at ~250-300 cycles per barrier (profiles/r02/c4_analysis.md).
    constexpr bool PREFETCH = NSTAGE >= 3;
    static_assert(KU % 2 == 0, "fragment sets alternate per k-slice");
    uint4 fa[2][4], fb[2][2];
    auto load_frags = [&](const uint4 *rd, uint32_t ks, int set) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t mi = 0; mi < 4; ++mi)
            fa[set][mi] = rd[((wave_m * 4u + mi) * KU + ks) * 64u + lane];
#pragma unroll
        for (uint32_t ni = 0; ni < 2; ++ni)
            fb[set][ni] = rd[(A_UNITS + (wave_n * 2u + ni) * KU + ks) * 64u + lane];
    };
    // FIRST: the tile's first K-step -- its first k-slice multiplies into C = 0 (an inline constant of the MFMA),
    // so the epilogue does not have to zero the 128 accumulator registers it has just read (one VALU operation per
    // element less: a seventh of the int8 fold, a fifth of the bf16 one)
    auto kstep = [&](auto jc, auto first_c) {
        constexpr int J = decltype(jc)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int JW = (J + NSTAGE - 1) % NSTAGE;
        constexpr int JN = (J + 1) % NSTAGE;
        const uint4 *rd = slot(J);
        // Every K-step issues exactly one stage (past the end: a redundant copy of the last one into
        // the free slot), so the number of younger stages in flight here is a constant.  It also keeps
        // hipcc's own (path-insensitive) LDS-DMA bookkeeping off vmcnt(0): with conditional stages it must
        // assume none was issued and drains the ring once per unrolled round.
        constexpr uint32_t IN_FLIGHT = G * (PREFETCH ? NSTAGE - 3 : NSTAGE - 2);
        VFSX_STAMP(t0);
        if (IN_FLIGHT == 4)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (IN_FLIGHT == 8)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VFSX_STAMP(t1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory"); // no fragment read of this stage (PREFETCH: the next) may be hoisted above
        VFSX_STAMP(t2);
        const __amdgpu_buffer_rsrc_t rsrc_n = next_rsrc();
        const uint32_t koff_n = nx_koff;
        const uint32_t wr = lds_addr(slot(JW));
        advance();
#pragma unroll
        for (uint32_t ks = 0; ks < (uint32_t)KU; ++ks) {
            const int set = (int)(ks & 1u);
            if (!PREFETCH)
                load_frags(rd, ks, set);
            else if (ks + 1u < (uint32_t)KU)
                load_frags(rd, ks + 1u, set ^ 1);
            else
                load_frags(slot(JN), 0u, set ^ 1);
            // next stage's pieces: spread over the K-step when the ring gives them NSTAGE-1 >= 2 K-steps
            // to land, front-loaded (first half of the K-step) when the very next barrier needs them: any later
            // piece of a 2-slot ring lands late (3,3,2,0 / 2,2,2,2 per k-slice: +4..5 %, measured)
            constexpr uint32_t PPK = NSTAGE > 2 ? (G + KU - 1u) / KU : 4u;
            if (ks * PPK < G) {
#pragma unroll
                for (uint32_t u = 0; u < PPK; ++u)
                    if (ks * PPK + u < G)
                        stage_piece(rsrc_n, koff_n, wr, ks * PPK + u);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (ZERO_C && FIRST && ks == 0u)
                        acc[mi][ni] = mfma_32x32(fa[set][mi], fb[set][ni], AccT{});
                    else
                        acc[mi][ni] = mfma_32x32(fa[set][mi], fb[set][ni], acc[mi][ni]);
                }
        }
        VFSX_STAMP(t3);
#ifdef VFSX_EXP_STAMPS
        st_vm += t1 - t0, st_bar += t2 - t1, st_body += t3 - t2, st_n += 1;
#endif
        if (!ZERO_C) { // bf16: the fold sits behind the tile's last K-step, inside the flat loop (see the main loop)
            if (++cur_kt == KT) {
                const uint32_t ts = cur_ts;
                cur_kt = 0;
                ++cur_ts;
                fold_one_tile(ts);
#ifdef VFSX_EXP_STAMPS
                st_fold += __builtin_amdgcn_s_memtime() - t3;
#endif
            }
        }
    };

    stage(lds_s0); // stages 0 .. NSTAGE-2 (advance() clamps at the last one)
    if (NSTAGE > 2)
        stage(lds_s1);
    if (NSTAGE > 3)
        stage(lds_s2);
    if (PREFETCH) { // stage 0 landed everywhere -> the first fragments of K-step 0
        if (G * (NSTAGE - 2) == 4)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (G * (NSTAGE - 2) == 8)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frags(lds_s0, 0u, 0);
    }
    // KT % NSTAGE == 0 (the host pads K so that KT is a multiple of NSTAGE): every tile starts on ring slot 0 and is
    // a whole number of straight-line rounds.
    if (ZERO_C) {
        // int8: tile by tile, the tile's first K-step is its own instantiation (C = 0), the fold follows the last
        for (uint32_t ts = 0; ts < ntiles; ++ts) {
            kstep(std::integral_constant<int, 0>{}, std::true_type{}); // (peeled: no branch inside the rounds below)
            kstep(std::integral_constant<int, 1>{}, std::false_type{});
            if (NSTAGE > 2)
                kstep(std::integral_constant<int, (NSTAGE > 2 ? 2 : 0)>{}, std::false_type{});
            if (NSTAGE > 3)
                kstep(std::integral_constant<int, (NSTAGE > 3 ? 3 : 0)>{}, std::false_type{});
            for (uint32_t kt = NSTAGE; kt < KT; kt += NSTAGE) {
                kstep(std::integral_constant<int, 0>{}, std::false_type{});
                kstep(std::integral_constant<int, 1>{}, std::false_type{});
                if (NSTAGE > 2)
                    kstep(std::integral_constant<int, (NSTAGE > 2 ? 2 : 0)>{}, std::false_type{});
                if (NSTAGE > 3)
                    kstep(std::integral_constant<int, (NSTAGE > 3 ? 3 : 0)>{}, std::false_type{});
            }
            VFSX_STAMP(tf0);
            fold_one_tile(ts);
#ifdef VFSX_EXP_STAMPS
            st_fold += __builtin_amdgcn_s_memtime() - tf0;
#endif
        }
    } else {
        // bf16: one flat loop over the strip's K-steps, the fold inside the K-step that ends a tile -- the
        // tile-structured form above measured 1-1.6 % slower here (profiles/r03/c4_analysis.md)
        for (uint32_t it = 0; it < n_it; it += NSTAGE) {
            kstep(std::integral_constant<int, 0>{}, std::false_type{});
            kstep(std::integral_constant<int, 1>{}, std::false_type{});
            if (NSTAGE > 2)
                kstep(std::integral_constant<int, (NSTAGE > 2 ? 2 : 0)>{}, std::false_type{});
            if (NSTAGE > 3)
                kstep(std::integral_constant<int, (NSTAGE > 3 ? 3 : 0)>{}, std::false_type{});
        }
    }

    // the redundant tail stages are still in flight: they must land before this workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef VFSX_EXP_STAMPS
    if (lane == 0) {
        atomicAdd(&exp_stamps[0], (unsigned long long)st_vm);
        atomicAdd(&exp_stamps[1], (unsigned long long)st_bar);
        atomicAdd(&exp_stamps[2], (unsigned long long)st_body);
        atomicAdd(&exp_stamps[3], (unsigned long long)st_fold);
        atomicAdd(&exp_stamps[4], (unsigned long long)st_n);
        atomicAdd(&exp_stamps[5], (unsigned long long)(__builtin_amdgcn_s_memtime() - st_begin));
        atomicAdd(&exp_stamps[6], 1ull);
        atomicAdd(&exp_stamps[7], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - st_rbegin));
    }
#endif

    const uint32_t list = strip * 4u + wave_m * 2u + (lane >> 5);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const uint32_t q = qtile * TQ + wave_n * 64u + (uint32_t)ni * 32u + (lane & 31u);
        if (q < nq) {
            cand[((size_t)q * kListDepth + 0u) * n_lists + list] = m1[ni];
            cand[((size_t)q * kListDepth + 1u) * n_lists + list] = m2[ni];
            cand[((size_t)q * kListDepth + 2u) * n_lists + list] = m3[ni];
        }
    }
}

// ---- prefilter GEMM, rows in registers ("rv"): 256 rows x 256 queries per workgroup, 8 waves as 8 (M) x 1 (N) ------
// What held mfma_prefilter256_kernel at ~0.5 of the matrix peak is the global -> LDS fill beside the MFMAs' own
// ds_reads (profiles/r02/c4_analysis.md: 8 LDS-DMA pieces per wave and K-step, 60-185 issue cycles each, in front of
// that wave's 32 MFMAs).  Here a wave owns 32 rows of the tile outright, so its row fragments never need to be
// shared: the mirror (and an rb64 int8 corpus) is stored in fragment order, one global load per lane IS the A operand
// of one MFMA k-slice -- 1 KiB contiguous (rb64: two 512-byte runs) per wave-instruction, global -> VGPR, no LDS.
// Only the query operand goes through LDS, where all 8 waves read it: per K-step a wave issues 4 LDS-DMA pieces
// (not 8) + 4 plain loads, the CU stages 32 KiB (not 64) into a 4-slot ring (4 x 32 KiB, three stages ahead), and a
// wave reads 8 query fragments per 8 MFMAs (one row fragment x 8 query groups, 128 accumulator registers as before).
// Ring protocol per K-step `it` (one barrier):
//     s_waitcnt vmcnt(8)   this wave's ops older than the 8 of K-step it-1 (row fragments of it+1, pieces of stage
//                          it+2) are done: the row fragments of K-step it and its pieces of the stages <= it+1
//     s_barrier            => stages <= it+1 are visible to every wave; every wave is done reading stage it-1
//     16 groups of {2 ds_read_b128 for group g+2, a VMEM op every other group, s_waitcnt lgkmcnt(4), 2 MFMAs}:
//         VMEM ops: first the row fragments of K-step it+2 into register set (it+2) % 3, then this wave's 4 pieces of
//                   stage it+3 into slot (it+3) % 4 = the slot of stage it-1
//         the last two groups' look-ahead reads are the first fragments of stage it+1 (visible: see the barrier)
// All loads that the ring counts are inline asm or laundered (hipcc would otherwise order every ds_read behind the
// pending LDS-DMAs with vmcnt(0), and any plain VGPR load beside LDS-DMAs the same way); the waits carry the
// registers they release as "+v" operands, so no MFMA can be scheduled above the wait that covers its fragments.
// Candidate lists: bit for bit those of mfma_prefilter256_kernel (wave w covers what (wave_m, mi) = (w >> 2, w & 3)
// covers there; the four waves of one wave_m merge their top-3 through LDS at the end, as in the skinny kernel).
template <int MODE, bool I8>
__global__ __launch_bounds__(512) void mfma_prefilter_rv_kernel(const uint4 *__restrict__ A, const uint4 *__restrict__ B,
                                                                float *__restrict__ cand, uint32_t n_rows,
                                                                uint32_t n_rowtiles, uint32_t KS, uint32_t n_qtiles,
                                                                uint32_t nq, uint32_t n_lists, uint32_t strip_tiles,
                                                                const float2 *__restrict__ rn2,
                                                                const float *__restrict__ q_norm,
                                                                const float *__restrict__ q_eps, uint32_t a_chunks,
                                                                uint64_t a_bytes, const float *__restrict__ inv_rn,
                                                                const float *__restrict__ q_scale, uint32_t qpx,
                                                                uint32_t n_strips)
{
    constexpr uint32_t KU = 4u, NSLOT = 4u, TROWS = 256u, TQ = 256u, NI = 8u;
    constexpr uint32_t SLOT_BYTES = NI * KU * 1024u; // 32 KiB: 8 query groups x 4 k-slices
    __shared__ __align__(16) uint4 ring[NSLOT * SLOT_BYTES / 16u];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // SGPR
    const uint32_t wave_m = wave >> 2, mi = wave & 3u;

    // XCD-aware mapping: as mfma_prefilter256_kernel
    const uint32_t id = blockIdx.x, xcd = id & 7u, within = id >> 3;
    const uint32_t n_groups = n_qtiles / qpx, lanes_x = 8u / n_groups;
    const uint32_t qtile = (xcd % n_groups) * qpx + within % qpx;
    const uint32_t strip = (within / qpx) * lanes_x + xcd / n_groups;
    if (strip >= n_strips)
        return;
    const uint32_t tile0 = strip * strip_tiles;
    const uint32_t ntiles = min(strip_tiles, n_rowtiles - tile0);
    const uint32_t KT = KS / KU;
    const uint32_t n_it = ntiles * KT;

    // ---- row fragments: global -> VGPR ----
    // bf16 mirror: 32-row group (tile * 8 + wave) is KS consecutive 1 KiB units; rb64 int8: row-block (tile * 4 +
    // (wave >> 1)), rows 32 (wave & 1) .., k-slice = a chunk pair (2 KiB apart), lane l reads chunk (l >> 5) of row l & 31
    const uint32_t lane16a = I8 ? (lane >> 5) * 1024u + (lane & 31u) * 16u : lane * 16u;
    const uint32_t a_tile_bytes = I8 ? 4u * a_chunks * 1024u : 8u * KS * 1024u;
    const uint32_t a_piece = I8 ? 2048u : 1024u;
    const uintptr_t a_end = (uintptr_t)A + a_bytes;
    uintptr_t a_base = (uintptr_t)A + (uintptr_t)tile0 * a_tile_bytes +
                       (I8 ? (uintptr_t)(wave >> 1) * a_chunks * 1024u + (wave & 1u) * 512u : (uintptr_t)wave * KS * 1024u);
    uint32_t a_idx = 0, a_koff = 0; // next K-step whose row fragments are to be loaded
    const uint32_t a_kstep = KU * a_piece, a_kend = KT * a_kstep;
    auto a_rsrc = [&]() -> u32x4 {
        // raw buffer resource: base, stride 0, num_records (rb64: bounded by the allocation -- the K padding and the
        // last tile's missing row-blocks read past it and must get zeros, not a fault), flags as make_buffer_rsrc
        uint32_t bound = 0xffffffffu;
        if (I8) {
            const uintptr_t left = a_base < a_end ? a_end - a_base : 0;
            bound = left < 0xffffffffu ? (uint32_t)left : 0xffffffffu;
        }
        u32x4 r;
        r[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a_base);
        r[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)a_base >> 32)) & 0xffffu;
        r[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)bound);
        r[3] = 0x00020000u;
        return r;
    };
    auto a_advance = [&]() {
        if (a_idx + 1u < n_it) { // past the last K-step: stay (a redundant reload keeps the vmcnt arithmetic constant)
            ++a_idx;
            a_koff += a_kstep;
            if (a_koff == a_kend) {
                a_koff = 0;
                a_base += a_tile_bytes;
            }
        }
    };
#define VFSX_RV_ALOAD(dst, rsrc, soff)                                                                             \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(lane16a), "s"(rsrc), "s"(soff) : "memory")

    // ---- query pieces: LDS-DMA, wave w stages query group w (its KU k-slices: 4 KiB contiguous) ----
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)ring;
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((uintptr_t)B + ((uintptr_t)qtile * 8u + wave) * KS * 1024u), 0, 0xffffffff, 0x00020000);
    const uint32_t lane16 = lane * 16u;
    uint32_t b_idx = 0, b_koff = 0, b_slot = 0; // next stage to issue
    auto b_piece = [&](uint32_t u) {
        uint32_t a = ring_base + b_slot * SLOT_BYTES + (wave * KU + u) * 1024u;
        asm volatile("" : "+s"(a)); // LDS destination laundered: see mfma_prefilter256_kernel
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void *)(uintptr_t)a, 16,
                                                 (int)lane16, (int)(b_koff + u * 1024u), 0, 0);
    };
    auto b_advance = [&]() {
        b_slot = (b_slot + 1u) & (NSLOT - 1u);
        if (b_idx + 1u < n_it) {
            ++b_idx;
            b_koff += KU * 1024u;
            if (b_koff == KS * 1024u)
                b_koff = 0;
        }
    };

    typename std::conditional<I8, i32x16, f32x16>::type acc[NI];
#pragma unroll
    for (uint32_t ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[ni][r] = 0;
    const float ninf = __uint_as_float(0xff800000u);
    float m1[NI], m2[NI], m3[NI];
#pragma unroll
    for (uint32_t ni = 0; ni < NI; ++ni)
        m1[ni] = m2[ni] = m3[ni] = ninf;
    uint32_t keep_mask = ~kIdxMask;
    asm volatile("" : "+v"(keep_mask));
    auto insert3 = [&](uint32_t ni, float sp) __attribute__((always_inline)) {
        m3[ni] = __builtin_amdgcn_fmed3f(m2[ni], m3[ni], sp);
        m2[ni] = __builtin_amdgcn_fmed3f(m1[ni], m2[ni], sp);
        asm("v_max_f32 %0, %1, %2" : "=v"(m1[ni]) : "v"(m1[ni]), "v"(sp));
    };
    auto fold_tile = [&](auto ragged_c, uint32_t ts, uint32_t tile_row0) __attribute__((always_inline)) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
        // MODE 1, per query column (as mfma_prefilter256_kernel): read here, once per tile, rather than held in 16
        // registers through the K loop
        float cq[NI], ce[NI];
        if (MODE == 1) {
#pragma unroll
            for (uint32_t ni = 0; ni < NI; ++ni) {
                const uint32_t q = qtile * TQ + ni * 32u + (lane & 31u);
                cq[ni] = ce[ni] = 0.0f;
                if (q < nq) {
                    const float a2 = 2.0f * q_norm[q];
                    cq[ni] = I8 ? a2 * q_scale[q] : a2;
                    ce[ni] = I8 ? a2 * q_eps[q] : q_eps[q];
                }
            }
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const uint32_t row4 = tile_row0 + wave_m * 128u + mi * 32u + 8u * (uint32_t)rq + 4u * (lane >> 5);
            float bn[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) {
                const float4 p0 = *(const float4 *)(rn2 + row4);
                const float4 p1 = *(const float4 *)(rn2 + row4 + 2u);
                bn[0] = p0.x, bs[0] = p0.y, bn[1] = p0.z, bs[1] = p0.w;
                bn[2] = p1.x, bs[2] = p1.y, bn[3] = p1.z, bs[3] = p1.w;
            }
            float rf[4] = {0.f, 0.f, 0.f, 0.f};
            if (I8) {
                const float4 p = *(const float4 *)(inv_rn + row4);
                rf[0] = p.x, rf[1] = p.y, rf[2] = p.z, rf[3] = p.w;
                if (MODE == 1) {
#pragma unroll
                    for (int rl = 0; rl < 4; ++rl)
                        rf[rl] *= bn[rl];
                }
            }
#pragma unroll
            for (int rl = 0; rl < 4; ++rl) {
                const int r = rq * 4 + rl;
                // the index is wave-uniform (an SGPR): mask (VGPR) + index is ONE v_and_or_b32 per element.  A row beyond
                // the corpus (only in the last, ragged tile) scores -inf: its mask keeps nothing of the score and its
                // "index" is the bit pattern of -inf.
                const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((ts << 6) | (mi << 4) | (uint32_t)r));
                const bool valid = !RAGGED || row4 + (uint32_t)rl < n_rows;
                const uint32_t km = RAGGED ? (valid ? keep_mask : 0u) : keep_mask;
                const uint32_t ix = RAGGED ? (valid ? idx : 0xff800000u) : idx;
#pragma unroll
                for (uint32_t ni = 0; ni < NI; ++ni) {
                    float v = (float)acc[ni][r];
                    if (I8 && MODE == 1)
                        v = __builtin_fmaf(v * rf[rl], cq[ni], __builtin_fmaf(bn[rl], ce[ni], -bs[rl]));
                    else if (I8)
                        v = v * rf[rl];
                    else if (MODE == 1)
                        v = cq[ni] * (bn[rl] * (v + ce[ni])) - bs[rl];
                    insert3(ni, __uint_as_float((__float_as_uint(v) & km) | ix));
                    acc[ni][r] = 0;
                }
            }
        }
    };

    // Registers.  Row fragments: a ring of three sets (set = K-step % 3), loaded TWO K-steps ahead -- they stream from
    // HBM / the Infinity Cache (each row tile is new to this XCD's L2 for the first workgroup that touches it), and a
    // timing experiment puts them at 19 % of the kernel when they only have one K-step to arrive, against 6 % for the
    // L2-resident query pieces (profiles/r03/c4_analysis.md).  Query fragments: a ring of three sets of 2 (one k-slice
    // of two query groups), read two groups ahead of their MFMAs; 16 groups per K-step and 16 = 1 (mod 3), so group g
    // of K-step `it` uses set (it + g) % 3: both rings turn with PH = it % 3 and everything below is static per PH.
    u32x4 fa[3][4];
    u32x4 fb[2][2];
    uint32_t rd_cur = ring_base + lane16, rd_slot = 0;
#define VFSX_RV_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // the 2 query fragments of group g = 4 ks + qq: query groups 2 qq, 2 qq + 1 at k-slice ks (LDS unit ni * KU + ks)
#define VFSX_RV_READ_PAIR(set, addr, g)                                                                            \
    do {                                                                                                           \
        VFSX_RV_DSR(fb[set][0], addr, ((2 * ((g) % 4) + 0) * 4 + (g) / 4) * 1024);                                 \
        VFSX_RV_DSR(fb[set][1], addr, ((2 * ((g) % 4) + 1) * 4 + (g) / 4) * 1024);                                 \
    } while (0)
#if defined(VFSX_EXP_RV) // timing experiments (WRONG results): 2 = no row loads in the loop, 3 = no query pieces
#if VFSX_EXP_RV == 2
#undef VFSX_RV_ALOAD
#define VFSX_RV_ALOAD(dst, rsrc, soff) asm volatile("" : "+v"(dst) : "s"(rsrc), "s"(soff))
#endif
#endif

#ifdef VFSX_EXP_STAMPS
    uint64_t st_vm = 0, st_bar = 0, st_body = 0, st_fold = 0, st_n = 0;
    const uint64_t st_begin = __builtin_amdgcn_s_memtime(), st_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
    auto kstep = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        VFSX_STAMP(t0);
        // all of this wave's operations but those of the previous K-step (4 row loads + 4 pieces) have completed:
        // the row fragments of this K-step and this wave's pieces of the stages <= it+1
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(fa[PH][0]), "+v"(fa[PH][1]), "+v"(fa[PH][2]), "+v"(fa[PH][3])::"memory");
        VFSX_STAMP(t1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        VFSX_STAMP(t2);
        const u32x4 ra = a_rsrc();
        const uint32_t ak = a_koff;
        a_advance();
        const uint32_t rd_next = ring_base + ((rd_slot + 1u) & (NSLOT - 1u)) * SLOT_BYTES + lane16;
        // group g: the reads of group g + 2, one VMEM operation every other group (the row fragments of K-step it+2
        // first, then this wave's pieces of stage it+3), the wait for this group's own reads, 2 MFMAs
#define VFSX_RV_GROUP(g, VMEM)                                                                                      \
    do {                                                                                                           \
        constexpr int S_ = (g) & 1, N_ = ((g) + 1) & 1;                                                             \
        if ((g) + 1 < 16)                                                                                          \
            VFSX_RV_READ_PAIR(N_, rd_cur, ((g) + 1) % 16);                                                         \
        else                                                                                                       \
            VFSX_RV_READ_PAIR(N_, rd_next, ((g) + 1) % 16);                                                        \
        VMEM;                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[S_][0]), "+v"(fb[S_][1])::"memory");                          \
        acc[2 * ((g) % 4) + 0] = mfma_32x32(__builtin_bit_cast(uint4, fa[PH][(g) / 4]),                             \
                                            __builtin_bit_cast(uint4, fb[S_][0]), acc[2 * ((g) % 4) + 0]);          \
        acc[2 * ((g) % 4) + 1] = mfma_32x32(__builtin_bit_cast(uint4, fa[PH][(g) / 4]),                             \
                                            __builtin_bit_cast(uint4, fb[S_][1]), acc[2 * ((g) % 4) + 1]);          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
        constexpr int AN = (PH + 2) % 3;
        VFSX_RV_GROUP(0, VFSX_RV_ALOAD(fa[AN][0], ra, ak + 0u * a_piece));
        VFSX_RV_GROUP(1, (void)0);
        VFSX_RV_GROUP(2, VFSX_RV_ALOAD(fa[AN][1], ra, ak + 1u * a_piece));
        VFSX_RV_GROUP(3, (void)0);
        VFSX_RV_GROUP(4, VFSX_RV_ALOAD(fa[AN][2], ra, ak + 2u * a_piece));
        VFSX_RV_GROUP(5, (void)0);
        VFSX_RV_GROUP(6, VFSX_RV_ALOAD(fa[AN][3], ra, ak + 3u * a_piece));
        VFSX_RV_GROUP(7, (void)0);
#if defined(VFSX_EXP_RV) && VFSX_EXP_RV == 3
#define b_piece(u) ((void)0)
#endif
        VFSX_RV_GROUP(8, b_piece(0));
        VFSX_RV_GROUP(9, (void)0);
        VFSX_RV_GROUP(10, b_piece(1));
        VFSX_RV_GROUP(11, (void)0);
        VFSX_RV_GROUP(12, b_piece(2));
        VFSX_RV_GROUP(13, (void)0);
        VFSX_RV_GROUP(14, b_piece(3));
        VFSX_RV_GROUP(15, (void)0);
#if defined(VFSX_EXP_RV) && VFSX_EXP_RV == 3
#undef b_piece
#endif
#undef VFSX_RV_GROUP
        VFSX_STAMP(t3);
#ifdef VFSX_EXP_STAMPS
        st_vm += t1 - t0, st_bar += t2 - t1, st_body += t3 - t2, st_n += 1;
#endif
        b_advance();
        rd_slot = (rd_slot + 1u) & (NSLOT - 1u);
        rd_cur = rd_next;
    };

    // prologue: stages 0 and 1, the row fragments of K-steps 0 and 1, stage 2 -- the last eight outstanding operations
    // are then 4 row loads + 4 pieces, as in the steady state
#pragma unroll
    for (uint32_t u = 0; u < KU; ++u)
        b_piece(u);
    b_advance();
#pragma unroll
    for (uint32_t u = 0; u < KU; ++u)
        b_piece(u);
    b_advance();
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const u32x4 ra = a_rsrc();
        const uint32_t ak = a_koff;
        a_advance();
        VFSX_RV_ALOAD(fa[st][0], ra, ak + 0u * a_piece);
        VFSX_RV_ALOAD(fa[st][1], ra, ak + 1u * a_piece);
        VFSX_RV_ALOAD(fa[st][2], ra, ak + 2u * a_piece);
        VFSX_RV_ALOAD(fa[st][3], ra, ak + 3u * a_piece);
    }
#pragma unroll
    for (uint32_t u = 0; u < KU; ++u)
        b_piece(u);
    b_advance();
    // stage 0 visible to everybody -> the query fragments of the first two groups
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    VFSX_RV_READ_PAIR(0, rd_cur, 0);

    // Tile by tile: within a tile the rings turn with kt % 3 (static); a tile of KT K-steps leaves them turned by
    // KT % 3, and behind the fold -- ~4 us of VALU work in which everything in flight lands anyway -- the two sets that
    // hold the next K-steps' row fragments are moved back to sets 0 and 1, so that every tile starts at phase 0 and the
    // fold exists at ONE site (three sites made hipcc spill 150+ registers into the loop).
    const uint32_t k_rem = KT % 3u;
    for (uint32_t ts = 0; ts < ntiles; ++ts) {
        uint32_t kt = 0;
        for (; kt + 3u <= KT; kt += 3u) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            kstep(std::integral_constant<int, 2>{});
        }
#ifndef VFSX_RV_T1
        if (k_rem >= 1u)
            kstep(std::integral_constant<int, 0>{});
        if (k_rem == 2u)
            kstep(std::integral_constant<int, 1>{});
#endif
        VFSX_STAMP(tf);
        const uint32_t tile_row0 = (tile0 + ts) * TROWS;
        if (tile_row0 + TROWS > n_rows)
            fold_tile(std::true_type{}, ts, tile_row0); // last, ragged tile: rows >= n_rows masked
        else
            fold_tile(std::false_type{}, ts, tile_row0);
        if (k_rem != 0u) {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]),
                           "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[2][2]), "+v"(fa[2][3])::"memory");
            if (k_rem == 1u) { // next K-step's fragments sit in set 1, the one after in set 2
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fa[0][u] = fa[1][u];
                    fa[1][u] = fa[2][u];
                }
            } else { // in set 2 and set 0
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fa[1][u] = fa[0][u];
                    fa[0][u] = fa[2][u];
                }
            }
        }
#ifdef VFSX_EXP_STAMPS
        st_fold += __builtin_amdgcn_s_memtime() - tf;
#endif
    }
#undef VFSX_RV_READ_PAIR
#undef VFSX_RV_DSR
#undef VFSX_RV_ALOAD

    // the redundant tail loads / stages and the last look-ahead reads must land before the ring is reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]),
                   "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[2][2]), "+v"(fa[2][3]),
                   "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])::"memory");
    __syncthreads();
#ifdef VFSX_EXP_STAMPS
    if (lane == 0) {
        atomicAdd(&exp_stamps[0], (unsigned long long)st_vm);
        atomicAdd(&exp_stamps[1], (unsigned long long)st_bar);
        atomicAdd(&exp_stamps[2], (unsigned long long)st_body);
        atomicAdd(&exp_stamps[3], (unsigned long long)st_fold);
        atomicAdd(&exp_stamps[4], (unsigned long long)st_n);
        atomicAdd(&exp_stamps[5], (unsigned long long)(__builtin_amdgcn_s_memtime() - st_begin));
        atomicAdd(&exp_stamps[6], 1ull);
        atomicAdd(&exp_stamps[7], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - st_rbegin));
    }
#endif

    // merge the four waves of one wave_m: [wave][ni][3][64] floats in the (now idle) ring
    float *xch = (float *)ring;
#pragma unroll
    for (uint32_t ni = 0; ni < NI; ++ni) {
        xch[((wave * NI + ni) * 3u + 0u) * 64u + lane] = m1[ni];
        xch[((wave * NI + ni) * 3u + 1u) * 64u + lane] = m2[ni];
        xch[((wave * NI + ni) * 3u + 2u) * 64u + lane] = m3[ni];
    }
    __syncthreads();
    if (mi == 0u) {
        for (uint32_t w = wave + 1u; w < wave + 4u; ++w)
#pragma unroll
            for (uint32_t ni = 0; ni < NI; ++ni)
#pragma unroll
                for (uint32_t j = 0; j < 3u; ++j)
                    insert3(ni, xch[((w * NI + ni) * 3u + j) * 64u + lane]);
        const uint32_t list = strip * 4u + wave_m * 2u + (lane >> 5);
#pragma unroll
        for (uint32_t ni = 0; ni < NI; ++ni) {
            const uint32_t q = qtile * TQ + ni * 32u + (lane & 31u);
            if (q < nq) {
                cand[((size_t)q * kListDepth + 0u) * n_lists + list] = m1[ni];
                cand[((size_t)q * kListDepth + 1u) * n_lists + list] = m2[ni];
                cand[((size_t)q * kListDepth + 2u) * n_lists + list] = m3[ni];
            }
        }
    }
}

// ---- prefilter for small batches (<= 64 queries): 256 rows x 64 queries per workgroup ---------------------
// One partly filled 256-query tile makes mfma_prefilter256_kernel latency-bound: a single workgroup per CU
// with one 64 KiB stage in flight moves the mirror at ~3.7 TB/s, and 3/4 of its MFMAs multiply zero columns.
// Here a stage is 32 KiB of rows + 8 KiB of queries (BK = 64) in a ring of NSTAGE slots with NSTAGE-1 stages
// in flight, each of the 8 waves multiplies its own 32 rows by the 64 queries (2 MFMAs per k-slice), and the
// kernel runs at the mirror's HBM rate.  Output: the candidate lists of mfma_prefilter256_kernel, bit for bit
// the same format (wave w covers what (wave_m, mi) = (w >> 2, w & 3) covers there; the four waves of one
// wave_m merge their top-3 through LDS at the end), so select / re-score are shared.
// I8: as in mfma_prefilter256_kernel -- the row pieces come straight out of the rb64 corpus (a_chunks chunks per
// row, a_bytes allocated), the MFMA is v_mfma_i32_32x32x32_i8, the score S / (scale ||x||).
// NT: the row pieces are staged with the nontemporal policy -- every row is read once per batch here, and like the
// exact scan the stream runs faster past the caches (1M x 768 int8, 8 queries: 0.259 -> 0.229 ms per batch; sc0 alone
// changes nothing).  The 256-query kernel re-reads rows across query tiles and keeps the default policy.
template <int NSTAGE, int MODE, int NCOL, bool I8, bool NT>
__global__ __launch_bounds__(512) void mfma_prefilter_skinny_kernel(const uint4 *__restrict__ A,
                                                                    const uint4 *__restrict__ B,
                                                                    float *__restrict__ cand, uint32_t n_rows,
                                                                    uint32_t n_rowtiles, uint32_t KS, uint32_t nq,
                                                                    uint32_t n_lists, uint32_t strip_tiles,
                                                                    const float2 *__restrict__ rn2,
                                                                    const float *__restrict__ q_norm,
                                                                    const float *__restrict__ q_eps, uint32_t a_chunks,
                                                                    uint64_t a_bytes, const float *__restrict__ inv_rn,
                                                                    const float *__restrict__ q_scale)
{
    constexpr uint32_t KU = 4u, TROWS = 256u;
    // NCOL = 64-query columns per workgroup (1: <= 64 queries, 2: <= 128): 32 row units + 8 NCOL query units of
    // 1 KiB per stage; every wave multiplies its 32 rows by all 64 NCOL queries
    constexpr uint32_t A_UNITS = 8u * KU, UNITS = (8u + 2u * NCOL) * KU;
    constexpr int NI = 2 * NCOL;
    static_assert(NSTAGE >= 3 && NSTAGE <= 4, "ring depth"); // 3 and 4 measure the same (0.48 ms at 1M x 768)
    __shared__ __align__(16) uint4 ring[NSTAGE * UNITS * 64];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // SGPR
    const uint32_t wave_m = wave >> 2, mi = wave & 3u;
    const uint32_t strip = blockIdx.x;
    const bool own_lists = (strip_tiles & kStripOwnLists) != 0u; // 16 lists per strip: no merge of the four waves
    strip_tiles &= 0xffu;
    const uint32_t tile0 = strip * strip_tiles;
    const uint32_t ntiles = min(strip_tiles, n_rowtiles - tile0);
    const uint32_t KT = KS / KU;
    const uint32_t n_it = ntiles * KT;

    // staging: every wave moves 4 row pieces (its own 32-row group, the 4 k-slices) and 1 query piece per stage
    const uint32_t lane16 = lane * 16u;
    // rb64 source (I8): 32-row group `wave` = row-block wave >> 1, rows 32 (wave & 1) ..; k-slice = a chunk pair
    const uint32_t lane16a = I8 ? (lane >> 5) * 1024u + (lane & 31u) * 16u : lane16;
    const uint32_t tile_bytes = I8 ? 4u * a_chunks * 1024u : 8u * KS * 1024u; // one 256-row tile of the rows
    const uint32_t a_piece = I8 ? 2048u : 1024u;                              // one k-slice inside a row group
    const uint32_t a_soff = I8 ? (wave >> 1) * a_chunks * 1024u + (wave & 1u) * 512u : wave * KS * 1024u;
    const uint32_t a_lds = wave * KU * 1024u;
    const uintptr_t a_end = (uintptr_t)A + a_bytes;
    // query pieces: wave w moves (32-query group w >> 2 [+ 2 for the second column], k-slice w & 3)
    const uint32_t b_soff = ((wave >> 2) * KS + (wave & 3u)) * 1024u, b_lds = (A_UNITS + wave) * 1024u;
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)ring;
    uintptr_t nx_base = (uintptr_t)A + (uintptr_t)tile0 * tile_bytes;
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)B, 0, 0xffffffff, 0x00020000);
    uint32_t nx_idx = 0, nx_koff = 0, nx_slot = 0; // nx_koff: K offset in the query mirror (rows: x a_piece / 1024)
    auto stage = [&]() { // issue the next stage into the next ring slot (past the end: re-issue the last one)
        uint32_t bound = 0xffffffffu; // rb64 source: bounded by the allocation (K padding / missing row-blocks read 0)
        if (I8) {
            const uintptr_t left = nx_base < a_end ? a_end - nx_base : 0;
            bound = left < 0xffffffffu ? (uint32_t)left : 0xffffffffu;
        }
        const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)nx_base, 0, bound, 0x00020000);
        const uint32_t slot_base = ring_base + nx_slot * (UNITS * 1024u);
        const uint32_t a_koff = I8 ? nx_koff * 2u : nx_koff;
#pragma unroll
        for (uint32_t u = 0; u < KU; ++u) {
            uint32_t a = slot_base + a_lds + u * 1024u;
            asm volatile("" : "+s"(a)); // LDS destination laundered: see mfma_prefilter256_kernel
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void *)(uintptr_t)a, 16,
                                                     (int)lane16a, (int)(a_soff + u * a_piece + a_koff), 0,
                                                     NT ? 2 : 0); // (aux bit 1 = nt)
        }
#pragma unroll
        for (uint32_t col = 0; col < (uint32_t)NCOL; ++col) {
            uint32_t b = slot_base + b_lds + col * (8u * 1024u);
            asm volatile("" : "+s"(b));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void *)(uintptr_t)b, 16,
                                                     (int)lane16, (int)(b_soff + col * (2u * KS * 1024u) + nx_koff), 0, 0);
        }
        nx_slot = nx_slot + 1u == (uint32_t)NSTAGE ? 0u : nx_slot + 1u;
        if (nx_idx + 1u < n_it) {
            ++nx_idx;
            nx_koff += KU * 1024u;
            if (nx_koff == KS * 1024u) {
                nx_koff = 0;
                nx_base += tile_bytes;
            }
        }
    };

    typename std::conditional<I8, i32x16, f32x16>::type acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[ni][r] = 0;
    const float ninf = __uint_as_float(0xff800000u);
    float m1[NI], m2[NI], m3[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
        m1[ni] = m2[ni] = m3[ni] = ninf;
    uint32_t keep_mask = ~kIdxMask;
    asm volatile("" : "+v"(keep_mask));
    float cq[NI], ce[NI]; // MODE 1: 2a and eps_q per query column
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
        cq[ni] = ce[ni] = 0.0f;
    if (MODE == 1) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const uint32_t q = (uint32_t)ni * 32u + (lane & 31u);
            if (q < nq) { // as in mfma_prefilter256_kernel
                const float a2 = 2.0f * q_norm[q];
                cq[ni] = I8 ? a2 * q_scale[q] : a2;
                ce[ni] = I8 ? a2 * q_eps[q] : q_eps[q];
            }
        }
    }
    auto insert3 = [&](int ni, float sp) __attribute__((always_inline)) {
        m3[ni] = __builtin_amdgcn_fmed3f(m2[ni], m3[ni], sp);
        m2[ni] = __builtin_amdgcn_fmed3f(m1[ni], m2[ni], sp);
        asm("v_max_f32 %0, %1, %2" : "=v"(m1[ni]) : "v"(m1[ni]), "v"(sp));
    };
    auto fold_tile = [&](uint32_t ts, uint32_t tile_row0) __attribute__((always_inline)) {
        const bool ragged = tile_row0 + TROWS > n_rows;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const uint32_t row4 = tile_row0 + wave_m * 128u + mi * 32u + 8u * (uint32_t)rq + 4u * (lane >> 5);
            float bn[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) {
                const float4 p0 = *(const float4 *)(rn2 + row4);
                const float4 p1 = *(const float4 *)(rn2 + row4 + 2u);
                bn[0] = p0.x, bs[0] = p0.y, bn[1] = p0.z, bs[1] = p0.w;
                bn[2] = p1.x, bs[2] = p1.y, bn[3] = p1.z, bs[3] = p1.w;
            }
            float rf[4] = {0.f, 0.f, 0.f, 0.f}; // as in mfma_prefilter256_kernel
            if (I8) {
                const float4 p = *(const float4 *)(inv_rn + row4);
                rf[0] = p.x, rf[1] = p.y, rf[2] = p.z, rf[3] = p.w;
                if (MODE == 1) {
#pragma unroll
                    for (int rl = 0; rl < 4; ++rl)
                        rf[rl] *= bn[rl];
                }
            }
#pragma unroll
            for (int rl = 0; rl < 4; ++rl) {
                const int r = rq * 4 + rl;
                const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((ts << 6) | (mi << 4) | (uint32_t)r));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    float v = (float)acc[ni][r];
                    if (I8 && MODE == 1)
                        v = __builtin_fmaf(v * rf[rl], cq[ni], __builtin_fmaf(bn[rl], ce[ni], -bs[rl]));
                    else if (I8)
                        v = v * rf[rl];
                    else if (MODE == 1)
                        v = cq[ni] * (bn[rl] * (v + ce[ni])) - bs[rl];
                    float sp = __uint_as_float((__float_as_uint(v) & keep_mask) | idx);
                    if (ragged && row4 + (uint32_t)rl >= n_rows)
                        sp = ninf;
                    insert3(ni, sp);
                    acc[ni][r] = 0;
                }
            }
        }
    };

    for (int j = 0; j < NSTAGE - 1; ++j)
        stage();
    uint32_t cur_kt = 0, cur_ts = 0, rd_slot = 0;
    for (uint32_t it = 0; it < n_it; ++it) {
        // 4 + NCOL pieces per wave and stage; NSTAGE-2 younger stages stay in flight behind the one consumed now
        if ((4 + NCOL) * (NSTAGE - 2) == 5)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if ((4 + NCOL) * (NSTAGE - 2) == 6)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if ((4 + NCOL) * (NSTAGE - 2) == 10)
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(); // into the slot consumed in the previous step (every wave is past it: barrier above)
        // The 12 fragment reads of the K-step as asm: hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of
        // any ds_read it cannot prove independent of a pending LDS-DMA (here: the stage issued two lines up),
        // which would serialise every stage with its K-step.  The ring protocol above is what orders them.
        const uint32_t rd_a = ring_base + rd_slot * (UNITS * 1024u) + wave * (KU * 1024u) + lane16;
        const uint32_t rd_b = ring_base + rd_slot * (UNITS * 1024u) + A_UNITS * 1024u + lane16;
        rd_slot = rd_slot + 1u == (uint32_t)NSTAGE ? 0u : rd_slot + 1u;
        u32x4 fa[4], fb[4 * NI];
#define VFSX_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
        VFSX_DSR(fa[0], rd_a, 0);
        VFSX_DSR(fa[1], rd_a, 1024);
        VFSX_DSR(fa[2], rd_a, 2048);
        VFSX_DSR(fa[3], rd_a, 3072);
        VFSX_DSR(fb[0], rd_b, 0);
        VFSX_DSR(fb[1], rd_b, 1024);
        VFSX_DSR(fb[2], rd_b, 2048);
        VFSX_DSR(fb[3], rd_b, 3072);
        VFSX_DSR(fb[4], rd_b, 4096);
        VFSX_DSR(fb[5], rd_b, 5120);
        VFSX_DSR(fb[6], rd_b, 6144);
        VFSX_DSR(fb[7], rd_b, 7168);
        if (NCOL == 2) {
            VFSX_DSR(fb[4 * NI - 8], rd_b, 8192);
            VFSX_DSR(fb[4 * NI - 7], rd_b, 9216);
            VFSX_DSR(fb[4 * NI - 6], rd_b, 10240);
            VFSX_DSR(fb[4 * NI - 5], rd_b, 11264);
            VFSX_DSR(fb[4 * NI - 4], rd_b, 12288);
            VFSX_DSR(fb[4 * NI - 3], rd_b, 13312);
            VFSX_DSR(fb[4 * NI - 2], rd_b, 14336);
            VFSX_DSR(fb[4 * NI - 1], rd_b, 15360);
        }
#undef VFSX_DSR
        // fb[ni * 4 + ks] = query group ni, k-slice ks (LDS unit A_UNITS + ni * KU + ks)
        if (NCOL == 2)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]),
                           "+v"(fb[3]), "+v"(fb[4]), "+v"(fb[5]), "+v"(fb[6]), "+v"(fb[7]), "+v"(fb[4 * NI - 8]),
                           "+v"(fb[4 * NI - 7]), "+v"(fb[4 * NI - 6]), "+v"(fb[4 * NI - 5]), "+v"(fb[4 * NI - 4]),
                           "+v"(fb[4 * NI - 3]), "+v"(fb[4 * NI - 2]), "+v"(fb[4 * NI - 1])
                         :
                         : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]),
                           "+v"(fb[3]), "+v"(fb[4]), "+v"(fb[5]), "+v"(fb[6]), "+v"(fb[7])
                         :
                         : "memory");
#pragma unroll
        for (uint32_t ks = 0; ks < KU; ++ks) {
#pragma unroll
            for (uint32_t ni = 0; ni < (uint32_t)NI; ++ni)
                acc[ni] = mfma_32x32(__builtin_bit_cast(uint4, fa[ks]), __builtin_bit_cast(uint4, fb[ni * 4u + ks]),
                                     acc[ni]);
        }
        if (++cur_kt == KT) {
            const uint32_t ts = cur_ts;
            cur_kt = 0;
            ++cur_ts;
            fold_tile(ts, (tile0 + ts) * TROWS);
        }
    }
    // the redundant tail stages must land before the ring is reused / released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (own_lists) { // long strips: list = (strip, wave, 16-row half), 16 rows per tile each -- nothing to merge
        const uint32_t list = strip * 16u + wave * 2u + (lane >> 5);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const uint32_t q = (uint32_t)ni * 32u + (lane & 31u);
            if (q < nq) {
                cand[((size_t)q * kListDepth + 0u) * n_lists + list] = m1[ni];
                cand[((size_t)q * kListDepth + 1u) * n_lists + list] = m2[ni];
                cand[((size_t)q * kListDepth + 2u) * n_lists + list] = m3[ni];
            }
        }
        return;
    }
    // merge the four waves of one wave_m: [wave][ni][3][64] floats in the (now idle) ring
    float *xch = (float *)ring;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        xch[((wave * NI + ni) * 3u + 0u) * 64u + lane] = m1[ni];
        xch[((wave * NI + ni) * 3u + 1u) * 64u + lane] = m2[ni];
        xch[((wave * NI + ni) * 3u + 2u) * 64u + lane] = m3[ni];
    }
    __syncthreads();
    if (mi == 0u) {
        for (uint32_t w = wave + 1u; w < wave + 4u; ++w)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (uint32_t j = 0; j < 3u; ++j)
                    insert3(ni, xch[((w * NI + ni) * 3u + j) * 64u + lane]);
        const uint32_t list = strip * 4u + wave_m * 2u + (lane >> 5);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const uint32_t q = (uint32_t)ni * 32u + (lane & 31u);
            if (q < nq) {
                cand[((size_t)q * kListDepth + 0u) * n_lists + list] = m1[ni];
                cand[((size_t)q * kListDepth + 1u) * n_lists + list] = m2[ni];
                cand[((size_t)q * kListDepth + 2u) * n_lists + list] = m3[ni];
            }
        }
    }
}

// ---- select: kp best approximate scores of one query ------------------------------------------
// composite = key(-score) << 32 | list ; smaller = better.  One workgroup per query.
template <typename TopK, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mfma_select_kernel(const float *__restrict__ cand,
                                                                 uint64_t *__restrict__ sel, uint32_t n_lists,
                                                                 uint32_t kp, const int *__restrict__ only_flagged)
{
    if (only_flagged && only_flagged[blockIdx.x] == 0) // second, wider pass: flagged queries only
        return;
    constexpr uint32_t BLOCK = 64u * WAVES;
    extern __shared__ __align__(16) uint64_t dyn_lds[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x;
    const uint32_t list_stride = kp > 64u ? kp : 64u;
    const float *__restrict__ src = cand + (size_t)q * kListDepth * n_lists;
    const float ninf = __uint_as_float(0xff800000u);

    TopK top;
    top.init(kp, dyn_lds + wave * list_stride);
    // `bound`: an upper bound of the kp-th best key of all lists from the heads of the first 64 * WAVES lists, as in
    // select_kernel (vfsx_kernels.hip) -- the second and third entries of the lists, and every later head, are
    // dropped against it before they are sorted
    __shared__ unsigned long long s_bound;
    uint64_t bound = kSentinel;
    uint32_t lbase = wave * 64u;
    auto load3 = [&](uint32_t list, uint64_t(&comp)[kListDepth]) {
        const bool in = list < n_lists;
        float sc[kListDepth];
#pragma unroll
        for (uint32_t j = 0; j < kListDepth; ++j)
            sc[j] = in ? src[(size_t)j * n_lists + list] : ninf;
#pragma unroll
        for (uint32_t j = 0; j < kListDepth; ++j)
            comp[j] = (sc[j] == ninf) ? kSentinel : (((uint64_t)dist_key(-sc[j]) << 32) | list);
    };
    if (TopK::kInRegisters) { // (compile-time: every wave takes the barriers)
        if (threadIdx.x == 0)
            s_bound = 0ull;
        __syncthreads();
        const uint32_t full = min(n_lists / 64u, (uint32_t)WAVES), m = full ? (kp + full - 1u) / full : 65u;
        uint64_t comp[kListDepth];
        load3(lbase + lane, comp);
        top.offer_dense(comp[0]);
        if (m <= 64u && wave < full && lane == 0u)
            atomicMax(&s_bound, (unsigned long long)top.rank(m - 1u));
        __syncthreads();
        if (m <= 64u)
            bound = s_bound;
#pragma unroll
        for (uint32_t j = 1; j < kListDepth; ++j)
            top.offer_dense(comp[j] < bound ? comp[j] : kSentinel);
        lbase += BLOCK;
    }
    for (; lbase < n_lists; lbase += BLOCK) {
        uint64_t comp[kListDepth];
        load3(lbase + lane, comp);
#pragma unroll
        for (uint32_t j = 0; j < kListDepth; ++j)
            top.offer_dense(comp[j] < bound ? comp[j] : kSentinel);
    }
    uint64_t *mine = dyn_lds + wave * list_stride;
    __syncthreads();
    top.spill(mine);
    __syncthreads();
    if (TopK::kInRegisters) { // merge tree over the waves' sorted register lists (as in select_kernel)
        for (uint32_t s = 1; s < (uint32_t)WAVES; s <<= 1) {
            if ((wave & (2u * s - 1u)) == 0u && wave + s < (uint32_t)WAVES &&
                (wave + s) * 64u < n_lists) { // (a partner wave without lists has nothing to merge)
                top.merge_list(dyn_lds + (wave + s) * list_stride);
                top.spill(mine);
            }
            __syncthreads();
        }
    }
    if (wave == 0) {
        bool alive = !TopK::kInRegisters && lane >= 1 && lane < (uint32_t)WAVES;
        const uint64_t *other = dyn_lds + (alive ? lane : 0u) * list_stride;
        for (uint32_t j = 0; !TopK::kInRegisters && j < kp; ++j) {
            uint64_t c = alive ? other[j] : kSentinel;
            if (__ballot(c < top.tau) == 0)
                break;
            top.offer(c);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        top.store(sel + (size_t)q * kp);
    }
}

// ---- second pass: every candidate above the safety margin, unsorted ----------------------------------------
// A query that failed its first check because more than kp rows sit above the margin is decided again from a wider
// candidate set.  What that set has to hold is every listed row whose score reaches the margin -- not the 256 BEST
// scores in order (a 256-deep sorted select of ~16 000 candidates by ONE workgroup took 0.8 ms per flagged query:
// a third of a 1024-query batch with one such query, 1.4 ms of every C4 batch).  The margin is known: the first
// pass' k-th exact distance dk1 gives thr1, the wider pass can only find a smaller k-th distance, i.e. a higher
// margin thr2 >= thr1, so {score >= thr1} contains everything the second check can call "above".  One workgroup per
// flagged query filters the 3 n_lists scores against thr1 (the re-score kernels' own expressions) and compacts the
// survivors into sel[q][0..256) in no particular order -- the exact re-score ranks by exact distance, and the list
// rule counts entries per list: neither depends on the order.  More than 256 survivors: all slots are cleared, the
// re-score finds fewer than k hits and leaves the query flagged for the exact scan.
__global__ __launch_bounds__(256) void mfma_filter_kernel(const float *__restrict__ cand, uint64_t *__restrict__ sel,
                                                          uint32_t n_lists, const int *__restrict__ flags,
                                                          const int *__restrict__ q_bad, const float *__restrict__ dk1,
                                                          const float *__restrict__ q_eps,
                                                          const float *__restrict__ q_scale,
                                                          const float *__restrict__ q_norm, float gamma, int euclid)
{
    const uint32_t q = blockIdx.x;
    if (flags[q] == 0)
        return;
    __shared__ uint32_t s_n;
    uint64_t *__restrict__ out = sel + (size_t)q * 256u;
    out[threadIdx.x] = kSentinel;
    if (threadIdx.x == 0)
        s_n = 0;
    __syncthreads();
    if (q_bad[q]) // the path never ran for this query (zero / out-of-range norm): nothing to re-decide
        return;
    const float *__restrict__ src = cand + (size_t)q * kListDepth * n_lists;
    const float ninf = __uint_as_float(0xff800000u);
    const float dk = dk1[q];
    const float thr = (1.0f - dk) - (q_eps[q] + 0x1p-13f);
    const float qs = q_scale ? q_scale[q] : 1.0f;
    const double a = (double)q_norm[q], g = (double)gamma;
    const double vk_low = a * a * (1.0 - g) - (double)dk * (double)dk * (1.0 + g);
    const uint32_t total = kListDepth * n_lists, lane = threadIdx.x & 63u;
    for (uint32_t base = 0; base < total; base += 256u) { // (uniform trip count: the barrier-free loop has ballots)
        const uint32_t i = base + threadIdx.x;
        bool pass = false;
        uint64_t comp = kSentinel;
        if (i < total) {
            const float sc = src[i];
            if (sc != ninf) {
                comp = ((uint64_t)dist_key(-sc) << 32) | (i % n_lists);
                const float score = -key_dist((uint32_t)(comp >> 32)); // exactly what cand_row hands the re-score
                if (euclid) {
                    const double s2 = (double)score;
                    pass = s2 + 0x1p-12 * (s2 < 0.0 ? -s2 : s2) >= vk_low;
                } else {
                    pass = (q_scale ? qs * score : score) >= thr;
                }
            }
        }
        const uint64_t mask = __ballot(pass);
        if (mask) {
            uint32_t pos0 = 0;
            if (lane == 0)
                pos0 = atomicAdd(&s_n, (uint32_t)__popcll(mask));
            pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos0);
            const uint32_t my = pos0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (pass && my < 256u)
                out[my] = comp;
        }
    }
    __syncthreads();
    if (s_n > 256u) // does not fit: leave nothing, the re-score then keeps the query flagged
        out[threadIdx.x] = kSentinel;
}

// ---- exact re-score + safety check ----------------------------------------------------------------
// A candidate = (approximate score with the row position in its 10 low mantissa bits, list): the inverse of the
// prefilter kernels' list / index assignment.
__device__ __forceinline__ uint32_t cand_row(uint64_t comp, float &score, uint32_t strip_tiles)
{
    const uint32_t list = (uint32_t)comp;
    score = -key_dist((uint32_t)(comp >> 32));
    const uint32_t idx = __float_as_uint(score) & kIdxMask;
    const uint32_t ts = idx >> 6, mi = (idx >> 4) & 3u, r = idx & 15u;
    // 4 lists per strip: (wave_m, half); 16 (kStripOwnLists): (wave = 4 wave_m + mi, half) -- mi is in the index either way
    const bool own = (strip_tiles & kStripOwnLists) != 0u;
    const uint32_t strip = own ? list >> 4 : list >> 2, wave_m = own ? (list >> 3) & 1u : (list >> 1) & 1u, half = list & 1u;
    return (strip * (strip_tiles & 0xffu) + ts) * 256u + wave_m * 128u + mi * 32u + (r & 3u) + 8u * (r >> 2) + 4u * half;
}

template <int DTYPE, int METRIC, typename TopK>
__global__ __launch_bounds__(kBlock) void mfma_rescore_kernel(
    const uint4 *__restrict__ corpus, const float *__restrict__ row_norms, const float *__restrict__ queries,
    const float *__restrict__ lut, const uint64_t *__restrict__ sel, const int *__restrict__ q_bad,
    const uint64_t *__restrict__ ids, vfsx_hit *__restrict__ out, int *__restrict__ flags, uint32_t n_rows,
    uint32_t d_chunks, uint32_t q_stride, uint32_t k, uint32_t kp, const float *__restrict__ q_eps, uint64_t row_base,
    uint64_t id_base, uint32_t strip_tiles, const float *__restrict__ q_norm, float gamma, int only_flagged,
    const float *__restrict__ q_scale, float *__restrict__ dk_out)
{
    if (only_flagged && flags[blockIdx.x] == 0) // second, wider pass: flagged queries only (uniform exit)
        return;
    constexpr uint32_t EPC = DTYPE == VFSX_I8 ? 16u : 4u;
    extern __shared__ __align__(16) uint64_t dyn_lds[]; // [4][max(k,64)]
    __shared__ float lut_s[DTYPE == VFSX_I8 ? 256 * 32 : 1];
    __shared__ uint32_t s_list[256];
    __shared__ float s_dk;
    __shared__ int s_flag;

    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x;
    const uint32_t list_stride = k > 64u ? 128u : 64u; // (WaveTopKReg2 publishes 128 entries)
    if (DTYPE == VFSX_I8) {
        for (uint32_t i = threadIdx.x; i < 256u * 32u; i += kBlock)
            lut_s[i] = lut[i >> 5];
    }
    if (threadIdx.x == 0)
        s_flag = q_bad[q];
    __syncthreads();
    const float *lut_lane = lut_s + (lane & 31u);
    const float *__restrict__ qv = queries + (size_t)q * q_stride;

    TopK top;
    top.init(k, dyn_lds + wave * list_stride);

    const uint32_t ci = wave * 64u + lane;
    const uint64_t comp_a = ci < kp ? sel[(size_t)q * kp + ci] : kSentinel;
    const bool valid = comp_a != kSentinel;
    float score = 0.0f;
    uint32_t row = 0;
    if (valid)
        row = cand_row(comp_a, score, strip_tiles);
    const bool in_range = valid && row < n_rows;
    if (wave * 64u < kp) { // wave-uniform: this wave has candidates
        const uint32_t r = in_range ? row : 0u;
        const uint4 *__restrict__ p = corpus + (size_t)(r >> 6) * d_chunks * 64u + (r & 63u);
        Acc acc[1] = {Acc{0.0f, 0.0f}};
        // a gather: 16 B per 1 KiB of the row's block, latency-bound -> 8 loads in flight per lane (batches of
        // <= 256 queries take mfma_rescore_coop_kernel below; here thousands of workgroups keep the memory busy)
        uint32_t c = 0;
        for (; c + 8u <= d_chunks; c += 8u) {
            uint4 v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u)
                v[u] = p[(size_t)(c + u) * 64u];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                if (DTYPE == VFSX_I8)
                    acc_chunk_i8<METRIC, true, 1>(acc, v[u], qv + (c + u) * EPC, 0, lut_lane);
                else
                    acc_chunk_f32<METRIC, true, 1>(acc, v[u], qv + (c + u) * EPC, 0);
            }
        }
        for (; c < d_chunks; ++c) {
            const uint4 v = p[(size_t)c * 64u];
            if (DTYPE == VFSX_I8)
                acc_chunk_i8<METRIC, true, 1>(acc, v, qv + c * EPC, 0, lut_lane);
            else
                acc_chunk_f32<METRIC, true, 1>(acc, v, qv + c * EPC, 0);
        }
        uint64_t comp = kSentinel;
        if (in_range) {
            float dist;
            if (METRIC == VFSX_EUCLIDEAN) {
                dist = f_sqrt(acc[0].a); // rank.rs:286
            } else {
                const float qn = f_sqrt(acc[0].qq);
                const float den = f_mul(qn, row_norms[row]);
                dist = f_sub(1.0f, f_div(acc[0].a, den)); // rank.rs:299
            }
            comp = ((uint64_t)dist_key(dist) << 32) | row;
        }
        top.offer_dense(comp); // 64 exact distances into an empty list
    }

    uint64_t *mine = dyn_lds + wave * list_stride;
    __syncthreads();
    top.spill(mine); // (kSentinel padded to 64 / 128 entries; a no-op for a list that lives in LDS)
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w)
            top.merge_list(dyn_lds + w * list_stride);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        top.store(mine);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        for (uint32_t j = lane; j < k; j += 64) {
            const uint64_t comp = mine[j];
            vfsx_hit h;
            if (comp == kSentinel) {
                h.dist = __uint_as_float(0x7fc00000u);
                h.flags = 0;
                h.row = ~0ull;
                h.id = ~0ull;
            } else {
                const uint32_t r = (uint32_t)comp;
                h.dist = key_dist((uint32_t)(comp >> 32));
                h.flags = 1;
                h.row = row_base + r;
                h.id = ids ? ids[r] : id_base + r;
            }
            out[(size_t)q * k + j] = h;
        }
        if (lane == 0) {
            const uint64_t kth = mine[k - 1];
            s_dk = kth == kSentinel ? __uint_as_float(0x7fc00000u) : key_dist((uint32_t)(kth >> 32));
        }
    }
    __syncthreads();

    // safety check.  A row r that was not re-scored can only belong to the exact top-k if
    // dist_r <= D_k, i.e. cos_ref(r) >= c_k = 1 - D_k, hence (|s~ - cos_ref| <= eps) s~_r >= c_k - eps, with
    // eps = eps_q + 2^-13: the 10 low mantissa bits of a score below 2 hold the row position.
    const float dk = s_dk;
    if (dk_out && threadIdx.x == 0) // the wider second pass filters the candidate lists against this k-th distance
        dk_out[q] = dk;
    bool bad = !(dk == dk) || !(dk < __uint_as_float(0x7f800000u)); // fewer than k exact hits / NaN
    bool above;
    if (METRIC == VFSX_EUCLIDEAN) {
        // A row's key' (the prefilter's upper bound of V = 2 q.x - ||x||^2 with its 10 low mantissa bits
        // replaced) bounds V by key' + 2^-12 |key'|.  It can only enter the exact top-k if
        // V >= ||q||^2 - D_k^2, where (f32 error of the reference's sum, of a and of D_k, all <= gamma)
        // ||q||^2 >= a^2 (1 - gamma) and the true D_k^2 <= dk^2 (1 + gamma).  Evaluated in f64.
        const double a = (double)q_norm[q], g = (double)gamma;
        const double vk_low = a * a * (1.0 - g) - (double)dk * (double)dk * (1.0 + g);
        const double sc = (double)score;
        above = valid && (sc + 0x1p-12 * (sc < 0.0 ? -sc : sc) >= vk_low);
    } else {
        // int8 stores: the candidate score is S / (scale ||x||); the query's scale s_q makes it the cosine estimate
        const float thr = (1.0f - dk) - (q_eps[q] + 0x1p-13f);
        above = valid && (q_scale ? q_scale[q] * score : score) >= thr;
    }
    // (1) rows listed but not selected score <= the worst selected one (sel is sorted, full => index kp-1)
    if (ci == kp - 1u && valid && above)
        bad = true;
    // (2) rows dropped inside a list score <= that list's last (3rd) entry: a list with all three
    //     entries above the margin may hide a fourth row that matters
    s_list[threadIdx.x] = above ? (uint32_t)comp_a : (0xffffffffu - threadIdx.x);
    __syncthreads();
    if (above) {
        const uint32_t my = (uint32_t)comp_a;
        uint32_t same = 0;
        for (uint32_t j = 0; j < ci; ++j)
            same += (s_list[j] == my) ? 1u : 0u;
        if (same + 1u >= kListDepth)
            bad = true;
    }
    if (valid && !in_range) // cannot happen (rows >= n_rows are masked in the prefilter); be loud if it does
        bad = true;
    if (bad)
        atomicOr(&s_flag, 1);
    __syncthreads();
    if (threadIdx.x == 0)
        flags[q] = s_flag;
}

// ---- re-score for small batches: the lanes share a row, one lane adds ---------------------------------------
// mfma_rescore_kernel gives every candidate row to ONE lane, which walks the row's d elements alone: with a
// handful of queries that is one busy wave per query on an otherwise idle chip, 40 us (f32, d = 768) of
// round trips and dependent arithmetic.  Only the ADDITIONS have to be sequential for the result to be the
// reference's (rank.rs:281-300: a left fold); the products (q_j - x_j)^2 / q_j x_j are independent.  Here a wave
// takes 4 candidates at a time: its 64 lanes fetch 256 elements of each (f32: lane l chunk l of all four;
// int8: lane l chunk l & 15 of candidate l >> 4), form the products with the scan's own operations, park them in
// LDS in element order, and lanes 0..3 fold "their" candidate's 256 products in order (lane 4: the q_j^2 of
// Cosine).  16 waves per query, kp / 16 candidates each; the next block's loads are in flight during the fold.
// Same arithmetic, same order => same bits; select, top-k, hits and the safety check as in mfma_rescore_kernel.
constexpr uint32_t kCoopWaves = 16, kCoopBlockElems = 256;

template <int DTYPE, int METRIC, typename TopK>
__global__ __launch_bounds__(64 * kCoopWaves) void mfma_rescore_coop_kernel(
    const uint4 *__restrict__ corpus, const float *__restrict__ row_norms, const float *__restrict__ queries,
    const float *__restrict__ lut, const uint64_t *__restrict__ sel, const int *__restrict__ q_bad,
    const uint64_t *__restrict__ ids, vfsx_hit *__restrict__ out, int *__restrict__ flags, uint32_t n_rows,
    uint32_t d_chunks, uint32_t q_stride, uint32_t k, uint32_t kp, const float *__restrict__ q_eps, uint64_t row_base,
    uint64_t id_base, uint32_t strip_tiles, const float *__restrict__ q_norm, float gamma, int only_flagged,
    const float *__restrict__ q_scale, float *__restrict__ dk_out)
{
    if (only_flagged && flags[blockIdx.x] == 0) // second, wider pass: flagged queries only (uniform exit)
        return;
    constexpr uint32_t EPC = DTYPE == VFSX_I8 ? 16u : 4u;
    constexpr uint32_t CPB = kCoopBlockElems / EPC; // chunks per block and candidate: 64 (f32) / 16 (int8)
    extern __shared__ __align__(16) uint64_t dyn_lds[]; // [16][64 | 128] lists
    __shared__ __align__(16) float prod[kCoopWaves][5][kCoopBlockElems]; // [wave][candidate 0..3, q^2][element]
    __shared__ float lut_s[DTYPE == VFSX_I8 ? 256 : 1];
    __shared__ uint32_t s_list[256];
    __shared__ float s_dk;
    __shared__ int s_flag;

    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x;
    const uint32_t list_stride = k > 64u ? 128u : 64u;
    if (DTYPE == VFSX_I8 && threadIdx.x < 256u)
        lut_s[threadIdx.x] = lut[threadIdx.x];
    if (threadIdx.x == 0)
        s_flag = q_bad[q];
    __syncthreads();
    const float *__restrict__ qv = queries + (size_t)q * q_stride;

    // lane t < cpw of wave w owns candidate ci = w * cpw + t (its key, the check, its slot in the wave's offer)
    const uint32_t cpw = kp / kCoopWaves; // 4 / 8 / 16
    const uint32_t ci = wave * cpw + lane;
    const uint64_t comp_a = lane < cpw ? sel[(size_t)q * kp + ci] : kSentinel;
    const bool valid = comp_a != kSentinel;
    float score = 0.0f;
    uint32_t row = 0;
    if (valid)
        row = cand_row(comp_a, score, strip_tiles);
    const bool in_range = valid && row < n_rows;
    const uint32_t row_safe = in_range ? row : 0u;

    float(*my_prod)[kCoopBlockElems] = prod[wave];
    const uint32_t n_blocks = (d_chunks + CPB - 1u) / CPB;
    uint64_t mycomp = kSentinel;
    float qn = 0.0f;
    for (uint32_t g = 0; g < cpw; g += 4u) { // (wave-uniform)
        // the four rows of this group (wave-uniform: SGPRs)
        uint32_t r4[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
            r4[j] = (uint32_t)__builtin_amdgcn_readlane((int)row_safe, (int)(g + j));
        // what this lane fetches: f32 -- chunk `lane` of each of the four; int8 -- chunk lane & 15 of row lane >> 4
        const uint32_t my_c = DTYPE == VFSX_I8 ? (lane & 15u) : lane;
        const uint4 *p[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
            p[j] = corpus + (size_t)(r4[j] >> 6) * d_chunks * 64u + (r4[j] & 63u);
        const uint4 *p_i8 = (lane >> 4) == 0u ? p[0] : ((lane >> 4) == 1u ? p[1] : ((lane >> 4) == 2u ? p[2] : p[3]));
        constexpr uint32_t NV = DTYPE == VFSX_I8 ? 1u : 4u;
        auto fetch = [&](uint32_t blk, uint4(&v)[NV]) __attribute__((always_inline)) {
            const uint32_t c = blk * CPB + my_c;
            const bool in = c < d_chunks;
#pragma unroll
            for (uint32_t j = 0; j < NV; ++j) {
                const uint4 *src = DTYPE == VFSX_I8 ? p_i8 : p[j];
                v[j] = in ? src[(size_t)c * 64u] : make_uint4(0u, 0u, 0u, 0u);
            }
        };
        float acc = 0.0f; // lanes 0..3: the candidates' sums; lane 4: sum q_j^2
        uint4 cur[NV], nxt[NV];
        fetch(0, cur);
        for (uint32_t blk = 0; blk < n_blocks; ++blk) {
            if (blk + 1u < n_blocks)
                fetch(blk + 1u, nxt);
            // products of this block, in element order, into the wave's LDS rows
            const uint32_t c = blk * CPB + my_c;
            const bool in = c < d_chunks;
            float qe[EPC];
#pragma unroll
            for (uint32_t i = 0; i < EPC; i += 4u) {
                const float4 t = in ? *(const float4 *)(qv + (size_t)c * EPC + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                qe[i] = t.x, qe[i + 1] = t.y, qe[i + 2] = t.z, qe[i + 3] = t.w;
            }
            if (DTYPE == VFSX_I8) {
                const uint32_t w[4] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w};
                float *dst = &my_prod[lane >> 4][my_c * EPC];
#pragma unroll
                for (uint32_t i = 0; i < 16u; ++i) {
                    const float xv = lut_s[(w[i >> 2] >> (8u * (i & 3u))) & 0xffu]; // (q as f32) / scale
                    float pr;
                    if (METRIC == VFSX_EUCLIDEAN) {
                        const float diff = f_sub(qe[i], xv);
                        pr = f_mul(diff, diff);
                    } else {
                        pr = f_mul(qe[i], xv);
                    }
                    dst[i] = in ? pr : 0.0f;
                }
                if (METRIC != VFSX_EUCLIDEAN && (lane >> 4) == 0u) {
#pragma unroll
                    for (uint32_t i = 0; i < 16u; ++i)
                        my_prod[4][my_c * EPC + i] = f_mul(qe[i], qe[i]);
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    const float xs[4] = {__uint_as_float(cur[j].x), __uint_as_float(cur[j].y), __uint_as_float(cur[j].z),
                                         __uint_as_float(cur[j].w)};
                    float4 pr;
                    float *pp = &pr.x;
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        if (METRIC == VFSX_EUCLIDEAN) {
                            const float diff = f_sub(qe[i], xs[i]);
                            pp[i] = f_mul(diff, diff);
                        } else {
                            pp[i] = f_mul(qe[i], xs[i]);
                        }
                        pp[i] = in ? pp[i] : 0.0f;
                    }
                    *(float4 *)&my_prod[j][my_c * EPC] = pr;
                }
                if (METRIC != VFSX_EUCLIDEAN)
                    *(float4 *)&my_prod[4][my_c * EPC] =
                        make_float4(f_mul(qe[0], qe[0]), f_mul(qe[1], qe[1]), f_mul(qe[2], qe[2]), f_mul(qe[3], qe[3]));
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier(); // one wave: its LDS operations execute in program order
            asm volatile("" ::: "memory");
            // the fold: strictly in element order (rank.rs:281-300), one candidate per lane
            if (lane < (METRIC == VFSX_EUCLIDEAN ? 4u : 5u)) {
                const float *src = my_prod[lane];
#pragma unroll 8
                for (uint32_t e = 0; e < kCoopBlockElems; e += 4u) {
                    const float4 t = *(const float4 *)(src + e);
                    acc = f_add(acc, t.x);
                    acc = f_add(acc, t.y);
                    acc = f_add(acc, t.z);
                    acc = f_add(acc, t.w);
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (uint32_t j = 0; j < NV; ++j)
                cur[j] = nxt[j];
        }
        // lane j < 4: the exact distance of candidate g + j
        if (METRIC != VFSX_EUCLIDEAN)
            qn = f_sqrt(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(acc), 4))); // rank.rs:296
        const uint32_t my_row = lane < 4u ? (lane == 0u ? r4[0] : (lane == 1u ? r4[1] : (lane == 2u ? r4[2] : r4[3]))) : 0u;
        float dist;
        if (METRIC == VFSX_EUCLIDEAN)
            dist = f_sqrt(acc); // rank.rs:286
        else
            dist = f_sub(1.0f, f_div(acc, f_mul(qn, row_norms[my_row]))); // rank.rs:299
        const uint64_t comp_g = ((uint64_t)dist_key(dist) << 32) | my_row;
        // candidate g + j lives in lane g + j of the wave's offer vector
        const uint32_t from = lane & 3u;
        const uint64_t moved = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)(uint32_t)(comp_g >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)(uint32_t)comp_g);
        if (lane >= g && lane < g + 4u)
            mycomp = moved;
    }
    if (!in_range)
        mycomp = kSentinel;

    TopK top;
    top.init(k, dyn_lds + wave * list_stride);
    top.offer_dense(mycomp);
    uint64_t *mine = dyn_lds + wave * list_stride;
    __syncthreads();
    top.spill(mine);
    __syncthreads();
    for (uint32_t s = 1; s < kCoopWaves; s <<= 1) { // merge tree over the waves' sorted lists
        if ((wave & (2u * s - 1u)) == 0u) {
            top.merge_list(dyn_lds + (wave + s) * list_stride);
            top.spill(mine);
        }
        __syncthreads();
    }
    if (wave == 0) {
        for (uint32_t j = lane; j < k; j += 64) {
            const uint64_t comp = mine[j];
            vfsx_hit h;
            if (comp == kSentinel) {
                h.dist = __uint_as_float(0x7fc00000u);
                h.flags = 0;
                h.row = ~0ull;
                h.id = ~0ull;
            } else {
                const uint32_t r = (uint32_t)comp;
                h.dist = key_dist((uint32_t)(comp >> 32));
                h.flags = 1;
                h.row = row_base + r;
                h.id = ids ? ids[r] : id_base + r;
            }
            out[(size_t)q * k + j] = h;
        }
        if (lane == 0) {
            const uint64_t kth = mine[k - 1];
            s_dk = kth == kSentinel ? __uint_as_float(0x7fc00000u) : key_dist((uint32_t)(kth >> 32));
        }
    }
    if (threadIdx.x < 256u)
        s_list[threadIdx.x] = 0xffffffffu - threadIdx.x;
    __syncthreads();

    // safety check: as in mfma_rescore_kernel, candidate ci in lane ci % cpw of wave ci / cpw
    const float dk = s_dk;
    if (dk_out && threadIdx.x == 0) // the wider second pass filters the candidate lists against this k-th distance
        dk_out[q] = dk;
    bool bad = !(dk == dk) || !(dk < __uint_as_float(0x7f800000u));
    bool above;
    if (METRIC == VFSX_EUCLIDEAN) {
        const double a = (double)q_norm[q], gm = (double)gamma;
        const double vk_low = a * a * (1.0 - gm) - (double)dk * (double)dk * (1.0 + gm);
        const double sc = (double)score;
        above = valid && (sc + 0x1p-12 * (sc < 0.0 ? -sc : sc) >= vk_low);
    } else {
        const float thr = (1.0f - dk) - (q_eps[q] + 0x1p-13f);
        above = valid && (q_scale ? q_scale[q] * score : score) >= thr;
    }
    if (lane < cpw && ci == kp - 1u && valid && above)
        bad = true;
    if (lane < cpw && above)
        s_list[ci] = (uint32_t)comp_a;
    __syncthreads();
    if (lane < cpw && above) {
        const uint32_t my = (uint32_t)comp_a;
        uint32_t same = 0;
        for (uint32_t j = 0; j < ci; ++j)
            same += (s_list[j] == my) ? 1u : 0u;
        if (same + 1u >= kListDepth)
            bad = true;
    }
    if (lane < cpw && valid && !in_range)
        bad = true;
    if (bad)
        atomicOr(&s_flag, 1);
    __syncthreads();
    if (threadIdx.x == 0)
        flags[q] = s_flag;
}

} // namespace

// ---- launchers ----------------------------------------------------------------------------------------

hipError_t launch_mfma_mirror(int dtype, const uint4 *corpus, const float *row_norms, const float *lut,
                              uint4 *mirror, uint32_t n_rows, uint32_t dim, uint32_t d_chunks, const MfmaGeometry &g,
                              uint32_t *stats, hipStream_t s)
{
    const uint64_t total_units = (uint64_t)g.n_rowtiles * 8u * g.KS; // 32-row groups of the padded row count
    const uint64_t blocks = (total_units * 64u + 255) / 256;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    const dim3 res_grid((n_rows + 255u) / 256u);
    if (dtype == VFSX_I8) {
        hipLaunchKernelGGL((mfma_mirror_kernel<VFSX_I8>), dim3((uint32_t)blocks), dim3(256), 0, s, corpus, row_norms,
                           lut, mirror, n_rows, dim, d_chunks, g.KS, total_units, stats);
        hipLaunchKernelGGL((mfma_residual_kernel<VFSX_I8>), res_grid, dim3(256), 0, s, corpus, row_norms, lut, n_rows,
                           d_chunks, g.k_pad, stats);
    } else {
        hipLaunchKernelGGL((mfma_mirror_kernel<VFSX_F32>), dim3((uint32_t)blocks), dim3(256), 0, s, corpus, row_norms,
                           lut, mirror, n_rows, dim, d_chunks, g.KS, total_units, stats);
        hipLaunchKernelGGL((mfma_residual_kernel<VFSX_F32>), res_grid, dim3(256), 0, s, corpus, row_norms, lut, n_rows,
                           d_chunks, g.k_pad, stats);
    }
    return hipGetLastError();
}

hipError_t launch_mfma_query_prep(const float *d_queries, uint32_t nq, uint32_t dim, void *q_mirror, uint32_t KS,
                                  int *q_bad, float *q_norm, float *q_eps, const uint32_t *stats, hipStream_t s)
{
    hipLaunchKernelGGL(mfma_query_prep_kernel, dim3(nq), dim3(64), 0, s, d_queries, nq, dim, (uint16_t *)q_mirror, KS,
                       q_bad, q_norm, q_eps, stats, mfma_f32_slack(KS * 16u));
    return hipGetLastError();
}

#ifdef VFSX_EXP_STAMPS
} // namespace vfsx
extern "C" int vfsx_exp_stamps(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vfsx::exp_stamps), sizeof(unsigned long long) * 8) != hipSuccess)
        return 1;
    if (reset) {
        unsigned long long z[8] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(vfsx::exp_stamps), z, sizeof(z)) != hipSuccess;
    }
    return 0;
}
namespace vfsx {
#endif

hipError_t launch_mfma_rownorm_pairs(const float *row_norms, uint32_t n_rows, uint32_t n_pad, float gamma, float *out,
                                     float *inv, float i8_scale, uint32_t *stats, hipStream_t s, uint32_t row0)
{
    if (row0 >= n_pad)
        return hipSuccess;
    hipLaunchKernelGGL(mfma_rownorm_pairs_kernel, dim3((n_pad - row0 + 255u) / 256u), dim3(256), 0, s, row_norms, n_rows,
                       n_pad, gamma, (float2 *)out, inv, i8_scale, stats, row0);
    return hipGetLastError();
}

hipError_t launch_mfma_query_prep_i8(const float *d_queries, uint32_t nq, uint32_t dim, void *q_mirror, uint32_t KS,
                                     int *q_bad, float *q_norm, float *q_eps, float *q_scale, const uint32_t *stats,
                                     hipStream_t s)
{
    hipLaunchKernelGGL(mfma_query_prep_i8_kernel, dim3(nq), dim3(64), 0, s, d_queries, nq, dim, (int8_t *)q_mirror, KS,
                       q_bad, q_norm, q_eps, q_scale, stats, mfma_f32_slack(KS * 32u));
    return hipGetLastError();
}

hipError_t launch_mfma_mirror_i8(const uint4 *corpus, const float *row_norms, uint4 *mirror, float *inv_rn,
                                 uint32_t n_rows, uint32_t d_chunks, uint32_t d_chunks8, uint32_t k_pad,
                                 uint32_t *stats, hipStream_t s, uint32_t row0)
{
    if (row0 >= n_rows)
        return hipSuccess;
    hipLaunchKernelGGL(mfma_mirror_i8_kernel, dim3((n_rows - row0 + 255u) / 256u), dim3(256), 0, s, corpus, row_norms,
                       mirror, inv_rn, n_rows, d_chunks, d_chunks8, k_pad, stats, row0);
    return hipGetLastError();
}

// which 256 x 256 prefilter: VFSX_MFMA_PF = "rv" (rows in registers, mfma_prefilter_rv_kernel) or "lds" (both operands
// through LDS, mfma_prefilter256_kernel)
static bool pf_rows_in_registers()
{
    static const bool rv = [] {
        const char *e = std::getenv("VFSX_MFMA_PF");
        return e ? (e[0] == 'r') : false;
    }();
    return rv;
}

// euclid != 0: Euclidean upper-bound epilogue; rn2 / q_norm / q_eps as in the kernel
// Grid of mfma_prefilter256_kernel and its query tiles per XCD (see the kernel's mapping comment): the most query
// tiles whose share of Q~ fits half an XCD's L2 (4 MiB) among the groupings the mapping allows, or VFSX_MFMA_QPX.
static dim3 pf_grid(const MfmaGeometry &g, uint32_t n_qtiles, size_t qtile_bytes, uint32_t *qpx_out)
{
    static const uint32_t forced = std::getenv("VFSX_MFMA_QPX") ? (uint32_t)std::atoi(std::getenv("VFSX_MFMA_QPX")) : 0u;
    uint32_t qpx = 0;
    for (uint32_t groups = 8u; groups >= 1u; groups >>= 1) { // ascending qpx
        if (n_qtiles % groups)
            continue;
        const uint32_t cand = n_qtiles / groups;
        if (qpx == 0 || cand == forced || (forced == 0 && (size_t)cand * qtile_bytes <= (2u << 20)))
            qpx = cand;
        if (cand == forced)
            break;
    }
    const uint32_t lanes = 8u / (n_qtiles / qpx);
    *qpx_out = qpx;
    return dim3(8u * qpx * ((g.n_strips + lanes - 1u) / lanes));
}

hipError_t launch_mfma_prefilter(const uint4 *mirror, const void *q_mirror, float *cand, uint32_t n_rows,
                                 const MfmaGeometry &g, uint32_t nq, int euclid, const float *rn2,
                                 const float *q_norm, const float *q_eps, hipStream_t s)
{
    const uint32_t n_qtiles = (nq + 255u) / 256u;
    uint32_t qpx = 1;
    const dim3 grid = pf_grid(g, n_qtiles, (size_t)256u * g.k_pad * 2u, &qpx);
    const uint4 *qm = (const uint4 *)q_mirror;
    const float2 *rn = (const float2 *)rn2;
    // <= 128 queries: the skinny kernel (HBM-bound instead of latency-bound); VFSX_MFMA_SKINNY=0 turns it off
    if (mfma_small_batch(nq)) {
        // rows larger than what the Infinity Cache keeps between batches are streamed nontemporally (as the exact scan)
        const bool nt = (uint64_t)g.n_rowtiles * 256u * g.k_pad * 2u > (192ull << 20);
#define VFSX_SKINNY(MODE, NCOL)                                                                                    \
    do {                                                                                                           \
        if (nt)                                                                                                    \
            hipLaunchKernelGGL((mfma_prefilter_skinny_kernel<3, MODE, NCOL, false, true>), dim3(g.n_strips),       \
                               dim3(512), 0, s, mirror, qm, cand, n_rows, g.n_rowtiles, g.KS, nq, g.n_lists,       \
                               g.strip_code(), rn, q_norm, q_eps, 0u, (uint64_t)0, (const float *)nullptr,          \
                               (const float *)nullptr);                                                            \
        else                                                                                                       \
            hipLaunchKernelGGL((mfma_prefilter_skinny_kernel<3, MODE, NCOL, false, false>), dim3(g.n_strips),      \
                               dim3(512), 0, s, mirror, qm, cand, n_rows, g.n_rowtiles, g.KS, nq, g.n_lists,       \
                               g.strip_code(), rn, q_norm, q_eps, 0u, (uint64_t)0, (const float *)nullptr,          \
                               (const float *)nullptr);                                                            \
    } while (0)
        if (nq <= 64u) {
            if (euclid)
                VFSX_SKINNY(1, 1);
            else
                VFSX_SKINNY(0, 1);
        } else {
            if (euclid)
                VFSX_SKINNY(1, 2);
            else
                VFSX_SKINNY(0, 2);
        }
#undef VFSX_SKINNY
        return hipGetLastError();
    }
    if (pf_rows_in_registers()) {
#define VFSX_PFRV(MODE)                                                                                            \
    hipLaunchKernelGGL((mfma_prefilter_rv_kernel<MODE, false>), grid, dim3(512), 0, s, mirror, qm, cand, n_rows,     \
                       g.n_rowtiles, g.KS, n_qtiles, nq, g.n_lists, g.strip_tiles, rn, q_norm, q_eps, 0u,            \
                       (uint64_t)0, (const float *)nullptr, (const float *)nullptr, qpx, g.n_strips)
        if (euclid)
            VFSX_PFRV(1);
        else
            VFSX_PFRV(0);
#undef VFSX_PFRV
        return hipGetLastError();
    }
#define VFSX_PF256(MODE)                                                                                           \
    hipLaunchKernelGGL((mfma_prefilter256_kernel<VFSX_PF_KU, VFSX_PF_NSTAGE, MODE, false>), grid, dim3(512), 0, s,  \
                       mirror, qm, cand, n_rows, g.n_rowtiles, g.KS, n_qtiles, nq, g.n_lists, g.strip_tiles, rn,    \
                       q_norm, q_eps, 0u, (uint64_t)0, (const float *)nullptr, (const float *)nullptr, qpx,         \
                       g.n_strips)
    if (euclid)
        VFSX_PF256(1);
    else
        VFSX_PF256(0);
#undef VFSX_PF256
    return hipGetLastError();
}

// int8 stores: `corpus` is the rb64 corpus itself (d_chunks chunks per row, corpus_bytes allocated), q_mirror the
// int8 query mirror; g from mfma_geometry(..., true)
hipError_t launch_mfma_prefilter_i8(const uint4 *corpus, uint32_t d_chunks, uint64_t corpus_bytes, const void *q_mirror,
                                    float *cand, uint32_t n_rows, const MfmaGeometry &g, uint32_t nq, int euclid,
                                    const float *rn2, const float *inv_rn, const float *q_norm, const float *q_eps,
                                    const float *q_scale, hipStream_t s)
{
    const uint32_t n_qtiles = (nq + 255u) / 256u;
    uint32_t qpx = 1;
    const dim3 grid = pf_grid(g, n_qtiles, (size_t)256u * g.k_pad, &qpx);
    if (mfma_small_batch(nq)) {
        const bool nt = (uint64_t)g.n_rowtiles * 256u * g.k_pad > (192ull << 20); // (as above; 1 B per element)
#define VFSX_SKINNY(MODE, NCOL)                                                                                    \
    do {                                                                                                           \
        if (nt)                                                                                                    \
            hipLaunchKernelGGL((mfma_prefilter_skinny_kernel<3, MODE, NCOL, true, true>), dim3(g.n_strips),        \
                               dim3(512), 0, s, corpus, (const uint4 *)q_mirror, cand, n_rows, g.n_rowtiles, g.KS, \
                               nq, g.n_lists, g.strip_code(), (const float2 *)rn2, q_norm, q_eps, d_chunks,         \
                               corpus_bytes, inv_rn, q_scale);                                                     \
        else                                                                                                       \
            hipLaunchKernelGGL((mfma_prefilter_skinny_kernel<3, MODE, NCOL, true, false>), dim3(g.n_strips),       \
                               dim3(512), 0, s, corpus, (const uint4 *)q_mirror, cand, n_rows, g.n_rowtiles, g.KS, \
                               nq, g.n_lists, g.strip_code(), (const float2 *)rn2, q_norm, q_eps, d_chunks,         \
                               corpus_bytes, inv_rn, q_scale);                                                     \
    } while (0)
        if (nq <= 64u) {
            if (euclid)
                VFSX_SKINNY(1, 1);
            else
                VFSX_SKINNY(0, 1);
        } else {
            if (euclid)
                VFSX_SKINNY(1, 2);
            else
                VFSX_SKINNY(0, 2);
        }
#undef VFSX_SKINNY
        return hipGetLastError();
    }
    if (pf_rows_in_registers()) {
#define VFSX_PFRV(MODE)                                                                                            \
    hipLaunchKernelGGL((mfma_prefilter_rv_kernel<MODE, true>), grid, dim3(512), 0, s, corpus,                        \
                       (const uint4 *)q_mirror, cand, n_rows, g.n_rowtiles, g.KS, n_qtiles, nq, g.n_lists,          \
                       g.strip_tiles, (const float2 *)rn2, q_norm, q_eps, d_chunks, corpus_bytes, inv_rn, q_scale,  \
                       qpx, g.n_strips)
        if (euclid)
            VFSX_PFRV(1);
        else
            VFSX_PFRV(0);
#undef VFSX_PFRV
        return hipGetLastError();
    }
    // g.ku == 2: K is an odd number of 128-k half-steps (dim <= 128, 257..384, ...): the 64-k K-step variant takes
    // it without padding it to the next 256 (at dim 128 that padding would double the kernel's MFMA and fill work)
#define VFSX_PF256(KU, NS, MODE)                                                                                   \
    hipLaunchKernelGGL((mfma_prefilter256_kernel<KU, NS, MODE, true>), grid, dim3(512), 0, s, corpus,               \
                       (const uint4 *)q_mirror, cand, n_rows, g.n_rowtiles, g.KS, n_qtiles, nq, g.n_lists,          \
                       g.strip_tiles, (const float2 *)rn2, q_norm, q_eps, d_chunks, corpus_bytes, inv_rn, q_scale,  \
                       qpx, g.n_strips)
    if (g.ku == 2u) {
        if (euclid)
            VFSX_PF256(2, 2, 1);
        else
            VFSX_PF256(2, 2, 0);
    } else {
        if (euclid)
            VFSX_PF256(VFSX_PF_KU, VFSX_PF_NSTAGE, 1);
        else
            VFSX_PF256(VFSX_PF_KU, VFSX_PF_NSTAGE, 0);
    }
#undef VFSX_PF256
    return hipGetLastError();
}

hipError_t launch_mfma_select(const float *cand, uint64_t *sel, uint32_t nq, uint32_t n_lists, uint32_t kp,
                              const int *only_flagged, hipStream_t s)
{
    constexpr int W = 16;
    const uint32_t list_stride = kp > 64u ? kp : 64u;
    const size_t lds = (size_t)W * list_stride * sizeof(uint64_t);
    if (kp <= kMaxRegisterK)
        hipLaunchKernelGGL((mfma_select_kernel<WaveTopKReg, W>), dim3(nq), dim3(64 * W), lds, s, cand, sel, n_lists,
                           kp, only_flagged);
    else if (kp <= 128u)
        hipLaunchKernelGGL((mfma_select_kernel<WaveTopKReg2, W>), dim3(nq), dim3(64 * W), lds, s, cand, sel, n_lists,
                           kp, only_flagged);
    else
        hipLaunchKernelGGL((mfma_select_kernel<WaveTopKLds, W>), dim3(nq), dim3(64 * W), lds, s, cand, sel, n_lists,
                           kp, only_flagged);
    return hipGetLastError();
}

hipError_t launch_mfma_filter(const float *cand, uint64_t *sel, uint32_t nq, uint32_t n_lists, const int *flags,
                              const int *q_bad, const float *dk1, const float *q_eps, const float *q_scale,
                              const float *q_norm, float gamma, int euclid, hipStream_t s)
{
    hipLaunchKernelGGL(mfma_filter_kernel, dim3(nq), dim3(256), 0, s, cand, sel, n_lists, flags, q_bad, dk1, q_eps,
                       q_scale, q_norm, gamma, euclid);
    return hipGetLastError();
}

hipError_t launch_mfma_rescore(int dtype, const MfmaRescoreArgs &a, uint32_t nq, hipStream_t s)
{
    const uint32_t list_stride = a.k > 64u ? 128u : 64u;
    const size_t lds = (size_t)kWavesPerBlock * list_stride * sizeof(uint64_t);
    // k <= 64: one register per lane; k <= 128 (kMfmaMaxK): two
#define VFSX_RESCORE_T(DT, M, TOPK)                                                                                      \
    hipLaunchKernelGGL((mfma_rescore_kernel<DT, M, TOPK>), dim3(nq), dim3(kBlock), lds, s, a.corpus,                  \
                       a.row_norms,                                                                                  \
                       a.queries, a.lut, a.sel, a.q_bad, a.ids, a.out, a.flags, a.n_rows, a.d_chunks, a.q_stride,     \
                       a.k, a.kp, a.q_eps, a.row_base, a.id_base, a.strip_tiles, a.q_norm, a.gamma, a.only_flagged,   \
                       a.q_scale, a.dk_out)
    // batches of <= 256 queries: the cooperative kernel (16 waves per query, lanes share the rows)
    const size_t lds_coop = (size_t)kCoopWaves * list_stride * sizeof(uint64_t);
#define VFSX_RESCORE_C(DT, M, TOPK)                                                                                  \
    hipLaunchKernelGGL((mfma_rescore_coop_kernel<DT, M, TOPK>), dim3(nq), dim3(64 * kCoopWaves), lds_coop, s,        \
                       a.corpus, a.row_norms, a.queries, a.lut, a.sel, a.q_bad, a.ids, a.out, a.flags, a.n_rows,      \
                       a.d_chunks, a.q_stride, a.k, a.kp, a.q_eps, a.row_base, a.id_base, a.strip_tiles, a.q_norm,    \
                       a.gamma, a.only_flagged, a.q_scale, a.dk_out)
#define VFSX_RESCORE(DT, M)                                                                                          \
    do {                                                                                                             \
        if (a.k <= kMaxRegisterK && coop)                                                                            \
            VFSX_RESCORE_C(DT, M, WaveTopKReg);                                                                      \
        else if (a.k <= kMaxRegisterK)                                                                               \
            VFSX_RESCORE_T(DT, M, WaveTopKReg);                                                                      \
        else if (coop)                                                                                               \
            VFSX_RESCORE_C(DT, M, WaveTopKReg2);                                                                     \
        else                                                                                                         \
            VFSX_RESCORE_T(DT, M, WaveTopKReg2);                                                                     \
    } while (0)
    static const bool coop_on = !(std::getenv("VFSX_MFMA_COOP") && std::getenv("VFSX_MFMA_COOP")[0] == '0');
    const bool coop = coop_on && nq <= 256u && (a.kp == 64u || a.kp == 128u || a.kp == 256u);
    if (dtype == VFSX_I8) {
        if (a.euclid)
            VFSX_RESCORE(VFSX_I8, VFSX_EUCLIDEAN);
        else
            VFSX_RESCORE(VFSX_I8, VFSX_COSINE);
    } else {
        if (a.euclid)
            VFSX_RESCORE(VFSX_F32, VFSX_EUCLIDEAN);
        else
            VFSX_RESCORE(VFSX_F32, VFSX_COSINE);
    }
#undef VFSX_RESCORE
#undef VFSX_RESCORE_C
#undef VFSX_RESCORE_T
    return hipGetLastError();
}

// <= 128 queries: the skinny kernel (HBM-bound instead of latency-bound); VFSX_MFMA_SKINNY=0 turns it off
bool mfma_small_batch(uint32_t nq)
{
    static const bool skinny_on = !(std::getenv("VFSX_MFMA_SKINNY") && std::getenv("VFSX_MFMA_SKINNY")[0] == '0');
    return skinny_on && nq <= 128u;
}

MfmaGeometry mfma_geometry(uint32_t n_rows, uint32_t dim, bool int8, uint32_t n_cus, uint32_t nq)
{
    MfmaGeometry g;
    const uint32_t n_qtiles = (nq + 255u) / 256u;
    const bool small = mfma_small_batch(nq);
    g.strip_lists = 4u;
    // whole ring rounds only: K-steps (of 4 MFMA k-slices: 64 bf16 / 128 int8 k) per tile % ring depth (2) == 0, i.e.
    // K a multiple of 128 (bf16) or 256 (int8); the int8 kernel also exists with K-steps of 2 k-slices (ku = 2), which
    // takes the multiples of 128 in between
    g.k_pad = (dim + 127u) / 128u * 128u;
    g.KS = g.k_pad / (int8 ? 32u : 16u);
    g.ku = (int8 && g.KS % 8u != 0u) ? 2u : 4u;
    g.n_rowtiles = (n_rows + 255u) / 256u;
    // aim for >= ~4096 candidate lists per query so that three rows above the safety margin rarely share
    // one (small corpora get short strips -- and more workgroups); a list holds at most 1024 rows (10 index
    // bits): at most 16 tiles = 4096 rows per strip, 4 lists per strip
    g.strip_tiles = std::min<uint32_t>(kMaxStripTiles, std::max<uint32_t>(1u, (n_rows / 256u) * 4u / 4096u));
    if (n_cus) {
        // Both prefilters run ONE workgroup per CU (their LDS rings are 120-128 KiB) and all strips have the same
        // length, so a launch of W = query tiles x strips workgroups takes ceil(W / CUs) rounds of (strip_tiles tiles +
        // about half a tile of ring fill and epilogue).  With few rounds the rounding matters: C2 (3907 tiles) and one
        // query tile, 3 tiles per strip = 1303 workgroups = 5.09 -> 6 rounds = 18 tile times for 15.26 tiles of work
        // per CU; 4 per strip = 3.82 -> 4 rounds = 16 (measured 0.143 -> 0.128 ms, profiles/r03/c2m_strip_sweep.txt).
        // Pick the cheapest strip length that still leaves enough lists per query (the safety check flags a list
        // with three rows above the margin: its rate grows with 1 / lists^2, and ONE flagged query costs its batch an
        // exact scan, ~0.5 ms at this size): >= 2048 for one query tile, >= 3900 for more (measured: 2236 lists made
        // one query per 1024-query batch fall back, 1.07 -> 1.68 ms, and 3908 lists still did: calls of more than two
        // query tiles keep the size rule -- the more queries, the likelier one of them is flagged, and the less the
        // rounding costs).  Ties go to the shorter strip.  Launches of >= 32 rounds keep the size rule as well.
        // The small-batch kernel has a second form for long strips (strip_lists = 16): every wave keeps its own two
        // lists instead of merging with the three other waves of its 128-row half, so a 16-tile strip yields the lists
        // (and rows per list) of four 4-tile strips -- one workgroup per CU for the whole launch, one ring fill and
        // epilogue instead of four (C2: 245 workgroups, 1 round x 16 tiles; <= 8192 lists so that the select stays
        // short).
        const uint32_t strips0 = (g.n_rowtiles + g.strip_tiles - 1u) / g.strip_tiles;
        if (n_qtiles <= 2u && (uint64_t)strips0 * n_qtiles < 32ull * n_cus) {
            static const bool own_ok = !(std::getenv("VFSX_MFMA_OWN_LISTS") && std::getenv("VFSX_MFMA_OWN_LISTS")[0] == '0');
            uint32_t best = g.strip_tiles, best_lists = 4u;
            uint64_t best_cost = ~0ull;
            for (uint32_t lists = 4u; lists <= ((small && own_ok) ? 16u : 4u); lists *= 4u) {
                for (uint32_t st = 1; st <= kMaxStripTiles; ++st) {
                    const uint32_t strips = (g.n_rowtiles + st - 1u) / st;
                    if (strips * lists < (n_qtiles == 1u ? 2048u : 3900u) && st > 1u)
                        break;
                    if (lists == 16u && strips * lists > 8192u)
                        continue;
                    const uint64_t wgs = (uint64_t)strips * n_qtiles;
                    const uint64_t cost = ((wgs + n_cus - 1u) / n_cus) * (2u * st + 1u);
                    if (cost < best_cost)
                        best_cost = cost, best = st, best_lists = lists;
                }
            }
            g.strip_tiles = best;
            g.strip_lists = best_lists;
        }
    }
    static const uint32_t forced = std::getenv("VFSX_MFMA_STRIP") ? (uint32_t)std::atoi(std::getenv("VFSX_MFMA_STRIP")) : 0u;
    if (forced >= 1u && forced <= kMaxStripTiles) { // experiments (always the merged, 4-lists form)
        g.strip_tiles = forced;
        g.strip_lists = 4u;
    }
    g.n_strips = (g.n_rowtiles + g.strip_tiles - 1) / g.strip_tiles;
    g.n_lists = g.n_strips * g.strip_lists;
    g.mirror_bytes = int8 ? 0 : (size_t)g.n_rowtiles * 256u * g.k_pad * 2u; // int8 stores need no mirror
    return g;
}

// ---- error model of the prefilter score -------------------------------------------------------------------
// Notation: x a row, q a query (their f32 elements taken as exact reals), K = k_pad, u = 2^-24.
//   b   = the f32 row norm the scan uses (sequential sum, sqrt):   b = ||x|| (1 + e_b), |e_b| <= (K/2 + 2) u
//   a   = the f32 query norm of mfma_query_prep_kernel (tree sum):  a = ||q|| (1 + e_a), |e_a| <= (K/2 + 2) u
//   x^  = x / b,  q^ = q / a  (reals);   x~, q~ = the bf16 mirror rows (exact reals)
//   s^  = x^ . q^ = q.x / (a b);         s~ = what the MFMAs accumulate from the x~_j q~_j
// bf16 keeps 8 significand bits: ONE round-to-nearest moves a value by up to 2^-8 of itself (not 2^-9: the
// round-1 bound had that wrong and was half of what two adversarially aligned roundings need -- a row and a
// query whose normalised components all sit just under a bf16 midpoint lose 0.0077 of their cosine, see
// tests/test_gpu_mfma.py::test_mfma_adversarial_bf16_midpoints).  Instead of a worst-case constant the bound
// uses what the roundings actually did:
//   r_x = || x^ - x~ ||_2 per row, R_x = max over the corpus (mfma_residual_kernel, stats[1]), r_q per query.
//   Both are upper bounds: the residual of fl(x_j / b) is exact in f32, the division adds <= u |x^_j| per
//   element, the f32 sum / sqrt are inflated by (K + 8) u (residual_norm_bound).  r <= 2^-8 (1 + 2u) always.
// (1) s~_exact - s^ = (x~ - x^).q~ + x^.(q~ - q^):  |.| <= r_x ||q~|| + ||x^|| r_q
//       <= r_x (1 + |e_a| + r_q) + (1 + |e_b|) r_q <= (R_x + r_q)(1 + 2^-7)          for K <= 32768
// (2) MFMA accumulation of K exact products in f32, any order, each add rounded or truncated:
//       |s~ - s~_exact| <= 2K u ||x~|| ||q~|| <= 2.03 K u
//     => eps_q = (R_x + r_q)(1 + 2^-7) + mfma_f32_slack(K)  >=  |s~ - s^|   (what the Euclidean epilogue needs)
// (3) s^ vs the true cosine c:  s^ = c / ((1 + e_a)(1 + e_b)),  |s^ - c| <= (K + 5) u
// (4) the reference's computed cosine term c_ref = fl(dot / fl(n1 n2)) (rank.rs:291-299): sequential dot
//     <= K u sum |q_j x_j| <= K u ||q|| ||x||, two norms (K/2 + 2) u each, product and quotient u each:
//     |c_ref - c| <= (2K + 6) u; the final 1 - c_ref and the check's own 1 - D_k and thr roundings <= 6 u
//     => |s~ - c_ref| <= (R_x + r_q)(1 + 2^-7) + (5.03 K + 17) u  <=  eps_q;  the Cosine check adds 2^-13 for
//     the 10 mantissa bits of the score that hold the row position.
// mfma_f32_slack = (6K + 64) u + 2^-20 (the last term: bf16 subnormals the MFMA may flush, < K 2^-126, and the
// f32 roundings of eps_q's own evaluation).  The Euclidean path needs only (2) of it, 2.03 K u; the surplus
// (>= 3.9 K u) is what covers the three f32 roundings of its epilogue, 3 u |2ab (s~ + eps_q)| <= 3.1 u (2ab).
//
// int8 operands (v_mfma_i32_32x32x32_i8): x~ = s_x X, q~ = s_q Q with int8 vectors X, Q; S = X.Q is EXACT in int32
// (|S| <= 128 * 127 * 32768 < 2^31), so (2) vanishes and only the quantization residuals remain:
//   int8 store:   X = the stored bytes, s_x = 1 / (scale b) exactly the row (x^ = s_x X up to the f32 rounding of the
//                 stored value byte / scale, vector.rs:138, <= u per element): r_x = 0,
//                 |s_q s_x S - s^| = |x^.(q~ - q^)| <= ||x^|| r_q = r_q (1 + |e_b|)
//   int8 mirror:  X = round(x^ / s_x), r_x measured per row by mfma_mirror_i8_kernel, R_x its maximum:
//                 |s_x s_q S - s^| <= r_x ||q~|| + ||x^|| r_q <= R_x (1 + r_q + |e_a|) + (1 + |e_b|) r_q
//   => eps_q = (R_x + r_q)(1 + 2^-7) + R_x r_q + mfma_f32_slack(K)   (mfma_query_prep_i8_kernel; R_x = 0 for a store)
// r_q, r_x carry + 2^-22 for the f32 roundings inside their own measurement (d = y - s t: two roundings of <= u |y|).
// The epilogue's f32 operations on the score -- int32 -> f32 conversion of S (relative u), the product with the
// row factor inv_rn (and ||x||, Euclidean), the query factor -- are 3-4 relative roundings of a value <= ~1 in
// cosine units, inside the same surplus of mfma_f32_slack that covers the bf16 epilogue.  (3) and (4) are unchanged.
float mfma_f32_slack(uint32_t k_pad)
{
    return (6.0f * (float)k_pad + 64.0f) * 0x1p-24f + 0x1p-20f;
}

} // namespace vfsx
