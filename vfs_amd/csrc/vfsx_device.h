// vfsx_device.h -- device-side building blocks shared by the HIP translation units
// (vfsx_kernels.hip, vfsx_mfma.hip): exact f32 primitives, the distance key, the
// per-wave top-k lists and the sequential row accumulation.
#pragma once

#include <hip/hip_runtime.h>

#include "vfsx_internal.h"

// No FMA contraction in any file that includes this header.
#pragma clang fp contract(off)

namespace vfsx {
namespace {

// One IEEE-754 binary32 operation each, round-to-nearest-even.  NOT the HIP
// __f*_rn intrinsics: without OCML_BASIC_ROUNDED_OPERATIONS __fsqrt_rn is the
// *native* (1 ulp) sqrt.  sqrtf and `/` are correctly rounded because hipcc's
// -fhip-fp32-correctly-rounded-divide-sqrt is on by default (the Makefile also
// passes it explicitly); tests/test_gpu_parity.py checks both bit-for-bit.
__device__ __forceinline__ float f_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float f_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float f_div(float a, float b) { return a / b; }
__device__ __forceinline__ float f_sqrt(float a) { return __builtin_sqrtf(a); }

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// ---- key <-> distance ---------------------------------------------------------
// Monotone map of an f32 to a u32 so that u32 order == the reference's
// partial_cmp order (rank.rs:170) for non-NaN values, with -0.0 == +0.0.
// NaN (outside the reference's defined behaviour) sorts last.
__device__ __forceinline__ uint32_t dist_key(float d)
{
    uint32_t b = __float_as_uint(d);
    if ((b & 0x7fffffffu) > 0x7f800000u)
        return 0xffffffffu;
    if (b == 0x80000000u)
        b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float key_dist(uint32_t k)
{
    if (k == 0xffffffffu)
        return __uint_as_float(0x7fc00000u);
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

// lane l <- lane l-1 (lane 0 keeps its own value): one DPP move per dword
// (wave_shr:1, a full-wavefront shift on the gfx9 family) instead of ds_bpermute.
__device__ __forceinline__ uint64_t shfl_up1_u64(uint64_t v)
{
    constexpr int kDppWaveShr1 = 0x138;
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, kDppWaveShr1, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, kDppWaveShr1, 0xf, 0xf, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// ---- wave-wide bitonic network on composite keys ---------------------------------
// One insert into a register list costs a ballot -> popcount -> shift -> readlane chain (~150 cycles of dependent
// VALU/SALU hops).  That is the right tool while a row-block beats the k-th best once in a while; it is the wrong one
// when 64 fresh keys meet an empty list (a wave's first row-block; the select's first column: ~k ln(64 e / k) inserts)
// or when two sorted lists are merged (workgroup folds, select).  Those go through a compare-exchange network instead:
// 21 stages sort 64 keys across the wave, 6 stages merge two sorted 64-lists, 9 instructions per stage.

// value of lane (l ^ J): DPP inside a row of 16, the gfx950 permlane swaps across rows
template <int J>
__device__ __forceinline__ uint32_t xor_lane32(uint32_t v)
{
    const int x = (int)v;
    if constexpr (J == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false); // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        // banks 0 and 2 of a row (lanes 0-3, 8-11) read lane + 4 (row_shl:4), banks 1 and 3 lane - 4 (row_shr:4)
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);
        return (uint32_t)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);
    } else if constexpr (J == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false); // row_ror:8
    } else if constexpr (J == 16) {
        // v_permlane16_swap: odd rows of the first operand <-> even rows of the second.  With both = v the first
        // becomes (r0, r0, r2, r2), the second (r1, r1, r3, r3): a lane of an even row wants the second
        const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane_id() & 16u) ? p[0] : p[1];
    } else {
        static_assert(J == 32, "xor distance");
        // v_permlane32_swap: upper half of the first operand <-> lower half of the second
        const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane_id() & 32u) ? p[0] : p[1];
    }
}

// bit l set: lane l keeps the SMALLER key of (own, lane l ^ j) in the stage (size, j) of a bitonic sort whose
// blocks of `size` lanes alternate direction (size = 64: one block), ascending overall unless `desc`
constexpr uint64_t bitonic_keep_min(int size, int j, bool desc)
{
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) {
        const bool asc = ((l & size) == 0) != desc;
        const bool low = (l & j) == 0;
        if (low == asc)
            m |= 1ull << l;
    }
    return m;
}

template <int J, uint64_t KEEP_MIN>
__device__ __forceinline__ uint64_t bitonic_cex(uint64_t v)
{
    const uint64_t o = ((uint64_t)xor_lane32<J>((uint32_t)(v >> 32)) << 32) | xor_lane32<J>((uint32_t)v);
    const uint64_t lt = __builtin_amdgcn_ballot_w64(v < o);
    return __builtin_amdgcn_inverse_ballot_w64(~(lt ^ KEEP_MIN)) ? v : o; // (own < other) == keep_min ? own : other
}

// the j = SIZE/2 ... 1 stages of block size SIZE: sorts a sequence that is bitonic within each block
template <int SIZE, bool DESC>
__device__ __forceinline__ uint64_t bitonic_merge(uint64_t v)
{
    if constexpr (SIZE >= 64)
        v = bitonic_cex<32, bitonic_keep_min(SIZE, 32, DESC)>(v);
    if constexpr (SIZE >= 32)
        v = bitonic_cex<16, bitonic_keep_min(SIZE, 16, DESC)>(v);
    if constexpr (SIZE >= 16)
        v = bitonic_cex<8, bitonic_keep_min(SIZE, 8, DESC)>(v);
    if constexpr (SIZE >= 8)
        v = bitonic_cex<4, bitonic_keep_min(SIZE, 4, DESC)>(v);
    if constexpr (SIZE >= 4)
        v = bitonic_cex<2, bitonic_keep_min(SIZE, 2, DESC)>(v);
    return bitonic_cex<1, bitonic_keep_min(SIZE, 1, DESC)>(v);
}

// lane l <- the (l+1)-th largest of the wave's 64 keys
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t v)
{
    v = bitonic_merge<2, true>(v);
    v = bitonic_merge<4, true>(v);
    v = bitonic_merge<8, true>(v);
    v = bitonic_merge<16, true>(v);
    v = bitonic_merge<32, true>(v);
    return bitonic_merge<64, true>(v);
}

__device__ __forceinline__ uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t max_u64(uint64_t a, uint64_t b) { return a < b ? b : a; }

// more fresh keys than this below the k-th best: sort them and merge instead of inserting one by one
constexpr uint32_t kBulkOfferMin = 6;

// ---- per-wave sorted top-k of composite keys ----------------------------------
// Composite keys are unique (they embed the row), so insertion order is irrelevant
// and "strictly smaller than the current k-th" is the exact acceptance test: a row
// that ties the k-th distance has a larger row index than every list entry of that
// distance only if it was seen later -- and then it must lose (stable sort +
// truncate, rank.rs:170-173).  The composite order encodes exactly that.

// k <= 64: lane i holds the i-th smallest key; no memory traffic at all.
struct WaveTopKReg {
    static constexpr bool kInRegisters = true;
    static constexpr bool kWriteAll = false;
    uint64_t e;
    uint64_t tau;
    uint32_t k;
    __device__ __forceinline__ void init(uint32_t k_, uint64_t *)
    {
        k = k_;
        e = kSentinel;
        tau = kSentinel;
    }
    __device__ __forceinline__ void insert(uint64_t c)
    {
        const uint32_t lane = lane_id();
        uint32_t p = (uint32_t)__popcll(__ballot(e < c));
        uint64_t up = shfl_up1_u64(e);
        e = lane < p ? e : (lane == p ? c : up);
        tau = readlane64(e, k - 1);
    }
    __device__ __forceinline__ void offer_some(uint64_t m, uint64_t comp)
    {
        while (m) {
            uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            uint64_t c = readlane64(comp, l);
            if (c < tau)
                insert(c);
        }
    }
    __device__ __forceinline__ void offer(uint64_t comp) { offer_some(__ballot(comp < tau), comp); }
    __device__ __forceinline__ uint64_t rank(uint32_t i) const { return readlane64(e, i); } // i-th smallest, i < 64
    // 64 keys in descending order (lane l: the (l+1)-th largest): the list becomes the 64 smallest of the union.
    // (All 64 lanes of e stay sorted, entries beyond k included: min(ascending, descending) is bitonic.)
    __device__ __forceinline__ void merge_desc(uint64_t r)
    {
        e = bitonic_merge<64, false>(min_u64(e, r));
        tau = readlane64(e, k - 1);
    }
    // offer() for a vector that may hold many keys below the k-th best (a wave's first row-block, list heads)
    __device__ __forceinline__ void offer_dense(uint64_t comp)
    {
        const uint64_t m = __ballot(comp < tau);
        if ((uint32_t)__popcll(m) > kBulkOfferMin) // wave-uniform
            merge_desc(wave_sort_desc(comp));
        else
            offer_some(m, comp);
    }
    // merge a published list (ascending, kSentinel padded to 64 entries: spill())
    __device__ __forceinline__ void merge_list(const uint64_t *other) { merge_desc(other[63u - lane_id()]); }
    // entry i (i < 64) for lane i
    __device__ __forceinline__ uint64_t mine() const { return e; }
    __device__ __forceinline__ void store(uint64_t *dst, size_t stride = 1) const
    {
        if (lane_id() < k)
            dst[lane_id() * stride] = e;
    }
    // publish the list (kSentinel padded to 64 entries) into this wave's LDS row
    __device__ __forceinline__ void spill(uint64_t *mine) const
    {
        mine[lane_id()] = kSentinel;
        store(mine);
    }
};

// 64 < k <= 128: two registers per lane (e0 = ranks 0..63, e1 = ranks 64..127); an insert is two
// ballots, two DPP shifts and the carry of rank 63 into rank 64 -- about twice the k <= 64 cost and
// still several times cheaper than the LDS list below.
struct WaveTopKReg2 {
    static constexpr bool kInRegisters = true;
    static constexpr bool kWriteAll = false;
    uint64_t e0, e1;
    uint64_t tau;
    uint32_t k;
    __device__ __forceinline__ void init(uint32_t k_, uint64_t *)
    {
        k = k_;
        e0 = e1 = kSentinel;
        tau = kSentinel;
    }
    __device__ __forceinline__ void insert(uint64_t c)
    {
        const uint32_t lane = lane_id();
        const uint32_t p = (uint32_t)__popcll(__ballot(e0 < c)) + (uint32_t)__popcll(__ballot(e1 < c));
        if (p < 64u) { // wave-uniform
            const uint64_t carry = readlane64(e0, 63);
            const uint64_t up0 = shfl_up1_u64(e0), up1 = shfl_up1_u64(e1);
            e0 = lane < p ? e0 : (lane == p ? c : up0);
            e1 = lane == 0u ? carry : up1;
        } else {
            const uint32_t q = p - 64u;
            const uint64_t up1 = shfl_up1_u64(e1);
            e1 = lane < q ? e1 : (lane == q ? c : up1);
        }
        tau = k <= 64u ? readlane64(e0, k - 1u) : readlane64(e1, k - 65u);
    }
    __device__ __forceinline__ void offer_some(uint64_t m, uint64_t comp)
    {
        while (m) {
            uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            uint64_t c = readlane64(comp, l);
            if (c < tau)
                insert(c);
        }
    }
    __device__ __forceinline__ void offer(uint64_t comp) { offer_some(__ballot(comp < tau), comp); }
    __device__ __forceinline__ uint64_t rank(uint32_t i) const { return readlane64(e0, i); } // i-th smallest, i < 64
    // 128 keys in descending order (r_lo lane l: the (l+1)-th largest, r_hi: the (l+65)-th): the list becomes the
    // 128 smallest of the union -- elementwise min with the ascending list is bitonic over 128, one exchange between
    // the registers and a 64-merge of each sorts it
    __device__ __forceinline__ void merge_desc(uint64_t r_lo, uint64_t r_hi)
    {
        const uint64_t lo = min_u64(e0, r_lo), hi = min_u64(e1, r_hi);
        e0 = bitonic_merge<64, false>(min_u64(lo, hi));
        e1 = bitonic_merge<64, false>(max_u64(lo, hi));
        tau = k <= 64u ? readlane64(e0, k - 1u) : readlane64(e1, k - 65u);
    }
    __device__ __forceinline__ void offer_dense(uint64_t comp)
    {
        const uint64_t m = __ballot(comp < tau);
        if ((uint32_t)__popcll(m) > kBulkOfferMin) // wave-uniform
            merge_desc(kSentinel, wave_sort_desc(comp));
        else
            offer_some(m, comp);
    }
    // merge a published list (ascending, kSentinel padded to 128 entries: spill())
    __device__ __forceinline__ void merge_list(const uint64_t *other)
    {
        merge_desc(other[127u - lane_id()], other[63u - lane_id()]);
    }
    __device__ __forceinline__ void store(uint64_t *dst, size_t stride = 1) const
    {
        if (lane_id() < k)
            dst[lane_id() * stride] = e0;
        if (lane_id() + 64u < k)
            dst[(lane_id() + 64u) * stride] = e1;
    }
    // publish the list (kSentinel padded to 128 entries) into this wave's LDS row
    __device__ __forceinline__ void spill(uint64_t *mine) const
    {
        mine[lane_id()] = kSentinel;
        mine[lane_id() + 64u] = kSentinel;
        store(mine);
    }
};

// any k: the list lives in LDS (one per wave), ascending, kSentinel padded.
struct WaveTopKLds {
    static constexpr bool kInRegisters = false;
    static constexpr bool kWriteAll = false;
    uint64_t *L;
    uint64_t tau;
    uint32_t k;
    __device__ __forceinline__ void init(uint32_t k_, uint64_t *lds)
    {
        k = k_;
        L = lds;
        for (uint32_t i = lane_id(); i < k; i += 64)
            L[i] = kSentinel;
        tau = kSentinel;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void insert(uint64_t c)
    {
        const uint32_t lane = lane_id();
        uint32_t p = 0;
        for (uint32_t base = 0; base < k; base += 64) {
            uint32_t idx = base + lane;
            uint64_t v = idx < k ? L[idx] : kSentinel;
            p += (uint32_t)__popcll(__ballot(v < c));
        }
        // shift [p, k-2] up by one, top chunk first; one wave = one instruction
        // stream, LDS executes a wave's instructions in order, the barriers only
        // stop the compiler from moving a chunk's store above the next chunk's load.
        for (int base = (int)(((k - 1) / 64) * 64); base >= 0; base -= 64) {
            if ((uint32_t)base + 63u < p)
                break;
            uint32_t idx = (uint32_t)base + lane;
            uint64_t v = idx < k ? L[idx] : kSentinel;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (idx >= p && idx + 1 < k)
                L[idx + 1] = v;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0)
            L[p] = c;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        tau = L[k - 1];
    }
    __device__ __forceinline__ void offer(uint64_t comp)
    {
        uint64_t m = __ballot(comp < tau);
        while (m) {
            uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            uint64_t c = readlane64(comp, l);
            if (c < tau)
                insert(c);
        }
    }
    __device__ __forceinline__ void offer_dense(uint64_t comp) { offer(comp); }
    __device__ __forceinline__ uint64_t rank(uint32_t i) const { return L[i]; }
    // merge another wave's list of k entries (ascending)
    __device__ __forceinline__ void merge_list(const uint64_t *other)
    {
        for (uint32_t base = 0; base < k; base += 64) {
            const uint32_t idx = base + lane_id();
            offer(idx < k ? other[idx] : kSentinel);
        }
    }
    __device__ __forceinline__ void store(uint64_t *dst, size_t stride = 1) const
    {
        for (uint32_t i = lane_id(); i < k; i += 64)
            dst[i * stride] = L[i];
    }
    __device__ __forceinline__ void spill(uint64_t *) const {} // the list already lives in the wave's LDS row
};

// No list at all: the scan writes every row's composite key to HBM ([query][row], kSentinel for rows that do not
// exist) and a radix select over them finds the k smallest (result_limit > 128: vfsx_kernels.hip, "long results").
struct WaveWriteAll {
    static constexpr bool kInRegisters = true;
    static constexpr bool kWriteAll = true;
    uint64_t tau;
    __device__ __forceinline__ void init(uint32_t, uint64_t *) { tau = kSentinel; }
    __device__ __forceinline__ void offer(uint64_t) {}
    __device__ __forceinline__ void offer_dense(uint64_t) {}
    __device__ __forceinline__ uint64_t rank(uint32_t) const { return kSentinel; }
    __device__ __forceinline__ void merge_list(const uint64_t *) {}
    __device__ __forceinline__ void store(uint64_t *, size_t = 1) const {}
    __device__ __forceinline__ void spill(uint64_t *) const {}
};

// ---- distance accumulation -----------------------------------------------------
// `q` is wave-uniform (kernel argument + uniform offsets): the compiler emits
// s_load and feeds SGPR operands.  Strictly sequential per lane.
// Cosine also needs ||q|| (rank.rs:296, recomputed per candidate by the reference):
// with WITH_QQ every lane folds q_j*q_j into a second accumulator in the same j
// order while it walks its first row-block -- the VALU has >10x headroom under the
// HBM stream, so this costs no time and saves a kernel launch per query.
struct Acc {
    float a;  // sum (q-x)^2  or  sum q*x
    float qq; // sum q*q (cosine, first row-block of the wave only)
};

template <int METRIC, bool WITH_QQ>
__device__ __forceinline__ void acc_step(Acc &acc, float qv, float xv)
{
    if (METRIC == VFSX_EUCLIDEAN) { // rank.rs:284: (a - b).powi(2), a = query
        float diff = f_sub(qv, xv);
        acc.a = f_add(acc.a, f_mul(diff, diff));
    } else { // rank.rs:293: a * b
        acc.a = f_add(acc.a, f_mul(qv, xv));
        if (WITH_QQ)
            acc.qq = f_add(acc.qq, f_mul(qv, qv));
    }
}

// query n of a group, or the group's last real query when the batch ends inside the group (queries read
// straight from the caller's buffer: no row past the batch is touched)
__device__ __forceinline__ uint32_t q_slot(int n, uint32_t n_last) { return (uint32_t)n < n_last ? (uint32_t)n : n_last; }

// NQ queries share one pass over the row: element x_j is loaded once and folded into
// NQ independent accumulators.  `q` points at query 0's chunk, `qs` is the stride
// (floats) to the next query.  NQ == 1: q is global + wave-uniform -> s_load/SGPR.
// NQ > 1: q is the workgroup's LDS copy; every lane reads the same address (broadcast).
template <int METRIC, bool WITH_QQ, int NQ>
__device__ __forceinline__ void acc_chunk_f32(Acc (&acc)[NQ], const uint4 v, const float *__restrict__ q,
                                              uint32_t qs, uint32_t n_last = 0xffffffffu)
{
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const float4 qv = *(const float4 *)(q + (size_t)q_slot(n, n_last) * qs);
        acc_step<METRIC, WITH_QQ>(acc[n], qv.x, __uint_as_float(v.x));
        acc_step<METRIC, WITH_QQ>(acc[n], qv.y, __uint_as_float(v.y));
        acc_step<METRIC, WITH_QQ>(acc[n], qv.z, __uint_as_float(v.z));
        acc_step<METRIC, WITH_QQ>(acc[n], qv.w, __uint_as_float(v.w));
    }
}

// i8: the byte indexes a per-bank replica of the 256-entry dequant table in LDS
// (lut_s[u * 32 + (lane & 31)]): lane l always hits bank l & 31 -> conflict free.
// Dequantized once per element, shared by the NQ queries.
template <int METRIC, bool WITH_QQ, int NQ>
__device__ __forceinline__ void acc_word_i8(Acc (&acc)[NQ], uint32_t w, const float *__restrict__ q, uint32_t qs,
                                            const float *lut_lane, uint32_t n_last = 0xffffffffu)
{
    float xv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        xv[b] = lut_lane[((w >> (8 * b)) & 0xffu) * 32u]; // (q as f32) / scale, vector.rs:135-139
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const float4 qv = *(const float4 *)(q + (size_t)q_slot(n, n_last) * qs);
        acc_step<METRIC, WITH_QQ>(acc[n], qv.x, xv[0]);
        acc_step<METRIC, WITH_QQ>(acc[n], qv.y, xv[1]);
        acc_step<METRIC, WITH_QQ>(acc[n], qv.z, xv[2]);
        acc_step<METRIC, WITH_QQ>(acc[n], qv.w, xv[3]);
    }
}

template <int METRIC, bool WITH_QQ, int NQ>
__device__ __forceinline__ void acc_chunk_i8(Acc (&acc)[NQ], const uint4 v, const float *__restrict__ q,
                                             uint32_t qs, const float *lut_lane, uint32_t n_last = 0xffffffffu)
{
    acc_word_i8<METRIC, WITH_QQ, NQ>(acc, v.x, q + 0, qs, lut_lane, n_last);
    acc_word_i8<METRIC, WITH_QQ, NQ>(acc, v.y, q + 4, qs, lut_lane, n_last);
    acc_word_i8<METRIC, WITH_QQ, NQ>(acc, v.z, q + 8, qs, lut_lane, n_last);
    acc_word_i8<METRIC, WITH_QQ, NQ>(acc, v.w, q + 12, qs, lut_lane, n_last);
}

template <bool NT>
__device__ __forceinline__ uint4 load_chunk(const uint4 *p)
{
    if (NT) {
        // streamed once per query: nontemporal (global_load_dwordx4 ... nt)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 t = __builtin_nontemporal_load((const u32x4 *)p);
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    return *p;
}

// one row-block: lane l walks row 64*blk + l through all d_chunks chunks
template <int DTYPE, int METRIC, int UNROLL, bool NT, bool WITH_QQ, int NQ>
__device__ __forceinline__ void scan_rowblock(Acc (&acc)[NQ], const uint4 *__restrict__ p,
                                              const float *__restrict__ q, uint32_t qs, const float *lut_lane,
                                              uint32_t d_chunks, uint32_t n_last = 0xffffffffu)
{
    constexpr uint32_t EPC = DTYPE == VFSX_I8 ? 16u : 4u;
    uint32_t c = 0;
    for (; c + UNROLL <= d_chunks; c += UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            v[u] = load_chunk<NT>(p + (size_t)(c + u) * 64u);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (DTYPE == VFSX_I8)
                acc_chunk_i8<METRIC, WITH_QQ, NQ>(acc, v[u], q + (c + u) * EPC, qs, lut_lane, n_last);
            else
                acc_chunk_f32<METRIC, WITH_QQ, NQ>(acc, v[u], q + (c + u) * EPC, qs, n_last);
        }
    }
    for (; c < d_chunks; ++c) {
        uint4 v = load_chunk<NT>(p + (size_t)c * 64u);
        if (DTYPE == VFSX_I8)
            acc_chunk_i8<METRIC, WITH_QQ, NQ>(acc, v, q + c * EPC, qs, lut_lane, n_last);
        else
            acc_chunk_f32<METRIC, WITH_QQ, NQ>(acc, v, q + c * EPC, qs, n_last);
    }
}

} // namespace
} // namespace vfsx
