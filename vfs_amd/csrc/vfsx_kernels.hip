// vfsx_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the VFS Ranker
// linear-search path.  Built with -ffp-contract=off: every f32 operation below is
// one correctly-rounded IEEE-754 binary32 operation, in the reference's order
// (vfs/src/vfs/rank.rs:271-300), so distances are bit-identical to the reference's
// scalar (and ordered-reduce SIMD) arithmetic.
//
// Design (DESIGN.md has the long form):
//   * the corpus sits in HBM in the "rb64" layout (vfsx_internal.h): one wavefront
//     owns 64 rows, lane l accumulates row l strictly sequentially, and every
//     global_load_dwordx4 of the wave still reads 1 KiB contiguous;
//   * the query is wave-uniform -> scalar loads, SGPR operands;
//   * top-k is fused: each wave keeps a sorted list of composite keys
//     (monotone f32 key << 32 | local row) in registers (k <= 64) or LDS, and only
//     touches it when a ballot says a lane beat the current k-th best;
//   * one list per workgroup goes to HBM; a tiny select kernel merges them.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vfsx_device.h"
#include "vfsx_internal.h"

// No FMA contraction anywhere in this file, whatever the command line says.
#pragma clang fp contract(off)

namespace vfsx {

namespace {

// ---- the scan kernel -------------------------------------------------------------
// grid = (n_lists, ceil(nq / NQ)); block = 256 = 4 waves.  Wave w of the launch walks
// row-blocks w, w + W, w + 2W, ... in ascending order, NQ queries at a time.
// dynamic LDS: [4 waves][NQ][max(k,64)] u64 lists, then (NQ > 1) [NQ][q_stride] f32 queries.
template <int DTYPE, int METRIC, typename TopK, int UNROLL, bool NT, int NQ, bool PIPE>
__global__ __launch_bounds__(kBlock) void scan_kernel(
    const uint4 *__restrict__ corpus, const float *__restrict__ row_norms,
    const float *__restrict__ queries, const float *__restrict__ lut, uint64_t *__restrict__ lists,
    uint32_t n_rows, uint32_t d_chunks, uint32_t n_rowblocks, uint32_t q_stride, uint32_t k,
    uint32_t n_lists, uint32_t nq)
{
    extern __shared__ __align__(16) uint64_t dyn_lds[];
    __shared__ float lut_s[DTYPE == VFSX_I8 ? 256 * 32 : 1];

    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t q0 = blockIdx.y * NQ; // first query of this group
    const uint32_t list_stride = topk_list_stride(k);

    if (DTYPE == VFSX_I8) {
        for (uint32_t i = threadIdx.x; i < 256u * 32u; i += kBlock)
            lut_s[i] = lut[i >> 5];
    }
    const float *lut_lane = lut_s + (lane & 31u);

    // query source.  One query: wave-uniform global reads (s_load, SGPR operands).  Several: the same for
    // int8 stores and for Cosine (MI355X: 11-22 % faster than the LDS copy, whose broadcast reads cost full
    // LDS bandwidth), the LDS copy for f32 Euclidean (8-15 % faster there).
    constexpr bool QDIRECT = NQ > 1 && (DTYPE == VFSX_I8 || METRIC == VFSX_COSINE);
    const float *__restrict__ q;
    uint32_t qs;
    uint32_t n_last = 0xffffffffu;
    if (NQ == 1) {
        q = queries + (size_t)q0 * q_stride;
        qs = 0;
    } else if (QDIRECT) {
        q = queries + (size_t)q0 * q_stride;
        qs = q_stride;
        n_last = min((uint32_t)NQ, nq - q0) - 1u; // ragged last group: repeat the last query
    } else {
        float *q_s = (float *)(dyn_lds + (size_t)kWavesPerBlock * NQ * list_stride);
        for (uint32_t i = threadIdx.x; i < NQ * q_stride; i += kBlock) {
            uint32_t n = i / q_stride, j = i - n * q_stride;
            uint32_t qi = q0 + n < nq ? q0 + n : nq - 1; // ragged last group: repeat the last query
            q_s[i] = queries[(size_t)qi * q_stride + j];
        }
        q = q_s;
        qs = q_stride;
    }
    if (DTYPE == VFSX_I8 || (NQ > 1 && !QDIRECT))
        __syncthreads();

    TopK top[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n)
        top[n].init(k, dyn_lds + ((size_t)wave * NQ + n) * list_stride);

    const uint32_t n_waves = gridDim.x * kWavesPerBlock;
    const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wave;
    float qn[NQ]; // ||q||, identical in every lane
#pragma unroll
    for (int n = 0; n < NQ; ++n)
        qn[n] = 0.0f;

    // distance of row (64 blk + lane) from its finished accumulators, offered to the wave's lists
    auto finish_rowblock = [&](uint32_t blk, Acc(&acc)[NQ]) {
        const uint32_t row = blk * 64u + lane;
        float rn = 0.0f;
        if (METRIC == VFSX_COSINE && row < n_rows)
            rn = row_norms[row];
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            uint64_t comp = kSentinel;
            if (row < n_rows) {
                float dist;
                if (METRIC == VFSX_EUCLIDEAN) {
                    dist = f_sqrt(acc[n].a); // rank.rs:286
                } else {                     // rank.rs:299: 1.0 - (dot / (norm_1 * norm_2))
                    float den = f_mul(qn[n], rn);
                    dist = f_sub(1.0f, f_div(acc[n].a, den));
                }
                comp = ((uint64_t)dist_key(dist) << 32) | row;
            }
            if (TopK::kWriteAll) { // `lists` is the key buffer [nq][64 n_rowblocks]: one coalesced 512 B store per wave
                if (q0 + n < nq)
                    lists[(size_t)(q0 + n) * n_rowblocks * 64u + row] = comp;
            } else {
                top[n].offer_dense(comp); // (a wave's first row-block: 64 keys into an empty list)
            }
        }
    };

    uint32_t blk = wave_global;
    if (METRIC == VFSX_COSINE && blk < n_rowblocks) {
        // first row-block of the wave: also folds sum q_j^2 (see Acc)
        const uint4 *__restrict__ p = corpus + (size_t)blk * d_chunks * 64u + lane;
        Acc acc[NQ];
#pragma unroll
        for (int n = 0; n < NQ; ++n)
            acc[n] = Acc{0.0f, 0.0f};
        scan_rowblock<DTYPE, METRIC, UNROLL, NT, true, NQ>(acc, p, q, qs, lut_lane, d_chunks, n_last);
#pragma unroll
        for (int n = 0; n < NQ; ++n)
            qn[n] = f_sqrt(acc[n].qq); // rank.rs:296
        finish_rowblock(blk, acc);
        blk += n_waves;
    }
    if (PIPE && d_chunks % UNROLL == 0) {
        // software pipeline over the flat sequence of (row-block, chunk group): the next group's
        // UNROLL loads (possibly the next row-block's first group) are in flight while the current
        // group is accumulated, so a wave never sits with an empty memory queue.
        constexpr uint32_t EPC = DTYPE == VFSX_I8 ? 16u : 4u;
        const uint32_t groups = d_chunks / UNROLL;
        uint4 cur[UNROLL], nxt[UNROLL];
        if (blk < n_rowblocks) {
            const uint4 *__restrict__ p0 = corpus + (size_t)blk * d_chunks * 64u + lane;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                cur[u] = load_chunk<NT>(p0 + (size_t)u * 64u);
        }
        for (; blk < n_rowblocks; blk += n_waves) {
            const uint4 *__restrict__ p = corpus + (size_t)blk * d_chunks * 64u + lane;
            const uint32_t nblk = blk + n_waves;
            Acc acc[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n)
                acc[n] = Acc{0.0f, 0.0f};
            for (uint32_t g = 0; g < groups; ++g) {
                if (g + 1 < groups) {
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
                        nxt[u] = load_chunk<NT>(p + (size_t)((g + 1) * UNROLL + u) * 64u);
                } else if (nblk < n_rowblocks) {
                    const uint4 *__restrict__ pn = corpus + (size_t)nblk * d_chunks * 64u + lane;
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
                        nxt[u] = load_chunk<NT>(pn + (size_t)u * 64u);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if (DTYPE == VFSX_I8)
                        acc_chunk_i8<METRIC, false, NQ>(acc, cur[u], q + (g * UNROLL + u) * EPC, qs, lut_lane, n_last);
                    else
                        acc_chunk_f32<METRIC, false, NQ>(acc, cur[u], q + (g * UNROLL + u) * EPC, qs, n_last);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    cur[u] = nxt[u];
            }
            finish_rowblock(blk, acc);
        }
    } else {
        for (; blk < n_rowblocks; blk += n_waves) {
            const uint4 *__restrict__ p = corpus + (size_t)blk * d_chunks * 64u + lane;
            Acc acc[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n)
                acc[n] = Acc{0.0f, 0.0f};
            scan_rowblock<DTYPE, METRIC, UNROLL, NT, false, NQ>(acc, p, q, qs, lut_lane, d_chunks, n_last);
            finish_rowblock(blk, acc);
        }
    }

    if (TopK::kWriteAll)
        return;
    // fold the 4 waves' lists into wave 0's, then one list per workgroup and query to HBM
    __syncthreads(); // (TopKLds: lists already live there; Reg: publish them)
#pragma unroll
    for (int n = 0; n < NQ; ++n)
        top[n].spill(dyn_lds + ((size_t)wave * NQ + n) * list_stride); // (a no-op for lists that live in LDS)
    __syncthreads();
    for (uint32_t s = 1; s < (uint32_t)kWavesPerBlock; s <<= 1) { // merge tree: wave 2 takes 3's lists while 0 takes 1's
        if ((wave & (2u * s - 1u)) == 0u) {
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                top[n].merge_list(dyn_lds + ((size_t)(wave + s) * NQ + n) * list_stride);
                top[n].spill(dyn_lds + ((size_t)wave * NQ + n) * list_stride);
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            // column-major: entry j of every list is contiguous ([q][j][list]) for the select walk
            if (q0 + n < nq)
                top[n].store(lists + (size_t)(q0 + n) * n_lists * k + blockIdx.x, n_lists);
        }
    }
}

// ---- single-query int8 scan, 16 waves per workgroup ------------------------------------------
// The faithful int8 kernel above is bound by VALU issue: 5 lane-ops per element, two of them only to
// turn a byte into an LDS address (extract, then shift-or with the lane's bank offset).  With a 64 KiB
// table -- entry (byte, lane) at byte offset  byte * 256 + lane * 4, i.e. one private, conflict-free
// bank column per lane -- that address is a single v_perm_b32 (byte k of the data word into address
// byte 1, the lane offset into byte 0).  64 KiB of LDS per workgroup only pays with many waves behind
// it: 1024 threads = 16 waves share one table, one workgroup per CU.
constexpr int kWideWaves = 16;

template <int METRIC, bool WITH_QQ>
__device__ __forceinline__ void acc_chunk_i8w(Acc &acc, const uint4 v, const float *__restrict__ q,
                                              const char *lut_b, uint32_t lane4)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // address = byte << 8 | lane * 4: selector bytes (LSB first) = S1.byte0, S0.byte_b, 0, 0
        const uint32_t off = __builtin_amdgcn_perm(w[i >> 2], lane4, 0x0c0c0000u | ((4u + (uint32_t)(i & 3)) << 8));
        x[i] = *(const float *)(lut_b + off); // (q as f32) / scale, vector.rs:135-139
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 qv = *(const float4 *)(q + 4 * i);
        acc_step<METRIC, WITH_QQ>(acc, qv.x, x[4 * i + 0]);
        acc_step<METRIC, WITH_QQ>(acc, qv.y, x[4 * i + 1]);
        acc_step<METRIC, WITH_QQ>(acc, qv.z, x[4 * i + 2]);
        acc_step<METRIC, WITH_QQ>(acc, qv.w, x[4 * i + 3]);
    }
}

template <int METRIC, int UNROLL, bool NT>
__global__ __launch_bounds__(64 * kWideWaves) void scan_i8w_kernel(
    const uint4 *__restrict__ corpus, const float *__restrict__ row_norms, const float *__restrict__ queries,
    const float *__restrict__ lut, uint64_t *__restrict__ lists, uint32_t n_rows, uint32_t d_chunks,
    uint32_t n_rowblocks, uint32_t q_stride, uint32_t k, uint32_t n_lists)
{
    __shared__ __align__(16) float lut_s[256 * 64]; // exactly 64 KiB: two workgroups fit one CU's 160 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 256u * 64u; i += 64u * kWideWaves)
        lut_s[i] = lut[i >> 6];
    __syncthreads();
    const char *lut_b = (const char *)lut_s;
    const uint32_t lane4 = lane * 4u;
    const float *__restrict__ q = queries + (size_t)blockIdx.y * q_stride; // wave-uniform: s_load

    WaveTopKReg top;
    top.init(k, nullptr);
    const uint32_t n_waves = gridDim.x * kWideWaves;
    float qn = 0.0f;

    auto rowblock = [&](uint32_t blk, auto with_qq) __attribute__((always_inline)) {
        constexpr bool QQ = decltype(with_qq)::value;
        const uint4 *__restrict__ p = corpus + (size_t)blk * d_chunks * 64u + lane;
        Acc acc{0.0f, 0.0f};
        uint32_t c = 0;
        for (; c + UNROLL <= d_chunks; c += UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                v[u] = load_chunk<NT>(p + (size_t)(c + u) * 64u);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                acc_chunk_i8w<METRIC, QQ>(acc, v[u], q + (c + u) * 16u, lut_b, lane4);
        }
        for (; c < d_chunks; ++c)
            acc_chunk_i8w<METRIC, QQ>(acc, load_chunk<NT>(p + (size_t)c * 64u), q + c * 16u, lut_b, lane4);
        if (QQ)
            qn = f_sqrt(acc.qq); // rank.rs:296
        const uint32_t row = blk * 64u + lane;
        uint64_t comp = kSentinel;
        if (row < n_rows) {
            float dist;
            if (METRIC == VFSX_EUCLIDEAN)
                dist = f_sqrt(acc.a); // rank.rs:286
            else
                dist = f_sub(1.0f, f_div(acc.a, f_mul(qn, row_norms[row]))); // rank.rs:299
            comp = ((uint64_t)dist_key(dist) << 32) | row;
        }
        top.offer_dense(comp);
    };

    // Whole rounds: wave w of workgroup b takes row-block round * n_waves + 16 b + w, so a workgroup reads 16
    // adjacent row-blocks at a time (measured: wave-major strides through the corpus cost 10 % at C3).  The
    // n_rowblocks % n_waves row-blocks of the last, partial round go wave-major instead -- to wave 0 (1, ...) of
    // EVERY workgroup rather than to all 16 waves of the first few: the kernel is VALU-bound, a CU is done when
    // its slowest workgroup is, and this way no workgroup carries 16 extra row-blocks.
    const uint32_t n_whole = (n_rowblocks / n_waves) * n_waves;
#ifdef VFSX_EXP_I8W_BLOCKMAJOR
    const uint32_t tail = n_whole + blockIdx.x * kWideWaves + wave;
#else
    const uint32_t tail = n_whole + wave * gridDim.x + blockIdx.x;
#endif
    uint32_t blk = blockIdx.x * kWideWaves + wave;
    bool have_qn = METRIC != VFSX_COSINE;
    if (METRIC == VFSX_COSINE && blk < n_whole) { // first row-block also folds sum q_j^2
        rowblock(blk, std::true_type{});
        blk += n_waves;
        have_qn = true;
    }
    for (; blk < n_whole; blk += n_waves)
        rowblock(blk, std::false_type{});
    if (tail < n_rowblocks) {
        if (have_qn)
            rowblock(tail, std::false_type{});
        else
            rowblock(tail, std::true_type{});
    }

    // fold the 16 waves' lists into wave 0's, one list per workgroup to HBM ([q][j][list]); the
    // table is dead once every wave has left the scan, its LDS holds the 16 lists
    __syncthreads();
    uint64_t *dyn_lds = (uint64_t *)lut_s;
    uint64_t *mine = dyn_lds + (size_t)wave * 64u;
    top.spill(mine);
    __syncthreads();
    for (uint32_t s = 1; s < (uint32_t)kWideWaves; s <<= 1) { // merge tree: 4 levels of 6 stages instead of 15 merges by wave 0
        if ((wave & (2u * s - 1u)) == 0u) {
            top.merge_list(dyn_lds + (size_t)(wave + s) * 64u);
            top.spill(mine); // read at the next level; this level reads the rows of waves that do not merge
        }
        __syncthreads();
    }
    if (wave == 0)
        top.store(lists + (size_t)blockIdx.y * n_lists * k + blockIdx.x, n_lists);
}

// ---- select: merge the per-workgroup lists of one query, emit hits ----------------
// One workgroup per query.  Lists arrive column-major ([j][list], each list ascending).
// A lane owns one list and walks it only while its entries still beat the wave's
// current k-th best (tau only shrinks, entries only grow): with random data that is
// 2-3 columns, i.e. one HBM/L2 round trip of four columns, instead of reading and offering everything.
template <typename TopK, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void select_kernel(const uint64_t *__restrict__ lists,
                                                            const uint64_t *__restrict__ ids,
                                                            vfsx_hit *__restrict__ out, uint64_t row_base,
                                                            uint64_t id_base, uint32_t n_lists, uint32_t k,
                                                            uint32_t out_stride)
{
    constexpr uint32_t BLOCK = 64u * WAVES;
    extern __shared__ __align__(16) uint64_t dyn_lds[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t qi = blockIdx.x;
    const uint32_t list_stride = topk_list_stride(k);
    const uint64_t *__restrict__ src = lists + (size_t)qi * n_lists * k;

    TopK top;
    top.init(k, dyn_lds + wave * list_stride);
    // `bound`: an upper bound of the k-th smallest key of ALL lists, known after the heads -- a wave alone only
    // learns the k-th smallest of ITS 64 lists, which prunes nothing while k is of the order of 64 (k = 100: a wave
    // needs two columns just to fill its list).  With `full` waves holding 64 lists each and m = ceil(k / full),
    // the m smallest heads of every such wave are >= k keys, all <= the largest of those m-th heads: that maximum
    // bounds the global k-th smallest.  One LDS atomic per wave; entries >= bound are dropped before they are offered.
    __shared__ unsigned long long s_bound;
    uint64_t bound = kSentinel;
    uint32_t lbase = wave * 64u, j0 = 0;
    bool alive0 = true;
    if (TopK::kInRegisters) { // (compile-time: every wave takes the barriers)
        if (threadIdx.x == 0)
            s_bound = 0ull;
        __syncthreads();
        // (only this workgroup's WAVES waves hold heads here: lists beyond 64 * WAVES are walked later and cannot
        // vouch for the bound -- VFSX_BLOCKS_PER_CU = 8 gives 2048 lists)
        const uint32_t full = min(n_lists / 64u, (uint32_t)WAVES), m = full ? (k + full - 1u) / full : 65u;
        const uint32_t list = lbase + lane;
        const bool in = list < n_lists;
        uint64_t c[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            c[u] = (in && u < k) ? src[(size_t)u * n_lists + list] : kSentinel;
        top.offer_dense(c[0]); // the heads of 64 lists meet an empty list: sorted and merged, not inserted
        if (m <= 64u && wave < full && lane == 0u)
            atomicMax(&s_bound, (unsigned long long)top.rank(m - 1u));
        __syncthreads();
        if (m <= 64u)
            bound = s_bound;
#pragma unroll
        for (uint32_t u = 1; u < 4u; ++u)
            top.offer_dense(c[u] < bound ? c[u] : kSentinel);
        alive0 = in && c[3] < bound && c[3] < top.tau;
        j0 = 4;
    }
    for (; lbase < n_lists; lbase += BLOCK, j0 = 0, alive0 = true) { // wave-uniform
        const uint32_t list = lbase + lane;
        bool alive = alive0 && list < n_lists;
        for (uint32_t j = j0; j < k; j += 4) {
            if (__ballot(alive) == 0)
                break;
            // four columns per round trip (the lists were just written by other CUs: every trip is ~2 us)
            uint64_t c[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                c[u] = (alive && j + u < k) ? src[(size_t)(j + u) * n_lists + list] : kSentinel;
                c[u] = c[u] < bound ? c[u] : kSentinel;
            }
            if (__ballot(c[0] < top.tau) == 0)
                break;
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u)
                top.offer_dense(c[u]);
            alive = alive && (c[3] < top.tau);
        }
    }

    uint64_t *mine = dyn_lds + wave * list_stride;
    __syncthreads();
    top.spill(mine);
    __syncthreads();
    if (TopK::kInRegisters) {
        // sorted register lists: a merge tree over the waves, 6 (k <= 64) or 13 compare-exchange stages per level
        for (uint32_t s = 1; s < (uint32_t)WAVES; s <<= 1) {
            if ((wave & (2u * s - 1u)) == 0u && wave + s < (uint32_t)WAVES &&
                (wave + s) * 64u < n_lists) { // (a partner wave without lists has nothing to merge)
                top.merge_list(dyn_lds + (wave + s) * list_stride);
                top.spill(mine); // read by the next level only: this level reads the slots of waves that do not merge
            }
            __syncthreads();
        }
    }
    if (wave == 0) {
        // lists in LDS: lane w walks wave w's (ascending) list while it still beats the k-th best
        bool alive = !TopK::kInRegisters && lane >= 1 && lane < (uint32_t)WAVES;
        const uint64_t *other = dyn_lds + (alive ? lane : 0u) * list_stride;
        for (uint32_t j = 0; !TopK::kInRegisters && j < k; ++j) {
            uint64_t c = alive ? other[j] : kSentinel;
            if (__ballot(c < top.tau) == 0)
                break;
            top.offer(c);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        top.store(mine);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        for (uint32_t j = lane; j < k; j += 64) {
            uint64_t comp = mine[j];
            vfsx_hit h;
            if (comp == kSentinel) {
                h.dist = __uint_as_float(0x7fc00000u);
                h.flags = 0;
                h.row = ~0ull;
                h.id = ~0ull;
            } else {
                uint32_t r = (uint32_t)comp;
                h.dist = key_dist((uint32_t)(comp >> 32));
                h.flags = 1;
                h.row = row_base + r;
                h.id = ids ? ids[r] : id_base + r;
            }
            out[(size_t)qi * out_stride + j] = h;
        }
    }
}

// ---- long results (result_limit > 128): radix select over every row's composite key ----------------------
// A wave scanning 256 rows cannot keep a 1000-entry list profitably (every row is an insert, and the select
// then merges ~1000 lists of 1000): C2 at k = 1000 took 10 ms per query that way.  Instead scan_kernel<WaveWriteAll>
// writes all composite keys (8 B per row: 0.3 % of the bytes the scan reads at d = 768) and the k smallest are
// found by an 8-pass radix select, most significant byte first.  Keys are unique (they embed the row), so after the
// last pass `prefix` IS the k-th smallest key above the query's cursor and exactly k keys lie in (cursor, prefix]:
// they are compacted, sorted by one workgroup (bitonic, <= kLongChunk entries) and emitted.  Longer results repeat
// select / compact / sort from the new cursor -- the scan runs ONCE however long the result is.
struct LongState { // per query
    uint64_t lo;     // cursor: only keys > lo count (has_lo != 0)
    uint64_t prefix; // the bytes of the k-th key decided so far (most significant first)
    uint32_t has_lo;
    uint32_t k_rem;  // rank of the wanted key among the keys that match `prefix`
    uint32_t count;  // compaction cursor
    uint32_t pad;
};

__device__ __forceinline__ bool long_in_play(uint64_t key, const LongState &st, uint32_t byte)
{
    if (st.has_lo && key <= st.lo)
        return false;
    return byte == 7u || (key >> (8u * (byte + 1u))) == (st.prefix >> (8u * (byte + 1u)));
}

__global__ void long_set_rank_kernel(LongState *__restrict__ state, uint32_t nq, uint32_t kp)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) {
        state[q].k_rem = kp;
        state[q].prefix = 0;
        state[q].count = 0;
    }
}

// grid (blocks, queries), 256 threads, kLongKeysPerBlock keys per block: histogram of byte `byte` over the keys in play
constexpr uint32_t kLongKeysPerBlock = 16384;
__global__ __launch_bounds__(256) void long_hist_kernel(const uint64_t *__restrict__ keys, uint32_t n,
                                                        const LongState *__restrict__ state, uint32_t *__restrict__ hist,
                                                        uint32_t byte)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const LongState st = state[blockIdx.y];
    const uint64_t *kq = keys + (size_t)blockIdx.y * n;
    const uint32_t i0 = blockIdx.x * kLongKeysPerBlock;
    const uint32_t i1 = min(n, i0 + kLongKeysPerBlock);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256u) {
        const uint64_t key = kq[i];
        if (long_in_play(key, st, byte))
            atomicAdd(&h[(uint32_t)(key >> (8u * byte)) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x])
        atomicAdd(&hist[blockIdx.y * 256u + threadIdx.x], h[threadIdx.x]);
}

// one wave per query: the bin in which the k_rem-th key in play falls becomes the next byte of `prefix`
__global__ __launch_bounds__(64) void long_pick_kernel(LongState *__restrict__ state, uint32_t *__restrict__ hist,
                                                       uint32_t byte)
{
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    uint32_t *hq = hist + q * 256u;
    uint32_t c[4], own = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = hq[lane * 4u + j];
        hq[lane * 4u + j] = 0; // ready for the next pass
        own += c[j];
    }
    uint32_t incl = own; // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o)
            incl += v;
    }
    LongState st = state[q];
    const uint32_t excl = incl - own;
    // the first bin whose inclusive count reaches k_rem (none: fewer keys than asked for -> the last bin, and the
    // select ends on kSentinel, which the compaction drops)
    uint32_t bin = 256u, below = 0;
    uint32_t run = excl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (bin == 256u && run + c[j] >= st.k_rem) {
            bin = lane * 4u + (uint32_t)j;
            below = run;
        }
        run += c[j];
    }
    const uint64_t has = __ballot(bin != 256u);
    uint32_t pick_bin = 255u, pick_below = 0;
    if (has) {
        const int l = __builtin_ctzll(has);
        pick_bin = (uint32_t)__shfl((int)bin, l, 64);
        pick_below = (uint32_t)__shfl((int)below, l, 64);
    } else {
        pick_below = (uint32_t)__shfl((int)(incl), 63, 64) - (uint32_t)__shfl((int)c[3], 63, 64); // keys below bin 255
    }
    if (lane == 0) {
        st.prefix |= (uint64_t)pick_bin << (8u * byte);
        st.k_rem = st.k_rem > pick_below ? st.k_rem - pick_below : 1u;
        state[q] = st;
    }
}

// grid (blocks, queries): the keys in (cursor, prefix] (kSentinel excluded) -> sel[q][0 .. count), unordered
__global__ __launch_bounds__(256) void long_compact_kernel(const uint64_t *__restrict__ keys, uint32_t n,
                                                           LongState *__restrict__ state, uint64_t *__restrict__ sel,
                                                           uint32_t sel_stride)
{
    const LongState st = state[blockIdx.y];
    const uint64_t *kq = keys + (size_t)blockIdx.y * n;
    const uint32_t i0 = blockIdx.x * kLongKeysPerBlock;
    const uint32_t i1 = min(n, i0 + kLongKeysPerBlock);
    for (uint32_t base = i0; base < i1; base += 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const uint64_t key = i < i1 ? kq[i] : kSentinel;
        const bool take = key != kSentinel && key <= st.prefix && !(st.has_lo && key <= st.lo);
        const uint64_t m = __ballot(take);
        if (m) { // one atomic per wave
            uint32_t pos0 = 0;
            if (lane_id() == 0)
                pos0 = atomicAdd(&state[blockIdx.y].count, (uint32_t)__popcll(m));
            pos0 = (uint32_t)__shfl((int)pos0, 0, 64);
            const uint32_t pos = pos0 + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
            if (take && pos < sel_stride)
                sel[(size_t)blockIdx.y * sel_stride + pos] = key;
        }
    }
}

// one 1024-thread workgroup per query: sort the compacted keys (bitonic, kLongChunk slots in LDS), emit the hits of
// this chunk at out[q * out_stride + out_off ..], move the cursor to the chunk's last key
constexpr uint32_t kLongChunk = 4096;
__global__ __launch_bounds__(1024) void long_sort_emit_kernel(const uint64_t *__restrict__ sel, LongState *__restrict__ state,
                                                              const uint64_t *__restrict__ ids, vfsx_hit *__restrict__ out,
                                                              uint64_t row_base, uint64_t id_base, uint32_t kp,
                                                              uint32_t out_stride, uint32_t out_off)
{
    __shared__ uint64_t s_key[kLongChunk];
    const uint32_t q = blockIdx.x;
    const uint32_t count = min(state[q].count, kp);
    for (uint32_t i = threadIdx.x; i < kLongChunk; i += 1024u)
        s_key[i] = i < count ? sel[(size_t)q * kLongChunk + i] : kSentinel;
    __syncthreads();
    for (uint32_t size = 2; size <= kLongChunk; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < kLongChunk / 2u; t += 1024u) {
                const uint32_t i = 2u * t - (t & (stride - 1u)); // the lower index of pair t at this stride
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0u;
                const uint64_t a = s_key[i], b = s_key[j];
                if ((a > b) == up) {
                    s_key[i] = b;
                    s_key[j] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t j = threadIdx.x; j < kp; j += 1024u) {
        const uint64_t comp = s_key[j];
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
        out[(size_t)q * out_stride + out_off + j] = h;
    }
    if (threadIdx.x == 0) { // next chunk: everything after this chunk's last key (kSentinel: nothing is left)
        LongState st = state[q];
        st.lo = count == kp ? s_key[kp - 1u] : kSentinel;
        st.has_lo = 1;
        st.prefix = 0;
        st.count = 0;
        state[q] = st;
    }
}

// ---- query padding: zero-padded copy when dim is not a multiple of the chunk ----
__global__ void query_pad_kernel(const float *__restrict__ q_in, uint32_t nq, uint32_t dim,
                                 float *__restrict__ q_pad, uint32_t q_stride)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)nq * q_stride)
        return;
    uint32_t qi = (uint32_t)(t / q_stride), j = (uint32_t)(t % q_stride);
    q_pad[t] = j < dim ? q_in[(size_t)qi * dim + j] : 0.0f;
}

__global__ void lut_kernel(float scale, float *__restrict__ lut)
{
    uint32_t u = threadIdx.x;                      // raw byte
    float f = (float)(int)(int8_t)(uint8_t)u;      // val as f32
    lut[u] = f_div(f, scale);                  // vector.rs:138
}

// ---- ingest: row-major staging -> rb64 ------------------------------------------------
__device__ __forceinline__ int quantize_one(float v, float scale)
{ // vector.rs:106 with Some(scale): (val * scale).round() as i8  (saturating, NaN -> 0)
    float p = f_mul(v, scale);
    float r = roundf(p); // half away from zero, exact
    if (r != r)
        return 0;
    if (r >= 127.0f)
        return 127;
    if (r <= -128.0f)
        return -128;
    return (int)r;
}

template <int MODE>
__global__ void ingest_kernel(const void *__restrict__ staging, uint64_t n, uint32_t dim,
                              uint4 *__restrict__ corpus, uint32_t d_chunks, uint64_t row0,
                              float scale, uint64_t first_blk, uint64_t total)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total)
        return;
    uint32_t lane = (uint32_t)(t & 63u);
    uint64_t cb = t >> 6;
    uint32_t c = (uint32_t)(cb % d_chunks);
    uint64_t b = first_blk + cb / d_chunks;
    uint64_t row = b * 64u + lane;
    if (row < row0 || row >= row0 + n)
        return;
    uint64_t i = row - row0;
    uint4 out;
    if (MODE == 0) {
        const float *src = (const float *)staging + i * dim;
        uint32_t j = c * 4u;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            f[e] = (j + e < dim) ? src[j + e] : 0.0f;
        out = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                         __float_as_uint(f[3]));
    } else {
        uint32_t j = c * 16u;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            uint32_t byte = 0;
            if (j + e < dim) {
                if (MODE == 1)
                    byte = ((const uint8_t *)staging)[i * dim + j + e];
                else
                    byte = (uint32_t)quantize_one(((const float *)staging)[i * dim + j + e], scale) & 0xffu;
            }
            w[e >> 2] |= byte << (8 * (e & 3));
        }
        out = make_uint4(w[0], w[1], w[2], w[3]);
    }
    corpus[(b * d_chunks + c) * 64u + lane] = out;
}

// ---- row norms (cosine): sqrt(sum x_j^2), sequential, rank.rs:297 ----------------------
template <int DTYPE>
__global__ void row_norms_kernel(const uint4 *__restrict__ corpus, uint32_t d_chunks,
                                 const float *__restrict__ lut, uint64_t row0, uint64_t n,
                                 float *__restrict__ row_norms, uint64_t first_blk)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t row = first_blk * 64u + t;
    if (row < row0 || row >= row0 + n)
        return;
    uint64_t b = row >> 6;
    uint32_t lane = (uint32_t)(row & 63u);
    const uint4 *p = corpus + b * d_chunks * 64u + lane;
    float s = 0.0f;
    for (uint32_t c = 0; c < d_chunks; ++c) {
        uint4 v = p[(size_t)c * 64u];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (DTYPE == VFSX_I8) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = lut[(w[e >> 2] >> (8 * (e & 3))) & 0xffu];
                s = f_add(s, f_mul(x, x));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = __uint_as_float(w[e]);
                s = f_add(s, f_mul(x, x));
            }
        }
    }
    row_norms[row] = f_sqrt(s);
}

// ---- synthetic fill (benchmark utility) ---------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float synth_value(uint64_t seed, uint64_t row, uint32_t dim, uint32_t col)
{
    uint64_t u = splitmix64(seed + row * (uint64_t)dim + col);
    uint32_t m = (uint32_t)(u >> 40);
    return f_sub(f_mul((float)m, 1.0f / 8388608.0f), 1.0f);
}

template <int DTYPE>
__global__ void fill_synthetic_kernel(uint4 *__restrict__ corpus, uint32_t dim, uint32_t d_chunks,
                                      uint64_t row0_local, uint64_t n, uint64_t seed,
                                      uint64_t synth_row0, float scale, uint64_t first_blk,
                                      uint64_t total)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total)
        return;
    uint32_t lane = (uint32_t)(t & 63u);
    uint64_t cb = t >> 6;
    uint32_t c = (uint32_t)(cb % d_chunks);
    uint64_t b = first_blk + cb / d_chunks;
    uint64_t row = b * 64u + lane;
    if (row < row0_local || row >= row0_local + n)
        return;
    uint64_t srow = synth_row0 + (row - row0_local);
    uint32_t w[4] = {0, 0, 0, 0};
    if (DTYPE == VFSX_I8) {
        uint32_t j = c * 16u;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (j + e < dim)
                w[e >> 2] |= ((uint32_t)quantize_one(synth_value(seed, srow, dim, j + e), scale) & 0xffu)
                             << (8 * (e & 3));
    } else {
        uint32_t j = c * 4u;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j + e < dim)
                w[e] = __float_as_uint(synth_value(seed, srow, dim, j + e));
    }
    corpus[(b * d_chunks + c) * 64u + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- gather rows back (hydration) -----------------------------------------------------------
template <int DTYPE, int AS_F32>
__global__ void get_rows_kernel(const uint4 *__restrict__ corpus, uint32_t dim, uint32_t d_chunks,
                                const float *__restrict__ lut, const uint64_t *__restrict__ rows,
                                uint64_t n, void *__restrict__ out)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * dim)
        return;
    uint64_t i = t / dim;
    uint32_t j = (uint32_t)(t % dim);
    uint64_t row = rows[i];
    const uint8_t *base = (const uint8_t *)(corpus + (row >> 6) * d_chunks * 64u + (row & 63u));
    if (DTYPE == VFSX_I8) {
        uint8_t u = base[(size_t)(j >> 4) * 1024u + (j & 15u)];
        if (AS_F32)
            ((float *)out)[t] = lut[u];
        else
            ((uint8_t *)out)[t] = u;
    } else {
        float f = *(const float *)(base + (size_t)(j >> 2) * 1024u + (j & 3u) * 4u);
        ((float *)out)[t] = f;
    }
}

// ---- cross-shard merge of ranked hit lists ------------------------------------------------------
// in: [n_lists][nq][k], every list already ranked (ascending by (distance key, row), invalid entries last:
// what select / rescore / this kernel write).  Strict total order: (valid first, key(dist), row, list).

// n_lists <= 64: a G-way merge in one wave per query.  Lane l holds the head of list l; k times the wave
// takes the smallest head (a 6-step butterfly over the 3-word order tuple) and that lane advances: O(k log 64)
// wave steps instead of the (n_lists k)^2 comparisons of rank-by-counting below.
struct MergeHead {
    uint64_t a; // (invalid << 32) | key(dist): valid entries first, then by distance
    uint64_t row;
    uint32_t list;
};
__device__ __forceinline__ bool head_less(const MergeHead &x, const MergeHead &y)
{
    if (x.a != y.a)
        return x.a < y.a;
    if (x.row != y.row)
        return x.row < y.row;
    return x.list < y.list;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void merge_hits_wave_kernel(const vfsx_hit *__restrict__ in, uint32_t n_lists,
                                                            uint32_t nq, uint32_t k, vfsx_hit *__restrict__ out)
{
    const uint32_t qi = blockIdx.x, lane = threadIdx.x;
    const vfsx_hit *mine = in + ((size_t)(lane < n_lists ? lane : 0u) * nq + qi) * k;
    uint32_t pos = lane < n_lists ? 0u : k; // exhausted lists (and idle lanes) sort after everything
    vfsx_hit h{};
    auto load_head = [&]() -> MergeHead {
        MergeHead m;
        m.list = lane;
        if (pos < k) {
            h = mine[pos];
            m.a = ((uint64_t)((h.flags & 1u) ^ 1u) << 32) | dist_key(h.dist);
            m.row = h.row;
        } else {
            m.a = ~0ull;
            m.row = ~0ull;
        }
        return m;
    };
    MergeHead head = load_head();
    for (uint32_t j = 0; j < k; ++j) {
        MergeHead best = head;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            MergeHead o;
            o.a = shfl_xor_u64(best.a, m);
            o.row = shfl_xor_u64(best.row, m);
            o.list = (uint32_t)__shfl_xor((int)best.list, m, 64);
            if (head_less(o, best))
                best = o;
        }
        if (best.list == lane) { // exactly one lane: the tuple ends in the (unique) list index
            if (pos < k) {
                out[(size_t)qi * k + j] = h;
            } else { // every list is exhausted: fewer than k entries in all (cannot happen for k lists of k)
                vfsx_hit e;
                e.dist = __uint_as_float(0x7fc00000u);
                e.flags = 0;
                e.row = ~0ull;
                e.id = ~0ull;
                out[(size_t)qi * k + j] = e;
            }
            ++pos;
            head = load_head();
        }
    }
}

// any n_lists: rank-by-counting, one workgroup per query (only used beyond 64 lists)
__global__ __launch_bounds__(kBlock) void merge_hits_kernel(const vfsx_hit *__restrict__ in,
                                                            uint32_t n_lists, uint32_t nq, uint32_t k,
                                                            vfsx_hit *__restrict__ out)
{
    const uint32_t qi = blockIdx.x;
    const uint32_t total = n_lists * k;
    for (uint32_t a = threadIdx.x; a < total; a += kBlock) {
        const vfsx_hit ha = in[((size_t)(a / k) * nq + qi) * k + (a % k)];
        const uint32_t va = ha.flags & 1u;
        const uint32_t ka = dist_key(ha.dist);
        uint32_t rank = 0;
        for (uint32_t b = 0; b < total; ++b) {
            const vfsx_hit hb = in[((size_t)(b / k) * nq + qi) * k + (b % k)];
            const uint32_t vb = hb.flags & 1u;
            const uint32_t kb = dist_key(hb.dist);
            bool less;
            if (va != vb)
                less = vb > va; // valid before invalid
            else if (kb != ka)
                less = kb < ka;
            else if (hb.row != ha.row)
                less = hb.row < ha.row;
            else
                less = b < a;
            rank += less ? 1u : 0u;
        }
        if (rank < k)
            out[(size_t)qi * k + rank] = ha;
    }
}

// ---- compact / scatter whole records by index (dwords) ------------------------------------------
// gather: dst[i] = src[idx[i]]; scatter: dst[idx[i]] = src[i]; one workgroup per record
__global__ void move_records_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                    const uint32_t *__restrict__ idx, uint32_t words, int scatter)
{
    const uint32_t i = blockIdx.x, j = idx[i];
    const uint32_t *s = src + (size_t)(scatter ? i : j) * words;
    uint32_t *d = dst + (size_t)(scatter ? j : i) * words;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x)
        d[w] = s[w];
}

} // namespace

// ---- launchers --------------------------------------------------------------------------------------

size_t scan_lds_bytes(uint32_t k, int nq_per_pass, uint32_t q_stride);

template <int DTYPE, int METRIC, int UNROLL, bool NT, int NQ, bool PIPE = false>
static hipError_t launch_scan_v(const ScanArgs &a, uint32_t nq, uint32_t grid_x, hipStream_t s)
{
    dim3 grid(grid_x, (nq + NQ - 1) / NQ), block(kBlock);
    const size_t lds = scan_lds_bytes(a.k, NQ, a.q_stride);
    if (a.write_all) { // long results: every row's key to HBM, no lists (a.k = 1 sizes the unused list LDS)
        hipLaunchKernelGGL((scan_kernel<DTYPE, METRIC, WaveWriteAll, UNROLL, NT, NQ, false>), grid, block, lds, s,
                           a.corpus, a.row_norms, a.queries, a.lut, a.lists, a.n_rows, a.d_chunks, a.n_rowblocks,
                           a.q_stride, a.k, a.n_lists, nq);
    } else if (a.k <= kMaxRegisterK) {
        hipLaunchKernelGGL((scan_kernel<DTYPE, METRIC, WaveTopKReg, UNROLL, NT, NQ, PIPE>), grid, block, lds, s, a.corpus,
                           a.row_norms, a.queries, a.lut, a.lists, a.n_rows, a.d_chunks, a.n_rowblocks,
                           a.q_stride, a.k, a.n_lists, nq);
    } else if (a.k <= kMaxRegister2K && NQ == 1 && !PIPE) { // single-query scans with 64 < k <= 128 (the reference's own
        // benchmark asks for 100): two registers per lane instead of the LDS list (C2, k = 100: 0.54 -> 0.47 ms)
        hipLaunchKernelGGL((scan_kernel<DTYPE, METRIC, WaveTopKReg2, UNROLL, NT, NQ, false>), grid, block, lds, s,
                           a.corpus, a.row_norms, a.queries, a.lut, a.lists, a.n_rows, a.d_chunks, a.n_rowblocks,
                           a.q_stride, a.k, a.n_lists, nq);
    } else {
        hipLaunchKernelGGL((scan_kernel<DTYPE, METRIC, WaveTopKLds, UNROLL, NT, NQ, PIPE>), grid, block, lds, s, a.corpus,
                           a.row_norms, a.queries, a.lut, a.lists, a.n_rows, a.d_chunks, a.n_rowblocks,
                           a.q_stride, a.k, a.n_lists, nq);
    }
    return hipGetLastError();
}

template <int DTYPE, int METRIC>
static hipError_t launch_scan_t(const ScanArgs &a, uint32_t nq, uint32_t grid_x, int variant, int nq_per_pass,
                                hipStream_t s)
{
    // multi-query passes: unroll 4, nontemporal per `variant & 1`
    const bool nt = variant & 1;
    if (nq_per_pass == 8)
        return nt ? launch_scan_v<DTYPE, METRIC, 4, true, 8>(a, nq, grid_x, s)
                  : launch_scan_v<DTYPE, METRIC, 4, false, 8>(a, nq, grid_x, s);
    if (nq_per_pass == 4)
        return nt ? launch_scan_v<DTYPE, METRIC, 4, true, 4>(a, nq, grid_x, s)
                  : launch_scan_v<DTYPE, METRIC, 4, false, 4>(a, nq, grid_x, s);
    if (nq_per_pass == 2)
        return nt ? launch_scan_v<DTYPE, METRIC, 4, true, 2>(a, nq, grid_x, s)
                  : launch_scan_v<DTYPE, METRIC, 4, false, 2>(a, nq, grid_x, s);
    // single query: variant = unroll * 2 + nt (+ 64 = software-pipelined); the default is chosen in vfsx_api.cpp
    switch (variant) {
    case 64 + 4 * 2 + 0: return launch_scan_v<DTYPE, METRIC, 4, false, 1, true>(a, nq, grid_x, s);
    case 64 + 4 * 2 + 1: return launch_scan_v<DTYPE, METRIC, 4, true, 1, true>(a, nq, grid_x, s);
    case 64 + 8 * 2 + 0: return launch_scan_v<DTYPE, METRIC, 8, false, 1, true>(a, nq, grid_x, s);
    case 64 + 8 * 2 + 1: return launch_scan_v<DTYPE, METRIC, 8, true, 1, true>(a, nq, grid_x, s);
    case 4 * 2 + 0: return launch_scan_v<DTYPE, METRIC, 4, false, 1>(a, nq, grid_x, s);
    case 8 * 2 + 1: return launch_scan_v<DTYPE, METRIC, 8, true, 1>(a, nq, grid_x, s);
    case 8 * 2 + 0: return launch_scan_v<DTYPE, METRIC, 8, false, 1>(a, nq, grid_x, s);
    case 16 * 2 + 0: return launch_scan_v<DTYPE, METRIC, 16, false, 1>(a, nq, grid_x, s);
    case 16 * 2 + 1: return launch_scan_v<DTYPE, METRIC, 16, true, 1>(a, nq, grid_x, s);
    default: return launch_scan_v<DTYPE, METRIC, 4, true, 1>(a, nq, grid_x, s);
    }
}

size_t scan_lds_bytes(uint32_t k, int nq_per_pass, uint32_t q_stride)
{
    const uint32_t list_stride = topk_list_stride(k);
    size_t lds = (size_t)kWavesPerBlock * nq_per_pass * list_stride * sizeof(uint64_t);
    if (nq_per_pass > 1)
        lds += (size_t)nq_per_pass * q_stride * sizeof(float);
    return lds;
}

hipError_t launch_scan(int dtype, int metric, const ScanArgs &a, uint32_t nq, uint32_t grid_x, int variant,
                       int nq_per_pass, hipStream_t s)
{
    const bool cosine = (metric == VFSX_COSINE || metric == VFSX_SIMD_COSINE);
    if (dtype == VFSX_I8)
        return cosine ? launch_scan_t<VFSX_I8, VFSX_COSINE>(a, nq, grid_x, variant, nq_per_pass, s)
                      : launch_scan_t<VFSX_I8, VFSX_EUCLIDEAN>(a, nq, grid_x, variant, nq_per_pass, s);
    return cosine ? launch_scan_t<VFSX_F32, VFSX_COSINE>(a, nq, grid_x, variant, nq_per_pass, s)
                  : launch_scan_t<VFSX_F32, VFSX_EUCLIDEAN>(a, nq, grid_x, variant, nq_per_pass, s);
}

// single-query int8 scan with 16-wave workgroups (k <= 64): grid (grid_x, nq)
hipError_t launch_scan_i8_wide(int metric, const ScanArgs &a, uint32_t nq, uint32_t grid_x, bool nt, int unroll,
                               hipStream_t s)
{
    const bool cosine = (metric == VFSX_COSINE || metric == VFSX_SIMD_COSINE);
    const dim3 grid(grid_x, nq), block(64 * kWideWaves);
#define VFSX_I8W(M, U, NT)                                                                                         \
    hipLaunchKernelGGL((scan_i8w_kernel<M, U, NT>), grid, block, 0, s, a.corpus, a.row_norms, a.queries, a.lut,    \
                       a.lists, a.n_rows, a.d_chunks, a.n_rowblocks, a.q_stride, a.k, a.n_lists)
#define VFSX_I8W_U(M, NT)                                                                                          \
    do {                                                                                                           \
        if (unroll >= 8)                                                                                           \
            VFSX_I8W(M, 8, NT);                                                                                    \
        else                                                                                                       \
            VFSX_I8W(M, 4, NT);                                                                                    \
    } while (0)
    if (cosine) {
        if (nt)
            VFSX_I8W_U(VFSX_COSINE, true);
        else
            VFSX_I8W_U(VFSX_COSINE, false);
    } else {
        if (nt)
            VFSX_I8W_U(VFSX_EUCLIDEAN, true);
        else
            VFSX_I8W_U(VFSX_EUCLIDEAN, false);
    }
#undef VFSX_I8W_U
#undef VFSX_I8W
    return hipGetLastError();
}

hipError_t launch_select(const SelectArgs &a, uint32_t nq, hipStream_t s)
{
    const uint32_t list_stride = topk_list_stride(a.k);
    if (a.k <= kMaxRegisterK) {
        constexpr int W = 16;
        const size_t lds = (size_t)W * list_stride * sizeof(uint64_t);
        hipLaunchKernelGGL((select_kernel<WaveTopKReg, W>), dim3(nq), dim3(64 * W), lds, s, a.lists, a.ids,
                           a.out, a.row_base, a.id_base, a.n_lists, a.k, a.out_stride ? a.out_stride : a.k);
    } else if (a.k <= kMaxRegister2K) {
        constexpr int W = 16;
        const size_t lds = (size_t)W * list_stride * sizeof(uint64_t);
        hipLaunchKernelGGL((select_kernel<WaveTopKReg2, W>), dim3(nq), dim3(64 * W), lds, s, a.lists, a.ids,
                           a.out, a.row_base, a.id_base, a.n_lists, a.k, a.out_stride ? a.out_stride : a.k);
    } else {
        constexpr int W = 4;
        const size_t lds = (size_t)W * list_stride * sizeof(uint64_t);
        hipLaunchKernelGGL((select_kernel<WaveTopKLds, W>), dim3(nq), dim3(64 * W), lds, s, a.lists, a.ids,
                           a.out, a.row_base, a.id_base, a.n_lists, a.k, a.out_stride ? a.out_stride : a.k);
    }
    return hipGetLastError();
}

// long results: one chunk (kp <= kLongChunk hits per query from the current cursors) of the radix select over `keys`
// ([nq][n] composite keys written by scan_kernel<WaveWriteAll>); state / hist / sel are the caller's workspaces
// (state[q].k_rem is set here; hist must be zero on entry and is zero again on return)
hipError_t launch_long_select(const uint64_t *keys, uint32_t n, uint32_t nq, uint32_t kp, void *state, uint32_t *hist,
                              uint64_t *sel, const uint64_t *ids, vfsx_hit *out, uint64_t row_base, uint64_t id_base,
                              uint32_t out_stride, uint32_t out_off, hipStream_t s)
{
    LongState *st = (LongState *)state;
    const dim3 grid((n + kLongKeysPerBlock - 1u) / kLongKeysPerBlock, nq);
    hipLaunchKernelGGL(long_set_rank_kernel, dim3((nq + 255u) / 256u), dim3(256), 0, s, st, nq, kp);
    for (int byte = 7; byte >= 0; --byte) {
        hipLaunchKernelGGL(long_hist_kernel, grid, dim3(256), 0, s, keys, n, st, hist, (uint32_t)byte);
        hipLaunchKernelGGL(long_pick_kernel, dim3(nq), dim3(64), 0, s, st, hist, (uint32_t)byte);
    }
    hipLaunchKernelGGL(long_compact_kernel, grid, dim3(256), 0, s, keys, n, st, sel, kLongChunk);
    hipLaunchKernelGGL(long_sort_emit_kernel, dim3(nq), dim3(1024), 0, s, sel, st, ids, out, row_base, id_base, kp,
                       out_stride, out_off);
    return hipGetLastError();
}

size_t long_state_bytes() { return sizeof(LongState); }

__global__ void empty_hits_kernel(vfsx_hit *__restrict__ out, uint32_t nq, uint32_t k, uint32_t out_stride)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)nq * k)
        return;
    vfsx_hit h;
    h.dist = __uint_as_float(0x7fc00000u);
    h.flags = 0;
    h.row = ~0ull;
    h.id = ~0ull;
    out[(t / k) * out_stride + t % k] = h;
}

hipError_t launch_empty_hits(vfsx_hit *out, uint32_t nq, uint32_t k, uint32_t out_stride, hipStream_t s)
{
    const uint64_t total = (uint64_t)nq * k;
    hipLaunchKernelGGL(empty_hits_kernel, dim3((uint32_t)((total + 255u) / 256u)), dim3(256), 0, s, out, nq, k, out_stride);
    return hipGetLastError();
}

hipError_t launch_query_pad(const float *d_queries_in, uint32_t nq, uint32_t dim, float *d_q_padded,
                            uint32_t q_stride, hipStream_t s)
{
    const uint64_t total = (uint64_t)nq * q_stride;
    hipLaunchKernelGGL(query_pad_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, d_queries_in, nq,
                       dim, d_q_padded, q_stride);
    return hipGetLastError();
}

hipError_t launch_lut(float scale, float *d_lut, hipStream_t s)
{
    hipLaunchKernelGGL(lut_kernel, dim3(1), dim3(256), 0, s, scale, d_lut);
    return hipGetLastError();
}

hipError_t launch_ingest(int mode, const void *d_staging, uint64_t n, uint32_t dim, uint4 *corpus,
                         uint32_t d_chunks, uint64_t row0, float scale, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const uint64_t first_blk = row0 / 64u, last_blk = (row0 + n - 1) / 64u;
    const uint64_t total = (last_blk - first_blk + 1) * d_chunks * 64u;
    const dim3 grid((uint32_t)((total + 255) / 256)), block(256);
    if (mode == 0)
        hipLaunchKernelGGL((ingest_kernel<0>), grid, block, 0, s, d_staging, n, dim, corpus, d_chunks,
                           row0, scale, first_blk, total);
    else if (mode == 1)
        hipLaunchKernelGGL((ingest_kernel<1>), grid, block, 0, s, d_staging, n, dim, corpus, d_chunks,
                           row0, scale, first_blk, total);
    else
        hipLaunchKernelGGL((ingest_kernel<2>), grid, block, 0, s, d_staging, n, dim, corpus, d_chunks,
                           row0, scale, first_blk, total);
    return hipGetLastError();
}

hipError_t launch_row_norms(int dtype, const uint4 *corpus, uint32_t d_chunks, const float *d_lut,
                            uint64_t row0, uint64_t n, float *row_norms, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const uint64_t first_blk = row0 / 64u, last_blk = (row0 + n - 1) / 64u;
    const uint64_t total = (last_blk - first_blk + 1) * 64u;
    const dim3 grid((uint32_t)((total + 255) / 256)), block(256);
    if (dtype == VFSX_I8)
        hipLaunchKernelGGL((row_norms_kernel<VFSX_I8>), grid, block, 0, s, corpus, d_chunks, d_lut, row0,
                           n, row_norms, first_blk);
    else
        hipLaunchKernelGGL((row_norms_kernel<VFSX_F32>), grid, block, 0, s, corpus, d_chunks, d_lut, row0,
                           n, row_norms, first_blk);
    return hipGetLastError();
}

hipError_t launch_fill_synthetic(int dtype, uint4 *corpus, uint32_t dim, uint32_t d_chunks,
                                 uint64_t row0_local, uint64_t n, uint64_t seed, uint64_t synth_row0,
                                 float scale, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const uint64_t first_blk = row0_local / 64u, last_blk = (row0_local + n - 1) / 64u;
    const uint64_t total = (last_blk - first_blk + 1) * d_chunks * 64u;
    const uint64_t nblocks = (total + 255) / 256;
    if (nblocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    const dim3 grid((uint32_t)nblocks), block(256);
    if (dtype == VFSX_I8)
        hipLaunchKernelGGL((fill_synthetic_kernel<VFSX_I8>), grid, block, 0, s, corpus, dim, d_chunks,
                           row0_local, n, seed, synth_row0, scale, first_blk, total);
    else
        hipLaunchKernelGGL((fill_synthetic_kernel<VFSX_F32>), grid, block, 0, s, corpus, dim, d_chunks,
                           row0_local, n, seed, synth_row0, scale, first_blk, total);
    return hipGetLastError();
}

hipError_t launch_get_rows(int dtype, const uint4 *corpus, uint32_t dim, uint32_t d_chunks,
                           const float *d_lut, const uint64_t *d_rows, uint64_t n, int as_f32,
                           void *d_out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const uint64_t total = n * dim;
    const dim3 grid((uint32_t)((total + 255) / 256)), block(256);
    if (dtype == VFSX_I8) {
        if (as_f32)
            hipLaunchKernelGGL((get_rows_kernel<VFSX_I8, 1>), grid, block, 0, s, corpus, dim, d_chunks,
                               d_lut, d_rows, n, d_out);
        else
            hipLaunchKernelGGL((get_rows_kernel<VFSX_I8, 0>), grid, block, 0, s, corpus, dim, d_chunks,
                               d_lut, d_rows, n, d_out);
    } else {
        hipLaunchKernelGGL((get_rows_kernel<VFSX_F32, 1>), grid, block, 0, s, corpus, dim, d_chunks, d_lut,
                           d_rows, n, d_out);
    }
    return hipGetLastError();
}

hipError_t launch_move_records(const void *src, void *dst, const uint32_t *d_idx, uint32_t n, uint32_t words,
                               int scatter, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(move_records_kernel, dim3(n), dim3(256), 0, s, (const uint32_t *)src, (uint32_t *)dst, d_idx,
                       words, scatter);
    return hipGetLastError();
}

hipError_t launch_merge_hits(const vfsx_hit *d_in, uint32_t n_lists, uint32_t nq, uint32_t k,
                             vfsx_hit *d_out, hipStream_t s)
{
    if (n_lists <= 64u)
        hipLaunchKernelGGL(merge_hits_wave_kernel, dim3(nq), dim3(64), 0, s, d_in, n_lists, nq, k, d_out);
    else
        hipLaunchKernelGGL(merge_hits_kernel, dim3(nq), dim3(kBlock), 0, s, d_in, n_lists, nq, k, d_out);
    return hipGetLastError();
}

} // namespace vfsx
