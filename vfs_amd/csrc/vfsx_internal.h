// vfsx_internal.h -- shared between the HIP kernels (vfsx_kernels.hip) and the
// C-ABI host layer (vfsx_api.cpp).  Not installed; the public surface is include/vfsx.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "vfsx.h"
#include "vfsx_host.h" // error plumbing, log walker, host_dist_key: the host-only part (vfsx_host.cpp, g++)

namespace vfsx {

// ---- HBM layout ("rb64"): see DESIGN.md --------------------------------------
// Rows are grouped in row-blocks of 64 (one wavefront: lane l owns row 64*b + l).
// A row is cut into 16-byte chunks (4 f32 or 16 i8).  Chunk c of row-block b is the
// 1 KiB span  [b*d_chunks + c] * 1024 .. +1024, holding lane 0..63's chunk c back to
// back, so one global_load_dwordx4 per lane reads 1 KiB contiguous AND every lane
// receives its own row's elements 4c..4c+3 (16c..16c+15) in index order.
constexpr uint32_t kRowBlock = 64;
constexpr uint32_t kChunkBytes = 16;
constexpr uint64_t kSentinel = ~0ull;   // empty top-k slot
constexpr uint32_t kMaxRegisterK = 64;  // k <= 64: the wave's list lives in registers
constexpr uint32_t kMaxRegister2K = 128; // k <= 128: two registers per lane (WaveTopKReg2)
// u64 entries one wave's published list occupies in LDS: what its top-k class spills (64 / 128) or k itself
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t topk_list_stride(uint32_t k) { return k <= kMaxRegisterK ? 64u : (k <= kMaxRegister2K ? 128u : k); }

inline uint32_t elems_per_chunk(int dtype) { return dtype == VFSX_I8 ? 16u : 4u; }

struct ScanArgs {
    const uint4 *corpus;    // rb64 layout
    const float *row_norms; // [n_rows] sqrt(sum x^2), sequential order (cosine only)
    const float *queries;   // [nq][q_stride] f32, zero padded to d_chunks*epc
    const float *lut;       // [256] f32 dequant table indexed by the raw byte (i8 only)
    uint64_t *lists;        // out: [nq][k][n_lists] composite keys (column-major: entry j of every
                            // list contiguous), each list ascending, kSentinel padded
    uint32_t n_rows;
    uint32_t d_chunks;
    uint32_t n_rowblocks;
    uint32_t q_stride;
    uint32_t k;
    uint32_t n_lists; // = gridDim.x (one list per workgroup)
    bool write_all;         // scan_kernel only: no lists -- `lists` receives every row's composite key,
                            // [nq][64 n_rowblocks] (kSentinel past n_rows), for the radix select below
};
constexpr uint32_t kMaxKPerPass = 1024; // longest list a wave keeps (LDS): the limit of the list path (no-alloc entry, integer-domain metric)

struct SelectArgs {
    const uint64_t *lists; // [nq][k][n_lists]
    const uint64_t *ids;   // [n_rows] or nullptr (=> id_base + local row)
    vfsx_hit *out;         // [nq][k]
    uint64_t row_base;
    uint64_t id_base;
    uint32_t n_lists;
    uint32_t k;
    uint32_t out_stride;  // hits per query in `out` (0 = k)
};

// launchers (vfsx_kernels.hip). All enqueue on `s` and return the launch status.
// nq_per_pass in {1,2,4,8}: queries that share one pass over the corpus (one workgroup column)
hipError_t launch_scan(int dtype, int metric, const ScanArgs &a, uint32_t nq, uint32_t grid_x, int variant,
                       int nq_per_pass, hipStream_t s);
// dynamic LDS the scan kernel needs (the i8 kernels add 32 KiB of static LDS for the LUT)
size_t scan_lds_bytes(uint32_t k, int nq_per_pass, uint32_t q_stride);
// single-query int8 scan, 16 waves per workgroup sharing a 64 KiB per-lane dequant table (k <= 64)
constexpr uint32_t kWideScanWaves = 16;
hipError_t launch_scan_i8_wide(int metric, const ScanArgs &a, uint32_t nq, uint32_t grid_x, bool nt, int unroll,
                               hipStream_t s);
hipError_t launch_select(const SelectArgs &a, uint32_t nq, hipStream_t s);
// result_limit > kLongResultK: scan_kernel with ScanArgs::write_all, then chunks of launch_long_select (vfsx_kernels.hip)
constexpr uint32_t kLongResultK = 128;  // beyond this the lists give way to the radix select
constexpr uint32_t kLongChunkHits = 4096; // hits one select / sort chunk emits per query
size_t long_state_bytes();
hipError_t launch_empty_hits(vfsx_hit *out, uint32_t nq, uint32_t k, uint32_t out_stride, hipStream_t s);
hipError_t launch_long_select(const uint64_t *keys, uint32_t n, uint32_t nq, uint32_t kp, void *state, uint32_t *hist,
                              uint64_t *sel, const uint64_t *ids, vfsx_hit *out, uint64_t row_base, uint64_t id_base,
                              uint32_t out_stride, uint32_t out_off, hipStream_t s);
hipError_t launch_query_pad(const float *d_queries_in, uint32_t nq, uint32_t dim, float *d_q_padded,
                            uint32_t q_stride, hipStream_t s);
hipError_t launch_lut(float scale, float *d_lut, hipStream_t s);
// staging (row-major, n rows of `dim` src elements) -> rb64 rows [row0, row0+n)
//   mode 0: f32 -> f32   1: i8 -> i8   2: f32 -> i8 (quantize, vector.rs:94-108)
hipError_t launch_ingest(int mode, const void *d_staging, uint64_t n, uint32_t dim, uint4 *corpus,
                         uint32_t d_chunks, uint64_t row0, float scale, hipStream_t s);
hipError_t launch_row_norms(int dtype, const uint4 *corpus, uint32_t d_chunks, const float *d_lut,
                            uint64_t row0, uint64_t n, float *row_norms, hipStream_t s);
hipError_t launch_fill_synthetic(int dtype, uint4 *corpus, uint32_t dim, uint32_t d_chunks,
                                 uint64_t row0_local, uint64_t n, uint64_t seed, uint64_t synth_row0,
                                 float scale, hipStream_t s);
// gather LOCAL rows back to row-major; as_f32 != 0 dequantizes i8 through the LUT
hipError_t launch_get_rows(int dtype, const uint4 *corpus, uint32_t dim, uint32_t d_chunks,
                           const float *d_lut, const uint64_t *d_rows, uint64_t n, int as_f32,
                           void *d_out, hipStream_t s);
hipError_t launch_merge_hits(const vfsx_hit *d_in, uint32_t n_lists, uint32_t nq, uint32_t k,
                             vfsx_hit *d_out, hipStream_t s);
// n records of `words` dwords: gather dst[i] = src[idx[i]] (scatter = 0) or scatter dst[idx[i]] = src[i]
hipError_t launch_move_records(const void *src, void *dst, const uint32_t *d_idx, uint32_t n, uint32_t words,
                               int scatter, hipStream_t s);

// ---- integer-domain Euclidean on int8 corpora (vfsx_i8dom.hip; VFSX_EUCLIDEAN_I8DOMAIN) ----------
// d_qw[nq][qw_stride]: n_words packed quantized dwords, then sum qi^2; qw_stride % 4 == 0
hipError_t launch_query_quantize(const float *d_queries, uint32_t nq, uint32_t dim, float scale, uint32_t *d_qw,
                                 uint32_t n_words, uint32_t qw_stride, hipStream_t s);
// a.queries = d_qw (reinterpreted), a.q_stride = qw_stride
hipError_t launch_scan_i8dom(const ScanArgs &a, uint32_t nq, uint32_t grid_x, bool nt, float scale, hipStream_t s);

// ---- many-query batches on the matrix cores (vfsx_mfma.hip) ---------------------------------
constexpr uint32_t kStripOwnLists = 0x100u; // flag on a kernel's strip_tiles argument: 16 lists per strip
struct MfmaGeometry {
    uint32_t k_pad;      // dim rounded up to 128
    uint32_t KS;         // fragment k-slices: k_pad / 16 (bf16) or k_pad / 32 (int8)
    uint32_t ku;         // k-slices per K-step of the 256 x 256 kernel: 4, or 2 when an int8 K is an odd multiple of 128
    uint32_t n_rowtiles; // ceil(n_rows / 256)
    uint32_t strip_tiles; // row tiles per strip (1..16)
    uint32_t n_strips;   // ceil(n_rowtiles / strip_tiles)
    uint32_t strip_lists; // candidate lists per strip and query: 4 (the four waves of a 128-row half merged), or 16
                          // (small-batch kernel, long strips: every wave keeps its own two lists -- same rows per
                          // list as four merged waves of a strip a quarter as long)
    uint32_t n_lists;    // candidate lists per query = strip_lists * n_strips
    // what the kernels that decode a candidate (and the small-batch prefilter) receive as `strip_tiles`
    uint32_t strip_code() const { return strip_tiles | (strip_lists == 16u ? kStripOwnLists : 0u); }
    size_t mirror_bytes; // bf16 mirror of the corpus (rows padded to 256); 0 for int8 stores (rb64 is the operand)
};
constexpr uint32_t kMfmaMaxDim = 32768; // the error model's (1 + 2^-7) factor assumes K <= 32768
// rows the matrix-core path takes: its kernels pad the row count to 256-row tiles in 32-bit arithmetic (larger shards,
// up to the 2^32 - 66 the exact scan serves, keep the exact scan)
constexpr uint64_t kMfmaMaxRows = 0xfffffe00ull;
constexpr uint32_t kMfmaMaxK = 128; // result_limit the matrix-core batch path serves (256 candidates re-scored per query)
// n_cus != 0: the geometry of ONE prefilter launch of nq queries on a device with that many CUs -- the strip length
// (and, for the small-batch kernel, the lists per strip) is then chosen for whole rounds of workgroups
MfmaGeometry mfma_geometry(uint32_t n_rows, uint32_t dim, bool int8, uint32_t n_cus = 0, uint32_t nq = 256);
bool mfma_small_batch(uint32_t nq); // does a launch of nq queries take mfma_prefilter_skinny_kernel?
float mfma_f32_slack(uint32_t k_pad);

struct MfmaRescoreArgs {
    const uint4 *corpus;
    const float *row_norms;
    const float *queries; // [nq][q_stride] f32 (zero padded to the chunk)
    const float *lut;
    const uint64_t *sel;  // [nq][kp]
    const int *q_bad;     // [nq]
    const uint64_t *ids;
    vfsx_hit *out;        // [nq][k]
    int *flags;           // [nq] 1 = not verified, re-run exactly
    uint32_t n_rows, d_chunks, q_stride, k, kp;
    const float *q_eps;   // [nq] eps_q of mfma_query_prep_kernel
    uint64_t row_base, id_base;
    uint32_t strip_tiles;
    int euclid;           // 0 = Cosine, 1 = Euclidean (see vfsx_mfma.hip)
    const float *q_norm;  // [nq] the scalar each query mirror row was divided by
    float gamma;          // relative f32 error allowance of the Euclidean check
    int only_flagged;     // != 0: second pass, only queries whose flags[q] is set (they are re-decided)
    const float *q_scale; // int8 stores: [nq] s_q, the query's quantization step (nullptr on the bf16 path)
    float *dk_out;        // [nq] (or nullptr): the k-th exact distance this pass found (NaN: fewer than k hits)
};
// stats[2] (device): [0] != 0: a row norm outside the path's range; [1]: R_x, the mirror's largest rounding
// residual (float bits).  Both zeroed by the caller before the two launches.
hipError_t launch_mfma_mirror(int dtype, const uint4 *corpus, const float *row_norms, const float *lut,
                              uint4 *mirror, uint32_t n_rows, uint32_t dim, uint32_t d_chunks, const MfmaGeometry &g,
                              uint32_t *stats, hipStream_t s);
hipError_t launch_mfma_query_prep(const float *d_queries, uint32_t nq, uint32_t dim, void *q_mirror, uint32_t KS,
                                  int *q_bad, float *q_norm, float *q_eps, const uint32_t *stats, hipStream_t s);
// out[n_pad] float2 = (b, b^2 (1 - gamma)), zero beyond n_rows; inv[n_pad] (or nullptr) = 1 / (i8_scale b); stats[0]
// is raised for a norm outside the path's range
hipError_t launch_mfma_rownorm_pairs(const float *row_norms, uint32_t n_rows, uint32_t n_pad, float gamma, float *out,
                                     float *inv, float i8_scale, uint32_t *stats, hipStream_t s, uint32_t row0 = 0);
// int8 stores: no mirror -- the prefilter reads the rb64 corpus; the queries are quantized per batch
// stats: nullptr for an int8 store (exact rows), the mirror's stats for an int8 mirror of an f32 store (R_x)
hipError_t launch_mfma_query_prep_i8(const float *d_queries, uint32_t nq, uint32_t dim, void *q_mirror, uint32_t KS,
                                     int *q_bad, float *q_norm, float *q_eps, float *q_scale, const uint32_t *stats,
                                     hipStream_t s);
// int8 mirror of an f32 store: rb64 int8 rows (d_chunks8 = ceil(dim / 16) chunks per row; the buffer must cover
// whole 64-row blocks and be zeroed beforehand), inv_rn[row] = the row's scale, stats[0] / stats[1] as launch_mfma_mirror
hipError_t launch_mfma_mirror_i8(const uint4 *corpus, const float *row_norms, uint4 *mirror, float *inv_rn,
                                 uint32_t n_rows, uint32_t d_chunks, uint32_t d_chunks8, uint32_t k_pad,
                                 uint32_t *stats, hipStream_t s, uint32_t row0 = 0); // rows [row0, n_rows)
hipError_t launch_mfma_prefilter_i8(const uint4 *corpus, uint32_t d_chunks, uint64_t corpus_bytes, const void *q_mirror,
                                    float *cand, uint32_t n_rows, const MfmaGeometry &g, uint32_t nq, int euclid,
                                    const float *rn2, const float *inv_rn, const float *q_norm, const float *q_eps,
                                    const float *q_scale, hipStream_t s);
hipError_t launch_mfma_prefilter(const uint4 *mirror, const void *q_mirror, float *cand, uint32_t n_rows,
                                 const MfmaGeometry &g, uint32_t nq, int euclid, const float *rn2,
                                 const float *q_norm, const float *q_eps, hipStream_t s);
// only_flagged != nullptr: skip queries whose flag is 0
// second pass: sel[q][0..256) = every listed candidate of a flagged query whose score reaches the safety margin of the
// first pass' k-th distance dk1[q] (unsorted; all sentinels when more than 256 do)
hipError_t launch_mfma_filter(const float *cand, uint64_t *sel, uint32_t nq, uint32_t n_lists, const int *flags,
                              const int *q_bad, const float *dk1, const float *q_eps, const float *q_scale,
                              const float *q_norm, float gamma, int euclid, hipStream_t s);
hipError_t launch_mfma_select(const float *cand, uint64_t *sel, uint32_t nq, uint32_t n_lists, uint32_t kp,
                              const int *only_flagged, hipStream_t s);
hipError_t launch_mfma_rescore(int dtype, const MfmaRescoreArgs &a, uint32_t nq, hipStream_t s);

} // namespace vfsx
