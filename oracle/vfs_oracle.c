/*
 * vfs_oracle.c -- CPU restatement of the VFS Ranker linear-search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vfs_amd/ (the product) may include,
 * link, load or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the reported
 * CPU baseline -- never as the thing shipped.
 *
 * PARITY UNPINNED: the reference (RaulMoldes/VFS, Rust nightly) holds no golden
 * vectors or known-answer tests for this path and cannot be
 * compiled in this image (no rustc/cargo).  This restatement is therefore pinned
 * only by (a) hand-derived KATs from the cited reference lines and (b) bit-for-bit
 * agreement with a second, independently written numpy restatement
 * (oracle/vfs_oracle_np.py).
 *
 * All citations are relative to /root/reference/.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, x86-64 SSE2: every f32
 * operation is a single correctly-rounded IEEE-754 binary32 operation, which is
 * what rustc emits for the reference's scalar code).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__FAST_MATH__)
#error "the oracle must not be built with -ffast-math"
#endif

#define VFSO_EUCLIDEAN 0      /* vfs/src/vfs/rank.rs:67-72, enum order */
#define VFSO_COSINE 1
#define VFSO_SIMD_EUCLIDEAN 2
#define VFSO_SIMD_COSINE 3

/* ------------------------------------------------------------------------- */
/* Distances: vfs/src/vfs/rank.rs:271-300                                     */
/* ------------------------------------------------------------------------- */

/* rank.rs:281-287: zip(a,b).map(|(a,b)| (a-b).powi(2)).sum::<f32>().sqrt().
 * powi(2) is x*x (one rounding); sum::<f32>() is a strict left fold; a = query. */
float vfso_dist_euclidean(const float *q, const float *x, uint32_t d)
{
    float acc = 0.0f; /* strict left fold; no -ffast-math, so gcc may not reassociate */
    for (uint32_t j = 0; j < d; ++j) {
        float diff = q[j] - x[j];
        float sq = diff * diff;
        acc = acc + sq;
    }
    return sqrtf(acc);
}

/* rank.rs:290-300: three separate sequential f32 sums, then
 * 1.0 - (dot / (norm_1 * norm_2)).  A zero norm gives 0/0 = NaN (kept). */
float vfso_dist_cosine(const float *q, const float *x, uint32_t d)
{
    float dot = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (uint32_t j = 0; j < d; ++j) {
        float p = q[j] * x[j];
        dot = dot + p;
    }
    for (uint32_t j = 0; j < d; ++j) {
        float p = q[j] * q[j];
        s1 = s1 + p;
    }
    for (uint32_t j = 0; j < d; ++j) {
        float p = x[j] * x[j];
        s2 = s2 + p;
    }
    float n1 = sqrtf(s1);
    float n2 = sqrtf(s2);
    float den = n1 * n2;
    float quo = dot / den;
    return 1.0f - quo;
}

/* rank.rs:14-33, 36-56, 302-308: the core::simd variants compute the same formula
 * on one Simd<f32,N>; reduce_sum() is an ordered (lane 0..N-1) reduction, so the
 * bits equal the scalar fold.  The reference only accepts
 * d in {2,4,8,16,32,64} and panics otherwise. */
int vfso_simd_width_supported(uint32_t d)
{
    return d == 2 || d == 4 || d == 8 || d == 16 || d == 32 || d == 64;
}

float vfso_distance(int metric, const float *q, const float *x, uint32_t d)
{
    switch (metric) {
    case VFSO_EUCLIDEAN:
    case VFSO_SIMD_EUCLIDEAN:
        return vfso_dist_euclidean(q, x, d);
    default:
        return vfso_dist_cosine(q, x, d);
    }
}

/* ------------------------------------------------------------------------- */
/* Quantize / dequantize: vfs/src/vfs/vector.rs:94-108, 135-139, 281-290      */
/* ------------------------------------------------------------------------- */

/* Rust `f32 as i8`: saturating, NaN -> 0. */
static int8_t sat_cast_i8(float v)
{
    if (v != v)
        return 0;
    if (v >= 127.0f)
        return 127;
    if (v <= -128.0f)
        return -128;
    return (int8_t)v; /* truncation toward zero; v is already integral here */
}

/* vector.rs:94-108.  has_scale=0 models `None`: clamp to [-1,1], scale 127.0.
 * The store always calls it with Some(127.0) (vector.rs:188) => no clamp.
 * f32::round is round-half-away-from-zero == C roundf. */
void vfso_quantize(const float *v, uint64_t n, int has_scale, float scale, int8_t *out)
{
    float s = has_scale ? scale : 127.0f;
    for (uint64_t i = 0; i < n; ++i) {
        float val = v[i];
        if (!has_scale) {
            /* f32::clamp: NaN stays NaN */
            if (val < -1.0f)
                val = -1.0f;
            if (val > 1.0f)
                val = 1.0f;
        }
        float prod = val * s;
        out[i] = sat_cast_i8(roundf(prod));
    }
}

/* vector.rs:135-139 / 281-290: (val as f32) / scale_factor -- a true division. */
void vfso_dequantize(const int8_t *v, uint64_t n, float scale, float *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        float f = (float)v[i];
        out[i] = f / scale;
    }
}

/*
 * EXTENSION (not a reference DistanceMethod; SURVEY.md 8a "A6 decision" (ii)): the
 * integer-domain Euclidean distance of the product's labelled fast mode
 * VFSX_EUCLIDEAN_I8DOMAIN (= metric 4).  `qi` is the query quantized by the reference's
 * own rule, VFSVector::as_i8_vec(Some(scale)) -> Vector::quantize (vector.rs:293-302,
 * 94-108), i.e. vfso_quantize(q, d, 1, scale, qi).  S is exact (order independent);
 * the two f32 operations are correctly rounded.
 */
float vfso_dist_euclidean_i8dom(const int8_t *qi, const int8_t *x, uint32_t d, float scale)
{
    uint32_t S = 0;
    for (uint32_t j = 0; j < d; ++j) {
        int32_t t = (int32_t)qi[j] - (int32_t)x[j];
        S += (uint32_t)(t * t);
    }
    return sqrtf((float)S) / scale;
}

/* ------------------------------------------------------------------------- */
/* Ranking: vfs/src/vfs/rank.rs:119-186                                       */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint64_t id;
    float dist;
    uint64_t pos; /* scan position, only used to make the merge sort stable */
} vfso_hit;

/* a.1.partial_cmp(&b.1).unwrap_or(Equal) (rank.rs:170): NaN compares Equal to
 * everything; -0.0 == +0.0. */
static int hit_less(const vfso_hit *a, const vfso_hit *b)
{
    return a->dist < b->dist;
}

/* Stable insertion-merge sort (slice::sort_by is stable).  n is small
 * (limit + batch) in the faithful path, so O(n^2) insertion is fine. */
static void stable_sort_hits(vfso_hit *h, size_t n)
{
    for (size_t i = 1; i < n; ++i) {
        vfso_hit key = h[i];
        size_t j = i;
        while (j > 0 && hit_less(&key, &h[j - 1])) {
            h[j] = h[j - 1];
            --j;
        }
        h[j] = key;
    }
}

/*
 * Faithful restatement of exact_search (rank.rs:119-186) over an in-memory,
 * fully flushed corpus (parity is defined on the in-memory
 * algorithm).  `rows` is the sequence load_batch would yield, in file order,
 * already dequantized when the store is quantized (storage_manager.rs:202-210).
 *
 *   batch  = num_vectors_per_iteration (rank.rs:91); 0 => no rows are ever loaded
 *   limit  = result_limit.unwrap_or(5) (rank.rs:129)
 *
 * Returns the number of hits written to out_ids/out_dist (<= limit, rank.rs:172-178).
 * NaN distances are outside the parity domain (the reference's order is then
 * unspecified); this function still mirrors the code literally.
 */
uint64_t vfso_exact_search(const float *rows, const uint64_t *ids, uint64_t n_rows, uint32_t d,
                           const float *query, int metric, uint64_t batch, uint64_t limit,
                           uint64_t *out_ids, float *out_dist)
{
    if (batch == 0) /* load_batch(0) yields nothing (storage_manager.rs:198): empty result */
        return 0;
    size_t cap = (size_t)(limit + batch + 1);
    vfso_hit *res = (vfso_hit *)malloc(cap * sizeof(vfso_hit));
    size_t len = 0;
    uint64_t off = 0;
    for (;;) {
        /* load_batch(batch): next <= batch rows; empty => end (rank.rs:151-156) */
        uint64_t take = n_rows - off < batch ? n_rows - off : batch;
        if (take == 0)
            break;
        for (uint64_t i = 0; i < take; ++i) { /* rank.rs:159-168 */
            uint64_t r = off + i;
            res[len].id = ids ? ids[r] : r + 1; /* ids 1,2,3.. storage_manager.rs:76,126-130 */
            res[len].dist = vfso_distance(metric, query, rows + (size_t)r * d, d);
            res[len].pos = r;
            ++len;
        }
        off += take;
        stable_sort_hits(res, len); /* rank.rs:170 */
        if (len > limit) {          /* rank.rs:172-178 */
            len = (size_t)limit;
            if (len > 0) {
                float lowest = res[len - 1].dist;
                size_t w = 0;
                for (size_t i = 0; i < len; ++i)
                    if (res[i].dist <= lowest) /* retain: drops NaN only */
                        res[w++] = res[i];
                len = w;
            }
        }
    }
    for (size_t i = 0; i < len; ++i) {
        out_ids[i] = res[i].id;
        out_dist[i] = res[i].dist;
    }
    free(res);
    return (uint64_t)len;
}

/*
 * Same result as vfso_exact_search for NaN-free distances, any batch size
 * (the net semantics are "the `limit` smallest by (distance asc,
 * first-seen scan position asc)"), but O(N log limit): used for large corpora and
 * as the timed CPU baseline.  Sorted insertion into a bounded list; a later row
 * with an equal distance never displaces an earlier one.
 * `x_i8` != NULL selects an int8 corpus dequantized on the fly with `scale`
 * exactly as storage_manager.rs:202-210 + vector.rs:135-139 do.
 */
uint64_t vfso_topk_scan(const float *rows_f32, const int8_t *rows_i8, float scale,
                        const uint64_t *ids, uint64_t n_rows, uint32_t d, const float *query,
                        int metric, uint64_t limit, uint64_t *out_ids, float *out_dist,
                        uint64_t *out_pos)
{
    if (limit == 0)
        return 0;
    if (metric == 4 && !rows_i8)
        return 0;
    vfso_hit *res = (vfso_hit *)malloc((size_t)limit * sizeof(vfso_hit));
    float *tmp = rows_i8 ? (float *)malloc((size_t)d * sizeof(float)) : NULL;
    int8_t *qi = NULL;
    if (metric == 4) {
        qi = (int8_t *)malloc(d ? d : 1);
        vfso_quantize(query, d, 1, scale, qi);
    }
    size_t len = 0;
    for (uint64_t r = 0; r < n_rows; ++r) {
        const float *x = NULL;
        if (metric == 4) {
            ;
        } else if (rows_i8) {
            vfso_dequantize(rows_i8 + (size_t)r * d, d, scale, tmp);
            x = tmp;
        } else {
            x = rows_f32 + (size_t)r * d;
        }
        float dist = metric == 4 ? vfso_dist_euclidean_i8dom(qi, rows_i8 + (size_t)r * d, d, scale)
                                 : vfso_distance(metric, query, x, d);
        if (len == limit && !(dist < res[len - 1].dist))
            continue;
        size_t j = len < limit ? len : len - 1;
        while (j > 0 && dist < res[j - 1].dist) {
            res[j] = res[j - 1];
            --j;
        }
        res[j].id = ids ? ids[r] : r + 1;
        res[j].dist = dist;
        res[j].pos = r;
        if (len < limit)
            ++len;
    }
    for (size_t i = 0; i < len; ++i) {
        out_ids[i] = res[i].id;
        out_dist[i] = res[i].dist;
        if (out_pos)
            out_pos[i] = res[i].pos;
    }
    free(res);
    free(tmp);
    free(qi);
    return (uint64_t)len;
}

/* All N distances (for tie/near-tie analysis in the tests). */
void vfso_all_distances(const float *rows_f32, const int8_t *rows_i8, float scale, uint64_t n_rows,
                        uint32_t d, const float *query, int metric, float *out)
{
    float *tmp = rows_i8 ? (float *)malloc((size_t)d * sizeof(float)) : NULL;
    if (metric == 4) {
        int8_t *qi = (int8_t *)malloc(d ? d : 1);
        vfso_quantize(query, d, 1, scale, qi);
        for (uint64_t r = 0; rows_i8 && r < n_rows; ++r)
            out[r] = vfso_dist_euclidean_i8dom(qi, rows_i8 + (size_t)r * d, d, scale);
        free(qi);
        free(tmp);
        return;
    }
    for (uint64_t r = 0; r < n_rows; ++r) {
        const float *x;
        if (rows_i8) {
            vfso_dequantize(rows_i8 + (size_t)r * d, d, scale, tmp);
            x = tmp;
        } else {
            x = rows_f32 + (size_t)r * d;
        }
        out[r] = vfso_distance(metric, query, x, d);
    }
    free(tmp);
}

/* ------------------------------------------------------------------------- */
/* Synthetic data: counter-based, so any row can be    */
/* regenerated anywhere without shipping data.  Not part of the reference.    */
/* ------------------------------------------------------------------------- */

static uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* element (row i, col j) <- counter seed + i*d + j; value in [-1,1), exact in f32 */
float vfso_synth_value(uint64_t seed, uint64_t row, uint32_t d, uint32_t col)
{
    uint64_t u = splitmix64(seed + row * (uint64_t)d + col);
    uint32_t m = (uint32_t)(u >> 40); /* top 24 bits */
    return (float)m * (1.0f / 8388608.0f) - 1.0f;
}

void vfso_fill_synthetic_f32(uint64_t seed, uint64_t row0, uint64_t n, uint32_t d, float *out)
{
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < d; ++j)
            out[(size_t)i * d + j] = vfso_synth_value(seed, row0 + i, d, j);
}

void vfso_fill_synthetic_i8(uint64_t seed, uint64_t row0, uint64_t n, uint32_t d, float scale,
                            int8_t *out)
{
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < d; ++j) {
            float v = vfso_synth_value(seed, row0 + i, d, j);
            vfso_quantize(&v, 1, 1, scale, &out[(size_t)i * d + j]);
        }
}
