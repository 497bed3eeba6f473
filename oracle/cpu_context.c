/*
 * cpu_context.c -- TEST/BENCH INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU *context* baselines for bench.py: what a
 * competent CPU port of the reference's `core::simd` path would look like if it were
 * extended beyond its d <= 64 limit (rank.rs:14-33 builds ONE Simd<f32,N> per vector
 * and panics for other lengths, rank.rs:31).  Here a vector is walked in 64-lane chunks,
 * the 64 lanes accumulate element-wise across chunks and are reduced once at the end.
 *
 * This is NOT the reference's arithmetic: the summation order differs from the scalar
 * fold (rank.rs:271-300) that vfs_oracle.c restates, so distances agree only to
 * rounding (~1e-6 relative).  It is never used as a parity checker; it only answers
 * "how fast is a vectorised / multi-threaded CPU on the same scan", reported beside the
 * bit-faithful 1-core port.
 *
 * Portable build (no -march flags): GCC vector extensions, 8 x v8sf accumulators; the
 * AVX2 body is selected at run time when the host supports it.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float v8 __attribute__((vector_size(32), aligned(4)));

#define SCAN_BODY                                                                              \
    float best_d[VFSC_MAX_K];                                                                  \
    uint64_t best_r[VFSC_MAX_K];                                                               \
    size_t len = 0;                                                                            \
    float qq = 0.f;                                                                            \
    if (metric) {                                                                              \
        for (uint32_t j = 0; j < d; ++j)                                                       \
            qq += q[j] * q[j];                                                                 \
        qq = sqrtf(qq);                                                                        \
    }                                                                                          \
    float *tmp = (float *)malloc(((size_t)d + 64) * sizeof(float));                            \
    for (uint64_t r = r0; r < r1; ++r) {                                                       \
        const float *x;                                                                        \
        if (rows_i8) {                                                                         \
            const int8_t *xi = rows_i8 + (size_t)r * d;                                        \
            for (uint32_t j = 0; j < d; ++j)                                                   \
                tmp[j] = lut[(uint8_t)xi[j]];                                                  \
            x = tmp;                                                                           \
        } else {                                                                               \
            x = rows_f32 + (size_t)r * d;                                                      \
        }                                                                                      \
        v8 a[8], b[8];                                                                         \
        for (int i = 0; i < 8; ++i)                                                            \
            a[i] = b[i] = (v8){0, 0, 0, 0, 0, 0, 0, 0};                                        \
        uint32_t j = 0;                                                                        \
        for (; j + 64 <= d; j += 64)                                                           \
            for (int i = 0; i < 8; ++i) {                                                      \
                v8 xv = *(const v8 *)(x + j + 8 * i), qv = *(const v8 *)(q + j + 8 * i);       \
                if (metric) {                                                                  \
                    a[i] += xv * qv;                                                           \
                    b[i] += xv * xv;                                                           \
                } else {                                                                       \
                    v8 t = qv - xv;                                                            \
                    a[i] += t * t;                                                             \
                }                                                                              \
            }                                                                                  \
        float s = 0.f, s2 = 0.f;                                                               \
        for (int i = 0; i < 8; ++i)                                                            \
            for (int l = 0; l < 8; ++l) {                                                      \
                s += a[i][l];                                                                  \
                s2 += b[i][l];                                                                 \
            }                                                                                  \
        for (; j < d; ++j) {                                                                   \
            if (metric) {                                                                      \
                s += x[j] * q[j];                                                              \
                s2 += x[j] * x[j];                                                             \
            } else {                                                                           \
                float t = q[j] - x[j];                                                         \
                s += t * t;                                                                    \
            }                                                                                  \
        }                                                                                      \
        float dist = metric ? 1.0f - s / (qq * sqrtf(s2)) : sqrtf(s);                          \
        if (len == limit && !(dist < best_d[len - 1]))                                         \
            continue;                                                                          \
        size_t p = len < limit ? len : len - 1;                                                \
        while (p > 0 && dist < best_d[p - 1]) {                                                \
            best_d[p] = best_d[p - 1];                                                         \
            best_r[p] = best_r[p - 1];                                                         \
            --p;                                                                               \
        }                                                                                      \
        best_d[p] = dist;                                                                      \
        best_r[p] = r;                                                                         \
        if (len < limit)                                                                       \
            ++len;                                                                             \
    }                                                                                          \
    free(tmp);                                                                                 \
    memcpy(out_d, best_d, len * sizeof(float));                                                \
    memcpy(out_r, best_r, len * sizeof(uint64_t));                                             \
    return len;

#define VFSC_MAX_K 64

#define SCAN_ARGS                                                                              \
    const float *rows_f32, const int8_t *rows_i8, const float *lut, uint64_t r0, uint64_t r1, \
        uint32_t d, const float *q, int metric, size_t limit, float *out_d, uint64_t *out_r

__attribute__((target("avx2"))) static size_t scan_avx2(SCAN_ARGS) { SCAN_BODY }
static size_t scan_generic(SCAN_ARGS) { SCAN_BODY }

typedef struct {
    const float *rows_f32;
    const int8_t *rows_i8;
    const float *lut;
    uint64_t r0, r1;
    uint32_t d;
    const float *q;
    int metric;
    size_t limit, len;
    float out_d[VFSC_MAX_K];
    uint64_t out_r[VFSC_MAX_K];
    int avx2;
} job;

static void *run_job(void *p)
{
    job *j = (job *)p;
    j->len = (j->avx2 ? scan_avx2 : scan_generic)(j->rows_f32, j->rows_i8, j->lut, j->r0, j->r1, j->d, j->q,
                                                  j->metric, j->limit, j->out_d, j->out_r);
    return NULL;
}

int vfsc_have_avx2(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") ? 1 : 0;
}

/* k smallest by (distance, row) over rows [0,n_rows) using `n_threads` row ranges.
 * metric: 0/2 Euclidean, 1/3 Cosine.  Returns the number of results (<= limit <= 64). */
uint64_t vfsc_chunked_scan(const float *rows_f32, const int8_t *rows_i8, float scale, uint64_t n_rows, uint32_t d,
                           const float *query, int metric, uint64_t limit, int n_threads, uint64_t *out_pos,
                           float *out_dist)
{
    if (limit == 0 || limit > VFSC_MAX_K || n_threads < 1 || n_threads > 1024)
        return 0;
    float lut[256];
    for (int b = 0; b < 256; ++b)
        lut[b] = (float)(int8_t)b / scale;
    job *jobs = (job *)calloc((size_t)n_threads, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    int avx2 = vfsc_have_avx2();
    uint64_t per = (n_rows + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
    for (int t = 0; t < n_threads; ++t) {
        uint64_t r0 = per * (uint64_t)t, r1 = r0 + per;
        if (r0 > n_rows)
            r0 = n_rows;
        if (r1 > n_rows)
            r1 = n_rows;
        jobs[t] = (job){rows_f32, rows_i8, lut, r0, r1, d, query, metric & 1, (size_t)limit, 0, {0}, {0}, avx2};
        if (t > 0)
            pthread_create(&th[t], NULL, run_job, &jobs[t]);
    }
    run_job(&jobs[0]);
    for (int t = 1; t < n_threads; ++t)
        pthread_join(th[t], NULL);
    /* merge: ranges are in row order, so a stable merge keeps (distance, row) order */
    size_t idx[1024] = {0};
    uint64_t n = 0;
    while (n < limit) {
        int best = -1;
        for (int t = 0; t < n_threads; ++t)
            if (idx[t] < jobs[t].len && (best < 0 || jobs[t].out_d[idx[t]] < jobs[best].out_d[idx[best]]))
                best = t;
        if (best < 0)
            break;
        out_pos[n] = jobs[best].out_r[idx[best]];
        out_dist[n] = jobs[best].out_d[idx[best]];
        ++idx[best];
        ++n;
    }
    free(jobs);
    free(th);
    return n;
}
