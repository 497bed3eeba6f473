/*
 * vfsx.h -- C ABI of the MI355X-native exact-search engine that replaces the
 * linear-search path of RaulMoldes/VFS (Ranker::search -> exact_search ->
 * calculate_distance + sort/truncate top-k).
 *
 * The reference has no FFI/plugin boundary of its own (no `extern "C"`, no
 * `unsafe`).  The seam this ABI replaces is the Rust method
 *
 *   Ranker::search(&mut self, query: &VFSVector, num_vectors_per_iteration: usize,
 *                  result_limit: Option<usize>, manager: &mut VFSManager)
 *       -> io::Result<Vec<(u64, f32)>>                    vfs/src/vfs/rank.rs:90-105
 *
 * and the data feed it pulls from, VFSManager::load_batch
 * (vfs/src/vfs/storage_manager.rs:166-217).  Each entry point below cites the
 * reference lines it stands in for.  All citations are relative to the reference
 * repository root.
 *
 * Conventions
 *   - plain C types only; caller owns every host buffer, the library owns all
 *     device memory;
 *   - every function returns VFSX_OK (0) or a negative vfsx_status; it never
 *     unwinds or aborts across the ABI (the reference panics instead:
 *     rank.rs:31,151,273-275); vfsx_last_error() gives the message (thread-local);
 *   - there is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point fails with VFSX_ERR_NO_DEVICE;
 *   - one search at a time per corpus handle (the reference serialises every
 *     request behind one mutex: vfs/src/main.rs:49, tcp.rs:367).
 */
#ifndef VFSX_H
#define VFSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error classes mirror VFSError (vfs/src/vfs/err.rs:10-16) and the status table
 * at vfs/src/vfs/tcp.rs:330-348. */
typedef enum vfsx_status {
    VFSX_OK = 0,
    VFSX_ERR_IO = -1,             /* VFSError::IoError            -> HTTP 500 */
    VFSX_ERR_MEMTABLE = -2,       /* VFSError::MemtableError      -> HTTP 500 */
    VFSX_ERR_INVALID_VECTOR = -3, /* VFSError::InvalidVector      -> HTTP 400 */
    VFSX_ERR_ID_GENERATION = -4,  /* VFSError::IdGenerationError  -> HTTP 500 */
    VFSX_ERR_SERIALIZATION = -5,  /* VFSError::SerializationError -> HTTP 422 */
    VFSX_ERR_INTERNAL = -6,       /* HIP/runtime failure          -> HTTP 500 */
    VFSX_ERR_NO_DEVICE = -7       /* no gfx950 device / HIP extension unusable */
} vfsx_status;

/* Storage dtype of a corpus: VFSVector::Dense / VFSVector::Quantized
 * (vfs/src/vfs/vector.rs:39-43). */
typedef enum vfsx_dtype {
    VFSX_F32 = 0, /* Vector { vector: Vec<f32> }                     vector.rs:21-25 */
    VFSX_I8 = 1   /* QuantizedVector { vector: Vec<i8>, scale_factor } vector.rs:30-35 */
} vfsx_dtype;

/* DistanceMethod, in the reference's declaration order (rank.rs:67-72).
 * SIMD_* are arithmetically identical to the scalar variants (ordered reduce_sum,
 * rank.rs:25-26,46-48); unlike the reference they accept any dimension
 * (the reference panics unless d is one of 2,4,8,16,32,64: rank.rs:31,303,307). */
typedef enum vfsx_metric {
    VFSX_EUCLIDEAN = 0,
    VFSX_COSINE = 1,
    VFSX_SIMD_EUCLIDEAN = 2,
    VFSX_SIMD_COSINE = 3,
    /* EXTENSION, not a reference DistanceMethod ("A6 decision" (ii)): the
     * labelled integer-domain fast mode for VFSX_I8 corpora.  The query is quantized with
     * the reference's own rule (VFSVector::as_i8_vec(Some(scale)), vector.rs:293-302 ->
     * Vector::quantize, vector.rs:94-108), S = sum (qi - x)^2 is exact in integers and
     * dist = sqrt(S as f32) / scale.  Its results differ from the reference's (the query's
     * quantization error), so it is never selected implicitly; VFSX_EUCLIDEAN on an int8
     * corpus stays the reference-faithful definition.  dim <= 32768; f32 corpora are
     * rejected with VFSX_ERR_INVALID_VECTOR. */
    VFSX_EUCLIDEAN_I8DOMAIN = 4
} vfsx_metric;

/* One ranked hit as produced on the device.  `row` is the global scan position
 * (row_base + local row): the reference's tie-break is (distance asc, first-seen
 * scan position asc) (rank.rs:170, stable sort), so the position travels with the
 * distance through every merge stage, including the cross-GPU one. */
typedef struct vfsx_hit {
    float dist;    /* the f32 the reference would put in the (u64, f32) tuple */
    uint32_t flags; /* bit 0: valid */
    uint64_t row;  /* global scan position */
    uint64_t id;   /* VFSVector::id() (vector.rs:163-168) */
} vfsx_hit;

typedef struct vfsx_corpus vfsx_corpus;

/* ---- runtime ---------------------------------------------------------------- */

/* Number of usable HIP devices (0 when there is none; never fails). */
int vfsx_device_count(void);

/* Start-up check for a host that wants `n_devices` GPUs (one process per GPU passes 1): VFSX_OK when at
 * least that many gfx950 devices are visible, VFSX_ERR_NO_DEVICE otherwise -- the place to fail
 * loudly at boot (main.rs:47-62 has no counterpart: the reference needs no device).  Allocates nothing; it
 * touches devices 0 .. n_devices-1 once (context + the static workspace of vfsx_search_device_noalloc) and
 * leaves the caller's current device as it found it. */
int vfsx_init(int n_devices);

/* Message of the last failure on this thread ("" if none). */
const char *vfsx_last_error(void);

/* Library / layout version string, e.g. "vfsx 0.1 gfx950 rb64". */
const char *vfsx_version(void);

/* ---- corpus (replaces VFSManager as the thing the scan reads) ---------------- */

/*
 * Creates an empty device-resident corpus on HIP device `device`.
 *   dim            vector dimension (ServerState::vector_dimension, tcp.rs:60-63); > 0
 *   dtype          VFSX_F32 or VFSX_I8
 *   scale          QuantizedVector::scale_factor (vector.rs:33); the store always uses
 *                  127.0 (vector.rs:188).  Ignored for VFSX_F32.
 *   capacity_rows  rows to back with memory up front.  The store grows past it without moving a row: the
 *                  row data, norms and ids each live in ONE virtual address range reserved for the device's
 *                  worth of rows, and memory is mapped behind it in equal-size segments (64 MiB to 1 GiB for the
 *                  row data, fixed by the first growth; HIP virtual memory management) -- an append never copies what is
 *                  stored, a store that was not sized up front can fill the card (round 2 reallocated twice
 *                  the size and copied: a third of the card at most), and device addresses handed out earlier
 *                  stay valid.  The reference's feed is an unbounded append (storage_manager.rs:139-157).
 *                  Where the driver offers no virtual memory management (or VFSX_VMM=0) the arrays are single
 *                  blocks that grow by reallocation and copy.
 * Rows are kept in scan order = insertion order = ascending id
 * (storage_manager.rs:139-157: flush order).  Returns NULL on failure.
 */
vfsx_corpus *vfsx_corpus_create(uint32_t dim, int dtype, float scale, uint64_t capacity_rows,
                                int device);

void vfsx_corpus_destroy(vfsx_corpus *c);

/*
 * Appends `n` rows (row-major, `dim` elements each, of the corpus dtype) in scan
 * order.  Stands in for the records VFSManager::load_batch would yield
 * (storage_manager.rs:166-217): for VFSX_I8 the bytes are QuantizedVector::vector
 * and are dequantized on read exactly as storage_manager.rs:202-210 does.
 * `ids` NULL => ids continue 1,2,3.. (storage_manager.rs:76,126-130).
 * Host rows are staged through pinned double buffers and copied with
 * hipMemcpyAsync, then laid out on the device ("HBM layout").
 * One corpus (= one shard) holds at most 2^32 - 66 rows: the row's scan position travels in 32 bits of the key the
 * kernels rank by (larger stores are row-sharded, vfsx_corpus_set_row_base).  An append beyond that fails with
 * VFSX_ERR_MEMTABLE and leaves the corpus as it was.  (The matrix-core batch path serves shards of up to
 * 2^32 - 512 rows; larger ones keep the exact scan.)
 */
int vfsx_corpus_append(vfsx_corpus *c, const void *rows, const uint64_t *ids, uint64_t n);

/*
 * VFSX_I8 corpora only: appends f32 rows, quantizing them on the device with the
 * store's rule  q = (v * scale).round() as i8  (Vector::quantize(Some(127.0)),
 * vector.rs:94-108,188; flush path storage_manager.rs:139-157).
 */
int vfsx_corpus_append_quantize(vfsx_corpus *c, const float *rows, const uint64_t *ids,
                                uint64_t n);

/*
 * Online inserts (register_vector_from_vec one at a time, storage_manager.rs:317-335; the reference keeps up to 10
 * in its memtable, :12,310-312).  Off by default: every append reaches the device before it returns.  With
 * max_pending_rows > 0, appends of <= 16 rows are staged in host memory and committed together -- one H2D copy, one
 * layout / quantize launch, one norm launch for the lot -- as soon as max_pending_rows of them wait, and before
 * ANYTHING reads the store (every search, get_rows / get_ids, larger appends, log loads, enable_mfma ...): a vector
 * is searchable the moment its append returns, exactly once (the reference shows memtable vectors twice).
 * An error of the deferred device work (out of memory) is reported by the call that commits; the rows stay staged (their
 * appends returned success: nothing is dropped silently) and every later call that must commit reports the error again
 * until a commit goes through or the corpus is destroyed.
 * vfsx_corpus_rows counts staged rows.  vfsx_search_device_noalloc never commits: with rows staged it fails with
 * VFSX_ERR_MEMTABLE -- call vfsx_corpus_commit first.  0 switches batching off and commits.
 */
int vfsx_corpus_set_insert_batch(vfsx_corpus *c, uint32_t max_pending_rows);
/* Commit the staged appends now (no-op when none wait). */
int vfsx_corpus_commit(vfsx_corpus *c);

/*
 * The store's quantization rule on the host: out[i] = (values[i] * scale).round() as i8 (Vector::quantize(Some(scale)),
 * vector.rs:94-108: f32 multiply, round half away from zero, saturating cast, NaN -> 0) -- the bytes
 * vfsx_corpus_append_quantize leaves in HBM for the same input (tests/test_gpu_api.py compares them), for hosts that
 * write the stored form to the reference's log (storage_manager.rs:139-157) without reading it back.  Needs no GPU.
 */
int vfsx_quantize_rows_host(const float *values, uint64_t n, float scale, int8_t *out);

/* Number of rows currently searchable. */
uint64_t vfsx_corpus_rows(const vfsx_corpus *c);
uint32_t vfsx_corpus_dim(const vfsx_corpus *c);
int vfsx_corpus_dtype(const vfsx_corpus *c);

/*
 * Global scan position of this corpus' row 0 (default 0).  Used when the vector
 * set is row-sharded over several GPUs: shard g holds a contiguous range, so
 * (row_base + local row) order equals the reference's scan order.
 */
int vfsx_corpus_set_row_base(vfsx_corpus *c, uint64_t row_base);

/*
 * Copies rows back as f32 (as_f32_vec semantics, vector.rs:281-290: an int8 row
 * comes back dequantized).  `rows` are LOCAL row indices.  Used for result
 * hydration (get_vector_by_id, storage_manager.rs:235-296; tcp.rs:408-427).
 */
int vfsx_corpus_get_rows_f32(vfsx_corpus *c, const uint64_t *rows, uint64_t n, float *out);

/* Raw stored bytes of LOCAL rows (f32 or i8 per the corpus dtype), row-major. */
int vfsx_corpus_get_rows_raw(vfsx_corpus *c, const uint64_t *rows, uint64_t n, void *out);

/* id of LOCAL rows. */
int vfsx_corpus_get_ids(vfsx_corpus *c, const uint64_t *rows, uint64_t n, uint64_t *out);

/*
 * Benchmark utility (no reference counterpart): fills the corpus on the device
 * with `n` rows of the counter-based synthetic generator described in DESIGN.md
 * (element (i,j) <- splitmix64(seed + (row0+i)*dim + j), uniform in [-1,1); an
 * int8 corpus stores quantize(value, scale)).  The corpus must be empty (ids are id0, id0+1, ...) or
 * hold implicit ids that id0 continues (id0 = first id + rows: filling in pieces).
 */
int vfsx_corpus_fill_synthetic(vfsx_corpus *c, uint64_t seed, uint64_t row0, uint64_t n,
                               uint64_t id0);

/*
 * How the store is laid out in device memory right now (no reference counterpart): rows backed by memory, slabs
 * behind the row data (0 = one hipMalloc block: no virtual memory management), bytes ever copied because an array
 * grew (0 with slabs), device bytes mapped for the store's arrays (rows, norms, ids, batch-path operands).
 * Any pointer may be NULL.
 */
int vfsx_corpus_storage_info(vfsx_corpus *c, uint64_t *capacity_rows, uint64_t *segments, uint64_t *copied_bytes,
                             uint64_t *device_bytes);

/* ---- ingest from an existing VFS data directory ------------ */

/* What a log file holds (first record's shape; the reference never mixes shapes in a store). */
typedef struct vfsx_log_info {
    uint64_t records;   /* well-formed records found */
    uint32_t dim;       /* Vec length of the first record */
    int dtype;          /* VFSX_F32 (Dense) or VFSX_I8 (Quantized) */
    float scale;        /* scale_factor of the first record (Quantized only) */
    uint64_t first_id, last_id;
    uint64_t file_bytes;
} vfsx_log_info;

/*
 * Parses the reference's append-only log `data/vectors.dat`
 * (writer save_vector serializer.rs:25-71, reader load_vectors serializer.rs:77-153):
 * records of  DE AD BE EF | u64-LE length | bincode(VFSVector).  The whole file is walked
 * (not the reference's 1 KiB window, which loses every record larger than that):
 * non-marker bytes are skipped one at a time and an incomplete trailing record ends the walk,
 * as in the reference; a payload that does not decode is VFSX_ERR_SERIALIZATION.
 */
int vfsx_log_inspect(const char *path, vfsx_log_info *info);

/*
 * Appends every record of the log to the corpus in file order (= scan order), keeping the
 * records' ids and their VectorMetadata (vector.rs:11-17) host-side.  A Quantized record
 * loaded into a VFSX_F32 corpus is stored dequantized (as_f32_vec, vector.rs:284-288); a
 * Dense record cannot enter a VFSX_I8 corpus (VFSVector::dequantize fails, vector.rs:194-198).
 */
int vfsx_corpus_load_log(vfsx_corpus *c, const char *path, uint64_t *n_loaded);

/* The same for records [first_record, first_record + max_records) of the file, in file order: what
 * one rank of a row-sharded deployment loads (contiguous row ranges, so that
 * (shard, local row) order equals the reference's scan order).  Earlier records are stepped over by
 * their length field only. */
int vfsx_corpus_load_log_range(vfsx_corpus *c, const char *path, uint64_t first_record, uint64_t max_records,
                               uint64_t *n_loaded);

/*
 * VectorMetadata of a LOCAL row as JSON: {"name":"..","tags":[..],"created_at":".."}
 * (empty strings for rows appended without metadata).  Writes at most cap-1 bytes + NUL;
 * *needed receives the size required.  Feeds the /search hydration (tcp.rs:408-427).
 */
int vfsx_corpus_get_metadata_json(vfsx_corpus *c, uint64_t row, char *buf, uint64_t cap,
                                  uint64_t *needed);

/* ---- search (replaces Ranker::search / exact_search, rank.rs:90-186) --------- */

/*
 * Exact top-k of each of `nq` f32 queries (row-major, `dim` each; the reference's
 * query is always Dense f32: tcp.rs:377-379) against the corpus.
 *   k        result_limit.unwrap_or(5) (rank.rs:129); 1..VFSX_MAX_K
 *   metric   vfsx_metric
 * Outputs (host): out_ids[nq*k], out_dist[nq*k], ranked ascending by
 * (distance, scan position); out_count[q] = min(k, rows) hits are valid for query q
 * (rank.rs:172: fewer than `limit` only when the corpus is smaller).
 * num_vectors_per_iteration (rank.rs:91) has no counterpart: the reference's result
 * does not depend on it, except that 0 yields an empty result, which
 * the host wrappers reproduce.
 */
int vfsx_search(vfsx_corpus *c, const float *queries, uint32_t nq, uint32_t k, int metric,
                uint64_t *out_ids, float *out_dist, uint32_t *out_count);

/*
 * Same search with device-resident inputs and outputs, enqueued on HIP stream
 * `stream` (a hipStream_t passed as void*; NULL = the default stream) without any
 * host synchronisation:
 *   d_queries  device, nq*dim f32
 *   d_hits     device, nq*k vfsx_hit; entries past the valid count have flags = 0
 * Used by the multi-GPU path (local top-k -> all-gather -> vfsx_merge_hits_device)
 * and by the benchmark (inputs already resident in HBM).
 * Searches on one corpus share its workspaces: a call on a different stream than the previous call's first waits
 * for the whole device (hipDeviceSynchronize -- the previous stream may be gone by then and is never touched
 * again), so a stream per request is safe, and a host that wants overlap uses one corpus handle per stream.
 */
int vfsx_search_device(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k,
                       int metric, vfsx_hit *d_hits, void *stream);

/*
 * The entry point that allocates nothing.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same arguments, same results (every byte of every vfsx_hit) as vfsx_search_device, always through the exact
 * scan (never the matrix-core batch path, whose workspaces grow on demand).  The device memory the scan needs
 * beyond its inputs and outputs -- the per-workgroup top-k lists (8 MiB) and, when dim is not a whole number
 * of 16-byte chunks or the metric is VFSX_EUCLIDEAN_I8DOMAIN, one launch's staged queries (1 MiB) -- is a
 * static 9 MiB device array inside the library, one copy per GPU for the life of the process.  The call
 * makes no hipMalloc / hipFree / allocating library call; the array's device address is looked up once per
 * device (hipGetSymbolAddress) by vfsx_init, or by the first call on that device -- call vfsx_init first
 * when even that must not happen on the hot path.  Limits, reported as VFSX_ERR_INVALID_VECTOR with a
 * message naming the static workspace: k <= 1024 (one pass); a staged query must fit 1 MiB
 * (dim <= 262144 when staging applies); nq is unlimited (several launches share the workspace in stream
 * order).  Every call on one GPU uses that GPU's array: calls on the same GPU that overlap (different
 * streams without ordering, or different host threads) race on it.  Nothing in the array needs to start at
 * zero: each launch overwrites what its select kernel reads.  Profiling (vfsx_corpus_set_profiling) does
 * not time these calls.
 * Under HIP graph capture the launches bake in the corpus' device addresses and row count: a captured graph
 * is valid until the next vfsx_corpus_append* / vfsx_corpus_load_log* / vfsx_corpus_enable_mfma on that corpus
 * (a replay afterwards scans the old row count) -- capture again after such a call.  Storage is segmented
 * ("HBM layout"): rows already stored never move when the corpus grows.
 */
int vfsx_search_device_noalloc(vfsx_corpus *c, const float *d_queries, uint32_t nq, uint32_t k,
                               int metric, vfsx_hit *d_hits, void *stream);

/*
 * Cross-shard merge (no reference counterpart): for each of nq
 * queries, d_in holds n_lists ranked lists of k hits laid out [list][query][k]
 * (the layout an all-gather of every rank's [query][k] block produces).  Writes
 * the k best by (distance, row) to d_out[nq*k].  Device pointers, enqueued on `stream`.
 */
int vfsx_merge_hits_device(const vfsx_hit *d_in, uint32_t n_lists, uint32_t nq, uint32_t k,
                           vfsx_hit *d_out, int device, void *stream);

/* Host version of the same merge (pure host logic; needs no GPU). */
int vfsx_merge_hits_host(const vfsx_hit *in, uint32_t n_lists, uint32_t nq, uint32_t k,
                         vfsx_hit *out);

/* ---- row-sharded search over several GPUs (no reference counterpart) ------------------
 *
 * One process per GPU.  Rank g holds a contiguous range of the scan order (vfsx_corpus_set_row_base) and
 * scans it locally; the per-rank top-k lists are exchanged with ONE RCCL all-gather per query batch
 * (ncclAllGather over xGMI, nq * k * 24 bytes per rank) and merged on every rank by (distance, global row),
 * which is the reference's (distance, first-seen) order (rank.rs:170) because the ranges are contiguous.
 * RCCL (librccl.so.1) is loaded at run time on the first vfsx_comm_* call; without it these entry points
 * fail with VFSX_ERR_NO_DEVICE and everything else keeps working.
 *
 * Bootstrap, as with any NCCL/RCCL communicator: rank 0 calls vfsx_comm_unique_id and hands the 128 bytes to
 * every rank by whatever transport the host already has (a file, a socket, MPI, torch.distributed); then all
 * ranks call vfsx_comm_init (collective: it returns once every rank has joined).
 */
#define VFSX_COMM_ID_BYTES 128
typedef struct vfsx_comm vfsx_comm;

int vfsx_comm_unique_id(void *id_out /* VFSX_COMM_ID_BYTES */);
/* `device`: the HIP device this rank's corpus lives on.  NULL on failure (vfsx_last_error).
 * The wait is bounded: ncclCommInitRank cannot be cancelled, so a rank still inside it after
 * VFSX_COMM_INIT_TIMEOUT_S seconds (default 300; 0 = wait for ever) reports the peer that never joined on stderr and
 * ends the PROCESS with status 3 -- agree among the ranks that every one is healthy BEFORE calling this (bench.py
 * and vfs_amd.sharded do).  On success every buffer the exchange needs is allocated here, sized for
 * VFSX_COMM_HITS hits per rank and batch (default 4096 x 10) and VFSX_COMM_QUERY_FLOATS query floats of the host
 * entry point (default 4096 x 1024): searches within those sizes make no allocation in this layer afterwards. */
vfsx_comm *vfsx_comm_init(int rank, int world, const void *id, int device);
/* Widens those buffers (nq * k hits per rank and batch; query floats of vfsx_search_sharded_host, 0 = keep). */
int vfsx_comm_reserve(vfsx_comm *comm, uint64_t hits_per_rank, uint64_t query_floats);
/* Device / pinned allocations this communicator has made so far (tests: it must not move after init). */
uint64_t vfsx_comm_alloc_count(const vfsx_comm *comm);
void vfsx_comm_destroy(vfsx_comm *comm);
int vfsx_comm_rank(const vfsx_comm *comm);
int vfsx_comm_world(const vfsx_comm *comm);
/* RCCL's version code (e.g. 22703), 0 when RCCL cannot be loaded. */
int vfsx_comm_rccl_version(void);

/*
 * The collective search: vfsx_search_device on this rank's shard, the all-gather, the merge -- all enqueued
 * on `stream` (the matrix-core batch path, if enabled on the corpus, synchronises it once as documented
 * below).  Every rank must call it with the same nq / k; d_hits[nq*k] receives the same global ranking on
 * every rank.  It is a collective: a rank that does not call it leaves the others waiting in the all-gather.  A
 * rank whose LOCAL scan fails (bad arguments on that rank only, device memory that ran dry) still joins the
 * all-gather, with an empty list, and then returns its error -- the peers complete with the remaining shards' hits
 * and must learn of the failure from the host protocol.  Argument errors that every rank shares (k = 0, null
 * handles) return before the collective on every rank alike.  Within the sizes reserved at vfsx_comm_init /
 * vfsx_comm_reserve this layer allocates nothing; the corpus' own scan workspaces grow on its first searches.
 */
int vfsx_search_sharded(vfsx_corpus *c, vfsx_comm *comm, const float *d_queries, uint32_t nq, uint32_t k,
                        int metric, vfsx_hit *d_hits, void *stream);

/* The same with host buffers, shaped like vfsx_search (what a Ranker-level caller uses): queries[nq*dim] in,
 * out_ids / out_dist [nq*k] and out_count[nq] out; runs on a stream the communicator owns and returns when the
 * results are on the host. */
int vfsx_search_sharded_host(vfsx_corpus *c, vfsx_comm *comm, const float *queries, uint32_t nq, uint32_t k,
                             int metric, uint64_t *out_ids, float *out_dist, uint32_t *out_count);

/* The exchange alone, for hosts that produce the local lists themselves (e.g. one scan launch per query):
 * d_local[nq*k] = this rank's ranked lists; all-gather + merge into d_hits[nq*k]. */
int vfsx_comm_merge_hits(vfsx_comm *comm, const vfsx_hit *d_local, uint32_t nq, uint32_t k, vfsx_hit *d_hits,
                         void *stream);

/*
 * Many-query batches on the matrix cores (no reference counterpart: the reference has no
 * multi-query API, rank.rs:90; a batch is defined as the single-query result per query).
 * enable: 0 = off, 1 = on (mirror chosen by the library), 2 = on with a bf16 mirror, 3 = on with an int8 mirror.
 * An f32 store gets a mirror the matrix cores multiply: int8 with one scale per row (+1 byte per element, read by
 * v_mfma_i32_32x32x32_i8 at twice the bf16 rate) or bf16 (+2 bytes per element); with 1 the library starts on int8
 * and moves to bf16 for good when the rows' measured quantization residual is large or more than 1/8 of the queries
 * (counted over windows of >= 64) come back.  An int8 store needs no mirror: its rb64 rows are the int8 operand (+12 bytes per row).
 * Afterwards vfsx_search / vfsx_search_device route Cosine and Euclidean (and Simd*) batches of >= 4 queries -- on an
 * f32 store of >= 1.75e8 elements (3e8 with the bf16 mirror) every call with k <= 32, single queries included (while such
 * calls mostly pass the check): the prefilter reads
 * the mirror's 1 or 2 bytes per element where the exact scan reads 4 (VFSX_MFMA_MIN_BATCH overrides) -- with
 * k <= 128 on corpora of >= 16384 rows whose row norms all lie in [2^-30, 2^30] (queries outside that range take the
 * exact scan) through Q.X^T on MFMA + an exact sequential re-score of the surviving candidates + a bound check built
 * from the measured rounding / quantization residuals, and re-run the exact scan for any query the check cannot
 * clear: the results stay bit-identical to the single-query scan.  On this path vfsx_search_device synchronises
 * `stream` once (to read the per-query check flags).
 */
int vfsx_corpus_enable_mfma(vfsx_corpus *c, int enable);
/* Queries answered through the MFMA path and how many of them fell back to the exact scan. */
int vfsx_corpus_mfma_stats(vfsx_corpus *c, uint64_t *queries, uint64_t *fallbacks, int reset);
/*
 * What the matrix-core path of this corpus multiplies right now (no reference counterpart): *kind = 0 path off,
 * 1 bf16 mirror, 2 int8 mirror of an f32 store, 3 the int8 store itself; *residual = R_x, the largest measured
 * rounding / quantization residual || x/||x|| - mirror(x) ||_2 over the rows (0 for kind 3 and before the first build).
 */
int vfsx_corpus_mfma_mirror(vfsx_corpus *c, int *kind, float *residual);

/*
 * Profiling aid: when enabled, vfsx_search / vfsx_search_device bracket the scan
 * kernel (the prefilter GEMM on the MFMA path) with HIP events on the stream it is launched on; vfsx_corpus_scan_time_ms
 * returns the accumulated kernel milliseconds and launch count since the last reset
 * (synchronises on the last event).
 */
int vfsx_corpus_set_profiling(vfsx_corpus *c, int enabled);
int vfsx_corpus_scan_time_ms(vfsx_corpus *c, double *total_ms, uint64_t *launches, int reset);

/* Largest result_limit.  Any length comes from ONE pass over the corpus: up to 128 hits from per-wavefront lists,
 * longer results from a radix select over every row's (distance, row) key.  The reference's limit is unbounded
 * (rank.rs:129,172-173); this bound only keeps result buffers addressable.  vfsx_search_device_noalloc and
 * VFSX_EUCLIDEAN_I8DOMAIN serve k <= 1024. */
#define VFSX_MAX_K (1u << 20)

#ifdef __cplusplus
}
#endif
#endif /* VFSX_H */
