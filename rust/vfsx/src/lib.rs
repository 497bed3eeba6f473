//! Thin FFI over `include/vfsx.h` plus a `Ranker`-shaped safe wrapper with the
//! signatures of the reference's `vfs/src/vfs/rank.rs:61-105`.  No logic lives here
//! that is not also reachable through the C ABI.  NOT compiled in the build image
//! (no Rust toolchain): see INTEGRATION.md.  `tests/test_abi.py` parses the `extern "C"`
//! block below against `include/vfsx.h` and fails on a missing or mis-typed declaration.
use std::ffi::{CStr, CString};
use std::io;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct VfsxCorpus {
    _private: [u8; 0],
}

#[repr(C)]
pub struct VfsxComm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct VfsxHit {
    pub dist: f32,
    pub flags: u32,
    pub row: u64,
    pub id: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct VfsxLogInfo {
    pub records: u64,
    pub dim: u32,
    pub dtype: c_int,
    pub scale: f32,
    pub first_id: u64,
    pub last_id: u64,
    pub file_bytes: u64,
}

pub const VFSX_F32: c_int = 0;
pub const VFSX_I8: c_int = 1;
pub const VFSX_COMM_ID_BYTES: usize = 128;
pub const VFSX_MAX_K: u32 = 1 << 20;

extern "C" {
    // ---- runtime
    pub fn vfsx_device_count() -> c_int;
    pub fn vfsx_init(n_devices: c_int) -> c_int;
    pub fn vfsx_last_error() -> *const c_char;
    pub fn vfsx_version() -> *const c_char;
    // ---- corpus
    pub fn vfsx_corpus_create(dim: u32, dtype: c_int, scale: f32, capacity_rows: u64, device: c_int) -> *mut VfsxCorpus;
    pub fn vfsx_corpus_destroy(c: *mut VfsxCorpus);
    pub fn vfsx_corpus_append(c: *mut VfsxCorpus, rows: *const c_void, ids: *const u64, n: u64) -> c_int;
    pub fn vfsx_corpus_append_quantize(c: *mut VfsxCorpus, rows: *const f32, ids: *const u64, n: u64) -> c_int;
    pub fn vfsx_corpus_rows(c: *const VfsxCorpus) -> u64;
    pub fn vfsx_corpus_dim(c: *const VfsxCorpus) -> u32;
    pub fn vfsx_corpus_dtype(c: *const VfsxCorpus) -> c_int;
    pub fn vfsx_corpus_set_row_base(c: *mut VfsxCorpus, row_base: u64) -> c_int;
    pub fn vfsx_corpus_get_rows_f32(c: *mut VfsxCorpus, rows: *const u64, n: u64, out: *mut f32) -> c_int;
    pub fn vfsx_corpus_get_rows_raw(c: *mut VfsxCorpus, rows: *const u64, n: u64, out: *mut c_void) -> c_int;
    pub fn vfsx_corpus_get_ids(c: *mut VfsxCorpus, rows: *const u64, n: u64, out: *mut u64) -> c_int;
    pub fn vfsx_corpus_fill_synthetic(c: *mut VfsxCorpus, seed: u64, row0: u64, n: u64, id0: u64) -> c_int;
    pub fn vfsx_quantize_rows_host(values: *const f32, n: u64, scale: f32, out: *mut i8) -> c_int;
    pub fn vfsx_corpus_set_insert_batch(c: *mut VfsxCorpus, max_pending_rows: u32) -> c_int;
    pub fn vfsx_corpus_commit(c: *mut VfsxCorpus) -> c_int;
    pub fn vfsx_corpus_storage_info(c: *mut VfsxCorpus, capacity_rows: *mut u64, segments: *mut u64, copied_bytes: *mut u64, device_bytes: *mut u64) -> c_int;
    // ---- ingest from a VFS data directory
    pub fn vfsx_log_inspect(path: *const c_char, info: *mut VfsxLogInfo) -> c_int;
    pub fn vfsx_corpus_load_log(c: *mut VfsxCorpus, path: *const c_char, n_loaded: *mut u64) -> c_int;
    pub fn vfsx_corpus_load_log_range(c: *mut VfsxCorpus, path: *const c_char, first_record: u64, max_records: u64, n_loaded: *mut u64) -> c_int;
    pub fn vfsx_corpus_get_metadata_json(c: *mut VfsxCorpus, row: u64, buf: *mut c_char, cap: u64, needed: *mut u64) -> c_int;
    // ---- search
    pub fn vfsx_search(c: *mut VfsxCorpus, queries: *const f32, nq: u32, k: u32, metric: c_int, out_ids: *mut u64, out_dist: *mut f32, out_count: *mut u32) -> c_int;
    pub fn vfsx_search_device(c: *mut VfsxCorpus, d_queries: *const f32, nq: u32, k: u32, metric: c_int, d_hits: *mut VfsxHit, stream: *mut c_void) -> c_int;
    pub fn vfsx_search_device_noalloc(c: *mut VfsxCorpus, d_queries: *const f32, nq: u32, k: u32, metric: c_int, d_hits: *mut VfsxHit, stream: *mut c_void) -> c_int;
    pub fn vfsx_merge_hits_device(d_in: *const VfsxHit, n_lists: u32, nq: u32, k: u32, d_out: *mut VfsxHit, device: c_int, stream: *mut c_void) -> c_int;
    pub fn vfsx_merge_hits_host(r#in: *const VfsxHit, n_lists: u32, nq: u32, k: u32, out: *mut VfsxHit) -> c_int;
    // ---- row-sharded search over several GPUs (RCCL)
    pub fn vfsx_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn vfsx_comm_init(rank: c_int, world: c_int, id: *const c_void, device: c_int) -> *mut VfsxComm;
    pub fn vfsx_comm_destroy(comm: *mut VfsxComm);
    pub fn vfsx_comm_rank(comm: *const VfsxComm) -> c_int;
    pub fn vfsx_comm_world(comm: *const VfsxComm) -> c_int;
    pub fn vfsx_comm_rccl_version() -> c_int;
    pub fn vfsx_comm_reserve(comm: *mut VfsxComm, hits_per_rank: u64, query_floats: u64) -> c_int;
    pub fn vfsx_comm_alloc_count(comm: *const VfsxComm) -> u64;
    pub fn vfsx_search_sharded(c: *mut VfsxCorpus, comm: *mut VfsxComm, d_queries: *const f32, nq: u32, k: u32, metric: c_int, d_hits: *mut VfsxHit, stream: *mut c_void) -> c_int;
    pub fn vfsx_search_sharded_host(c: *mut VfsxCorpus, comm: *mut VfsxComm, queries: *const f32, nq: u32, k: u32, metric: c_int, out_ids: *mut u64, out_dist: *mut f32, out_count: *mut u32) -> c_int;
    pub fn vfsx_comm_merge_hits(comm: *mut VfsxComm, d_local: *const VfsxHit, nq: u32, k: u32, d_hits: *mut VfsxHit, stream: *mut c_void) -> c_int;
    // ---- many-query batches on the matrix cores, profiling
    pub fn vfsx_corpus_enable_mfma(c: *mut VfsxCorpus, enable: c_int) -> c_int;
    pub fn vfsx_corpus_mfma_stats(c: *mut VfsxCorpus, queries: *mut u64, fallbacks: *mut u64, reset: c_int) -> c_int;
    pub fn vfsx_corpus_mfma_mirror(c: *mut VfsxCorpus, kind: *mut c_int, residual: *mut f32) -> c_int;
    pub fn vfsx_corpus_set_profiling(c: *mut VfsxCorpus, enabled: c_int) -> c_int;
    pub fn vfsx_corpus_scan_time_ms(c: *mut VfsxCorpus, total_ms: *mut f64, launches: *mut u64, reset: c_int) -> c_int;
}

fn last_error(code: c_int) -> io::Error {
    let msg = unsafe { CStr::from_ptr(vfsx_last_error()) }.to_string_lossy().into_owned();
    // VFSError::InvalidVector -> InvalidInput (HTTP 400, tcp.rs:331); everything else -> Other (500)
    let kind = if code == -3 { io::ErrorKind::InvalidInput } else { io::ErrorKind::Other };
    io::Error::new(kind, msg)
}

fn check(rc: c_int) -> io::Result<()> {
    if rc == 0 {
        Ok(())
    } else {
        Err(last_error(rc))
    }
}

fn c_path(path: &str) -> io::Result<CString> {
    CString::new(path).map_err(|_| io::Error::new(io::ErrorKind::InvalidInput, "path contains a NUL byte"))
}

/// Boot-time check (no reference counterpart): at least `n_devices` gfx950 GPUs, or an error -- the engine has
/// no CPU fallback.
pub fn init(n_devices: i32) -> io::Result<()> {
    check(unsafe { vfsx_init(n_devices as c_int) })
}

pub fn version() -> String {
    unsafe { CStr::from_ptr(vfsx_version()) }.to_string_lossy().into_owned()
}

/// What a `data/vectors.dat` holds (serializer.rs:25-153).
pub fn log_inspect(path: &str) -> io::Result<VfsxLogInfo> {
    let p = c_path(path)?;
    let mut info = VfsxLogInfo::default();
    check(unsafe { vfsx_log_inspect(p.as_ptr(), &mut info) })?;
    Ok(info)
}

/// rank.rs:61-64
pub enum SearchType {
    Exact,
    Approximate,
}

/// rank.rs:67-72 (declaration order = the C enum's values)
#[derive(Clone, Copy)]
pub enum DistanceMethod {
    Euclidean = 0,
    Cosine = 1,
    SimdEuclidean = 2,
    SimdCosine = 3,
    /// Extension (VFSX_EUCLIDEAN_I8DOMAIN): integer-domain fast mode, quantized corpora only.
    EuclideanI8Domain = 4,
}

/// Device-resident stand-in for the part of `VFSManager` the scan reads.
pub struct Corpus {
    h: *mut VfsxCorpus,
    dim: usize,
    quantize: bool,
}

unsafe impl Send for Corpus {}

/// Argument of `vfsx_corpus_enable_mfma` beyond on / off (include/vfsx.h).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum MfmaMirror {
    Auto = 1,
    Bf16 = 2,
    Int8 = 3,
}

impl Corpus {
    pub fn new(dim: usize, quantize: bool, capacity_rows: u64, device: i32) -> io::Result<Self> {
        let h = unsafe { vfsx_corpus_create(dim as u32, if quantize { VFSX_I8 } else { VFSX_F32 }, 127.0, capacity_rows, device) };
        if h.is_null() {
            return Err(last_error(-7));
        }
        Ok(Corpus { h, dim, quantize })
    }

    pub fn handle(&self) -> *mut VfsxCorpus {
        self.h
    }
    pub fn dim(&self) -> usize {
        unsafe { vfsx_corpus_dim(self.h) as usize }
    }
    pub fn quantized(&self) -> bool {
        unsafe { vfsx_corpus_dtype(self.h) == VFSX_I8 }
    }
    pub fn rows(&self) -> u64 {
        unsafe { vfsx_corpus_rows(self.h) }
    }

    /// register_vector_from_vec (storage_manager.rs:317-335); ids continue 1,2,3..
    pub fn register_vector_from_vec(&mut self, data: &[f32]) -> io::Result<u64> {
        if data.is_empty() || data.len() != self.dim {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "Vector data cannot be empty"));
        }
        let rc = unsafe {
            if self.quantize {
                vfsx_corpus_append_quantize(self.h, data.as_ptr(), std::ptr::null(), 1)
            } else {
                vfsx_corpus_append(self.h, data.as_ptr() as *const c_void, std::ptr::null(), 1)
            }
        };
        check(rc)?;
        Ok(self.rows())
    }

    /// Many f32 rows at once (row-major), optional explicit ids; a quantizing store quantizes on the device.
    pub fn register_vectors(&mut self, rows: &[f32], ids: Option<&[u64]>) -> io::Result<()> {
        if self.dim == 0 || rows.len() % self.dim != 0 {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "rows must be a whole number of vectors"));
        }
        let n = (rows.len() / self.dim) as u64;
        if let Some(i) = ids {
            if i.len() as u64 != n {
                return Err(io::Error::new(io::ErrorKind::InvalidInput, "ids must have one entry per row"));
            }
        }
        let idp = ids.map_or(std::ptr::null(), |i| i.as_ptr());
        check(unsafe {
            if self.quantize {
                vfsx_corpus_append_quantize(self.h, rows.as_ptr(), idp, n)
            } else {
                vfsx_corpus_append(self.h, rows.as_ptr() as *const c_void, idp, n)
            }
        })
    }

    /// Already-quantized `QuantizedVector::vector` bytes (vector.rs:30-35).
    pub fn append_quantized(&mut self, rows: &[i8], ids: Option<&[u64]>) -> io::Result<()> {
        if !self.quantize || self.dim == 0 || rows.len() % self.dim != 0 {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "needs a quantized corpus and whole vectors"));
        }
        let idp = ids.map_or(std::ptr::null(), |i| i.as_ptr());
        check(unsafe { vfsx_corpus_append(self.h, rows.as_ptr() as *const c_void, idp, (rows.len() / self.dim) as u64) })
    }

    /// Stream an existing `data/vectors.dat` (or records [first, first + count) of it: one shard) into HBM.
    pub fn load_log(&mut self, path: &str, first: u64, count: Option<u64>) -> io::Result<u64> {
        let p = c_path(path)?;
        let mut n = 0u64;
        check(unsafe { vfsx_corpus_load_log_range(self.h, p.as_ptr(), first, count.unwrap_or(u64::MAX), &mut n) })?;
        Ok(n)
    }
    pub fn load_log_all(&mut self, path: &str) -> io::Result<u64> {
        let p = c_path(path)?;
        let mut n = 0u64;
        check(unsafe { vfsx_corpus_load_log(self.h, p.as_ptr(), &mut n) })?;
        Ok(n)
    }

    /// This shard's first global scan position (row-sharded stores).
    pub fn set_row_base(&mut self, row_base: u64) -> io::Result<()> {
        check(unsafe { vfsx_corpus_set_row_base(self.h, row_base) })
    }

    /// get_vector_by_id's data half (storage_manager.rs:235-296): LOCAL rows back as f32 (`as_f32_vec`).
    pub fn get_rows_f32(&mut self, rows: &[u64]) -> io::Result<Vec<f32>> {
        let mut out = vec![0f32; rows.len() * self.dim];
        check(unsafe { vfsx_corpus_get_rows_f32(self.h, rows.as_ptr(), rows.len() as u64, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// The stored bytes of LOCAL rows of a quantized corpus.
    pub fn get_rows_i8(&mut self, rows: &[u64]) -> io::Result<Vec<i8>> {
        if !self.quantize {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "corpus is not quantized"));
        }
        let mut out = vec![0i8; rows.len() * self.dim];
        check(unsafe { vfsx_corpus_get_rows_raw(self.h, rows.as_ptr(), rows.len() as u64, out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
    pub fn get_ids(&mut self, rows: &[u64]) -> io::Result<Vec<u64>> {
        let mut out = vec![0u64; rows.len()];
        check(unsafe { vfsx_corpus_get_ids(self.h, rows.as_ptr(), rows.len() as u64, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// VectorMetadata (vector.rs:11-17) of a LOCAL row as JSON text.
    pub fn get_metadata_json(&mut self, row: u64) -> io::Result<String> {
        let mut need = 0u64;
        let mut buf = vec![0u8; 512];
        check(unsafe { vfsx_corpus_get_metadata_json(self.h, row, buf.as_mut_ptr() as *mut c_char, buf.len() as u64, &mut need) })?;
        if need as usize > buf.len() {
            buf = vec![0u8; need as usize];
            check(unsafe { vfsx_corpus_get_metadata_json(self.h, row, buf.as_mut_ptr() as *mut c_char, buf.len() as u64, &mut need) })?;
        }
        let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
        Ok(String::from_utf8_lossy(&buf[..end]).into_owned())
    }

    /// Benchmark utility: counter-based synthetic rows generated on the device.
    pub fn fill_synthetic(&mut self, seed: u64, row0: u64, n: u64, id0: u64) -> io::Result<()> {
        check(unsafe { vfsx_corpus_fill_synthetic(self.h, seed, row0, n, id0) })
    }

    /// `nq` queries at once ("A2/A3 + A4 per query"): flat `[nq][k]` ids / distances and per-query counts.
    pub fn search_batch(&mut self, queries: &[f32], k: usize, metric: DistanceMethod) -> io::Result<(Vec<u64>, Vec<f32>, Vec<u32>)> {
        if self.dim == 0 || queries.len() % self.dim != 0 {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "queries must be a whole number of vectors"));
        }
        let nq = queries.len() / self.dim;
        let (mut ids, mut dist, mut cnt) = (vec![0u64; nq * k], vec![0f32; nq * k], vec![0u32; nq]);
        check(unsafe {
            vfsx_search(self.h, queries.as_ptr(), nq as u32, k as u32, metric as c_int, ids.as_mut_ptr(), dist.as_mut_ptr(), cnt.as_mut_ptr())
        })?;
        Ok((ids, dist, cnt))
    }

    /// Device-resident search enqueued on a HIP stream (`stream` = a `hipStream_t`, null = default stream).
    ///
    /// # Safety
    /// `d_queries` / `d_hits` must be device pointers of `nq * dim` floats / `nq * k` hits on this corpus' GPU.
    pub unsafe fn search_device(&mut self, d_queries: *const f32, nq: u32, k: u32, metric: DistanceMethod, d_hits: *mut VfsxHit,
                                stream: *mut c_void) -> io::Result<()> {
        check(vfsx_search_device(self.h, d_queries, nq, k, metric as c_int, d_hits, stream))
    }
    /// The same on the library's static device workspace: no allocation on the call (HIP graph capture, serving loops).
    ///
    /// # Safety
    /// As `search_device`; calls on one GPU must not overlap.
    pub unsafe fn search_device_noalloc(&mut self, d_queries: *const f32, nq: u32, k: u32, metric: DistanceMethod,
                                        d_hits: *mut VfsxHit, stream: *mut c_void) -> io::Result<()> {
        check(vfsx_search_device_noalloc(self.h, d_queries, nq, k, metric as c_int, d_hits, stream))
    }

    /// Many-query batches on the matrix cores (results unchanged).  f32 stores get a mirror chosen by the library
    /// (int8 with per-row scales, +1 byte per element, or bf16, +2); int8 stores need none.
    pub fn enable_mfma(&mut self, on: bool) -> io::Result<()> {
        check(unsafe { vfsx_corpus_enable_mfma(self.h, on as c_int) })
    }
    /// The same with the mirror of an f32 store forced: `MfmaMirror::Bf16` or `MfmaMirror::Int8`.
    pub fn enable_mfma_with(&mut self, mirror: MfmaMirror) -> io::Result<()> {
        check(unsafe { vfsx_corpus_enable_mfma(self.h, mirror as c_int) })
    }
    /// (what the matrix-core path multiplies: 0 off, 1 bf16 mirror, 2 int8 mirror, 3 the int8 store itself; the rows'
    /// largest measured rounding / quantization residual)
    pub fn mfma_mirror(&mut self) -> io::Result<(i32, f32)> {
        let (mut kind, mut res) = (0 as c_int, 0f32);
        check(unsafe { vfsx_corpus_mfma_mirror(self.h, &mut kind, &mut res) })?;
        Ok((kind as i32, res))
    }
    /// (queries answered through the matrix-core path, of which sent back to the exact scan)
    pub fn mfma_stats(&mut self, reset: bool) -> io::Result<(u64, u64)> {
        let (mut q, mut f) = (0u64, 0u64);
        check(unsafe { vfsx_corpus_mfma_stats(self.h, &mut q, &mut f, reset as c_int) })?;
        Ok((q, f))
    }
    pub fn set_profiling(&mut self, on: bool) -> io::Result<()> {
        check(unsafe { vfsx_corpus_set_profiling(self.h, on as c_int) })
    }
    /// (accumulated scan-kernel milliseconds, launches)
    pub fn scan_time_ms(&mut self, reset: bool) -> io::Result<(f64, u64)> {
        let (mut ms, mut n) = (0f64, 0u64);
        check(unsafe { vfsx_corpus_scan_time_ms(self.h, &mut ms, &mut n, reset as c_int) })?;
        Ok((ms, n))
    }
}

impl Drop for Corpus {
    fn drop(&mut self) {
        unsafe { vfsx_corpus_destroy(self.h) }
    }
}

/// Cross-shard merge of ranked lists laid out `[list][query][k]`, on the host.
pub fn merge_hits_host(lists: &[VfsxHit], n_lists: usize, nq: usize, k: usize) -> io::Result<Vec<VfsxHit>> {
    if lists.len() != n_lists * nq * k {
        return Err(io::Error::new(io::ErrorKind::InvalidInput, "lists must hold n_lists * nq * k hits"));
    }
    let mut out = vec![VfsxHit { dist: f32::NAN, flags: 0, row: u64::MAX, id: u64::MAX }; nq * k];
    check(unsafe { vfsx_merge_hits_host(lists.as_ptr(), n_lists as u32, nq as u32, k as u32, out.as_mut_ptr()) })?;
    Ok(out)
}

/// The same on the device (device pointers, enqueued on `stream`).
///
/// # Safety
/// `d_in` / `d_out` must be device pointers of `n_lists * nq * k` / `nq * k` hits on GPU `device`.
pub unsafe fn merge_hits_device(d_in: *const VfsxHit, n_lists: u32, nq: u32, k: u32, d_out: *mut VfsxHit, device: i32,
                                stream: *mut c_void) -> io::Result<()> {
    check(vfsx_merge_hits_device(d_in, n_lists, nq, k, d_out, device as c_int, stream))
}

/// One rank's handle on the RCCL communicator of a row-sharded store (no reference counterpart).
/// Rank 0 makes the id (`Comm::unique_id`), the host hands the 128 bytes to every rank, all ranks call `Comm::new`.
pub struct Comm {
    h: *mut VfsxComm,
}

unsafe impl Send for Comm {}

impl Comm {
    pub fn unique_id() -> io::Result<[u8; VFSX_COMM_ID_BYTES]> {
        let mut id = [0u8; VFSX_COMM_ID_BYTES];
        check(unsafe { vfsx_comm_unique_id(id.as_mut_ptr() as *mut c_void) })?;
        Ok(id)
    }
    pub fn rccl_version() -> i32 {
        unsafe { vfsx_comm_rccl_version() }
    }
    /// Collective: returns once every rank has joined.
    pub fn new(rank: i32, world: i32, id: &[u8; VFSX_COMM_ID_BYTES], device: i32) -> io::Result<Self> {
        let h = unsafe { vfsx_comm_init(rank, world, id.as_ptr() as *const c_void, device) };
        if h.is_null() {
            return Err(last_error(-6));
        }
        Ok(Comm { h })
    }
    pub fn rank(&self) -> i32 {
        unsafe { vfsx_comm_rank(self.h) }
    }
    pub fn world(&self) -> i32 {
        unsafe { vfsx_comm_world(self.h) }
    }
    /// Widens the exchange buffers (`nq * k` hits per rank and batch; query floats of the host entry point):
    /// searches within the reserved sizes never allocate in the communicator.
    pub fn reserve(&mut self, hits_per_rank: u64, query_floats: u64) -> io::Result<()> {
        check(unsafe { vfsx_comm_reserve(self.h, hits_per_rank, query_floats) })
    }
    /// Device / pinned allocations made by this communicator so far.
    pub fn alloc_count(&self) -> u64 {
        unsafe { vfsx_comm_alloc_count(self.h) }
    }

    /// Local scan + ncclAllGather + merge, device buffers, enqueued on `stream`.
    ///
    /// # Safety
    /// As `Corpus::search_device`; every rank must call with the same `nq` and `k`.
    pub unsafe fn search_device(&mut self, shard: &mut Corpus, d_queries: *const f32, nq: u32, k: u32, metric: DistanceMethod,
                                d_hits: *mut VfsxHit, stream: *mut c_void) -> io::Result<()> {
        check(vfsx_search_sharded(shard.h, self.h, d_queries, nq, k, metric as c_int, d_hits, stream))
    }
    /// The exchange alone: this rank's ranked lists in, the global ranking out.
    ///
    /// # Safety
    /// `d_local` / `d_hits` must be device pointers of `nq * k` hits on this rank's GPU.
    pub unsafe fn merge_hits(&mut self, d_local: *const VfsxHit, nq: u32, k: u32, d_hits: *mut VfsxHit, stream: *mut c_void) -> io::Result<()> {
        check(vfsx_comm_merge_hits(self.h, d_local, nq, k, d_hits, stream))
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { vfsx_comm_destroy(self.h) }
    }
}

/// rank.rs:76-81
pub struct Ranker {
    search_type: SearchType,
    distance_method: DistanceMethod,
    #[allow(dead_code)]
    ef_search: usize,
}

impl Ranker {
    /// rank.rs:85-87
    pub fn new(search_type: SearchType, distance_method: DistanceMethod, ef_search: Option<usize>) -> Self {
        Ranker { search_type, distance_method, ef_search: ef_search.unwrap_or(6) }
    }

    fn precheck(&self, query: &[f32], manager: &Corpus) -> io::Result<()> {
        match self.search_type {
            SearchType::Exact => {}
            SearchType::Approximate => {
                return Err(io::Error::new(io::ErrorKind::Unsupported, "approximate search is outside the vfsx hot path"))
            }
        }
        if query.len() != manager.dim {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "Los vectores deben tener la misma dimensión"));
        }
        Ok(())
    }

    /// rank.rs:90-92, with `&mut Corpus` where the reference takes `&mut VFSManager`
    /// and `&[f32]` where it takes `&VFSVector` (the query is always Dense, tcp.rs:379).
    pub fn search(
        &mut self, query: &[f32], num_vectors_per_iteration: usize, result_limit: Option<usize>, manager: &mut Corpus,
    ) -> io::Result<Vec<(u64, f32)>> {
        self.precheck(query, manager)?;
        let limit = result_limit.unwrap_or(5); // rank.rs:129
        if num_vectors_per_iteration == 0 {
            return Ok(Vec::new()); // load_batch(0) yields nothing
        }
        if limit == 0 || limit > VFSX_MAX_K as usize {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "result_limit must be in 1..=VFSX_MAX_K"));
        }
        let mut ids = vec![0u64; limit];
        let mut dist = vec![0f32; limit];
        let mut count = 0u32;
        let rc = unsafe {
            vfsx_search(manager.h, query.as_ptr(), 1, limit as u32, self.distance_method as c_int,
                        ids.as_mut_ptr(), dist.as_mut_ptr(), &mut count)
        };
        check(rc)?;
        Ok(ids.into_iter().zip(dist).take(count as usize).collect())
    }

    /// The same search over a row-sharded store: `shard` is this rank's contiguous range of the scan order,
    /// every rank calls with the same query and receives the same global ranking
    /// (local scan -> ncclAllGather -> merge by (distance, global row)).
    /// A collective: the argument checks below are the same on every rank (same query, same `dim` on every
    /// shard), so either every rank returns early or none does; a rank whose local scan fails later still joins
    /// the all-gather (with an empty list) before it reports -- see `vfsx_search_sharded` in include/vfsx.h.
    pub fn search_sharded(
        &mut self, query: &[f32], num_vectors_per_iteration: usize, result_limit: Option<usize>, shard: &mut Corpus,
        comm: &mut Comm,
    ) -> io::Result<Vec<(u64, f32)>> {
        self.precheck(query, shard)?;
        let limit = result_limit.unwrap_or(5);
        if limit == 0 || limit > VFSX_MAX_K as usize {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "result_limit must be in 1..=VFSX_MAX_K"));
        }
        let mut ids = vec![0u64; limit];
        let mut dist = vec![0f32; limit];
        let mut count = 0u32;
        // every rank joins the collective, also when its caller asked for nothing (num_vectors_per_iteration == 0)
        let rc = unsafe {
            vfsx_search_sharded_host(shard.h, comm.h, query.as_ptr(), 1, limit as u32, self.distance_method as c_int,
                                     ids.as_mut_ptr(), dist.as_mut_ptr(), &mut count)
        };
        check(rc)?;
        if num_vectors_per_iteration == 0 {
            return Ok(Vec::new());
        }
        Ok(ids.into_iter().zip(dist).take(count as usize).collect())
    }
}
