//! Thin FFI over `include/vfsx.h` plus a `Ranker`-shaped safe wrapper with the
//! signatures of the reference's `vfs/src/vfs/rank.rs:61-105`.  No logic lives here
//! that is not also reachable through the C ABI.  NOT compiled in the build image
//! (no Rust toolchain): see INTEGRATION.md.
use std::ffi::CStr;
use std::io;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct VfsxCorpus {
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

pub const VFSX_F32: c_int = 0;
pub const VFSX_I8: c_int = 1;

extern "C" {
    pub fn vfsx_device_count() -> c_int;
    pub fn vfsx_init(n_devices: c_int) -> c_int;
    pub fn vfsx_corpus_load_log_range(c: *mut VfsxCorpus, path: *const c_char, first_record: u64, max_records: u64, n_loaded: *mut u64) -> c_int;
    pub fn vfsx_last_error() -> *const c_char;
    pub fn vfsx_corpus_create(dim: u32, dtype: c_int, scale: f32, capacity_rows: u64, device: c_int) -> *mut VfsxCorpus;
    pub fn vfsx_corpus_destroy(c: *mut VfsxCorpus);
    pub fn vfsx_corpus_append(c: *mut VfsxCorpus, rows: *const c_void, ids: *const u64, n: u64) -> c_int;
    pub fn vfsx_corpus_append_quantize(c: *mut VfsxCorpus, rows: *const f32, ids: *const u64, n: u64) -> c_int;
    pub fn vfsx_corpus_rows(c: *const VfsxCorpus) -> u64;
    pub fn vfsx_corpus_get_rows_f32(c: *mut VfsxCorpus, rows: *const u64, n: u64, out: *mut f32) -> c_int;
    pub fn vfsx_search(
        c: *mut VfsxCorpus, queries: *const f32, nq: u32, k: u32, metric: c_int,
        out_ids: *mut u64, out_dist: *mut f32, out_count: *mut u32,
    ) -> c_int;
}

fn last_error(code: c_int) -> io::Error {
    let msg = unsafe { CStr::from_ptr(vfsx_last_error()) }.to_string_lossy().into_owned();
    // VFSError::InvalidVector -> InvalidInput (HTTP 400, tcp.rs:331); everything else -> Other (500)
    let kind = if code == -3 { io::ErrorKind::InvalidInput } else { io::ErrorKind::Other };
    io::Error::new(kind, msg)
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

impl Corpus {
    pub fn new(dim: usize, quantize: bool, capacity_rows: u64, device: i32) -> io::Result<Self> {
        let h = unsafe { vfsx_corpus_create(dim as u32, if quantize { VFSX_I8 } else { VFSX_F32 }, 127.0, capacity_rows, device) };
        if h.is_null() {
            return Err(last_error(-7));
        }
        Ok(Corpus { h, dim, quantize })
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
        if rc != 0 {
            return Err(last_error(rc));
        }
        Ok(unsafe { vfsx_corpus_rows(self.h) })
    }
}

impl Drop for Corpus {
    fn drop(&mut self) {
        unsafe { vfsx_corpus_destroy(self.h) }
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

    /// rank.rs:90-92, with `&mut Corpus` where the reference takes `&mut VFSManager`
    /// and `&[f32]` where it takes `&VFSVector` (the query is always Dense, tcp.rs:379).
    pub fn search(
        &mut self, query: &[f32], num_vectors_per_iteration: usize, result_limit: Option<usize>, manager: &mut Corpus,
    ) -> io::Result<Vec<(u64, f32)>> {
        match self.search_type {
            SearchType::Exact => {}
            SearchType::Approximate => {
                return Err(io::Error::new(io::ErrorKind::Unsupported, "approximate search is outside the vfsx hot path"))
            }
        }
        if query.len() != manager.dim {
            return Err(io::Error::new(io::ErrorKind::InvalidInput, "Los vectores deben tener la misma dimensión"));
        }
        let limit = result_limit.unwrap_or(5); // rank.rs:129
        if num_vectors_per_iteration == 0 {
            return Ok(Vec::new()); // load_batch(0) yields nothing
        }
        let mut ids = vec![0u64; limit];
        let mut dist = vec![0f32; limit];
        let mut count = 0u32;
        let rc = unsafe {
            vfsx_search(manager.h, query.as_ptr(), 1, limit as u32, self.distance_method as c_int,
                        ids.as_mut_ptr(), dist.as_mut_ptr(), &mut count)
        };
        if rc != 0 {
            return Err(last_error(rc));
        }
        Ok(ids.into_iter().zip(dist).take(count as usize).collect())
    }
}
