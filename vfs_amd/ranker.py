"""Host-side mirror of the reference's Ranker API over the vfsx C ABI.

Same names, argument meaning and error behaviour as vfs/src/vfs/rank.rs:61-105
(`SearchType`, `DistanceMethod`, `Ranker::new`, `Ranker::search`); `Corpus` stands
where `VFSManager` does for this path (the thing the scan pulls rows from,
storage_manager.rs:166-217), but device-resident.  Everything computes on the GPU
through libvfsx.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


class SearchType(enum.Enum):  # rank.rs:61-64
    Exact = 0
    Approximate = 1


class DistanceMethod(enum.IntEnum):  # rank.rs:67-72 (declaration order)
    Euclidean = 0
    Cosine = 1
    SimdEuclidean = 2
    SimdCosine = 3
    # EXTENSION (include/vfsx.h VFSX_EUCLIDEAN_I8DOMAIN): integer-domain fast mode for quantized
    # corpora; not one of the reference's methods and never chosen implicitly
    EuclideanI8Domain = 4

    @staticmethod
    def from_request(name: Optional[str]) -> "DistanceMethod":
        """tcp.rs:388-394: unknown / missing -> Euclidean."""
        return {"cosine": DistanceMethod.Cosine, "euclidean": DistanceMethod.Euclidean,
                "simd_cosine": DistanceMethod.SimdCosine,
                "simd_euclidean": DistanceMethod.SimdEuclidean,
                "euclidean_i8_domain": DistanceMethod.EuclideanI8Domain}.get(name or "", DistanceMethod.Euclidean)


SIMD_WIDTHS = (2, 4, 8, 16, 32, 64)  # rank.rs:303,307


class Corpus:
    """Device-resident vector set in scan order (what `VFSManager::load_batch` feeds
    the reference's scan).  `quantize=True` is the `QuantizedVector` store
    (scale_factor 127.0, vector.rs:188)."""

    def __init__(self, dim: int, quantize: bool = False, scale: float = 127.0, capacity_rows: int = 0,
                 device: int = 0):
        lib = _lib.load()
        self._lib = lib
        self.dim = int(dim)
        self.quantize = bool(quantize)
        self.scale = float(scale)
        self.device = int(device)
        self._h = lib.vfsx_corpus_create(self.dim, _lib.I8 if quantize else _lib.F32, self.scale,
                                         int(capacity_rows), self.device)
        if not self._h:
            msg = lib.vfsx_last_error().decode("utf-8", "replace")
            code = _lib.ERR_NO_DEVICE if ("device" in msg.lower() or "gfx" in msg) else _lib.ERR_INVALID_VECTOR
            raise _lib.VfsxError(code, msg)

    # -- lifetime --------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vfsx_corpus_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.vfsx_corpus_rows(self._h))

    @property
    def handle(self):
        return self._h

    # -- ingest ----------------------------------------------------------------
    @staticmethod
    def _ids_arg(ids, n):
        if ids is None:
            return None, None
        a = np.ascontiguousarray(ids, dtype=np.uint64)
        if a.shape != (n,):
            raise _lib.VfsxError(_lib.ERR_INVALID_VECTOR, "ids must have one entry per row")
        return a, a.ctypes.data_as(C.POINTER(C.c_uint64))

    def _check_rows(self, rows: np.ndarray):
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            # tcp.rs:307-312
            got = rows.shape[1] if rows.ndim == 2 else rows.shape
            raise _lib.VfsxError(_lib.ERR_INVALID_VECTOR,
                                 f"Vector dimension mismatch. Expected {self.dim}, got {got}")

    def append(self, rows, ids=None) -> None:
        """Append stored rows: float32 for a dense corpus, int8 (already quantized
        `QuantizedVector::vector` bytes) for a quantized one."""
        want = np.int8 if self.quantize else np.float32
        rows = np.ascontiguousarray(rows, dtype=want)
        self._check_rows(rows)
        keep, idp = self._ids_arg(ids, rows.shape[0])
        _lib.check(self._lib.vfsx_corpus_append(self._h, rows.ctypes.data_as(C.c_void_p), idp, rows.shape[0]))

    def append_f32(self, rows, ids=None) -> None:
        """Register f32 vectors the way the store does: a quantized corpus quantizes
        them (on the device) with Vector::quantize(Some(127.0)) (vector.rs:94-108)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self._check_rows(rows)
        keep, idp = self._ids_arg(ids, rows.shape[0])
        if self.quantize:
            _lib.check(self._lib.vfsx_corpus_append_quantize(
                self._h, rows.ctypes.data_as(C.POINTER(C.c_float)), idp, rows.shape[0]))
        else:
            _lib.check(self._lib.vfsx_corpus_append(self._h, rows.ctypes.data_as(C.c_void_p), idp, rows.shape[0]))

    def load_log(self, path: str, first: int = 0, count: Optional[int] = None) -> int:
        """Append the records of a reference data file (serializer.rs:25-153) -- all of them, or the
        `count` records starting at record `first` (one shard's contiguous range).  Returns the
        number of records loaded."""
        n = C.c_uint64(0)
        _lib.check(self._lib.vfsx_corpus_load_log_range(self._h, str(path).encode(), int(first),
                                                        (1 << 64) - 1 if count is None else int(count), C.byref(n)))
        return int(n.value)

    def get_metadata(self, row: int) -> dict:
        """VectorMetadata (vector.rs:11-17) of a local row: {name, tags, created_at}."""
        import json
        need = C.c_uint64(0)
        buf = C.create_string_buffer(512)
        _lib.check(self._lib.vfsx_corpus_get_metadata_json(self._h, int(row), buf, len(buf), C.byref(need)))
        if need.value > len(buf):
            buf = C.create_string_buffer(int(need.value))
            _lib.check(self._lib.vfsx_corpus_get_metadata_json(self._h, int(row), buf, len(buf), C.byref(need)))
        return json.loads(buf.value.decode("utf-8"))

    def fill_synthetic(self, seed: int, n: int, row0: int = 0, id0: int = 1) -> None:
        _lib.check(self._lib.vfsx_corpus_fill_synthetic(self._h, seed, row0, n, id0))

    def set_insert_batch(self, max_pending_rows: int) -> None:
        """Stage appends of <= 16 rows on the host and commit them together (vfsx_corpus_set_insert_batch): they stay
        searchable at once -- every read of the store commits first."""
        _lib.check(self._lib.vfsx_corpus_set_insert_batch(self._h, int(max_pending_rows)))

    def commit(self) -> None:
        _lib.check(self._lib.vfsx_corpus_commit(self._h))

    def storage_info(self) -> dict:
        """Rows backed by memory, slabs behind the row data (0 = one block grown by copy), bytes copied by growth,
        device bytes mapped (vfsx_corpus_storage_info)."""
        cap, seg, cp, dev = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.vfsx_corpus_storage_info(self._h, C.byref(cap), C.byref(seg), C.byref(cp), C.byref(dev)))
        return {"capacity_rows": cap.value, "segments": seg.value, "copied_bytes": cp.value, "device_bytes": dev.value}

    def set_row_base(self, row_base: int) -> None:
        _lib.check(self._lib.vfsx_corpus_set_row_base(self._h, int(row_base)))

    # -- read back ---------------------------------------------------------------
    def get_rows_f32(self, rows: Sequence[int]) -> np.ndarray:
        idx = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty((idx.size, self.dim), dtype=np.float32)
        _lib.check(self._lib.vfsx_corpus_get_rows_f32(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), idx.size,
                                                      out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def get_rows_raw(self, rows: Sequence[int]) -> np.ndarray:
        idx = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty((idx.size, self.dim), dtype=np.int8 if self.quantize else np.float32)
        _lib.check(self._lib.vfsx_corpus_get_rows_raw(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), idx.size,
                                                      out.ctypes.data_as(C.c_void_p)))
        return out

    def get_ids(self, rows: Sequence[int]) -> np.ndarray:
        idx = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty(idx.size, dtype=np.uint64)
        _lib.check(self._lib.vfsx_corpus_get_ids(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), idx.size,
                                                 out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    # -- search ------------------------------------------------------------------
    def search_batch(self, queries, k: int, metric: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Exact top-k of each query row. Returns (ids[nq,k], dist[nq,k], count[nq])."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            got = q.shape[1] if q.ndim == 2 else q.shape
            # tcp.rs:371-376
            raise _lib.VfsxError(_lib.ERR_INVALID_VECTOR,
                                 f"Query vector dimension mismatch. Expected {self.dim}, got {got}")
        nq = q.shape[0]
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.full((nq, max(k, 1)), np.nan, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _lib.check(self._lib.vfsx_search(self._h, q.ctypes.data_as(C.POINTER(C.c_float)), nq, k, int(metric),
                                         ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         dist.ctypes.data_as(C.POINTER(C.c_float)),
                                         cnt.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ids, dist, cnt

    def search_device(self, d_queries_ptr: int, nq: int, k: int, metric: int, d_hits_ptr: int,
                      stream_ptr: int = 0) -> None:
        _lib.check(self._lib.vfsx_search_device(self._h, C.c_void_p(d_queries_ptr), nq, k, int(metric),
                                                C.c_void_p(d_hits_ptr), C.c_void_p(stream_ptr)))

    def search_device_noalloc(self, d_queries_ptr: int, nq: int, k: int, metric: int, d_hits_ptr: int,
                              stream_ptr: int = 0) -> None:
        """search_device for callers that cannot allocate (HIP graph capture, tight serving loops): the same
        exact scan on the library's static device workspace; no hipMalloc / hipFree on this call."""
        _lib.check(self._lib.vfsx_search_device_noalloc(self._h, C.c_void_p(d_queries_ptr), nq, k, int(metric),
                                                        C.c_void_p(d_hits_ptr), C.c_void_p(stream_ptr)))

    def enable_mfma(self, enabled: bool = True, mirror: str = "auto") -> None:
        """Let Cosine / Euclidean batches run on the matrix cores (results stay those of the exact scan).  f32 stores
        get a mirror: "int8" (1 B / element, per-row scales), "bf16" (2 B / element) or "auto" (int8 unless the rows'
        measured quantization residual or a batch's fallbacks say bf16).  int8 stores need none."""
        mode = {"auto": 1, "bf16": 2, "int8": 3}[mirror]
        _lib.check(self._lib.vfsx_corpus_enable_mfma(self._h, mode if enabled else 0))

    def mfma_mirror(self) -> Tuple[str, float]:
        """(what the matrix-core path multiplies: "off" | "bf16" | "int8" | "int8-store", the rows' largest measured
        rounding / quantization residual)."""
        kind, res = C.c_int(0), C.c_float(0.0)
        _lib.check(self._lib.vfsx_corpus_mfma_mirror(self._h, C.byref(kind), C.byref(res)))
        return ("off", "bf16", "int8", "int8-store")[kind.value], float(res.value)

    def mfma_stats(self, reset: bool = False) -> Tuple[int, int]:
        q, f = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.vfsx_corpus_mfma_stats(self._h, C.byref(q), C.byref(f), 1 if reset else 0))
        return q.value, f.value

    def set_profiling(self, enabled: bool) -> None:
        _lib.check(self._lib.vfsx_corpus_set_profiling(self._h, 1 if enabled else 0))

    def scan_time_ms(self, reset: bool = True) -> Tuple[float, int]:
        ms, n = C.c_double(0.0), C.c_uint64(0)
        _lib.check(self._lib.vfsx_corpus_scan_time_ms(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value


class Ranker:
    """rank.rs:76-105.  `Ranker(search_type, distance_method, ef_search)` then
    `search(query, num_vectors_per_iteration, result_limit, manager)`."""

    def __init__(self, search_type: SearchType, distance_method: DistanceMethod, ef_search: Optional[int] = None,
                 strict_simd_widths: bool = False):
        self.search_type = search_type
        self.distance_method = DistanceMethod(distance_method)
        self.ef_search = 6 if ef_search is None else int(ef_search)  # rank.rs:86
        # the reference panics for Simd* unless d in {2,..,64} (rank.rs:31); opt in to that
        self.strict_simd_widths = strict_simd_widths

    def search(self, query, num_vectors_per_iteration: int, result_limit: Optional[int],
               manager: Corpus) -> List[Tuple[int, float]]:
        if self.search_type is not SearchType.Exact:
            # the HNSW path (rank.rs:189-268) is out of scope for this engine
            raise NotImplementedError("approximate (HNSW) search is not part of the vfsx hot path")
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if q.size != manager.dim:  # rank.rs:273-275 panics
            raise _lib.VfsxError(_lib.ERR_INVALID_VECTOR, "Los vectores deben tener la misma dimensión")
        if (self.strict_simd_widths and self.distance_method in (DistanceMethod.SimdEuclidean,
                                                                 DistanceMethod.SimdCosine)
                and manager.dim not in SIMD_WIDTHS):
            raise _lib.VfsxError(_lib.ERR_INVALID_VECTOR, f"No se puede usar SIMD con longitud {manager.dim}")
        limit = 5 if result_limit is None else int(result_limit)  # rank.rs:129
        if int(num_vectors_per_iteration) <= 0:
            return []  # load_batch(0) yields nothing (storage_manager.rs:198): empty result
        if limit <= 0:
            # rank.rs:175: results.last().unwrap() on an empty Vec panics
            raise _lib.VfsxError(_lib.ERR_INVALID_VECTOR, "result_limit must be >= 1")
        ids, dist, cnt = manager.search_batch(q, limit, int(self.distance_method))
        n = int(cnt[0])
        return [(int(ids[0, i]), float(dist[0, i])) for i in range(n)]
