"""ctypes loader for oracle/libvfs_oracle.so (the C restatement) -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvfs_oracle.so")

EUCLIDEAN, COSINE, SIMD_EUCLIDEAN, SIMD_COSINE = 0, 1, 2, 3
EUCLIDEAN_I8DOMAIN = 4  # product extension (labelled fast mode), not a reference DistanceMethod


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    f32p, i8p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_uint64)
    lib.vfso_dist_euclidean.restype = C.c_float
    lib.vfso_dist_euclidean.argtypes = [f32p, f32p, C.c_uint32]
    lib.vfso_dist_cosine.restype = C.c_float
    lib.vfso_dist_cosine.argtypes = [f32p, f32p, C.c_uint32]
    lib.vfso_quantize.argtypes = [f32p, C.c_uint64, C.c_int, C.c_float, i8p]
    lib.vfso_dequantize.argtypes = [i8p, C.c_uint64, C.c_float, f32p]
    lib.vfso_exact_search.restype = C.c_uint64
    lib.vfso_exact_search.argtypes = [f32p, u64p, C.c_uint64, C.c_uint32, f32p, C.c_int,
                                      C.c_uint64, C.c_uint64, u64p, f32p]
    lib.vfso_topk_scan.restype = C.c_uint64
    lib.vfso_topk_scan.argtypes = [f32p, i8p, C.c_float, u64p, C.c_uint64, C.c_uint32, f32p,
                                   C.c_int, C.c_uint64, u64p, f32p, u64p]
    lib.vfso_all_distances.argtypes = [f32p, i8p, C.c_float, C.c_uint64, C.c_uint32, f32p,
                                       C.c_int, f32p]
    lib.vfso_fill_synthetic_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, f32p]
    lib.vfso_fill_synthetic_i8.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                           C.c_float, i8p]
    lib.vfso_simd_width_supported.argtypes = [C.c_uint32]
    lib.vfsc_chunked_scan.restype = C.c_uint64
    lib.vfsc_chunked_scan.argtypes = [f32p, i8p, C.c_float, C.c_uint64, C.c_uint32, f32p, C.c_int,
                                      C.c_uint64, C.c_int, u64p, f32p]
    lib.vfsc_have_avx2.restype = C.c_int
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def dist(metric: int, q: np.ndarray, x: np.ndarray) -> np.float32:
    q = np.ascontiguousarray(q, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    fn = lib().vfso_dist_euclidean if metric in (0, 2) else lib().vfso_dist_cosine
    return np.float32(fn(_p(q, C.c_float), _p(x, C.c_float), q.size))


def quantize(v: np.ndarray, scale=127.0) -> np.ndarray:
    v = np.ascontiguousarray(v, np.float32)
    out = np.empty(v.shape, np.int8)
    lib().vfso_quantize(_p(v, C.c_float), v.size, 0 if scale is None else 1,
                        127.0 if scale is None else float(scale), _p(out, C.c_int8))
    return out


def dequantize(q: np.ndarray, scale=127.0) -> np.ndarray:
    q = np.ascontiguousarray(q, np.int8)
    out = np.empty(q.shape, np.float32)
    lib().vfso_dequantize(_p(q, C.c_int8), q.size, float(scale), _p(out, C.c_float))
    return out


def exact_search(rows, query, metric, batch, limit=5, ids=None):
    rows = np.ascontiguousarray(rows, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    n, d = rows.shape
    ids = None if ids is None else np.ascontiguousarray(ids, np.uint64)
    oi = np.empty(max(limit, 1), np.uint64)
    od = np.empty(max(limit, 1), np.float32)
    m = lib().vfso_exact_search(_p(rows, C.c_float), _p(ids, C.c_uint64), n, d,
                                _p(query, C.c_float), metric, batch, limit,
                                _p(oi, C.c_uint64), _p(od, C.c_float))
    return oi[:m].copy(), od[:m].copy()


def topk_scan(rows, query, metric, limit, ids=None, scale=127.0):
    """rows: float32 (N,d) or int8 (N,d) [dequantized on the fly with `scale`]."""
    query = np.ascontiguousarray(query, np.float32)
    n, d = rows.shape
    ids = None if ids is None else np.ascontiguousarray(ids, np.uint64)
    oi = np.empty(max(limit, 1), np.uint64)
    od = np.empty(max(limit, 1), np.float32)
    op = np.empty(max(limit, 1), np.uint64)
    if rows.dtype == np.int8:
        rows = np.ascontiguousarray(rows)
        f, i8 = None, _p(rows, C.c_int8)
    else:
        rows = np.ascontiguousarray(rows, np.float32)
        f, i8 = _p(rows, C.c_float), None
    m = lib().vfso_topk_scan(f, i8, float(scale), _p(ids, C.c_uint64), n, d,
                             _p(query, C.c_float), metric, limit, _p(oi, C.c_uint64),
                             _p(od, C.c_float), _p(op, C.c_uint64))
    return oi[:m].copy(), od[:m].copy(), op[:m].copy()


def chunked_scan(rows, query, metric, limit, n_threads=1, scale=127.0):
    """cpu_context.c: 64-lane-chunked (vectorised, optionally threaded) scan.  NOT the
    reference's summation order -- timing context for bench.py only.  Returns (pos, dist)."""
    query = np.ascontiguousarray(query, np.float32)
    n, d = rows.shape
    op = np.empty(max(limit, 1), np.uint64)
    od = np.empty(max(limit, 1), np.float32)
    if rows.dtype == np.int8:
        rows = np.ascontiguousarray(rows)
        f, i8 = None, _p(rows, C.c_int8)
    else:
        rows = np.ascontiguousarray(rows, np.float32)
        f, i8 = _p(rows, C.c_float), None
    m = lib().vfsc_chunked_scan(f, i8, float(scale), n, d, _p(query, C.c_float), metric, limit,
                                int(n_threads), _p(op, C.c_uint64), _p(od, C.c_float))
    return op[:m].copy(), od[:m].copy()


def all_distances(rows, query, metric, scale=127.0):
    query = np.ascontiguousarray(query, np.float32)
    n, d = rows.shape
    out = np.empty(n, np.float32)
    if rows.dtype == np.int8:
        rows = np.ascontiguousarray(rows)
        f, i8 = None, _p(rows, C.c_int8)
    else:
        rows = np.ascontiguousarray(rows, np.float32)
        f, i8 = _p(rows, C.c_float), None
    lib().vfso_all_distances(f, i8, float(scale), n, d, _p(query, C.c_float), metric,
                             _p(out, C.c_float))
    return out


def synth_f32(seed, row0, n, d):
    out = np.empty((n, d), np.float32)
    lib().vfso_fill_synthetic_f32(seed, row0, n, d, _p(out, C.c_float))
    return out


def synth_i8(seed, row0, n, d, scale=127.0):
    out = np.empty((n, d), np.int8)
    lib().vfso_fill_synthetic_i8(seed, row0, n, d, float(scale), _p(out, C.c_int8))
    return out
