"""Second, independent CPU restatement of the VFS Ranker linear-search path (numpy).

TEST INFRASTRUCTURE ONLY -- see the header of ``oracle/vfs_oracle.c``.  The product
(``vfs_amd/``) never imports this module.  PARITY UNPINNED: the reference holds no
golden vectors for this path; this file exists so that the C oracle is cross-checked
bit-for-bit by code written separately, in a different language, from the same
reference lines (all citations relative to ``/root/reference/``).

Every arithmetic step is an explicit ``np.float32`` operation (one IEEE-754
binary32 rounding each).  Rows are processed as whole columns, but each row's
accumulator still receives its terms strictly in index order j = 0..d-1, which is
the reference's summation order (``rank.rs:282-286``).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

EUCLIDEAN, COSINE, SIMD_EUCLIDEAN, SIMD_COSINE = 0, 1, 2, 3  # rank.rs:67-72
SIMD_WIDTHS = (2, 4, 8, 16, 32, 64)  # rank.rs:303,307


def _seq_sum_cols(terms_col_fn, n_rows: int, d: int) -> np.ndarray:
    """acc[i] = (((0 + t(i,0)) + t(i,1)) + ...) in float32, per row i."""
    acc = np.zeros(n_rows, dtype=F32)
    for j in range(d):
        acc = (acc + terms_col_fn(j)).astype(F32, copy=False)
    return acc


def distances(rows: np.ndarray, query: np.ndarray, metric: int) -> np.ndarray:
    """All N distances of ``query`` to ``rows`` (N x d float32): rank.rs:271-300."""
    rows = np.ascontiguousarray(rows, dtype=F32)
    q = np.asarray(query, dtype=F32)
    n, d = rows.shape
    assert q.shape == (d,)  # rank.rs:273-275 panics on mismatch
    cols = np.ascontiguousarray(rows.T)  # column j contiguous
    with np.errstate(all="ignore"):
        if metric in (EUCLIDEAN, SIMD_EUCLIDEAN):  # rank.rs:281-287 / 14-33
            def term(j):
                diff = (q[j] - cols[j]).astype(F32, copy=False)
                return (diff * diff).astype(F32, copy=False)
            return np.sqrt(_seq_sum_cols(term, n, d)).astype(F32, copy=False)
        # rank.rs:290-300 / 36-56
        dot = _seq_sum_cols(lambda j: (q[j] * cols[j]).astype(F32, copy=False), n, d)
        s1 = F32(0.0)
        for j in range(d):
            s1 = F32(s1 + F32(q[j] * q[j]))
        n1 = np.sqrt(s1).astype(F32)
        n2 = np.sqrt(_seq_sum_cols(lambda j: (cols[j] * cols[j]).astype(F32, copy=False), n, d))
        den = (n1 * n2).astype(F32, copy=False)
        quo = (dot / den).astype(F32, copy=False)
        return (F32(1.0) - quo).astype(F32, copy=False)


def quantize(v: np.ndarray, scale=127.0) -> np.ndarray:
    """vector.rs:94-108. ``scale=None`` models ``None`` (clamp to [-1,1], scale 127)."""
    v = np.asarray(v, dtype=F32)
    with np.errstate(all="ignore"):
        if scale is None:
            v = np.where(np.isnan(v), v, np.clip(v, F32(-1.0), F32(1.0))).astype(F32)
            s = F32(127.0)
        else:
            s = F32(scale)
        prod = (v * s).astype(F32, copy=False)
        # f32::round = half away from zero (np.round is half-to-even: do not use it).
        # trunc + exact fractional part: prod - trunc(prod) is exact in binary32.
        t = np.trunc(prod).astype(F32, copy=False)
        frac = (prod - t).astype(F32, copy=False)
        r = np.where(np.abs(frac) >= F32(0.5), t + np.copysign(F32(1.0), prod), t)
        r = np.where(np.isnan(prod), F32(0.0), r)  # `as i8`: NaN -> 0
        r = np.clip(r, -128.0, 127.0)  # `as i8` saturates
    return r.astype(np.int8)


def dequantize(q: np.ndarray, scale=127.0) -> np.ndarray:
    """vector.rs:135-139: (val as f32) / scale_factor."""
    return (np.asarray(q, dtype=np.int8).astype(F32) / F32(scale)).astype(F32)


def distances_i8dom(rows_i8: np.ndarray, query: np.ndarray, scale=127.0) -> np.ndarray:
    """EXTENSION (product metric 4, not a reference DistanceMethod):
    query quantized by the reference's rule (vector.rs:293-302 -> 94-108), S exact in
    integers, dist = sqrt(S as f32) / scale."""
    x = np.asarray(rows_i8, dtype=np.int8).astype(np.int64)
    qi = quantize(np.asarray(query, dtype=F32), scale).astype(np.int64)
    S = ((qi[None, :] - x) ** 2).sum(axis=1)
    assert S.max(initial=0) < 2 ** 32
    return (np.sqrt(S.astype(np.uint32).astype(F32)).astype(F32) / F32(scale)).astype(F32)


def exact_search(rows, query, metric, batch, limit=5, ids=None):
    """rank.rs:119-186 mirrored literally (stable sort, truncate, retain), per batch.

    ``rows`` is what load_batch yields: float32, already dequantized for a quantized
    store (storage_manager.rs:202-210).  Returns (ids, dists).
    """
    rows = np.ascontiguousarray(rows, dtype=F32)
    n = rows.shape[0]
    ids = np.arange(1, n + 1, dtype=np.uint64) if ids is None else np.asarray(ids, dtype=np.uint64)
    dist_all = distances(rows, query, metric) if n else np.zeros(0, dtype=F32)
    res = []  # list of (id, dist)
    off = 0
    while batch > 0:  # load_batch(0) returns nothing
        take = min(batch, n - off)
        if take <= 0:
            break
        res.extend((int(ids[r]), dist_all[r]) for r in range(off, off + take))
        off += take
        # stable sort with comparator partial_cmp().unwrap_or(Equal): insertion sort
        # moving an element left only while it is strictly less
        for i in range(1, len(res)):
            key = res[i]
            j = i
            while j > 0 and key[1] < res[j - 1][1]:
                res[j] = res[j - 1]
                j -= 1
            res[j] = key
        if len(res) > limit:
            del res[limit:]
            lowest = res[-1][1]  # limit == 0 panics in the reference (last() on empty)
            res = [h for h in res if h[1] <= lowest]
    return (np.array([h[0] for h in res], dtype=np.uint64),
            np.array([h[1] for h in res], dtype=F32))


def topk(rows, query, metric, limit, ids=None):
    """Net semantics: `limit` smallest by (distance, scan position)."""
    rows = np.ascontiguousarray(rows, dtype=F32)
    n = rows.shape[0]
    ids = np.arange(1, n + 1, dtype=np.uint64) if ids is None else np.asarray(ids, dtype=np.uint64)
    dist = distances(rows, query, metric)
    order = np.argsort(dist, kind="stable")[:limit]  # NaN-free inputs only
    return ids[order], dist[order], order.astype(np.uint64)


_M64 = (1 << 64) - 1


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return z ^ (z >> np.uint64(31))


def synth_f32(seed: int, row0: int, n: int, d: int) -> np.ndarray:
    """Counter-based synthetic rows in [-1,1) (not part of the reference)."""
    with np.errstate(over="ignore"):
        ctr = (np.uint64(seed) + (np.uint64(row0) + np.arange(n, dtype=np.uint64))[:, None]
               * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :])
        u = _splitmix64(ctr)
    m = (u >> np.uint64(40)).astype(np.uint32)
    return (m.astype(F32) * F32(1.0 / 8388608.0) - F32(1.0)).astype(F32)


def synth_i8(seed: int, row0: int, n: int, d: int, scale=127.0) -> np.ndarray:
    return quantize(synth_f32(seed, row0, n, d), scale)
