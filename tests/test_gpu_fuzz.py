"""GPU: randomized parity sweep -- random shapes, stores, methods, limits, duplicate rows, explicit
ids, ragged appends and query batches, each case checked bit for bit against the oracle.  Seeds are
fixed, so a failure names a reproducible case."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("seed", range(12))
def test_random_cases_match_the_oracle(oracle, seed):
    import vfs_amd
    rng = np.random.default_rng(1000 + seed)
    for case in range(12):
        d = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 64, 100, 129, 256, 300]))
        n = int(rng.choice([1, 2, 63, 64, 65, 200, 1000, 4097, 9000]))
        quantize = bool(rng.integers(0, 2))
        scale = float(rng.choice([127.0, 127.0, 64.0, 100.0])) if quantize else 127.0
        spread = float(rng.choice([1.0, 1.0, 3.0, 0.01]))          # saturation / tiny values
        rows = (rng.uniform(-1, 1, (n, d)) * spread).astype(np.float32)
        if n > 10:                                                 # exact duplicates -> ties
            dup = rng.integers(0, n, n // 5)
            rows[dup] = rows[rng.integers(0, n, n // 5)]
        ids = None
        if rng.integers(0, 3) == 0:
            ids = rng.permutation(np.arange(10_000, 10_000 + n)).astype(np.uint64)
        c = vfs_amd.Corpus(d, quantize=quantize, scale=scale)
        cuts = sorted(set([0, n] + list(rng.integers(0, n + 1, 2))))   # ragged appends
        for a, b in zip(cuts[:-1], cuts[1:]):
            part_ids = None if ids is None else ids[a:b]
            (c.append_f32 if quantize else c.append)(rows[a:b], ids=part_ids)
        ref_rows = oracle.quantize(rows, scale) if quantize else rows
        nq = int(rng.choice([1, 1, 2, 3, 9]))
        queries = (rng.uniform(-1, 1, (nq, d)) * float(rng.choice([1.0, 2.0]))).astype(np.float32)
        k = int(rng.choice([1, 5, 10, 64, 65, 200]))
        metrics = [0, 1, 2, 3] + ([4] if quantize and scale == 127.0 else [])
        metric = int(rng.choice(metrics))
        if metric in (1, 3):  # zero-norm rows / queries give NaN: outside parity
            if quantize:
                zero = np.abs(ref_rows.astype(np.int32)).sum(axis=1) == 0
            else:
                zero = np.abs(ref_rows).sum(axis=1) == 0
            if zero.any() or (np.abs(queries).sum(axis=1) == 0).any():
                c.close()
                continue
        gi, gd, gc = c.search_batch(queries, k, metric)
        for qi in range(nq):
            oi, od, _ = oracle.topk_scan(ref_rows, queries[qi], metric, k, ids=ids, scale=scale)
            m = int(gc[qi])
            what = (seed, case, n, d, quantize, scale, metric, k, qi)
            assert m == len(oi) == min(k, n), what
            assert (gi[qi, :m] == oi).all(), what
            assert (_bits(gd[qi, :m]) == _bits(od)).all(), what
        c.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_mfma_batches_equal_the_exact_scan(seed):
    """Random corpora / batch sizes / k / methods through the matrix-core batch path vs the exact scan
    of the same corpus (which the sweep above ties to the oracle): ids and distance bits identical."""
    import vfs_amd
    rng = np.random.default_rng(2000 + seed)
    for case in range(4):
        d = int(rng.choice([24, 64, 100, 128, 200, 384]))
        n = int(rng.integers(16384, 70000))
        quantize = bool(rng.integers(0, 2))
        rows = (rng.uniform(-1, 1, (n, d)) * np.exp(rng.uniform(-1, 1, (n, 1)))).astype(np.float32)
        nq = int(rng.choice([64, 100, 257, 300]))
        queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
        k = int(rng.choice([1, 5, 10, 33, 64]))
        metric = int(rng.choice([0, 1, 2, 3]))
        c = vfs_amd.Corpus(d, quantize=quantize)
        (c.append_f32 if quantize else c.append)(rows)
        c.enable_mfma(True)
        c.mfma_stats(reset=True)
        im, dm, cm = c.search_batch(queries, k, metric)
        mq, fb = c.mfma_stats()
        c.enable_mfma(False)
        ie, de, ce = c.search_batch(queries, k, metric)
        what = (seed, case, n, d, quantize, nq, k, metric, mq, fb)
        assert mq in (0, nq), what  # 0: a zero-norm row disabled the path (saturated int8 rows can be all zero)
        assert (cm == ce).all() and (im == ie).all(), what
        assert (_bits(dm) == _bits(de)).all(), what
        c.close()


@pytest.mark.parametrize("seed", range(4))
def test_wide_int8_scan_matches_the_oracle(oracle, seed):
    """Single-query scans of int8 stores above 16 row-blocks per CU run the 16-wave kernel
    (scan_i8w_kernel): ragged row counts, odd dimensions, duplicates, every reference method."""
    import vfs_amd
    rng = np.random.default_rng(3000 + seed)
    d = int([7, 100, 128, 200][seed])
    n = int(rng.integers(270_000, 330_000))
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    dup = rng.integers(0, n, 2000)
    rows[dup] = rows[rng.integers(0, n, 2000)]
    c = vfs_amd.Corpus(d, quantize=True)
    c.append_f32(rows)
    ref_rows = oracle.quantize(rows, 127.0)
    nonzero = np.abs(ref_rows.astype(np.int32)).sum(axis=1) != 0
    for metric in (0, 1, 2, 3):
        if metric in (1, 3) and not nonzero.all():
            continue
        for k in (1, 10, 64):
            q = rows[int(rng.integers(0, n))] if k == 10 else rng.uniform(-1, 1, d).astype(np.float32)
            gi, gd, gc = c.search_batch(q[None, :], k, metric)
            oi, od, _ = oracle.topk_scan(ref_rows, q, metric, k, scale=127.0)
            what = (seed, n, d, metric, k)
            assert int(gc[0]) == k, what
            assert (gi[0] == oi).all(), what
            assert (_bits(gd[0]) == _bits(od)).all(), what
    c.close()
