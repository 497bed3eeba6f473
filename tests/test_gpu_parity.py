"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): top-k ids identical; int8 distances bit-exact; f32
distances within 1e-5 relative.  The rb64 kernels accumulate each row strictly in
the reference's order, so these tests demand BIT-EXACT f32 distances as well.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EUCLIDEAN, COSINE, SIMD_EUCLIDEAN, SIMD_COSINE = 0, 1, 2, 3


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check(corpus, oracle, rows, q, metric, k, ids=None, scale=127.0):
    gi, gd, gc = corpus.search_batch(q, k, metric)
    oi, od, _ = oracle.topk_scan(rows, q, metric, k, ids=ids, scale=scale)
    n = int(gc[0])
    assert n == len(oi) == min(k, rows.shape[0])
    assert (gi[0, :n] == oi).all(), (gi[0, :n], oi)
    assert (_bits(gd[0, :n]) == _bits(od)).all(), (gd[0, :n], od)


@pytest.mark.parametrize("d", [1, 3, 4, 16, 64, 128, 131, 768])
@pytest.mark.parametrize("metric", [EUCLIDEAN, COSINE])
def test_f32_bit_exact(oracle, d, metric):
    import vfs_amd
    rng = np.random.default_rng(d * 7 + metric)
    n = 5000 if d <= 131 else 3000
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    assert len(c) == n
    for t in range(3):
        q = rng.uniform(-1, 1, d).astype(np.float32)
        for k in (1, 5, 10, 64, 100):
            _check(c, oracle, rows, q, metric, k)
    c.close()


@pytest.mark.parametrize("d", [1, 15, 16, 17, 128, 200])
@pytest.mark.parametrize("metric", [EUCLIDEAN, COSINE])
def test_i8_bit_exact(oracle, d, metric):
    import vfs_amd
    rng = np.random.default_rng(100 + d * 7 + metric)
    n = 6000
    rows = rng.integers(-128, 128, (n, d), dtype=np.int8)
    if metric == COSINE:
        rows[np.abs(rows).sum(axis=1) == 0, 0] = 1  # zero-norm rows give NaN: outside parity
    c = vfs_amd.Corpus(d, quantize=True)
    c.append(rows)
    for t in range(3):
        q = rng.uniform(-1, 1, d).astype(np.float32)
        for k in (1, 10, 65, 300):
            _check(c, oracle, rows, q, metric, k)
    c.close()


def test_ties_scan_order(oracle):
    """Saturated corpus (the reference's own benchmark data is almost all +-127/-128):
    equal distances must come back in ascending scan order (rank.rs:170, stable sort)."""
    import vfs_amd
    rng = np.random.default_rng(5)
    d, n = 16, 20000
    base = rng.uniform(-10, 10, (40, d)).round(4).astype(np.float32)
    rows_f = base[rng.integers(0, 40, n)]  # many exact duplicates
    rows = oracle.quantize(rows_f, 127.0)
    c = vfs_amd.Corpus(d, quantize=True)
    c.append(rows)
    q = rng.uniform(-1, 1, d).astype(np.float32)
    for metric in (EUCLIDEAN, COSINE, SIMD_EUCLIDEAN, SIMD_COSINE):
        for k in (10, 100, 1000):
            _check(c, oracle, rows, q, metric, k)
    c.close()


def test_small_and_ragged(oracle):
    import vfs_amd
    rng = np.random.default_rng(9)
    d = 8
    for n in (1, 2, 63, 64, 65, 127, 129, 257):
        rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
        c = vfs_amd.Corpus(d)
        # ragged appends across row-block boundaries
        cuts = sorted(set([0, n] + list(rng.integers(0, n + 1, 3))))
        for a, b in zip(cuts[:-1], cuts[1:]):
            c.append(rows[a:b])
        assert len(c) == n
        q = rng.uniform(-1, 1, d).astype(np.float32)
        for k in (1, 5, 300):
            _check(c, oracle, rows, q, EUCLIDEAN, k)
            _check(c, oracle, rows, q, COSINE, k)
        assert (c.get_rows_f32(np.arange(n)) == rows).all()
        c.close()


def test_explicit_ids_and_quantize_ingest(oracle):
    import vfs_amd
    rng = np.random.default_rng(11)
    d, n = 32, 3000
    rows = rng.uniform(-1.3, 1.3, (n, d)).astype(np.float32)
    rows[0, :4] = [0.5 / 127 * 127, 63.5 / 127, -63.5 / 127, np.nan]
    ids = rng.permutation(np.arange(1000, 1000 + n)).astype(np.uint64)
    c = vfs_amd.Corpus(d, quantize=True)
    c.append_f32(rows, ids=ids)
    qrows = oracle.quantize(rows, 127.0)
    assert (c.get_rows_raw(np.arange(n)) == qrows).all()
    assert (_bits(c.get_rows_f32(np.arange(n))) == _bits(oracle.dequantize(qrows, 127.0))).all()
    q = rng.uniform(-1, 1, d).astype(np.float32)
    _check(c, oracle, qrows, q, EUCLIDEAN, 10, ids=ids)
    c.close()


def test_synthetic_fill_matches_oracle_generator(oracle):
    import vfs_amd
    d, n, seed = 24, 1000, 0x5EED0001
    c = vfs_amd.Corpus(d)
    c.fill_synthetic(seed, n)
    ref = oracle.synth_f32(seed, 0, n, d)
    assert (_bits(c.get_rows_f32(np.arange(n))) == _bits(ref)).all()
    c.close()
    c = vfs_amd.Corpus(d, quantize=True)
    c.fill_synthetic(seed, n)
    assert (c.get_rows_raw(np.arange(n)) == oracle.synth_i8(seed, 0, n, d)).all()
    c.close()


@pytest.mark.parametrize("magnitude", [1e-19, 3e-21, 1e-23, 1e18])
@pytest.mark.parametrize("metric", [EUCLIDEAN, COSINE])
def test_subnormal_and_huge_magnitudes(oracle, magnitude, metric):
    """f32 denormals must stay enabled and sqrt / divide correctly rounded on the device: with
    rows of magnitude 1e-19 .. 1e-23 the squares, their sums and the norms are subnormal (or flush to zero
    only where IEEE says so); 1e18 takes the sums close to overflow.  Compared bit for bit."""
    import vfs_amd
    rng = np.random.default_rng(int(abs(np.log10(magnitude))) + metric)
    n, d = 3000, 24
    rows = (rng.uniform(-1, 1, (n, d)) * magnitude).astype(np.float32)
    q = (rng.uniform(-1, 1, d) * magnitude).astype(np.float32)
    if metric == COSINE:  # zero norms give NaN: outside parity (the oracle's norms decide)
        sq = (rows.astype(np.float32) ** 2).astype(np.float32)
        acc = np.zeros(n, np.float32)
        for j in range(d):
            acc = (acc + sq[:, j]).astype(np.float32)
        keep = np.sqrt(acc) > 0
        qacc = np.float32(0)
        for j in range(d):
            qacc = np.float32(qacc + np.float32(q[j] * q[j]))
        if not (np.sqrt(qacc) > 0):
            pytest.skip("query norm underflows to zero: NaN distances, outside parity")
        rows = rows[keep]
        if len(rows) < 100:
            pytest.skip("row norms underflow to zero: NaN distances, outside parity")
    c = vfs_amd.Corpus(d)
    c.append(rows)
    for k in (1, 10, 64):
        _check(c, oracle, rows, q, metric, k)
    gi, gd, gc = c.search_batch(np.stack([q, q * np.float32(0.5), -q]), 10, metric)
    for qi, qq in enumerate([q, q * np.float32(0.5), -q]):
        oi, od, _ = oracle.topk_scan(rows, qq.astype(np.float32), metric, 10)
        assert (gi[qi] == oi).all() and (_bits(gd[qi]) == _bits(od)).all()
    c.close()


@pytest.mark.parametrize("quantize", [False, True])
def test_result_limit_beyond_one_pass(oracle, quantize):
    """The reference's result_limit is unbounded (rank.rs:129,172-173).  Up to 1024 hits come from one pass; a longer
    result takes ceil(k / 1024) passes, pass p ranking what sorts after pass p-1's last (distance, row).  Saturated
    int8 rows make thousands of exact ties straddle the pass boundaries."""
    import vfs_amd
    rng = np.random.default_rng(61)
    n, d = 9000, 24
    if quantize:
        base = rng.uniform(-10, 10, (40, d)).round(4).astype(np.float32)
        rows = oracle.quantize(base[rng.integers(0, 40, n)], 127.0)  # 40 distinct rows: ties everywhere
    else:
        rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (3, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, quantize=quantize)
    c.append(rows)
    kw = {"scale": 127.0} if quantize else {}
    for k in (1025, 2048, 3000, 9000, 12000):  # 12000 > rows: every row comes back, ranked
        for metric in (0, 1):
            ids, dist, cnt = c.search_batch(queries, k, metric)
            for qi in range(len(queries)):
                oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k, **kw)
                assert cnt[qi] == len(oi) == min(k, n)
                assert (ids[qi, :cnt[qi]] == oi).all()
                assert (_bits(dist[qi, :cnt[qi]]) == _bits(od)).all()
    c.close()


@pytest.mark.parametrize("quantize", [False, True])
def test_long_results_radix_select(oracle, quantize):
    """result_limit > 128 leaves the per-wave lists: the scan writes every row's composite key and an 8-pass radix
    select + one-workgroup sort emits the hits in chunks of 4096 from a moving cursor.  Batches of queries (they share
    corpus passes), chunk boundaries (4096, 4097, 8192), k beyond the rows, explicit ids, NaN / duplicate / tied
    distances -- all equal to the oracle's stable ranking."""
    import vfs_amd
    rng = np.random.default_rng(77 + quantize)
    n, d = 20011, 20
    if quantize:
        base = rng.uniform(-6, 6, (300, d)).round(3).astype(np.float32)
        rows = oracle.quantize(base[rng.integers(0, 300, n)], 127.0)  # 300 distinct rows: ties everywhere
        kw = {"scale": 127.0}
    else:
        rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
        rows[100:140] = rows[50]  # duplicates
        kw = {}
    ids = rng.permutation(np.arange(5, 5 + n)).astype(np.uint64)
    queries = rng.uniform(-1, 1, (11, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, quantize=quantize)
    c.append(rows, ids=ids)
    for k in (129, 1000, 4096, 4097, 8192, n, n + 500):
        for metric in (0, 1):
            gi, gd, gc = c.search_batch(queries, k, metric)
            for qi in (0, 5, 10):
                oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k, ids=ids, **kw)
                assert gc[qi] == len(oi) == min(k, n)
                assert (gi[qi, :gc[qi]] == oi).all(), (k, metric, qi)
                assert (_bits(gd[qi, :gc[qi]]) == _bits(od)).all(), (k, metric, qi)
    c.close()


@pytest.mark.parametrize("n,balanced", [(16384 * 64, "1"), (600_037, "0")])
def test_select_bound_with_more_lists_than_the_select_has_waves(oracle, monkeypatch, n, balanced):
    """VFSX_BLOCKS_PER_CU=8 gives the scan up to 2048 workgroups = 2048 lists, twice what the select's 16 waves
    hold as heads: the pruning bound may only count the waves that exist (with full = n_lists / 64
    = 32 and k in 17..32 only 16 keys were vouched for and true hits were dropped)."""
    import vfs_amd
    monkeypatch.setenv("VFSX_BLOCKS_PER_CU", "8")
    monkeypatch.setenv("VFSX_BALANCED_GRID", balanced)
    d = 8
    rows = oracle.synth_f32(77, 0, n, d)
    queries = oracle.synth_f32(78, 0, 3, d)
    c = vfs_amd.Corpus(d, capacity_rows=n)
    c.append(rows)
    for metric in (0, 1):
        for k in (17, 20, 32, 64, 100, 128):
            ids, dist, cnt = c.search_batch(queries, k, metric)  # 3 queries: one pass of 2 + one single
            for qi in range(3):
                oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k)
                assert cnt[qi] == k and (ids[qi] == oi).all(), (metric, k, qi)
                assert (dist[qi].view(np.uint32) == od.view(np.uint32)).all(), (metric, k, qi)
    c.close()


@pytest.mark.parametrize("d,quantize", [(4099, False), (65536, False), (200_001, False), (70_001, True)])
def test_very_long_rows(oracle, d, quantize):
    """Dimensions far beyond the BASELINE shapes (the reference accepts any Vec length): rows of up to 800 KB, ragged
    against the 16-byte chunk, single queries and a 3-query call (queries longer than the staging LDS take one pass
    each), every reference method, against the oracle bit for bit."""
    import vfs_amd
    rng = np.random.default_rng(d)
    n = 700 if d > 100_000 else 1500
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    rows[n - 3] = rows[5]  # an exact duplicate: tie by scan order
    queries = rng.uniform(-1, 1, (3, d)).astype(np.float32)
    queries[1] = rows[5]
    c = vfs_amd.Corpus(d, quantize=quantize)
    (c.append_f32 if quantize else c.append)(rows)
    ref = oracle.quantize(rows, 127.0) if quantize else rows
    kw = {"scale": 127.0} if quantize else {}
    for metric in (0, 1, 2, 3):
        ids, dist, cnt = c.search_batch(queries, 7, metric)
        one = c.search_batch(queries[2:3], 7, metric)
        for qi in range(3):
            oi, od, _ = oracle.topk_scan(ref, queries[qi], metric, 7, **kw)
            assert cnt[qi] == 7 and (ids[qi] == oi).all(), (d, metric, qi, ids[qi], oi)
            assert (_bits(dist[qi]) == _bits(od)).all(), (d, metric, qi)
        assert (one[0][0] == ids[2]).all() and (_bits(one[1][0]) == _bits(dist[2])).all()
    c.close()
