"""GPU: many-query Cosine batches on the matrix cores (vfsx_mfma.hip) must return exactly
what the single-query scan returns -- ids identical and distances bit-exact -- because the
MFMA scores only pre-filter: survivors are re-scored in the reference's sequential order and
a bound check sends every doubtful query back to the exact scan."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _compare(c, queries, k, oracle=None, rows=None, metric=1):
    """MFMA batch result vs the exact scan of the same corpus (and vs the oracle on a sample)."""
    c.enable_mfma(True)
    c.mfma_stats(reset=True)
    ids_m, dist_m, cnt_m = c.search_batch(queries, k, metric)
    nq_m, fb = c.mfma_stats()
    assert nq_m == len(queries), "the batch did not take the MFMA path"
    c.enable_mfma(False)
    ids_e, dist_e, cnt_e = c.search_batch(queries, k, metric)
    assert (cnt_m == cnt_e).all()
    assert (ids_m == ids_e).all()
    assert (bits(dist_m) == bits(dist_e)).all()
    if oracle is not None:
        for qi in range(0, len(queries), max(1, len(queries) // 5)):
            oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k)
            assert (ids_m[qi] == oi).all() and (bits(dist_m[qi]) == bits(od)).all()
    return fb


@pytest.mark.parametrize("d,n,k", [(128, 40000, 10), (100, 33333, 5), (768, 20000, 10), (64, 70000, 40)])
def test_mfma_batch_equals_exact_f32(oracle, d, n, k):
    import vfs_amd
    rng = np.random.default_rng(d + n)
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (200, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    fb = _compare(c, queries, k, oracle, rows)
    assert fb <= 100  # the check may send queries back; the result must not change either way
    c.close()


def test_mfma_batch_equals_exact_int8(oracle):
    import vfs_amd
    rng = np.random.default_rng(77)
    n, d = 50000, 96
    rows = rng.integers(-128, 128, (n, d), dtype=np.int8)
    queries = rng.uniform(-1, 1, (130, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, quantize=True)
    c.append(rows)
    _compare(c, queries, 10, oracle, rows)
    c.close()


def test_mfma_clustered_corpus_forces_fallbacks(oracle):
    """Near-duplicate rows packed next to each other: many rows of one candidate list sit above
    the safety margin, the check must flag those queries, and the exact scan must answer them."""
    import vfs_amd
    rng = np.random.default_rng(5)
    n, d = 30000, 64
    centers = rng.uniform(-1, 1, (30, d)).astype(np.float32)
    rows = np.repeat(centers, n // 30, axis=0) + rng.normal(0, 1e-3, (n, d)).astype(np.float32)
    rows = rows.astype(np.float32)
    queries = (centers[rng.integers(0, 30, 128)] + rng.normal(0, 1e-3, (128, d))).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    fb = _compare(c, queries, 10, oracle, rows)
    assert fb > 0
    c.close()


def test_mfma_zero_norm_rows_disable_the_path():
    import vfs_amd
    rng = np.random.default_rng(6)
    n, d = 20000, 64
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    rows[1234] = 0.0  # cosine distance NaN in the reference (rank.rs:299)
    queries = rng.uniform(-1, 1, (128, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    c.enable_mfma(True)
    ids_m, dist_m, _ = c.search_batch(queries, 10, 1)
    assert c.mfma_stats()[0] == 0  # mirror flagged bad -> exact scan
    c.enable_mfma(False)
    ids_e, dist_e, _ = c.search_batch(queries, 10, 1)
    assert (ids_m == ids_e).all() and (bits(dist_m) == bits(dist_e)).all()
    c.close()


def test_mfma_bad_queries_and_append_after_enable(oracle):
    import vfs_amd
    rng = np.random.default_rng(8)
    n, d = 25000, 80
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (150, d)).astype(np.float32)
    queries[3] = 0.0  # zero query: NaN distances everywhere, handled by the exact scan
    c = vfs_amd.Corpus(d)
    c.append(rows[:20000])
    c.enable_mfma(True)
    c.append(rows[20000:])  # mirror goes stale and is rebuilt
    c.mfma_stats(reset=True)
    ids_m, dist_m, cnt = c.search_batch(queries, 10, 1)
    assert c.mfma_stats()[0] == 150 and c.mfma_stats()[1] >= 1
    for qi in (0, 1, 2, 77, 149):
        oi, od, _ = oracle.topk_scan(rows, queries[qi], 1, 10)
        assert (ids_m[qi] == oi).all() and (bits(dist_m[qi]) == bits(od)).all()
    c.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_mfma_every_prefilter_kernel(oracle, monkeypatch, variant):
    """The five prefilter kernels (256x256 tile with a 4-slot / 2-slot ring, 256x128, 128x128, 256x256 on 16x16x32 MFMA;
    VFSX_MFMA_VARIANT, read when the corpus is created) all feed the same exact re-score."""
    import vfs_amd
    monkeypatch.setenv("VFSX_MFMA_VARIANT", str(variant))
    rng = np.random.default_rng(900 + variant)
    n, d = 70001, 200  # ragged last tile; K padded to 256
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (300, d)).astype(np.float32)  # ragged query tile as well
    c = vfs_amd.Corpus(d)
    c.append(rows)
    fb = _compare(c, queries, 10, oracle, rows)
    assert fb <= 30
    c.close()


# ---- Euclidean batches: same contraction, upper-bound epilogue  key = 2ab(s~ + eps) - b^2(1 - gamma) ----

@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("d,n,k", [(128, 40000, 10), (100, 33333, 5), (768, 20000, 10)])
def test_mfma_euclidean_batch_equals_exact_f32(oracle, d, n, k, metric):
    import vfs_amd
    rng = np.random.default_rng(d + n + metric)
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (200, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    fb = _compare(c, queries, k, oracle, rows, metric=metric)
    assert fb <= 100
    c.close()


def test_mfma_euclidean_wildly_varying_norms(oracle):
    """Row norms over three decades and queries of different lengths: the epilogue's bound is per row
    (b) and per query (a); the nearest rows are the short ones, not the best-aligned ones."""
    import vfs_amd
    rng = np.random.default_rng(21)
    n, d = 60000, 96
    rows = (rng.uniform(-1, 1, (n, d)) * np.exp(rng.uniform(np.log(0.01), np.log(10.0), (n, 1)))).astype(np.float32)
    queries = (rng.uniform(-1, 1, (160, d)) * np.exp(rng.uniform(np.log(0.05), np.log(5.0), (160, 1)))).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    _compare(c, queries, 10, oracle, rows, metric=0)
    c.close()


def test_mfma_euclidean_int8_and_clustered(oracle):
    import vfs_amd
    rng = np.random.default_rng(22)
    n, d = 50000, 96
    rows = rng.integers(-128, 128, (n, d), dtype=np.int8)
    queries = rng.uniform(-1, 1, (130, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, quantize=True)
    c.append(rows)
    _compare(c, queries, 10, oracle, rows, metric=0)
    c.close()
    # near-duplicates next to each other: the check must flag, the exact scan must answer
    n, d = 30000, 64
    centers = rng.uniform(-1, 1, (30, d)).astype(np.float32)
    rows = (np.repeat(centers, n // 30, axis=0) + rng.normal(0, 1e-3, (n, d))).astype(np.float32)
    queries = (centers[rng.integers(0, 30, 128)] + rng.normal(0, 1e-3, (128, d))).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    fb = _compare(c, queries, 10, oracle, rows, metric=0)
    assert fb > 0
    c.close()


def test_mfma_euclidean_needs_the_256_kernels(monkeypatch, oracle):
    """The older prefilter kernels have no Euclidean epilogue: such batches use the exact scan."""
    import vfs_amd
    monkeypatch.setenv("VFSX_MFMA_VARIANT", "2")
    rng = np.random.default_rng(23)
    rows = rng.uniform(-1, 1, (20000, 64)).astype(np.float32)
    queries = rng.uniform(-1, 1, (128, 64)).astype(np.float32)
    c = vfs_amd.Corpus(64)
    c.append(rows)
    c.enable_mfma(True)
    c.mfma_stats(reset=True)
    ids, dist, _ = c.search_batch(queries, 10, 0)
    assert c.mfma_stats()[0] == 0
    oi, od, _ = oracle.topk_scan(rows, queries[5], 0, 10)
    assert (ids[5] == oi).all() and (bits(dist[5]) == bits(od)).all()
    c.close()


def test_mfma_backs_off_when_the_check_keeps_failing(oracle):
    """Every row a near-duplicate of its neighbours: (almost) every query fails the bound check, so the
    prefilter is wasted work.  The next batches skip it (1, then 3, ... batches) and stay exact."""
    import vfs_amd
    rng = np.random.default_rng(9)
    n, d = 20000, 64
    center = rng.uniform(-1, 1, d).astype(np.float32)
    rows = (center + rng.normal(0, 1e-4, (n, d))).astype(np.float32)
    queries = (center + rng.normal(0, 1e-4, (128, d))).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    c.enable_mfma(True)
    c.mfma_stats(reset=True)
    ids1, dist1, _ = c.search_batch(queries, 10, 1)
    mq, fb = c.mfma_stats()
    assert mq == 128 and fb > 64
    ids2, dist2, _ = c.search_batch(queries, 10, 1)          # backed off: the exact scan, no prefilter
    assert c.mfma_stats()[0] == 128
    ids3, dist3, _ = c.search_batch(queries, 10, 1)          # probes again
    assert c.mfma_stats()[0] == 256
    for ids, dist in ((ids1, dist1), (ids2, dist2), (ids3, dist3)):
        oi, od, _ = oracle.topk_scan(rows, queries[3], 1, 10)
        assert (ids[3] == oi).all() and (bits(dist[3]) == bits(od)).all()
    c.close()


def test_mfma_scattered_fallbacks_are_compacted(oracle):
    """Every other query sits inside a near-duplicate cluster (fails the check), the rest are ordinary:
    the flagged ones are gathered into one exact batch and their hits scattered back in place."""
    import vfs_amd
    rng = np.random.default_rng(33)
    n, d = 40000, 64
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    center = rng.uniform(-1, 1, d).astype(np.float32)
    rows[5000:9000] = (center + rng.normal(0, 1e-4, (4000, d))).astype(np.float32)
    queries = rng.uniform(-1, 1, (140, d)).astype(np.float32)
    queries[::2] = (center + rng.normal(0, 1e-4, (70, d))).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    fb = _compare(c, queries, 10, oracle, rows, metric=1)
    assert 0 < fb < 140
    fb = _compare(c, queries, 10, oracle, rows, metric=0)
    assert 0 < fb < 140
    c.close()


@pytest.mark.parametrize("nq", [8, 33, 64, 65, 100, 128, 129])
def test_mfma_skinny_prefilter_int8_and_ragged(oracle, nq):
    """Batches of <= 128 queries run mfma_prefilter_skinny_kernel (one 64-query column up to 64, two up to 128;
    129: the 256-query kernel): int8 store, a ragged last row tile, a dimension that needs K padding, both
    metrics, against the oracle."""
    import vfs_amd
    rng = np.random.default_rng(900 + nq)
    n, d = 16384 + 256 * 37 + 113, 100
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, quantize=True)
    c.append_f32(rows)
    ref = oracle.quantize(rows, 127.0)
    c.enable_mfma(True)
    for metric in (0, 1):
        c.mfma_stats(reset=True)
        ids_m, dist_m, cnt_m = c.search_batch(queries, 10, metric)
        assert c.mfma_stats()[0] == nq
        for qi in range(0, nq, 7):
            oi, od, _ = oracle.topk_scan(ref, queries[qi], metric, 10, scale=127.0)
            assert (ids_m[qi] == oi).all() and (bits(dist_m[qi]) == bits(od)).all()
    c.close()


@pytest.mark.parametrize("nq", [3, 4, 7, 8, 9, 31, 63])
def test_mfma_small_batches(oracle, nq):
    """Batches from 4 queries on (f32 stores; 8 for int8) take the matrix-core path (it overtakes the
    multi-query scan there, tools/batch_crossover.py); 3 stay on the exact scan.  Same results either way."""
    import vfs_amd
    rng = np.random.default_rng(500 + nq)
    n, d = 45000, 200
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    for metric in (0, 1):
        c.enable_mfma(True)
        c.mfma_stats(reset=True)
        ids_m, dist_m, cnt_m = c.search_batch(queries, 10, metric)
        assert c.mfma_stats()[0] == (nq if nq >= 4 else 0)
        for qi in range(nq):
            oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, 10)
            assert (ids_m[qi] == oi).all() and (bits(dist_m[qi]) == bits(od)).all()
    c.close()


@pytest.mark.parametrize("mag,qmag,on_mfma", [(1e-6, 1e-6, True), (1e6, 1e6, True), (1e-6, 1.0, True), (1e-6, 1e-12, False),
                                               (1e-12, 1.0, False), (1e12, 1e12, False), (3e-21, 3e-21, False)])
def test_mfma_magnitude_guard(mag, qmag, on_mfma):
    """The batch path's relative error bounds need norms in [2^-30, 2^30] (no subnormal / overflowing
    intermediate in the reference's sums): rows outside disable the path for the corpus, queries outside take
    the exact scan.  Either way the result equals the exact scan's bit for bit."""
    import vfs_amd
    rng = np.random.default_rng(int(abs(np.log10(mag))) * 31 + int(abs(np.log10(qmag))))
    n, d = 40000, 64
    rows = (rng.uniform(-1, 1, (n, d)) * mag).astype(np.float32)
    q = (rng.uniform(-1, 1, (100, d)) * qmag).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    for metric in (0, 1):
        c.enable_mfma(True)
        c.mfma_stats(reset=True)
        im, dm, cm = c.search_batch(q, 10, metric)
        nq_m, fb = c.mfma_stats()
        if on_mfma:  # attempted; with |q| >> |x| the Euclidean check rightly sends every query back
            assert nq_m == 100 and (fb == 0 or mag != qmag)
        else:
            assert nq_m == 0 or fb == nq_m  # path disabled for the corpus, or every query sent back
        c.enable_mfma(False)
        ie, de, ce = c.search_batch(q, 10, metric)
        assert (cm == ce).all() and (im == ie).all() and (bits(dm) == bits(de)).all()
    c.close()
