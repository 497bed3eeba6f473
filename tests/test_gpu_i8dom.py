"""GPU parity for the labelled integer-domain fast mode (metric 4 = VFSX_EUCLIDEAN_I8DOMAIN,
an EXTENSION: SURVEY.md 8a "A6 decision" (ii), not a reference DistanceMethod).  The bar is
the same as everywhere: ids identical and distances bit-exact against the CPU restatement
(oracle/vfs_oracle.c vfso_dist_euclidean_i8dom, cross-checked with numpy on the CPU side)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

I8DOM = 4


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check(corpus, oracle, rows, q, k):
    gi, gd, gc = corpus.search_batch(q, k, I8DOM)
    q2 = np.atleast_2d(q)
    for qi in range(q2.shape[0]):
        oi, od, _ = oracle.topk_scan(rows, q2[qi], I8DOM, k)
        n = int(gc[qi])
        assert n == len(oi) == min(k, rows.shape[0])
        assert (gi[qi, :n] == oi).all(), (qi, gi[qi, :n], oi)
        assert (_bits(gd[qi, :n]) == _bits(od)).all(), (qi, gd[qi, :n], od)


@pytest.mark.parametrize("d", [1, 15, 16, 17, 128, 200, 1030])
def test_i8dom_bit_exact(oracle, d):
    import vfs_amd
    rng = np.random.default_rng(300 + d)
    n = 6000
    rows = rng.integers(-128, 128, (n, d), dtype=np.int8)
    c = vfs_amd.Corpus(d, quantize=True)
    c.append(rows)
    for t in range(2):
        # beyond [-1,1] too: the query saturates exactly like the store does (vector.rs:106)
        q = rng.uniform(-1.3, 1.3, d).astype(np.float32)
        for k in (1, 10, 65, 300):
            _check(c, oracle, rows, q, k)
    _check(c, oracle, rows, rng.uniform(-1, 1, (5, d)).astype(np.float32), 10)  # a batch
    c.close()


def test_i8dom_ties_and_ragged(oracle):
    """Integer S ties constantly on saturated data: order must be ascending scan position."""
    import vfs_amd
    rng = np.random.default_rng(7)
    d = 16
    for n in (1, 63, 64, 65, 257, 20000):
        base = rng.uniform(-10, 10, (40, d)).round(4).astype(np.float32)
        rows = oracle.quantize(base[rng.integers(0, 40, n)], 127.0)
        c = vfs_amd.Corpus(d, quantize=True)
        c.append(rows)
        q = rng.uniform(-1, 1, d).astype(np.float32)
        for k in (5, 100, 1000):
            _check(c, oracle, rows, q, k)
        c.close()


def test_i8dom_rejects_f32_corpus():
    import vfs_amd
    from vfs_amd import _lib
    c = vfs_amd.Corpus(8)
    c.append(np.zeros((4, 8), np.float32))
    with pytest.raises(_lib.VfsxError) as e:
        c.search_batch(np.zeros(8, np.float32), 2, I8DOM)
    assert e.value.code == _lib.ERR_INVALID_VECTOR
    with pytest.raises(_lib.VfsxError):
        c.search_batch(np.zeros(8, np.float32), 2, 5)  # unknown method
    c.close()


def test_i8dom_synthetic_matches_and_tracks_faithful(oracle):
    """On the bench's synthetic int8 corpus: bit-exact vs the restatement, and within the query's
    quantization error of the reference-faithful distance (what the label promises)."""
    import vfs_amd
    from vfs_amd import synth
    n, d = 200_000, 128
    c = vfs_amd.Corpus(d, quantize=True, capacity_rows=n)
    c.fill_synthetic(synth.CORPUS_SEED, n)
    rows = oracle.synth_i8(synth.CORPUS_SEED, 0, n, d)
    q = synth.rows_f32(synth.QUERY_SEED, 0, 3, d)
    _check(c, oracle, rows, q, 10)
    gi, gd, _ = c.search_batch(q, 10, I8DOM)
    fi, fd, _ = c.search_batch(q, 10, 0)
    assert np.max(np.abs(gd - fd)) <= np.sqrt(d) * 0.5 / 127.0
    c.close()
