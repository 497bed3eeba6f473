"""CPU: the oracle against the committed golden fixtures and hand-derived KATs, and
the C restatement against the independent numpy restatement.
PARITY UNPINNED: none of these vectors come from the reference itself."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import vfs_oracle_np as on
from tests import golden_util

CASES, QUANT = golden_util.load()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_exact_search(case):
    rows_f = co.dequantize(case["rows"], 127.0) if case["quantized"] else case["rows"]
    ids, dist = co.exact_search(rows_f, case["query"], case["metric"], case["batch"], case["limit"],
                                ids=case["ids"])
    assert (ids == case["expect_ids"]).all()
    assert (bits(dist) == bits(case["expect_dist"])).all()
    # the O(N log k) form used for big corpora / the CPU baseline gives the same answer
    ids2, dist2, _ = co.topk_scan(case["rows"], case["query"], case["metric"], case["limit"], ids=case["ids"])
    assert (ids2 == ids).all() and (bits(dist2) == bits(dist)).all()


@pytest.mark.parametrize("case", CASES[:12], ids=[c["name"] for c in CASES[:12]])
def test_golden_numpy_restatement(case):
    rows_f = on.dequantize(case["rows"], 127.0) if case["quantized"] else case["rows"]
    ids, dist = on.exact_search(rows_f, case["query"], case["metric"], case["batch"], case["limit"],
                                ids=case["ids"])
    assert (ids == case["expect_ids"]).all()
    assert (bits(dist) == bits(case["expect_dist"])).all()


def test_kat_quantize():
    v = golden_util.unhex(QUANT["in"])
    assert co.quantize(v, 127.0).astype(int).tolist() == QUANT["some127"]
    assert co.quantize(v, None).astype(int).tolist() == QUANT["none"]
    assert on.quantize(v, 127.0).astype(int).tolist() == QUANT["some127"]
    # vector.rs:106: (val*scale).round() as i8 -- half away from zero, NaN -> 0, saturating, no clamp
    assert QUANT["some127"][:5] == [64, -64, 0, -128, 127]
    # with None the value is clamped to [-1,1] first (vector.rs:105): 2.0 -> 127, -2.0 -> -127
    assert QUANT["none"][3] == -127 and QUANT["none"][4] == 127
    allq = np.arange(-128, 128).astype(np.int8)
    assert (bits(co.dequantize(allq, 127.0)) == bits(golden_util.unhex(QUANT["dequant_all_127"]))).all()


def test_kat_tests_sh_vector():
    """tests.sh:81-104: [1,2,3,4] into a quantizing store, searched with [1,2,3,4]."""
    q = np.array([1, 2, 3, 4], np.float32)
    stored = co.quantize(q[None, :], 127.0)
    assert stored.tolist() == [[127, 127, 127, 127]]  # saturation, no clamp
    back = co.dequantize(stored, 127.0)
    assert back.tolist() == [[1.0, 1.0, 1.0, 1.0]]
    ids, dist = co.exact_search(back, q, 0, 2, 6)
    assert ids.tolist() == [1] and bits(dist)[0] == bits(np.sqrt(np.float32(14.0)))[()]


@pytest.mark.parametrize("d", [1, 2, 3, 4, 16, 64, 128, 131])
@pytest.mark.parametrize("metric", [0, 1, 2, 3])
def test_c_vs_numpy_distances(d, metric):
    rng = np.random.default_rng(d * 11 + metric)
    x = rng.uniform(-1, 1, (300, d)).astype(np.float32)
    q = rng.uniform(-1, 1, d).astype(np.float32)
    assert (bits(co.all_distances(x, q, metric)) == bits(on.distances(x, q, metric))).all()
    xq = rng.integers(-128, 128, (300, d), dtype=np.int8)
    a = co.all_distances(xq, q, metric)
    b = on.distances(on.dequantize(xq, 127.0), q, metric)
    assert (bits(a) == bits(b)).all()


def test_batch_size_independence():
    """A4: the result of exact_search does not depend on num_vectors_per_iteration."""
    rng = np.random.default_rng(3)
    x = co.dequantize(co.quantize(rng.uniform(-10, 10, (500, 16)).round(4).astype(np.float32)), 127.0)
    q = rng.uniform(-1, 1, 16).astype(np.float32)
    ref = co.exact_search(x, q, 0, 1, 100)
    for batch in (2, 5, 7, 100, 499, 500, 10_000):
        got = co.exact_search(x, q, 0, batch, 100)
        assert (got[0] == ref[0]).all() and (bits(got[1]) == bits(ref[1])).all()
    assert len(co.exact_search(x, q, 0, 0, 100)[0]) == 0  # load_batch(0) -> nothing


def test_zero_norm_is_nan_and_cosine_self():
    z = np.zeros((1, 8), np.float32)
    q = np.ones(8, np.float32)
    assert np.isnan(co.all_distances(z, q, 1)[0])  # rank.rs:299: 0/0
    v = np.random.default_rng(1).uniform(-1, 1, (1, 37)).astype(np.float32)
    assert abs(float(co.all_distances(v, v[0], 1)[0])) < 3e-7


def test_synth_generators_agree():
    from vfs_amd import synth
    a = co.synth_f32(synth.CORPUS_SEED, 12345, 40, 33)
    assert (bits(a) == bits(on.synth_f32(synth.CORPUS_SEED, 12345, 40, 33))).all()
    assert (bits(a) == bits(synth.rows_f32(synth.CORPUS_SEED, 12345, 40, 33))).all()
    assert a.min() >= -1.0 and a.max() < 1.0


@pytest.mark.parametrize("dtype", ["f32", "i8"])
@pytest.mark.parametrize("metric", [0, 1])
def test_cpu_context_scan_tracks_oracle(dtype, metric):
    """oracle/cpu_context.c is only a timing baseline (chunked sums), but it must rank the same
    corpus sensibly: distances within 1e-5 of the bit-faithful port, same rows where the
    port's neighbours are separated by more than that, threaded == single-threaded."""
    d = 200  # exercises the 64-lane body and the scalar tail
    x = co.synth_f32(11, 0, 5000, d)
    if dtype == "i8":
        x = co.quantize(x)
    q = co.synth_f32(12, 0, 1, d)[0]
    _, dist, pos = co.topk_scan(x, q, metric, 10)
    p1, d1 = co.chunked_scan(x, q, metric, 10, 1)
    p3, d3 = co.chunked_scan(x, q, metric, 10, 3)
    assert np.array_equal(p1, p3) and np.array_equal(d1, d3)
    assert np.allclose(d1, dist, rtol=1e-5, atol=1e-6)
    if np.min(np.diff(dist)) > 1e-5:
        assert np.array_equal(p1, pos)


@pytest.mark.parametrize("d", [1, 16, 40, 128, 515])
def test_i8dom_extension_c_vs_numpy(d):
    """Metric 4 (integer-domain fast mode, an extension -- not the reference's arithmetic): the two
    restatements agree bit for bit, ties resolve by scan position, and it differs from the
    reference-faithful distance only by the query's quantization error."""
    x = co.synth_i8(21, 0, 3000, d)
    x[100] = x[7]  # an exact duplicate: tie broken by position
    q = co.synth_f32(22, 0, 1, d)[0]
    all_c = co.all_distances(x, q, 4)
    all_n = on.distances_i8dom(x, q)
    assert np.array_equal(all_c.view(np.uint32), all_n.view(np.uint32))
    ids, dist, pos = co.topk_scan(x, q, 4, 25)
    order = np.argsort(all_n, kind="stable")[:25]
    assert np.array_equal(pos, order.astype(np.uint64))
    assert np.array_equal(dist.view(np.uint32), all_n[order].view(np.uint32))
    faithful = co.all_distances(x, q, 0)
    assert np.max(np.abs(faithful - all_c)) <= np.sqrt(d) * (0.5 / 127.0) + 1e-5


def test_i8dom_kats():
    # query [1,2,3,4] saturates to [127]*4 (vector.rs:106, no clamp with Some(scale)); row = the same
    x = np.array([[127, 127, 127, 127], [0, 0, 0, 0], [-128, 127, 0, 1]], np.int8)
    q = np.array([1, 2, 3, 4], np.float32)
    got = co.all_distances(x, q, 4)
    want = np.array([0.0, np.sqrt(np.float32(4 * 127 * 127)) / np.float32(127.0),
                     np.sqrt(np.float32(255 * 255 + 127 * 127 + 126 * 126)) / np.float32(127.0)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # f32 rows are not a valid operand for this mode
    assert len(co.topk_scan(np.zeros((4, 4), np.float32), q, 4, 2)[0]) == 0
