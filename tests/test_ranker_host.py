"""CPU: host-side mirror of Ranker::new / Ranker::search (rank.rs:76-105) -- argument
meaning, defaults and error behaviour, exercised against a stub corpus (no compute)."""
import numpy as np
import pytest

import vfs_amd
from vfs_amd import DistanceMethod, Ranker, SearchType


class StubCorpus:
    """Records what Ranker asks for; returns 3 fake hits."""
    dim = 4

    def __init__(self):
        self.calls = []

    def search_batch(self, q, k, metric):
        self.calls.append((np.array(q), k, metric))
        ids = np.zeros((1, k), np.uint64)
        dist = np.zeros((1, k), np.float32)
        n = min(k, 3)
        ids[0, :n] = [7, 8, 9][:n]
        dist[0, :n] = [0.1, 0.2, 0.3][:n]
        return ids, dist, np.array([n], np.uint32)


def test_enum_order_matches_reference():
    # rank.rs:67-72
    # the reference's four methods keep its declaration order; the one extension comes after them
    assert [m.name for m in DistanceMethod][:4] == ["Euclidean", "Cosine", "SimdEuclidean", "SimdCosine"]
    assert [int(m) for m in DistanceMethod] == [0, 1, 2, 3, 4]
    assert DistanceMethod.EuclideanI8Domain == 4 and len(DistanceMethod) == 5
    # tcp.rs:388-394
    assert DistanceMethod.from_request("cosine") is DistanceMethod.Cosine
    assert DistanceMethod.from_request("simd_euclidean") is DistanceMethod.SimdEuclidean
    assert DistanceMethod.from_request(None) is DistanceMethod.Euclidean
    assert DistanceMethod.from_request("manhattan") is DistanceMethod.Euclidean


def test_defaults():
    r = Ranker(SearchType.Exact, DistanceMethod.Euclidean)
    assert r.ef_search == 6  # rank.rs:86
    c = StubCorpus()
    out = r.search([1, 2, 3, 4], 5, None, c)
    assert c.calls[0][1] == 5  # result_limit.unwrap_or(5), rank.rs:129
    assert out == [(7, pytest.approx(0.1)), (8, pytest.approx(0.2)), (9, pytest.approx(0.3))]
    assert all(isinstance(i, int) and isinstance(d, float) for i, d in out)


def test_batch_size_is_ignored_except_zero():
    c = StubCorpus()
    r = Ranker(SearchType.Exact, DistanceMethod.Cosine)
    a = r.search([1, 2, 3, 4], 1, 2, c)
    b = r.search([1, 2, 3, 4], 1000, 2, c)
    assert a == b and c.calls[0][2] == 1
    assert r.search([1, 2, 3, 4], 0, 2, c) == []  # load_batch(0) -> empty, storage_manager.rs:198
    assert len(c.calls) == 2


def test_errors():
    c = StubCorpus()
    r = Ranker(SearchType.Exact, DistanceMethod.Euclidean)
    with pytest.raises(vfs_amd.VfsxError) as e:  # rank.rs:273-275 panics
        r.search([1, 2, 3], 5, 5, c)
    assert e.value.kind == "invalid_vector"
    with pytest.raises(vfs_amd.VfsxError):  # rank.rs:175 panics on limit 0
        r.search([1, 2, 3, 4], 5, 0, c)
    with pytest.raises(NotImplementedError):
        Ranker(SearchType.Approximate, DistanceMethod.Euclidean).search([1, 2, 3, 4], 5, 5, c)


def test_strict_simd_widths_opt_in():
    c = StubCorpus()
    c.dim = 5
    q = [1, 2, 3, 4, 5]
    assert Ranker(SearchType.Exact, DistanceMethod.SimdEuclidean).search(q, 5, 5, c)
    with pytest.raises(vfs_amd.VfsxError) as e:  # rank.rs:31
        Ranker(SearchType.Exact, DistanceMethod.SimdEuclidean, strict_simd_widths=True).search(q, 5, 5, c)
    assert "No se puede usar SIMD con longitud 5" in str(e.value)
    c.dim = 4
    assert Ranker(SearchType.Exact, DistanceMethod.SimdCosine, strict_simd_widths=True).search(q[:4], 5, 5, c)
