"""GPU: golden fixtures, the Ranker mirror, device-resident search and the hit merge,
all through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from tests import golden_util

pytestmark = pytest.mark.gpu

CASES, QUANT = golden_util.load()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_on_gpu(case):
    import vfs_amd
    rows = case["rows"]
    c = vfs_amd.Corpus(rows.shape[1], quantize=case["quantized"])
    c.append(rows, ids=case["ids"])
    r = vfs_amd.Ranker(vfs_amd.SearchType.Exact, vfs_amd.DistanceMethod(case["metric"]))
    got = r.search(case["query"], case["batch"], case["limit"], c)
    assert [g[0] for g in got] == case["expect_ids"].tolist()
    assert (bits([g[1] for g in got]) == bits(case["expect_dist"])).all()
    c.close()


def test_device_lut_is_ieee_division():
    import vfs_amd
    for scale in (127.0, 100.0, 3.0, 0.37):
        c = vfs_amd.Corpus(256, quantize=True, scale=scale)
        allq = np.arange(-128, 128).astype(np.int8)[None, :]
        c.append(allq)
        want = (allq.astype(np.float32) / np.float32(scale)).astype(np.float32)
        assert (bits(c.get_rows_f32([0])) == bits(want)).all()
        c.close()


def test_search_device_and_merge(oracle):
    import torch
    import vfs_amd
    from vfs_amd import sharded
    rng = np.random.default_rng(21)
    n, d, k, nq = 9001, 48, 10, 5
    rows = oracle.quantize(rng.uniform(-10, 10, (n, d)).round(4).astype(np.float32), 127.0)
    queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    cuts = sharded.split_rows(n, 3)
    shards = []
    for lo, hi in cuts:
        c = vfs_amd.Corpus(d, quantize=True)
        c.append(rows[lo:hi], ids=np.arange(lo + 1, hi + 1, dtype=np.uint64))
        c.set_row_base(lo)
        shards.append(c)
    dq = torch.from_numpy(queries).cuda()
    for metric in (0, 1):
        lists = torch.zeros((3, nq, k, 6), dtype=torch.int32, device="cuda")
        for g, c in enumerate(shards):
            c.search_device(dq.data_ptr(), nq, k, metric, lists[g].data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
        out = torch.zeros((nq, k, 6), dtype=torch.int32, device="cuda")
        lib = vfs_amd.load()
        vfs_amd._lib.check(lib.vfsx_merge_hits_device(C.c_void_p(lists.data_ptr()), 3, nq, k,
                                                      C.c_void_p(out.data_ptr()), 0,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        ids, dist, cnt = sharded.hits_to_arrays(out)
        # host merge of the same lists agrees with the device merge
        g = lists.cpu().numpy().view(np.uint8).reshape(3, nq, k, 24).view(sharded.HIT_DTYPE).reshape(3, nq, k)
        hm = sharded.merge_hits_host(g, k)
        assert (hm["id"] == ids).all() and (bits(hm["dist"]) == bits(dist)).all()
        for qi in range(nq):
            oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k)
            assert (ids[qi] == oi).all() and (bits(dist[qi]) == bits(od)).all()
    for c in shards:
        c.close()


def test_multi_query_host_api(oracle):
    import vfs_amd
    rng = np.random.default_rng(33)
    n, d, k, nq = 4000, 20, 7, 9
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    for metric in (0, 1, 2, 3):
        ids, dist, cnt = c.search_batch(queries, k, metric)
        for qi in range(nq):
            oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k)
            assert cnt[qi] == k and (ids[qi] == oi).all() and (bits(dist[qi]) == bits(od)).all()
    c.close()


def test_growth_beyond_capacity(oracle):
    import vfs_amd
    rng = np.random.default_rng(34)
    d = 12
    rows = rng.uniform(-1, 1, (1000, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, capacity_rows=64)
    for a in range(0, 1000, 137):
        c.append(rows[a:a + 137])
    q = rng.uniform(-1, 1, d).astype(np.float32)
    ids, dist, cnt = c.search_batch(q, 10, 1)
    oi, od, _ = oracle.topk_scan(rows, q, 1, 10)
    assert (ids[0] == oi).all() and (bits(dist[0]) == bits(od)).all()
    c.close()


def test_error_behaviour():
    import vfs_amd
    c = vfs_amd.Corpus(8)
    with pytest.raises(vfs_amd.VfsxError) as e:
        c.append(np.zeros((3, 7), np.float32))  # tcp.rs:307-312 -> 400
    assert e.value.kind == "invalid_vector" and "Expected 8, got 7" in str(e.value)
    with pytest.raises(vfs_amd.VfsxError) as e:
        c.search_batch(np.zeros(9, np.float32), 5, 0)  # tcp.rs:371-376 -> 400
    assert "Query vector dimension mismatch" in str(e.value)
    with pytest.raises(vfs_amd.VfsxError):
        c.search_batch(np.zeros(8, np.float32), 0, 0)
    ids, dist, cnt = c.search_batch(np.zeros(8, np.float32), 5000, 0)  # beyond one pass: served (empty corpus: 0 hits)
    assert cnt[0] == 0
    with pytest.raises(vfs_amd.VfsxError):
        c.search_batch(np.zeros(8, np.float32), (1 << 20) + 1, 0)  # VFSX_MAX_K
    with pytest.raises(vfs_amd.VfsxError):
        c.search_batch(np.zeros(8, np.float32), 5, 9)
    # empty corpus: zero hits, no error
    ids, dist, cnt = c.search_batch(np.zeros(8, np.float32), 5, 0)
    assert cnt[0] == 0
    c.close()


def test_nan_sorts_last_documented_deviation():
    """Zero-norm rows give NaN under cosine (rank.rs:299); the reference's order is then
    unspecified.  vfsx: NaN sorts after every number, and is only returned when k > #non-NaN."""
    import vfs_amd
    rows = np.array([[1, 0], [0, 0], [0, 1], [0, 0], [1, 1]], np.float32)
    c = vfs_amd.Corpus(2)
    c.append(rows)
    ids, dist, cnt = c.search_batch(np.array([1, 0], np.float32), 5, 1)
    assert cnt[0] == 5 and ids[0].tolist() == [1, 5, 3, 2, 4]
    assert np.isnan(dist[0, 3:]).all() and not np.isnan(dist[0, :3]).any()
    c.close()


@pytest.mark.parametrize("quantize", [False, True])
@pytest.mark.parametrize("nq", [2, 3, 4, 5, 8, 17])
def test_multi_query_passes_bit_exact(oracle, quantize, nq):
    """nq > 1: 2/4/8 queries share one pass over the corpus; every query's result must
    equal its own single-query result (ragged last group included)."""
    import vfs_amd
    rng = np.random.default_rng(nq * 3 + quantize)
    n, d = 7000, 72 if not quantize else 40
    if quantize:
        rows = rng.integers(-128, 128, (n, d), dtype=np.int8)
        rows[np.abs(rows).sum(axis=1) == 0, 0] = 1
    else:
        rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, quantize=quantize)
    c.append(rows)
    for metric in (0, 1):
        for k in (10, 100):
            ids, dist, cnt = c.search_batch(queries, k, metric)
            for qi in range(nq):
                oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k)
                assert cnt[qi] == k and (ids[qi] == oi).all(), (metric, k, qi)
                assert (bits(dist[qi]) == bits(od)).all(), (metric, k, qi)
    c.close()


def test_large_k_batch_splits_into_several_launches(oracle):
    """k = 700 over >= 4096 row-blocks makes one query's list workspace 5.7 MB, so a 60-query batch
    exceeds the 256 MiB workspace and is scanned in several launches: results must not depend on it."""
    import vfs_amd
    n, d, k, nq = 270_000, 4, 700, 60
    rows = oracle.synth_f32(41, 0, n, d)
    queries = oracle.synth_f32(42, 0, nq, d)
    c = vfs_amd.Corpus(d, capacity_rows=n)
    c.append(rows)
    for metric in (0, 1):
        ids, dist, cnt = c.search_batch(queries, k, metric)
        assert (cnt == k).all()
        for qi in (0, 31, 45, 46, 59):  # both sides of the launch boundary
            oi, od, _ = oracle.topk_scan(rows, queries[qi], metric, k)
            assert (ids[qi] == oi).all()
            assert (dist[qi].view(np.uint32) == od.view(np.uint32)).all()
    c.close()


def test_vfsx_init_counts_devices():
    import vfs_amd
    from vfs_amd import _lib
    lib = vfs_amd.load()
    n = lib.vfsx_device_count()
    assert n >= 1 and lib.vfsx_init(1) == _lib.OK and lib.vfsx_init(n) == _lib.OK
    assert lib.vfsx_init(n + 1) == _lib.ERR_NO_DEVICE


def test_rccl_collective_and_device_merge_single_rank():
    """The N > 1 data path (one RCCL all_gather_into_tensor of the hit lists + vfsx_merge_hits_device) on
    the real `nccl` backend, with the one GPU this box has: world size 1, collective forced.  Run in a
    child process so the process group does not leak into the test session."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_world1_check.py")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "OK rccl" in r.stdout


def test_implicit_ids_continue_after_the_largest_explicit_id(oracle):
    """A store whose ids are not its row positions (e.g. loaded from a log with gaps): rows appended without ids
    continue at max(id) + 1 (next_id, storage_manager.rs:126-130), never at row position + 1."""
    import vfs_amd
    rng = np.random.default_rng(12)
    rows = rng.uniform(-1, 1, (6, 8)).astype(np.float32)
    c = vfs_amd.Corpus(8)
    c.append(rows[:3], ids=np.array([10, 50, 20], dtype=np.uint64))
    c.append(rows[3:5])          # -> 51, 52
    c.append(rows[5:], ids=np.array([7], dtype=np.uint64))
    assert c.get_ids(range(6)).tolist() == [10, 50, 20, 51, 52, 7]
    ids, dist, cnt = c.search_batch(rows[4], 1, 0)
    assert ids[0, 0] == 52 and dist[0, 0] == 0.0
    c.close()


def test_searches_on_different_streams_share_the_workspaces_safely(oracle):
    """vfsx_search_device on two streams without any synchronisation in between: the second call's stream is made
    to wait for the first call's kernels (they share the corpus' list workspace), so both results are right."""
    import torch
    import vfs_amd
    from vfs_amd import sharded
    rng = np.random.default_rng(13)
    n, d, k = 400000, 64, 10
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    q = rng.uniform(-1, 1, (2, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    dq = torch.from_numpy(q).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = torch.zeros((40, k, 6), dtype=torch.int32, device="cuda")
    for i in range(40):  # alternate streams, never synchronise
        st = s1 if i % 2 == 0 else s2
        c.search_device(dq[i % 2].data_ptr(), 1, k, i % 2, outs[i].data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    ids, dist, cnt = sharded.hits_to_arrays(outs)
    want = [oracle.topk_scan(rows, q[m], m, k) for m in (0, 1)]
    for i in range(40):
        oi, od, _ = want[i % 2]
        assert (ids[i] == oi).all() and (bits(dist[i]) == bits(od)).all()
    c.close()


def test_search_after_the_previous_stream_was_destroyed():
    """A host that makes a stream per request: search on stream A, destroy A, search on stream B.  The corpus
    remembers the previous search's stream to order the shared workspaces; a dead handle there must not fail this
    search or wedge the corpus (chain_search_stream)."""
    import torch
    import vfs_amd
    from vfs_amd import sharded
    hip = C.CDLL("libamdhip64.so")  # the runtime torch already loaded
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    rng = np.random.default_rng(5)
    n, d, k = 50000, 32, 10
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    q = rng.uniform(-1, 1, (1, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    dq = torch.from_numpy(q).cuda()
    outs = torch.zeros((6, k, 6), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for i in range(6):
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        c.search_device(dq.data_ptr(), 1, k, 0, outs[i].data_ptr(), s.value)
        assert hip.hipStreamSynchronize(s) == 0
        assert hip.hipStreamDestroy(s) == 0  # the next search meets a dead `last_search_stream`
    c.search_device(dq.data_ptr(), 1, k, 0, outs[0].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ids, dist, cnt = sharded.hits_to_arrays(outs)
    for i in range(1, 6):
        assert (ids[i] == ids[0]).all() and (bits(dist[i]) == bits(dist[0])).all() and cnt[i] == k
    c.close()


def test_vfsx_init_leaves_the_current_device_alone():
    import torch
    import vfs_amd
    lib = vfs_amd.load()
    before = torch.cuda.current_device()
    assert lib.vfsx_init(lib.vfsx_device_count()) == 0
    assert torch.cuda.current_device() == before


def test_insert_batching_keeps_every_row_searchable_at_once(oracle):
    """vfsx_corpus_set_insert_batch: one-row appends are staged on the host and committed together, but every read
    of the store commits first -- a vector is searchable the moment its append returns, once, in scan order, with
    the ids it was given (storage_manager.rs:317-335 / SURVEY F3)."""
    import torch
    import vfs_amd
    rng = np.random.default_rng(44)
    d = 24
    rows = rng.uniform(-1, 1, (3000, d)).astype(np.float32)
    for quantize in (False, True):
        c = vfs_amd.Corpus(d, quantize=quantize)
        c.set_insert_batch(512)
        ref = oracle.quantize(rows, 127.0) if quantize else rows
        for i in range(700):  # single rows, explicit ids
            c.append_f32(rows[i:i + 1], ids=[1000 + i])
            assert len(c) == i + 1
            if i in (0, 1, 9, 511, 512, 699):  # searched while rows wait: the newest row is its own nearest neighbour
                probe = ref[i].astype(np.float32) / np.float32(127.0) if quantize else rows[i]
                ids, dist, cnt = c.search_batch(probe, 1, 0)
                assert cnt[0] == 1 and dist[0, 0] == 0.0 and ids[0, 0] == 1000 + int(np.flatnonzero((ref[:i + 1] == ref[i]).all(axis=1))[0])
        c.append_f32(rows[700:716], ids=np.arange(1700, 1716, dtype=np.uint64))  # 16 rows: still staged
        c.append_f32(rows[716:2000], ids=np.arange(1716, 3000, dtype=np.uint64))  # a large append commits what waits, in order
        for i in range(2000, 2010):
            c.append_f32(rows[i:i + 1])  # no ids on a store that has explicit ones: continue after the largest
        assert len(c) == 2010 and c.get_ids([0, 699, 700, 1999, 2000, 2009]).tolist() == [1000, 1699, 1700, 2999, 3000, 3009]
        q = rng.uniform(-1, 1, d).astype(np.float32)
        for metric in (0, 1):
            ids, dist, cnt = c.search_batch(q, 10, metric)
            oi, od, op = oracle.topk_scan(ref[:2010], q, metric, 10)
            assert (ids[0] == op + 1000).all() and (bits(dist[0]) == bits(od)).all()
        # the entry point that allocates nothing never commits: it refuses while rows wait
        c.append_f32(rows[2010:2011])
        dq = torch.from_numpy(q[None, :]).cuda()
        out = torch.zeros((1, 10, 6), dtype=torch.int32, device="cuda")
        with pytest.raises(vfs_amd.VfsxError) as e:
            c.search_device_noalloc(dq.data_ptr(), 1, 10, 0, out.data_ptr(), 0)
        assert e.value.code == vfs_amd._lib.ERR_MEMTABLE and "vfsx_corpus_commit" in str(e.value)
        c.commit()
        c.search_device_noalloc(dq.data_ptr(), 1, 10, 0, out.data_ptr(), 0)
        torch.cuda.synchronize()
        c.set_insert_batch(0)
        c.append_f32(rows[2011:2012])
        assert len(c) == 2012
        c.close()


def test_host_quantizer_equals_the_device_quantizer(oracle):
    """vfsx_quantize_rows_host (what the server writes to the log) == the bytes vfsx_corpus_append_quantize leaves in
    HBM == the oracle's quantize, on halves, saturation, NaN, infinities, subnormals and random values."""
    import vfs_amd
    rng = np.random.default_rng(3)
    special = np.array([0.5 / 127, -0.5 / 127, 1.5 / 127, 2.5 / 127, -2.5 / 127, 63.5 / 127, 0.49999997 / 127, 1.0, -1.0,
                        1.004, -1.008, 2.0, -2.0, 1e9, -1e9, np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-45, -1e-45,
                        126.5 / 127, 127.5 / 127, -127.5 / 127, -128.5 / 127], dtype=np.float32)
    vals = np.concatenate([special, rng.uniform(-1.2, 1.2, 64 * 300 - special.size).astype(np.float32)]).reshape(300, 64)
    vals[1:, :] += (rng.integers(-128, 128, vals[1:, :].shape) + 0.5).astype(np.float32) / np.float32(127) * (rng.random(vals[1:, :].shape) < 0.2)
    lib = vfs_amd.load()
    host = np.empty(vals.shape, dtype=np.int8)
    vfs_amd._lib.check(lib.vfsx_quantize_rows_host(vals.ctypes.data_as(C.POINTER(C.c_float)), vals.size, 127.0,
                                                   host.ctypes.data_as(C.POINTER(C.c_int8))))
    c = vfs_amd.Corpus(64, quantize=True)
    c.append_f32(vals)
    dev = c.get_rows_raw(np.arange(300))
    assert (host == dev).all() and (host == oracle.quantize(vals, 127.0)).all()
    c.close()
