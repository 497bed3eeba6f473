"""GPU: the reference's tests.sh sequence (tests.sh:72-113) against vfs_amd.server over real
sockets, checking the bodies as well as the status codes, plus log ingest + /search hydration."""
import http.client
import json
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HEAD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_head.dat")


def _req(port, method, path, body=None):
    conn = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
    data = None if body is None else json.dumps(body)
    conn.request(method, path, body=data, headers={"Content-Type": "application/json"} if data else {})
    r = conn.getresponse()
    text = r.read().decode()
    conn.close()
    return r.status, text


@pytest.fixture
def srv(tmp_path):
    from vfs_amd import server
    s = server.VfsServer(port=0, state_dir=str(tmp_path / "state"))
    t = threading.Thread(target=s.serve_forever, daemon=True)
    t.start()
    yield s, s.server_address[1]
    s.shutdown()
    s.server_close()
    if s.app.state is not None:
        s.app.state.corpus.close()


def test_tests_sh_sequence(srv):
    s, port = srv
    assert _req(port, "POST", "/init", {"vector_dimension": 4, "storage_name": "my_vfs", "truncate_data": True,
                                        "quantize": True}) == (200, '{"status":"initialized"}')
    assert _req(port, "POST", "/init", {"vector_dimension": 4, "truncate_data": True, "quantize": True})[0] == 400
    assert _req(port, "POST", "/vectors", {"values": [1.0, 2.0, 3.0, 4.0], "name": "Vector de ejemplo",
                                           "tags": ["demo", "test"]}) == (201, '{"id":1,"status":"success"}')
    st, body = _req(port, "POST", "/vectors", {"values": [1.0, 2.0], "name": "x", "tags": ["error"]})
    assert (st, body) == (400, '{"error":"Vector dimension mismatch. Expected 4, got 2"}')
    # quantized store: [1,2,3,4]*127 saturates to 127 -> dequantized [1,1,1,1] (vector.rs:94-108,135-139)
    st, body = _req(port, "GET", "/vectors/1")
    assert st == 200
    assert body == ('{"id":1,"values":[1.0,1.0,1.0,1.0],"name":"Vector de ejemplo_quantized",'
                    '"tags":["demo","test","quantized"]}')
    assert _req(port, "GET", "/vectors/99") == (404, '{"error":"Vector not found"}')
    st, body = _req(port, "POST", "/search", {"values": [1.0, 2.0, 3.0, 4.0], "top_k": 3, "ef_search": 6,
                                              "search_type": "approximate", "distance_method": "euclidean"})
    assert st == 200
    doc = json.loads(body)
    assert list(doc.keys()) == ["query_time_ms", "results"]             # alphabetical (BTreeMap)
    assert list(doc["results"][0].keys()) == ["distance", "id", "vector"]
    assert list(doc["results"][0]["vector"].keys()) == ["id", "name", "tags", "values"]
    assert '"distance":3.7416574954986572,"id":1' in body               # sqrt(14) as f32, widened to f64
    assert doc["results"][0]["vector"]["values"] == [1.0, 1.0, 1.0, 1.0]
    st, body = _req(port, "POST", "/search", {"values": [1.0, 2.0, 3.0], "top_k": 3})
    assert (st, body) == (400, '{"error":"Query vector dimension mismatch. Expected 4, got 3"}')
    assert _req(port, "POST", "/flush") == (200, '{"message":"Memtable flushed to disk","status":"success"}')
    assert _req(port, "POST", "/snapshot") == (200, '{"status":"State saved successfully"}')
    assert _req(port, "POST", "/restore") == (200, '{"status":"State loaded successfully"}')


def test_search_limit_is_ef_search_and_batch_zero(srv, oracle):
    """tcp.rs:397-403: ef_search (default 6) is the result limit, top_k the scan batch size."""
    s, port = srv
    _req(port, "POST", "/init", {"vector_dimension": 8, "truncate_data": True, "quantize": False})
    rng = np.random.default_rng(1)
    rows = rng.uniform(-1, 1, (40, 8)).astype(np.float32)
    for i, r in enumerate(rows):
        assert _req(port, "POST", "/vectors", {"values": [float(x) for x in r], "name": f"v{i}", "tags": []})[0] == 201
    q = rng.uniform(-1, 1, 8).astype(np.float32)
    body = {"values": [float(x) for x in q], "top_k": 3, "search_type": "exact", "distance_method": "cosine"}
    doc = json.loads(_req(port, "POST", "/search", body)[1])
    assert len(doc["results"]) == 6                                        # default ef_search
    oi, od, _ = oracle.topk_scan(rows, q, 1, 6)
    assert [r["id"] for r in doc["results"]] == [int(i) for i in oi]
    assert [np.float32(r["distance"]) for r in doc["results"]] == list(od)
    body["ef_search"] = 2
    assert len(json.loads(_req(port, "POST", "/search", body)[1])["results"]) == 2
    body["top_k"] = 0                                                      # load_batch(0): nothing scanned
    assert json.loads(_req(port, "POST", "/search", body)[1])["results"] == []


def test_log_ingest_and_hydration(oracle):
    """F1: the reference's sample log -> device corpus; ids, metadata and search parity."""
    import vfs_amd
    from vfs_amd import vfslog
    c = vfslog.open_corpus(HEAD)
    assert len(c) == 8 and c.quantize and c.dim == 16
    first = c.get_rows_raw([0])[0]
    assert first.tolist() == [127, -128, 127, -128, 127, -128, 127, 127, -128, 127, -2, 127, 127, 127, -23, 127]
    assert c.get_metadata(0) == {"name": "Vec_0_quantized", "tags": ["benchmark", "quantized"],
                                 "created_at": "2025-04-17T14:11:10.265600617Z"}
    assert c.get_ids(np.arange(8)).tolist() == list(range(1, 9))
    rows = c.get_rows_raw(np.arange(8))
    q = np.linspace(-1, 1, 16).astype(np.float32)
    for metric in (0, 1):
        ids, dist, cnt = c.search_batch(q, 5, metric)
        oi, od, _ = oracle.topk_scan(rows, q, metric, 5)
        assert (ids[0] == oi).all() and (dist[0].view(np.uint32) == od.view(np.uint32)).all()
    c.close()


def test_log_ingest_large_dense_records(tmp_path, oracle):
    """Records larger than the reference's 1 KiB read window load fine."""
    import vfs_amd
    from tests import logfmt
    from vfs_amd import vfslog
    rng = np.random.default_rng(3)
    rows = rng.uniform(-1, 1, (300, 300)).astype(np.float32)
    p = tmp_path / "vectors.dat"
    with open(p, "wb") as f:
        for i, r in enumerate(rows):
            f.write(logfmt.record(1000 + i, r.tolist(), name=f"n{i}", tags=("a", "b")))
    c = vfslog.open_corpus(str(p))
    assert len(c) == 300 and not c.quantize
    assert (c.get_rows_f32(np.arange(300)).view(np.uint32) == rows.view(np.uint32)).all()
    assert c.get_metadata(299)["name"] == "n299"
    q = rng.uniform(-1, 1, 300).astype(np.float32)
    ids, dist, cnt = c.search_batch(q, 10, 0)
    oi, od, _ = oracle.topk_scan(rows, q, 0, 10, ids=np.arange(1000, 1300, dtype=np.uint64))
    assert (ids[0] == oi).all() and (dist[0].view(np.uint32) == od.view(np.uint32)).all()
    # quantized records into a dense corpus come back as as_f32_vec would give them
    p2 = tmp_path / "q.dat"
    qrows = rng.integers(-128, 128, (20, 300), dtype=np.int8)
    with open(p2, "wb") as f:
        for i, r in enumerate(qrows):
            f.write(logfmt.record(i + 1, r.tolist(), scale=127.0))
    d = vfs_amd.Corpus(300)
    assert d.load_log(str(p2)) == 20
    assert (d.get_rows_f32(np.arange(20)).view(np.uint32) == oracle.dequantize(qrows, 127.0).view(np.uint32)).all()
    e = vfs_amd.Corpus(300, quantize=True)
    with pytest.raises(vfs_amd.VfsxError):
        e.load_log(str(p))  # Dense records cannot enter a quantized corpus
    for x in (c, d, e):
        x.close()


def _mk_server(tmp_path):
    from vfs_amd import server
    s = server.VfsServer(port=0, state_dir=str(tmp_path / "state"))
    threading.Thread(target=s.serve_forever, daemon=True).start()
    return s, s.server_address[1]


def _stop(s):
    s.shutdown()
    s.server_close()
    if s.app.state is not None:
        s.app.state.corpus.close()


@pytest.mark.parametrize("quantize", [False, True])
def test_flush_snapshot_restart_restore(tmp_path, oracle, quantize):
    """F3/F4: registered vectors reach data/vectors.dat in the reference's record format (auto flush
    every 10, storage_manager.rs:12,310-312), /snapshot writes the reference's bincode VFSState, and
    a restarted server serves the same data directory: same /search answer, ids continue."""
    from tests.test_log_cpu import _parse_records
    from vfs_amd import vfslog
    rng = np.random.default_rng(11)
    rows = rng.uniform(-1, 1, (23, 8)).astype(np.float32)
    q = rng.uniform(-1, 1, 8).astype(np.float32)
    sbody = {"values": [float(x) for x in q], "top_k": 5, "ef_search": 7, "distance_method": "cosine"}
    data = tmp_path / "data" / "vectors.dat"

    a, port = _mk_server(tmp_path)
    try:
        assert _req(port, "POST", "/init", {"vector_dimension": 8, "storage_name": "shop", "truncate_data": True,
                                            "quantize": quantize})[0] == 200
        for i, r in enumerate(rows):
            assert _req(port, "POST", "/vectors", {"values": [float(x) for x in r], "name": f"v{i}",
                                                   "tags": ["t", str(i)]})[0] == 201
        assert len(_parse_records(data.read_bytes())) == 20           # two automatic flushes, 3 still pending
        before = _req(port, "POST", "/search", sbody)
        assert before[0] == 200
        assert _req(port, "POST", "/snapshot")[0] == 200
    finally:
        _stop(a)

    recs = _parse_records(data.read_bytes())
    assert [r[0] for r in recs] == list(range(1, 24))
    if quantize:
        assert all(r[5] == 127.0 for r in recs)
        assert np.array_equal(np.array([r[1] for r in recs], np.int8), oracle.quantize(rows, 127.0))
        assert recs[4][2] == "v4_quantized" and recs[4][3] == ["t", "4", "quantized"]
    else:
        assert np.array_equal(np.array([r[1] for r in recs], np.float32).view(np.uint32), rows.view(np.uint32))
        assert recs[4][2] == "v4" and recs[4][3] == ["t", "4"]
    st = vfslog.decode_state((tmp_path / "state" / "vfs_state.bin").read_bytes())
    assert st["name"] == "shop" and st["next_id"] == 24 and st["quantize"] == quantize
    offs, pos = [], 0
    for r in recs:  # index_map = id -> offset of its record
        offs.append(pos)
        pos += len(vfslog.encode_record(r[0], r[1], r[2], r[3], r[4], scale=r[5]))
    assert st["index_map"] == {i + 1: o for i, o in enumerate(offs)}

    b, port = _mk_server(tmp_path)
    try:
        assert _req(port, "POST", "/init", {"vector_dimension": 8, "truncate_data": False, "quantize": quantize})[0] == 200
        after = _req(port, "POST", "/search", sbody)
        strip = lambda body: [(r["id"], r["distance"], r["vector"]) for r in json.loads(body)["results"]]
        assert after[0] == 200 and strip(after[1]) == strip(before[1])
        assert _req(port, "POST", "/restore") == (200, '{"status":"State loaded successfully"}')
        assert b.app.state.name == "shop"
        st, body = _req(port, "POST", "/vectors", {"values": [0.0] * 8, "name": "late", "tags": []})
        assert (st, body) == (201, '{"id":24,"status":"success"}')
    finally:
        _stop(b)


def test_restore_without_snapshot_is_500(srv):
    s, port = srv
    _req(port, "POST", "/init", {"vector_dimension": 4, "truncate_data": True, "quantize": False})
    st, body = _req(port, "POST", "/restore")
    assert st == 500 and body.startswith('{"error":"Failed to load state: I/O error:')


def test_log_ranges_as_shards_equal_the_whole_file(tmp_path, oracle):
    """F1 x 8e: three corpora each load their contiguous record range of one reference-format log
    (what the ranks of a row-sharded deployment do, vfs_amd.sharded.ShardedCorpus.load_log), search
    locally and merge by (distance, global row): identical to one corpus holding the whole file."""
    import torch
    import vfs_amd
    from vfs_amd import sharded, vfslog
    rng = np.random.default_rng(31)
    n, d = 1000, 24
    base = rng.uniform(-1, 1, (50, d)).astype(np.float32)
    rows = base[rng.integers(0, 50, n)]                      # many exact duplicates: ties across shards
    p = str(tmp_path / "vectors.dat")
    vfslog.append_records(p, [vfslog.encode_record(5000 + i, r, f"v{i}", ["t"], "2025-04-17T14:11:10Z")
                              for i, r in enumerate(rows)])
    whole = vfslog.open_corpus(p)
    assert len(whole) == n
    q = rng.uniform(-1, 1, (4, d)).astype(np.float32)
    k = 20
    shards = []
    for lo, hi in sharded.split_rows(n, 3):
        c = vfs_amd.Corpus(d)
        assert c.load_log(p, first=lo, count=hi - lo) == hi - lo
        c.set_row_base(lo)
        assert c.get_ids([0])[0] == 5000 + lo and c.get_metadata(0)["name"] == f"v{lo}"
        shards.append(c)
    dq = torch.from_numpy(q).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for metric in (0, 1):
        lists = torch.zeros((3, 4, k, 6), dtype=torch.int32, device="cuda")
        for g, c in enumerate(shards):
            c.search_device(dq.data_ptr(), 4, k, metric, lists[g].data_ptr(), st)
        torch.cuda.synchronize()
        g = lists.cpu().numpy().view(np.uint8).reshape(3, 4, k, 24).view(sharded.HIT_DTYPE).reshape(3, 4, k)
        merged = sharded.merge_hits_host(g, k)
        ids, dist, cnt = whole.search_batch(q, k, metric)
        assert (merged["id"] == ids).all()
        assert (merged["dist"].view(np.uint32) == dist.view(np.uint32)).all()
        oi, od, _ = oracle.topk_scan(rows, q[0], metric, k, ids=np.arange(5000, 5000 + n, dtype=np.uint64))
        assert (ids[0] == oi).all()
    assert shards[2].load_log(p, first=n + 5, count=10) == 0  # a range past the end loads nothing
    for c in shards + [whole]:
        c.close()
