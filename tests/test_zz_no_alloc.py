"""GPU: vfsx_search_device_noalloc -- the entry point that allocates nothing -- must produce exactly what
vfsx_search_device produces (every byte of every vfsx_hit), through the C ABI, on two calls in a row that share
the library's static device workspace, the second with a smaller input.  Collected last on purpose (the
round's end runs pytest -x)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(c, d_q, nq, k, metric):
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    ref = torch.zeros((nq, k, 6), dtype=torch.int32, device="cuda")
    got = torch.full((nq, k, 6), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
    c.search_device(d_q.data_ptr(), nq, k, metric, ref.data_ptr(), stream)
    c.search_device_noalloc(d_q.data_ptr(), nq, k, metric, got.data_ptr(), stream)
    torch.cuda.synchronize()
    return ref.cpu().numpy(), got.cpu().numpy()


@pytest.mark.parametrize("quantize", [False, True])
def test_noalloc_equals_search_device_two_calls_in_a_row(oracle, quantize):
    import torch
    import vfs_amd
    from vfs_amd import sharded
    rng = np.random.default_rng(99 + quantize)
    # first call: 50 001 rows of dimension 100 (not a whole number of chunks: the queries are staged in the
    # workspace), 5 queries, k = 10; second call: a smaller corpus, fewer queries, smaller k
    big = rng.uniform(-1, 1, (50001, 100)).astype(np.float32)
    small = rng.uniform(-1, 1, (3000, 100)).astype(np.float32)
    q_big = torch.from_numpy(rng.uniform(-1, 1, (5, 100)).astype(np.float32)).cuda()
    q_small = torch.from_numpy(rng.uniform(-1, 1, (2, 100)).astype(np.float32)).cuda()
    cb = vfs_amd.Corpus(100, quantize=quantize)
    cs = vfs_amd.Corpus(100, quantize=quantize)
    cb.append_f32(big)
    cs.append_f32(small)
    metrics = (0, 1, 4) if quantize else (0, 1)
    for metric in metrics:
        ref1, got1 = _both(cb, q_big, 5, 10, metric)
        ref2, got2 = _both(cs, q_small, 2, 4, metric)
        assert (ref1 == got1).all() and (ref2 == got2).all()
        # and both equal the oracle (ids identical, distances bit-exact)
        rows = oracle.quantize(small, 127.0) if quantize else small
        ids, dist, _ = sharded.hits_to_arrays(torch.from_numpy(got2))
        for qi in range(2):
            kw = {"scale": 127.0} if quantize else {}
            if metric == 4:
                continue
            oi, od, _ = oracle.topk_scan(rows, q_small[qi].cpu().numpy(), metric, 4, **kw)
            assert (ids[qi] == oi).all() and (dist[qi].view(np.uint32) == od.view(np.uint32)).all()
    cb.close()
    cs.close()


def test_noalloc_large_k_many_queries_and_limits():
    import torch
    import vfs_amd
    rng = np.random.default_rng(5)
    rows = rng.uniform(-1, 1, (20000, 64)).astype(np.float32)
    c = vfs_amd.Corpus(64)
    c.append(rows)
    # k = 1024 fills most of the list workspace per query: 40 queries run as several launches
    q = torch.from_numpy(rng.uniform(-1, 1, (40, 64)).astype(np.float32)).cuda()
    ref, got = _both(c, q, 40, 1024, 0)
    assert (ref == got).all()
    ref, got = _both(c, q, 3, 7, 1)  # the next, smaller call sees no leftovers
    assert (ref == got).all()
    # beyond the stated maximum: one staged query of a chunk-ragged dimension above 262 144 floats
    huge = vfs_amd.Corpus(262147, capacity_rows=64)
    qh = torch.zeros((1, 262147), dtype=torch.float32, device="cuda")
    out = torch.zeros((1, 1, 6), dtype=torch.int32, device="cuda")
    with pytest.raises(vfs_amd.VfsxError) as e:
        huge.search_device_noalloc(qh.data_ptr(), 1, 1, 0, out.data_ptr(), 0)
    assert e.value.code == vfs_amd._lib.ERR_INVALID_VECTOR and "static workspace" in str(e.value)
    huge.close()
    with pytest.raises(vfs_amd.VfsxError) as e:  # longer results need the key buffer of the radix select
        c.search_device_noalloc(q.data_ptr(), 1, 2000, 0, out.data_ptr(), 0)
    assert "limited to 1024" in str(e.value)
    c.close()
