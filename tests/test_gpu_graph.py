"""GPU: vfsx_search_device_noalloc under HIP graph capture -- the caller it exists for (include/vfsx.h).  The calls
enqueue kernels only (no allocation, no synchronisation, no host-side read), so a stream capture records them and a
replay re-runs the searches; results must equal the directly launched ones bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("quantize,metric", [(False, 1), (False, 2), (True, 0)])
def test_noalloc_search_is_graph_capturable(oracle, quantize, metric):
    import torch
    import vfs_amd
    rng = np.random.default_rng(5 + metric)
    n, d, k, nq = 50_000, 96, 10, 24
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    c = vfs_amd.Corpus(d, quantize=quantize)
    (c.append_f32 if quantize else c.append)(rows)
    ref_rows = oracle.quantize(rows, 127.0) if quantize else rows
    dev = torch.device("cuda:0")
    queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    q = torch.from_numpy(queries).to(dev)
    direct = torch.zeros((nq, k, 6), dtype=torch.int32, device=dev)
    replayed = torch.zeros((nq, k, 6), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for j in range(nq):
            c.search_device_noalloc(q[j].data_ptr(), 1, k, metric, direct[j].data_ptr(), s.cuda_stream)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for j in range(nq):
            c.search_device_noalloc(q[j].data_ptr(), 1, k, metric, replayed[j].data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
    assert not bool(replayed.any())  # captured, not run
    g.replay()
    torch.cuda.synchronize()
    assert bool((replayed == direct).all())
    # new queries in the same buffers: the replay reads them (the graph holds pointers, not values)
    queries2 = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    q.copy_(torch.from_numpy(queries2).to(dev))
    g.replay()
    torch.cuda.synchronize()
    from vfs_amd import sharded
    ids, dist, cnt = sharded.hits_to_arrays(replayed)
    for qi in (0, nq - 1):
        oi, od, _ = oracle.topk_scan(ref_rows, queries2[qi], metric, k)
        assert (ids[qi] == oi).all()
        assert (np.ascontiguousarray(dist[qi]).view(np.uint32) == np.ascontiguousarray(od).view(np.uint32)).all()
    del g
    c.close()
