"""GPU: the library's own RCCL communicator (vfsx_comm_*, include/vfsx.h) through ctypes, with NO
torch.distributed anywhere: world size 1 on the box's one GPU runs the real ncclCommInitRank / ncclAllGather +
device merge that an N-GPU job runs per rank."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_comm_world1_sharded_search_equals_oracle(oracle):
    import torch
    import vfs_amd
    from vfs_amd import sharded
    assert vfs_amd.load().vfsx_comm_rccl_version() > 0
    comm = sharded.Comm(0, 1, sharded.Comm.unique_id(), 0)
    assert comm.rank == 0 and comm.world == 1
    rng = np.random.default_rng(17)
    n, d, k, nq = 20011, 72, 10, 6
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    q = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    c.set_row_base(1000)  # a shard in the middle of a larger scan order: rows are global positions
    dq = torch.from_numpy(q).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    for metric in (0, 1):
        local = torch.zeros((nq, k, 6), dtype=torch.int32, device="cuda")
        glob = torch.zeros((nq, k, 6), dtype=torch.int32, device="cuda")
        glob2 = torch.zeros((nq, k, 6), dtype=torch.int32, device="cuda")
        c.search_device(dq.data_ptr(), nq, k, metric, local.data_ptr(), stream)
        comm.search(c, dq.data_ptr(), nq, k, metric, glob.data_ptr(), stream)       # scan + all-gather + merge
        comm.merge_hits(local.data_ptr(), nq, k, glob2.data_ptr(), stream)           # the exchange alone
        torch.cuda.synchronize()
        assert (local.cpu() == glob.cpu()).all() and (local.cpu() == glob2.cpu()).all()
        h = glob.cpu().numpy().view(np.uint8).reshape(nq, k, 24).view(sharded.HIT_DTYPE).reshape(nq, k)
        ids_h, dist_h, cnt_h = comm.search_host(c, q, k, metric)                     # host buffers (Ranker level)
        for qi in range(nq):
            oi, od, op = oracle.topk_scan(rows, q[qi], metric, k)
            assert (h["id"][qi] == oi).all() and (bits(h["dist"][qi]) == bits(od)).all()
            assert (h["row"][qi] == op + 1000).all()
            assert (ids_h[qi] == oi).all() and (bits(dist_h[qi]) == bits(od)).all() and cnt_h[qi] == k
    comm.close()
    c.close()


def test_comm_allocates_at_init_only(oracle):
    """vfsx_comm_init sizes every exchange buffer (gather, result, pinned result, queries) for 4096 x 10 hits per
    rank; searches within that never allocate in the communicator (vfsx_comm_alloc_count stays put), a larger batch
    grows them once, vfsx_comm_reserve does it ahead of time."""
    import torch
    import vfs_amd
    from vfs_amd import sharded
    comm = sharded.Comm(0, 1, sharded.Comm.unique_id(), 0)
    a0 = comm.alloc_count
    assert a0 == 4  # gather, result, pinned result, query buffer
    rng = np.random.default_rng(3)
    n, d, k = 30000, 64, 10
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    c = vfs_amd.Corpus(d)
    c.append(rows)
    stream = torch.cuda.current_stream().cuda_stream
    for nq in (1, 7, 256, 4096):
        q = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
        dq = torch.from_numpy(q).cuda()
        out = torch.zeros((nq, k, 6), dtype=torch.int32, device="cuda")
        comm.search(c, dq.data_ptr(), nq, k, 0, out.data_ptr(), stream)
        comm.merge_hits(out.data_ptr(), nq, k, out.data_ptr(), stream)
        torch.cuda.synchronize()
        ids_h, dist_h, cnt_h = comm.search_host(c, q, k, 0)
        assert comm.alloc_count == a0, nq
        oi, od, _ = oracle.topk_scan(rows, q[nq - 1], 0, k)
        assert (ids_h[nq - 1] == oi).all() and (bits(dist_h[nq - 1]) == bits(od)).all()
    comm.reserve(8192 * 10)
    a1 = comm.alloc_count
    assert a1 == a0 + 3  # gather, result, pinned result grew; the query buffer was left alone (0 = keep)
    q = rng.uniform(-1, 1, (8192, d)).astype(np.float32)
    ids_h, dist_h, cnt_h = comm.search_host(c, q, k, 0)
    assert comm.alloc_count == a1 and (cnt_h == k).all()
    comm.close()
    c.close()


def test_comm_sharded_search_joins_the_gather_when_the_local_scan_fails():
    """A rank whose local scan fails (here: an unknown metric, which vfsx_search_device rejects) must still take part
    in the all-gather -- with an empty list -- before it reports, or its peers wait for ever."""
    import torch
    import vfs_amd
    from vfs_amd import sharded
    comm = sharded.Comm(0, 1, sharded.Comm.unique_id(), 0)
    c = vfs_amd.Corpus(16)
    c.append(np.ones((100, 16), np.float32))
    dq = torch.ones((2, 16), dtype=torch.float32, device="cuda")
    out = torch.full((2, 5, 6), 7, dtype=torch.int32, device="cuda")
    with pytest.raises(vfs_amd.VfsxError) as e:
        comm.search(c, dq.data_ptr(), 2, 5, 9, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert "joined the all-gather with an empty list" in str(e.value)
    torch.cuda.synchronize()
    ids, dist, cnt = sharded.hits_to_arrays(out)
    assert (cnt == 0).all()  # the merged result of this world-1 job: nothing, written by the merge kernel
    comm.close()
    c.close()


def test_comm_init_wait_is_bounded():
    """ncclCommInitRank cannot be cancelled: a rank whose peer never joins is ended by the library's watchdog after
    VFSX_COMM_INIT_TIMEOUT_S with status 3 and a message, instead of holding its GPU for ever.  Child process: rank 0
    of a world of 2 whose rank 1 does not exist."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from vfs_amd import sharded\n"
            "sharded.Comm(0, 2, sharded.Comm.unique_id(), 0)\n"
            "print('init returned')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VFSX_COMM_INIT_TIMEOUT_S="8")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 3, (r.returncode, r.stdout[-800:], r.stderr[-1500:])
    assert "still waits in ncclCommInitRank after 8 s" in r.stderr and "init returned" not in r.stdout


def test_merge_hits_device_many_lists_and_large_k(oracle):
    """The device merge on its two kernels (<= 64 lists: one-wave G-way merge; more: rank by counting) against the
    host merge, with ties across lists, short lists and k up to 1024."""
    import ctypes as C
    import torch
    import vfs_amd
    from vfs_amd import sharded
    lib = vfs_amd.load()
    rng = np.random.default_rng(4)
    for n_lists, nq, k in ((2, 3, 10), (8, 5, 10), (64, 2, 33), (8, 2, 1024), (70, 2, 12), (3, 4, 1)):
        lists = np.zeros((n_lists, nq, k), dtype=sharded.HIT_DTYPE)
        for g in range(n_lists):
            for qi in range(nq):
                m = int(rng.integers(0, k + 1))  # this shard has only m rows
                dist = np.sort(rng.integers(0, 40, m).astype(np.float32) * np.float32(0.25))  # many exact ties
                row = np.zeros(m, np.uint64)
                # rows ascend within equal distances (what a ranked list looks like); global range of shard g
                for v in np.unique(dist):
                    idx = np.where(dist == v)[0]
                    row[idx] = np.sort(rng.choice(100000, idx.size, replace=False)) + g * 100000
                lists[g, qi, :m]["dist"], lists[g, qi, :m]["row"] = dist, row
                lists[g, qi, :m]["id"], lists[g, qi, :m]["flags"] = row + 1, 1
                lists[g, qi, m:]["dist"], lists[g, qi, m:]["row"], lists[g, qi, m:]["id"] = np.nan, 2**64 - 1, 2**64 - 1
        want = sharded.merge_hits_host(lists, k)
        d_in = torch.from_numpy(lists.view(np.int32).reshape(n_lists, nq, k, 6)).cuda()
        d_out = torch.zeros((nq, k, 6), dtype=torch.int32, device="cuda")
        vfs_amd._lib.check(lib.vfsx_merge_hits_device(C.c_void_p(d_in.data_ptr()), n_lists, nq, k,
                                                      C.c_void_p(d_out.data_ptr()), 0, C.c_void_p(0)))
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint8).reshape(nq, k, 24).view(sharded.HIT_DTYPE).reshape(nq, k)
        valid = (want["flags"] & 1).astype(bool)
        assert ((got["flags"] & 1).astype(bool) == valid).all(), (n_lists, nq, k)
        for f in ("row", "id"):
            assert (got[f][valid] == want[f][valid]).all(), (n_lists, nq, k, f)
        assert (bits(got["dist"][valid]) == bits(want["dist"][valid])).all()
