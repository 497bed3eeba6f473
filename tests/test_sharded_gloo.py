"""CPU, world_size 2 (gloo): the N>1 path -- contiguous row shards, one all-gather of
the per-shard top-k lists, merge by (distance, global row) -- reproduces the oracle's
global result, ties included.  Local lists come from the oracle here (no GPU); the
exchange and the merge are the product's (vfs_amd.sharded)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle as co
        from vfs_amd import sharded
        rng = np.random.default_rng(42)  # same data on every rank
        n, d, k, nq = 1003, 16, 10, 3
        base = rng.uniform(-10, 10, (30, d)).round(4).astype(np.float32)
        rows = co.quantize(base[rng.integers(0, 30, n)], 127.0)  # saturated, many exact ties
        queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
        lo, hi = sharded.split_rows(n, world)[rank]
        for metric in (0, 1):
            local = np.zeros((nq, k), dtype=sharded.HIT_DTYPE)
            for qi in range(nq):
                ids, dd, pos = co.topk_scan(rows[lo:hi], queries[qi], metric, k)
                m = len(ids)
                local[qi, :m]["dist"] = dd
                local[qi, :m]["row"] = pos + lo          # global scan position
                local[qi, :m]["id"] = pos + lo + 1       # ids 1..N in scan order
                local[qi, :m]["flags"] = 1
            t = torch.from_numpy(local.view(np.int32).reshape(nq, k, sharded.HIT_WORDS))
            merged = sharded.exchange_and_merge(t, k)
            ids, dd, cnt = sharded.hits_to_arrays(merged)
            for qi in range(nq):
                oi, od, _ = co.topk_scan(rows, queries[qi], metric, k)
                assert cnt[qi] == len(oi)
                assert (ids[qi, :cnt[qi]] == oi).all(), (rank, metric, qi, ids[qi], oi)
                assert (dd[qi, :cnt[qi]].view(np.uint32) == od.view(np.uint32)).all()
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_merge_matches_global(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


def _agree_worker(rank, world, port, ret, bad_rank):
    """Comm.agree / Comm.bootstrap with a mocked communicator: vfsx_comm_init (Comm.__init__) must be entered by
    every rank or by none."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfs_amd import sharded
        entered = []

        class MockComm(sharded.Comm):
            def __init__(self, rank, world, uid, device):  # stands where ncclCommInitRank would block
                entered.append((rank, world, uid, device))
                self._h = None

            @staticmethod
            def preflight(device):
                return (False, "injected: device fell off the bus") if rank == bad_rank else (True, "")

            @staticmethod
            def unique_id():
                return bytes(range(128))

        try:
            MockComm.bootstrap(rank)
            ret[rank] = ("joined", entered[0][2] == bytes(range(128)) and entered[0][:2] == (rank, world))
        except RuntimeError as e:
            ret[rank] = ("refused", str(e), len(entered))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank", [-1, 0, 2])
def test_comm_bootstrap_is_all_or_nothing(bad_rank):
    world = 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_agree_worker, args=(world, port, ret, bad_rank), nprocs=world, join=True)
    got = dict(ret)
    if bad_rank < 0:
        assert got == {r: ("joined", True) for r in range(world)}
    else:
        for r in range(world):  # the same refusal on every rank, naming the culprit; nobody entered the init
            kind, msg, n_entered = got[r]
            assert kind == "refused" and n_entered == 0
            assert f"rank {bad_rank}: injected" in msg and "no rank enters ncclCommInitRank" in msg
