"""Row-sharded search over the GPUs of one node: one process per GPU, each holding a
contiguous range of the scan order; every rank produces its local top-k with the HIP
scan, the lists are exchanged with ONE all-gather per query batch (RCCL over xGMI on
GPUs; gloo on CPU for the tests) and merged by (distance, global row).

No reference counterpart (the reference is single-process).  Shard g
holds global rows [row_base_g, row_base_g + n_g): contiguity makes (shard, local row)
order equal the reference's scan order, so the first-seen tie-break (rank.rs:170)
survives the merge.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

HIT_WORDS = 6  # sizeof(vfsx_hit) / 4

HIT_DTYPE = np.dtype([("dist", np.float32), ("flags", np.uint32), ("row", np.uint64), ("id", np.uint64)])
assert HIT_DTYPE.itemsize == C.sizeof(_lib.Hit) == 24


def split_rows(n_total: int, world: int) -> list:
    """Contiguous row ranges: shard g = [g*ceil(N/G), min(N, (g+1)*ceil(N/G)))  (SURVEY.md 8e)."""
    per = -(-n_total // world)
    return [(min(n_total, g * per), min(n_total, (g + 1) * per)) for g in range(world)]


def merge_hits_host(gathered: np.ndarray, k: int) -> np.ndarray:
    """gathered: structured HIT_DTYPE array [n_lists, nq, k] -> [nq, k] best by (distance, row)."""
    lib = _lib.load()
    g = np.ascontiguousarray(gathered)
    n_lists, nq, kk = g.shape
    assert kk == k
    out = np.zeros((nq, k), dtype=HIT_DTYPE)
    _lib.check(lib.vfsx_merge_hits_host(g.ctypes.data_as(C.POINTER(_lib.Hit)), n_lists, nq, k,
                                        out.ctypes.data_as(C.POINTER(_lib.Hit))))
    return out


class Comm:
    """vfsx_comm (include/vfsx.h): the library's own RCCL communicator -- ncclAllGather + device merge behind
    the C ABI, the path a C / C++ / Rust host takes.  `Comm.bootstrap()` distributes rank 0's 128-byte id over
    an existing torch.distributed group (any backend); `Comm(rank, world, id, device)` takes an id obtained
    any other way."""

    def __init__(self, rank: int, world: int, uid: bytes, device: int):
        self._lib = _lib.load()
        assert len(uid) == _lib.COMM_ID_BYTES
        buf = C.create_string_buffer(uid, _lib.COMM_ID_BYTES)
        self._h = self._lib.vfsx_comm_init(int(rank), int(world), buf, int(device))
        if not self._h:
            raise _lib.VfsxError(_lib.ERR_INTERNAL, self._lib.vfsx_last_error().decode("utf-8", "replace"))
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    @staticmethod
    def unique_id() -> bytes:
        lib = _lib.load()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(lib.vfsx_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def preflight(device: int):
        """What this rank can check by itself before anyone enters the (uncancellable) ncclCommInitRank:
        RCCL loads, the device exists.  Returns (ok, message)."""
        try:
            lib = _lib.load()
            if lib.vfsx_comm_rccl_version() <= 0:
                return False, "RCCL cannot be loaded: " + lib.vfsx_last_error().decode("utf-8", "replace")
            n = lib.vfsx_device_count()
            if not 0 <= int(device) < n:
                return False, f"HIP device {device} is not visible ({n} device(s))"
            return True, ""
        except Exception as e:  # noqa: BLE001 -- reported to every rank by bootstrap
            return False, f"{type(e).__name__}: {e}"

    @classmethod
    def agree(cls, device: int, group=None) -> bytes:
        """Collective over an existing torch.distributed group (any backend): every rank reports its pre-flight,
        rank 0 adds the communicator id; returns the id, or raises THE SAME RuntimeError on every rank when any
        rank failed -- so either all ranks go on to ncclCommInitRank or none does."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ok, msg = cls.preflight(device)
        uid = None
        if rank == 0 and ok:
            try:
                uid = cls.unique_id()
            except Exception as e:  # noqa: BLE001 -- re-raised on every rank below
                ok, msg = False, f"vfsx_comm_unique_id: {type(e).__name__}: {e}"
        reports = [None] * world
        dist.all_gather_object(reports, (bool(ok), msg, uid), group=group)
        bad = [f"rank {r}: {m}" for r, (o, m, _) in enumerate(reports) if not o]
        if bad:
            raise RuntimeError("vfsx_comm pre-flight failed, no rank enters ncclCommInitRank -- " + "; ".join(bad))
        return reports[0][2]

    @classmethod
    def bootstrap(cls, device: int, group=None) -> "Comm":
        """`agree`, then vfsx_comm_init on every rank.  A rank that cannot join raises in `agree`, on every rank
        alike, before anybody is inside ncclCommInitRank (which has no way out while a peer is missing;
        vfsx_comm_init's own watchdog, VFSX_COMM_INIT_TIMEOUT_S, bounds what is left: a rank that dies between
        the agreement and the init)."""
        import torch.distributed as dist
        uid = cls.agree(device, group)
        return cls(dist.get_rank(group), dist.get_world_size(group), uid, device)

    @property
    def handle(self):
        return self._h

    def merge_hits(self, d_local_ptr: int, nq: int, k: int, d_out_ptr: int, stream_ptr: int = 0) -> None:
        _lib.check(self._lib.vfsx_comm_merge_hits(self._h, C.c_void_p(d_local_ptr), nq, k, C.c_void_p(d_out_ptr),
                                                  C.c_void_p(stream_ptr)))

    def search(self, corpus, d_queries_ptr: int, nq: int, k: int, metric: int, d_hits_ptr: int,
               stream_ptr: int = 0) -> None:
        _lib.check(self._lib.vfsx_search_sharded(corpus.handle, self._h, C.c_void_p(d_queries_ptr), nq, k, int(metric),
                                                 C.c_void_p(d_hits_ptr), C.c_void_p(stream_ptr)))

    def search_host(self, corpus, queries, k: int, metric: int):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, corpus.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint64)
        dist = np.full((nq, k), np.nan, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _lib.check(self._lib.vfsx_search_sharded_host(
            corpus.handle, self._h, q.ctypes.data_as(C.POINTER(C.c_float)), nq, k, int(metric),
            ids.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(C.POINTER(C.c_float)),
            cnt.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ids, dist, cnt

    def reserve(self, hits_per_rank: int, query_floats: int = 0) -> None:
        """Widen the exchange buffers now, so that no later call allocates (vfsx_comm_reserve)."""
        _lib.check(self._lib.vfsx_comm_reserve(self._h, int(hits_per_rank), int(query_floats)))

    @property
    def alloc_count(self) -> int:
        return int(self._lib.vfsx_comm_alloc_count(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vfsx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def exchange_and_merge(local_hits, k: int, group=None, device_index: Optional[int] = None,
                       force_collective: bool = False, comm: Optional[Comm] = None):
    """local_hits: torch int32 tensor [nq, k, 6] (a view of vfsx_hit[nq*k]) on this
    rank's device.  With `comm` (the library's RCCL communicator): vfsx_comm_merge_hits, i.e. ncclAllGather +
    device merge inside libvfsx.so.  Otherwise through torch.distributed: cuda + nccl backend -> RCCL
    all-gather + device merge; anything else (gloo: CPU tests, rehearsals) -> host tensors + host merge.
    Returns the merged [nq, k, 6] tensor (same device), identical on every rank."""
    import torch
    import torch.distributed as dist

    if comm is not None:
        out = torch.empty_like(local_hits)
        src = local_hits.contiguous()
        stream = torch.cuda.current_stream(local_hits.device).cuda_stream
        comm.merge_hits(src.data_ptr(), src.shape[0], k, out.data_ptr(), stream)
        return out
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq = local_hits.shape[0]
    if world == 1 and not (force_collective and dist.is_initialized()):
        return local_hits  # (force_collective: run the all-gather + merge anyway -- single-GPU RCCL test)
    # RCCL moves device tensors; any other backend (gloo, used for CPU tests and
    # rehearsals) gets host tensors and the host merge.
    on_device = local_hits.is_cuda and dist.get_backend(group) == "nccl"
    src = local_hits if on_device or not local_hits.is_cuda else local_hits.cpu()
    flat = src.contiguous().view(-1)
    gathered = torch.empty(world * flat.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(gathered, flat, group=group)  # rank-major: [list][query][k]
    gathered = gathered.view((world,) + tuple(local_hits.shape))
    if on_device:
        out = torch.empty_like(local_hits)
        lib = _lib.load()
        stream = torch.cuda.current_stream(local_hits.device).cuda_stream
        dev = local_hits.device.index if device_index is None else device_index
        _lib.check(lib.vfsx_merge_hits_device(C.c_void_p(gathered.data_ptr()), world, nq, k,
                                              C.c_void_p(out.data_ptr()), dev, C.c_void_p(stream)))
        return out
    g = gathered.numpy().view(np.uint8).reshape(world, nq, k, 24).view(HIT_DTYPE).reshape(world, nq, k)
    merged = merge_hits_host(g, k)
    out = torch.from_numpy(merged.view(np.int32).reshape(nq, k, HIT_WORDS))
    return out.to(local_hits.device) if local_hits.is_cuda else out


def hits_to_arrays(hits_tensor) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """[nq, k, 6] int32 tensor -> (ids[nq,k] u64, dist[nq,k] f32, count[nq])."""
    h = hits_tensor.cpu().numpy()
    nq, k, _ = h.shape
    s = np.ascontiguousarray(h).view(np.uint8).reshape(nq, k, 24).view(HIT_DTYPE).reshape(nq, k)
    valid = (s["flags"] & 1).astype(bool)
    return s["id"].copy(), s["dist"].copy(), valid.sum(axis=1).astype(np.uint32)


class ShardedCorpus:
    """This rank's shard of a row-sharded corpus + the collective search."""

    def __init__(self, dim: int, quantize: bool = False, scale: float = 127.0, capacity_rows: int = 0,
                 device: int = 0, group=None, use_vfsx_comm: Optional[bool] = None):
        """use_vfsx_comm: exchange through the library's own RCCL communicator (the C-ABI path).  Default:
        yes when the process group's backend is nccl (= RCCL), no otherwise (gloo rehearsals on CPU)."""
        import torch.distributed as dist
        from .ranker import Corpus
        self.local = Corpus(dim, quantize=quantize, scale=scale, capacity_rows=capacity_rows, device=device)
        self.device = device
        self.group = group
        self.comm = None
        if use_vfsx_comm is None:
            use_vfsx_comm = dist.is_initialized() and dist.get_backend(group) == "nccl"
        if use_vfsx_comm:
            self.comm = Comm.bootstrap(device, group)

    def set_row_base(self, row_base: int) -> None:
        self.local.set_row_base(row_base)

    def load_log(self, path: str) -> int:
        """Every rank loads its contiguous range of the records of one reference data file
        (split_rows over the record count) and takes that range's start as its row base, so the
        merged ranking equals a single scan of the whole file.  Returns this rank's record count."""
        import torch.distributed as dist
        from . import vfslog
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        lo, hi = split_rows(vfslog.inspect(path)["records"], world)[rank]
        n = self.local.load_log(path, first=lo, count=hi - lo)
        self.local.set_row_base(lo)
        return n

    def search(self, d_queries, k: int, metric: int):
        """d_queries: torch float32 cuda tensor [nq, dim]. Returns merged hits tensor [nq,k,6]."""
        import torch
        nq = d_queries.shape[0]
        local = torch.empty((nq, k, HIT_WORDS), dtype=torch.int32, device=d_queries.device)
        stream = torch.cuda.current_stream(d_queries.device).cuda_stream
        if self.comm is not None:  # local scan + ncclAllGather + merge, all inside libvfsx.so
            self.comm.search(self.local, d_queries.data_ptr(), nq, k, int(metric), local.data_ptr(), stream)
            return local
        self.local.search_device(d_queries.data_ptr(), nq, k, int(metric), local.data_ptr(), stream)
        return exchange_and_merge(local, k, group=self.group, device_index=self.device)

    def close(self) -> None:
        if self.comm is not None:
            self.comm.close()
        self.local.close()
