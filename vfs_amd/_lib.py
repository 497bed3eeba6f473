"""ctypes binding of libvfsx.so (include/vfsx.h).  No CPU fallback: if the HIP
library is missing or unusable, importing/using it raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VFSX_LIB_PATH") or os.path.join(_HERE, "lib", "libvfsx.so")  # (override: A/B of builds)

# vfsx_status (include/vfsx.h), mirroring VFSError (vfs/src/vfs/err.rs:10-16)
OK = 0
ERR_IO = -1
ERR_MEMTABLE = -2
ERR_INVALID_VECTOR = -3
ERR_ID_GENERATION = -4
ERR_SERIALIZATION = -5
ERR_INTERNAL = -6
ERR_NO_DEVICE = -7

F32, I8 = 0, 1
MAX_K = 1 << 20  # VFSX_MAX_K; results beyond 128 hits: one scan + a radix select over every row's key

# every symbol include/vfsx.h declares
SYMBOLS = (
    "vfsx_device_count", "vfsx_init", "vfsx_last_error", "vfsx_version", "vfsx_corpus_create",
    "vfsx_corpus_destroy", "vfsx_corpus_append", "vfsx_corpus_append_quantize",
    "vfsx_corpus_rows", "vfsx_corpus_dim", "vfsx_corpus_dtype", "vfsx_corpus_set_row_base",
    "vfsx_corpus_get_rows_f32", "vfsx_corpus_get_rows_raw", "vfsx_corpus_get_ids",
    "vfsx_corpus_fill_synthetic", "vfsx_search", "vfsx_search_device", "vfsx_search_device_noalloc",
    "vfsx_merge_hits_device", "vfsx_merge_hits_host", "vfsx_corpus_set_profiling",
    "vfsx_corpus_scan_time_ms", "vfsx_corpus_enable_mfma", "vfsx_corpus_mfma_stats", "vfsx_corpus_mfma_mirror",
    "vfsx_log_inspect", "vfsx_corpus_load_log", "vfsx_corpus_load_log_range", "vfsx_corpus_get_metadata_json",
    "vfsx_comm_unique_id", "vfsx_comm_init", "vfsx_comm_destroy", "vfsx_comm_rank", "vfsx_comm_world",
    "vfsx_comm_rccl_version", "vfsx_search_sharded", "vfsx_search_sharded_host", "vfsx_comm_merge_hits",
    "vfsx_comm_reserve", "vfsx_comm_alloc_count", "vfsx_corpus_storage_info",
    "vfsx_corpus_set_insert_batch", "vfsx_corpus_commit", "vfsx_quantize_rows_host",
)
COMM_ID_BYTES = 128


class Hit(C.Structure):
    """vfsx_hit (24 bytes)."""
    _fields_ = [("dist", C.c_float), ("flags", C.c_uint32), ("row", C.c_uint64), ("id", C.c_uint64)]


class LogInfo(C.Structure):
    """vfsx_log_info."""
    _fields_ = [("records", C.c_uint64), ("dim", C.c_uint32), ("dtype", C.c_int), ("scale", C.c_float),
                ("first_id", C.c_uint64), ("last_id", C.c_uint64), ("file_bytes", C.c_uint64)]


class VfsxError(RuntimeError):
    """A negative vfsx_status; `.code` holds it, `.kind` names the VFSError class."""

    KINDS = {ERR_IO: "io_error", ERR_MEMTABLE: "storage_error", ERR_INVALID_VECTOR: "invalid_vector",
             ERR_ID_GENERATION: "id_generation_error", ERR_SERIALIZATION: "serialization_error",
             ERR_INTERNAL: "internal_error", ERR_NO_DEVICE: "no_device"}

    def __init__(self, code: int, msg: str):
        super().__init__(f"vfsx error {code} ({self.KINDS.get(code, 'unknown_error')}): {msg}")
        self.code = code
        self.kind = self.KINDS.get(code, "unknown_error")


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(vfs_amd has no CPU fallback)")
    # PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's).  Two HIP
    # runtimes in one process cannot both own the GPU ("No HIP GPUs are available"), so in
    # a Python host let torch's copy load first; libvfsx.so then binds to that one.
    if not os.environ.get("VFSX_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    vp, u64p, f32p, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    hitp = C.POINTER(Hit)
    lib.vfsx_device_count.restype = C.c_int
    lib.vfsx_init.restype = C.c_int
    lib.vfsx_init.argtypes = [C.c_int]
    lib.vfsx_last_error.restype = C.c_char_p
    lib.vfsx_version.restype = C.c_char_p
    lib.vfsx_corpus_create.restype = vp
    lib.vfsx_corpus_create.argtypes = [C.c_uint32, C.c_int, C.c_float, C.c_uint64, C.c_int]
    lib.vfsx_corpus_destroy.restype = None
    lib.vfsx_corpus_destroy.argtypes = [vp]
    lib.vfsx_corpus_append.argtypes = [vp, vp, u64p, C.c_uint64]
    lib.vfsx_corpus_append_quantize.argtypes = [vp, f32p, u64p, C.c_uint64]
    lib.vfsx_corpus_rows.restype = C.c_uint64
    lib.vfsx_corpus_rows.argtypes = [vp]
    lib.vfsx_corpus_dim.restype = C.c_uint32
    lib.vfsx_corpus_dim.argtypes = [vp]
    lib.vfsx_corpus_dtype.argtypes = [vp]
    lib.vfsx_corpus_set_row_base.argtypes = [vp, C.c_uint64]
    lib.vfsx_corpus_get_rows_f32.argtypes = [vp, u64p, C.c_uint64, f32p]
    lib.vfsx_corpus_get_rows_raw.argtypes = [vp, u64p, C.c_uint64, vp]
    lib.vfsx_corpus_get_ids.argtypes = [vp, u64p, C.c_uint64, u64p]
    lib.vfsx_corpus_fill_synthetic.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.vfsx_corpus_set_insert_batch.argtypes = [vp, C.c_uint32]
    lib.vfsx_corpus_commit.argtypes = [vp]
    lib.vfsx_quantize_rows_host.argtypes = [f32p, C.c_uint64, C.c_float, C.POINTER(C.c_int8)]
    lib.vfsx_corpus_storage_info.argtypes = [vp, u64p, u64p, u64p, u64p]
    lib.vfsx_search.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, C.c_int, u64p, f32p, u32p]
    lib.vfsx_search_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp]
    lib.vfsx_search_device_noalloc.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp]
    lib.vfsx_merge_hits_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_int, vp]
    lib.vfsx_merge_hits_host.argtypes = [hitp, C.c_uint32, C.c_uint32, C.c_uint32, hitp]
    lib.vfsx_corpus_set_profiling.argtypes = [vp, C.c_int]
    lib.vfsx_log_inspect.argtypes = [C.c_char_p, C.POINTER(LogInfo)]
    lib.vfsx_corpus_load_log.argtypes = [vp, C.c_char_p, u64p]
    lib.vfsx_corpus_load_log_range.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_uint64, u64p]
    lib.vfsx_corpus_get_metadata_json.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_uint64, u64p]
    lib.vfsx_corpus_enable_mfma.argtypes = [vp, C.c_int]
    lib.vfsx_corpus_mfma_stats.argtypes = [vp, u64p, u64p, C.c_int]
    lib.vfsx_corpus_mfma_mirror.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    lib.vfsx_corpus_scan_time_ms.argtypes = [vp, C.POINTER(C.c_double), u64p, C.c_int]
    lib.vfsx_comm_unique_id.argtypes = [vp]
    lib.vfsx_comm_init.restype = vp
    lib.vfsx_comm_init.argtypes = [C.c_int, C.c_int, vp, C.c_int]
    lib.vfsx_comm_destroy.restype = None
    lib.vfsx_comm_destroy.argtypes = [vp]
    lib.vfsx_comm_rank.argtypes = [vp]
    lib.vfsx_comm_world.argtypes = [vp]
    lib.vfsx_comm_reserve.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.vfsx_comm_alloc_count.restype = C.c_uint64
    lib.vfsx_comm_alloc_count.argtypes = [vp]
    lib.vfsx_search_sharded.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp]
    lib.vfsx_comm_merge_hits.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    lib.vfsx_search_sharded_host.argtypes = [vp, vp, f32p, C.c_uint32, C.c_uint32, C.c_int, u64p, f32p, u32p]
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError if the library does not export it
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != OK:
        raise VfsxError(rc, load().vfsx_last_error().decode("utf-8", "replace"))
