"""The reference's on-disk vector log (`data/vectors.dat`, serializer.rs:25-153): inspection
through the C ABI.  Parsing lives in libvfsx.so (vfsx_log_inspect / vfsx_corpus_load_log)."""
from __future__ import annotations

import ctypes as C

from . import _lib


def inspect(path: str) -> dict:
    info = _lib.LogInfo()
    _lib.check(_lib.load().vfsx_log_inspect(str(path).encode(), C.byref(info)))
    return {"records": int(info.records), "dim": int(info.dim), "quantized": info.dtype == _lib.I8,
            "scale": float(info.scale), "first_id": int(info.first_id), "last_id": int(info.last_id),
            "file_bytes": int(info.file_bytes)}


def open_corpus(path: str, device: int = 0):
    """Create a Corpus shaped like the log (dim, dense/quantized, scale) and load it."""
    from .ranker import Corpus
    meta = inspect(path)
    if meta["records"] == 0:
        raise _lib.VfsxError(_lib.ERR_IO, f"{path}: no records")
    c = Corpus(meta["dim"], quantize=meta["quantized"], scale=meta["scale"] or 127.0,
               capacity_rows=meta["records"], device=device)
    c.load_log(path)
    return c
