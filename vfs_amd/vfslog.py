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


# ---- writer (serializer.rs:25-71) and the VFSState snapshot (storage_manager.rs:18-26,367-427) -----
# Host-side bookkeeping for the F3/F4 rows of SURVEY.md 8f: what `/flush` and `/snapshot` leave on
# disk keeps the reference's formats, so a data directory can move between the two servers.

import os
import struct
import time

MARKER = b"\xde\xad\xbe\xef"  # serializer.rs:13


def _bstr(text: str) -> bytes:  # bincode String: u64 LE length + UTF-8
    b = text.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def utc_now_rfc3339(ns: int = None) -> str:
    """chrono `DateTime<Utc>` as serde writes it (RFC 3339, `SecondsFormat::AutoSi`, 'Z')."""
    ns = time.time_ns() if ns is None else int(ns)
    sec, frac = divmod(ns, 1_000_000_000)
    t = time.gmtime(sec)
    head = f"{t.tm_year:04d}-{t.tm_mon:02d}-{t.tm_mday:02d}T{t.tm_hour:02d}:{t.tm_min:02d}:{t.tm_sec:02d}"
    if frac == 0:
        return head + "Z"
    if frac % 1_000_000 == 0:
        return f"{head}.{frac // 1_000_000:03d}Z"
    if frac % 1_000 == 0:
        return f"{head}.{frac // 1_000:06d}Z"
    return f"{head}.{frac:09d}Z"


def encode_record(vid: int, values, name: str, tags, created_at: str, scale=None) -> bytes:
    """One log record: marker | u64 payload length | bincode 1.3 (fixint, LE) of VFSVector:
    u32 variant (0 Dense, 1 Quantized) | u64 id | u64 n | n x f32 or n x i8 | [f32 scale_factor] |
    String name | Vec<String> tags | String created_at   (vector.rs:11-43)."""
    import numpy as np
    if scale is None:
        v = np.ascontiguousarray(values, dtype="<f4")
        body = struct.pack("<IQQ", 0, vid, v.size) + v.tobytes()
    else:
        v = np.ascontiguousarray(values, dtype=np.int8)
        body = struct.pack("<IQQ", 1, vid, v.size) + v.tobytes() + struct.pack("<f", scale)
    body += _bstr(name) + struct.pack("<Q", len(tags)) + b"".join(_bstr(t) for t in tags) + _bstr(created_at)
    return MARKER + struct.pack("<Q", len(body)) + body


def append_records(path: str, records) -> list:
    """save_vector (serializer.rs:25-71) for each encoded record; returns the offsets written."""
    parent = os.path.dirname(path)
    if parent:
        os.makedirs(parent, exist_ok=True)
    offsets = []
    with open(path, "ab") as f:
        f.seek(0, os.SEEK_END)
        for rec in records:
            offsets.append(f.tell())
            f.write(rec)
    return offsets


def encode_state(name: str, next_id: int, current_offset: int, index_map: dict, quantize: bool) -> bytes:
    """bincode of VFSState { name, next_id, current_offset, index_map: BTreeMap<u64, usize>, quantize }
    in declaration order (storage_manager.rs:18-26); BTreeMap = u64 count + (key, value) ascending."""
    out = _bstr(name) + struct.pack("<QQ", next_id, current_offset) + struct.pack("<Q", len(index_map))
    for key in sorted(index_map):
        out += struct.pack("<QQ", key, index_map[key])
    return out + struct.pack("<B", 1 if quantize else 0)


def decode_state(buf: bytes) -> dict:
    try:
        n, = struct.unpack_from("<Q", buf, 0)
        if n > len(buf):
            raise ValueError("name length beyond the buffer")
        name = buf[8:8 + n].decode("utf-8")
        pos = 8 + n
        next_id, current_offset, count = struct.unpack_from("<QQQ", buf, pos)
        pos += 24
        if count * 16 > len(buf) - pos:
            raise ValueError("index_map length beyond the buffer")
        index_map = {}
        for _ in range(count):
            key, off = struct.unpack_from("<QQ", buf, pos)
            index_map[key] = off
            pos += 16
        flag, = struct.unpack_from("<B", buf, pos)
        if flag > 1:
            raise ValueError("invalid bool")
    except (struct.error, UnicodeDecodeError) as e:
        raise ValueError(str(e))
    return {"name": name, "next_id": next_id, "current_offset": current_offset, "index_map": index_map,
            "quantize": bool(flag)}
