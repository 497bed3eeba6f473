"""Test-side encoder of the reference's log record format (serializer.rs:25-71 +
bincode 1.3 fixint LE of VFSVector, vector.rs:11-43).  Test infrastructure only."""
import struct

MARKER = b"\xde\xad\xbe\xef"


def _s(text: str) -> bytes:
    b = text.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def record(vid, values, name="v", tags=("t",), created="2025-04-17T14:11:10.265600617Z", scale=None) -> bytes:
    if scale is None:  # Dense
        body = struct.pack("<IQQ", 0, vid, len(values)) + struct.pack(f"<{len(values)}f", *values)
    else:              # Quantized
        body = (struct.pack("<IQQ", 1, vid, len(values)) + struct.pack(f"<{len(values)}b", *values)
                + struct.pack("<f", scale))
    body += _s(name) + struct.pack("<Q", len(tags)) + b"".join(_s(t) for t in tags) + _s(created)
    return MARKER + struct.pack("<Q", len(body)) + body
