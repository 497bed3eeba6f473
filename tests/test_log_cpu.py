"""CPU: the log parser (host logic in libvfsx.so) against the reference's own sample data
(first 8 records of vfs/data/vectors.dat) and against synthetic logs."""
import os

import pytest

from tests import logfmt
from vfs_amd import vfslog
import vfs_amd

HEAD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_head.dat")


def test_inspect_reference_sample():
    info = vfslog.inspect(HEAD)
    # Quantized, d=16, scale 127.0, ids 1.., payloads 142-145 B
    assert info == {"records": 8, "dim": 16, "quantized": True, "scale": 127.0, "first_id": 1, "last_id": 8,
                    "file_bytes": 1240}


def test_inspect_synthetic(tmp_path):
    p = tmp_path / "vectors.dat"
    recs = [logfmt.record(i + 1, [0.5 * i] * 300, name=f"Vec_{i}") for i in range(5)]  # 1.2 KB records
    blob = recs[0] + b"\x00garbage\xde\xad" + recs[1] + recs[2] + recs[3] + recs[4][:-7]  # truncated tail
    p.write_bytes(blob)
    info = vfslog.inspect(str(p))
    assert info["records"] == 4 and info["dim"] == 300 and not info["quantized"]
    assert info["first_id"] == 1 and info["last_id"] == 4


def test_corrupt_payload_is_serialization_error(tmp_path):
    p = tmp_path / "bad.dat"
    good = logfmt.record(1, [1.0, 2.0])
    bad = bytearray(logfmt.record(2, [1.0, 2.0]))
    bad[12:16] = (7).to_bytes(4, "little")  # variant 7 does not exist
    p.write_bytes(good + bytes(bad))
    with pytest.raises(vfs_amd.VfsxError) as e:
        vfslog.inspect(str(p))
    assert e.value.kind == "serialization_error"


def test_missing_file_is_io_error():
    with pytest.raises(vfs_amd.VfsxError) as e:
        vfslog.inspect("/nonexistent/vectors.dat")
    assert e.value.kind == "io_error"


# ---- the writer and the VFSState snapshot (F3/F4) ---------------------------------------------------

def _parse_records(buf):
    """Test-side decoder of the record format (independent of the product's C parser)."""
    import struct
    pos, out = 0, []
    while pos < len(buf):
        assert buf[pos:pos + 4] == b"\xde\xad\xbe\xef"
        size, = struct.unpack_from("<Q", buf, pos + 4)
        p = pos + 12
        variant, vid, n = struct.unpack_from("<IQQ", buf, p)
        p += 20
        if variant == 0:
            values = list(struct.unpack_from(f"<{n}f", buf, p)); p += 4 * n; scale = None
        else:
            values = list(struct.unpack_from(f"<{n}b", buf, p)); p += n
            scale, = struct.unpack_from("<f", buf, p); p += 4

        def s(p):
            ln, = struct.unpack_from("<Q", buf, p)
            return buf[p + 8:p + 8 + ln].decode(), p + 8 + ln
        name, p = s(p)
        nt, = struct.unpack_from("<Q", buf, p); p += 8
        tags = []
        for _ in range(nt):
            t, p = s(p); tags.append(t)
        created, p = s(p)
        assert p == pos + 12 + size
        out.append((vid, values, name, tags, created, scale))
        pos = p
    return out


def test_writer_reproduces_the_reference_log_bytes():
    """KAT: re-encoding the fields of the first 8 records of the reference's own data/vectors.dat with
    the product's writer gives back the reference's bytes."""
    from vfs_amd import vfslog
    buf = open(HEAD, "rb").read()
    recs = _parse_records(buf)
    assert len(recs) == 8 and recs[0][0] == 1 and recs[0][5] == 127.0
    again = b"".join(vfslog.encode_record(vid, values, name, tags, created, scale=scale)
                     for vid, values, name, tags, created, scale in recs)
    assert again == buf
    dense = vfslog.encode_record(7, [0.5, -1.25], "n", ["a"], "2025-04-17T14:11:10Z")
    assert dense == logfmt.record(7, [0.5, -1.25], name="n", tags=("a",), created="2025-04-17T14:11:10Z")


def test_append_records_offsets(tmp_path):
    from vfs_amd import vfslog
    p = str(tmp_path / "data" / "vectors.dat")
    r1 = vfslog.encode_record(1, [1.0], "a", [], vfslog.utc_now_rfc3339())
    r2 = vfslog.encode_record(2, [2.0], "b", ["t"], vfslog.utc_now_rfc3339())
    assert vfslog.append_records(p, [r1, r2]) == [0, len(r1)]
    assert vfslog.append_records(p, [r1]) == [len(r1) + len(r2)]           # save_vector: offset = old file end
    info = vfslog.inspect(p)
    assert info["records"] == 3 and info["dim"] == 1 and not info["quantized"]


def test_state_codec_matches_the_reference_snapshot():
    """KAT: the reference's own state/vfs_state.bin (55 bytes; tests/golden/README.md)."""
    from vfs_amd import vfslog
    buf = open(os.path.join(os.path.dirname(HEAD), "vfs_state.bin"), "rb").read()
    st = vfslog.decode_state(buf)
    assert st == {"name": "my_vfs", "next_id": 2, "current_offset": 162, "index_map": {1: 0}, "quantize": True}
    assert vfslog.encode_state(**st) == buf
    big = vfslog.encode_state("x", 10, 0, {5: 500, 1: 0, 3: 250}, False)
    assert list(vfslog.decode_state(big)["index_map"].items()) == [(1, 0), (3, 250), (5, 500)]   # BTreeMap order
    for bad in (b"", buf[:20], buf[:-1], b"\xff" * 8 + buf[8:]):
        with pytest.raises(ValueError):
            vfslog.decode_state(bad)


def test_rfc3339_autosi():
    from vfs_amd import vfslog
    assert vfslog.utc_now_rfc3339(1744899070265600617) == "2025-04-17T14:11:10.265600617Z"
    assert vfslog.utc_now_rfc3339(1744899070265600000) == "2025-04-17T14:11:10.265600Z"
    assert vfslog.utc_now_rfc3339(1744899070265000000) == "2025-04-17T14:11:10.265Z"
    assert vfslog.utc_now_rfc3339(1744899070000000000) == "2025-04-17T14:11:10Z"
