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
