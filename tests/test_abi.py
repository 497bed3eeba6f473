"""CPU: the C-ABI library loads, exports every symbol include/vfsx.h declares, fails
loudly without a GPU, and its pure-host logic (hit merge) is right.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vfs_amd
from vfs_amd import _lib, sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "vfsx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vfsx_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol():
    lib = vfs_amd.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SYMBOLS)


def test_hit_layout():
    assert C.sizeof(_lib.Hit) == 24 and sharded.HIT_DTYPE.itemsize == 24
    assert _lib.Hit.row.offset == 8 and _lib.Hit.id.offset == 16


def test_no_gpu_fails_loudly():
    lib = vfs_amd.load()
    if lib.vfsx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(vfs_amd.VfsxError) as e:
        vfs_amd.Corpus(16)
    assert e.value.code == _lib.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)
    assert lib.vfsx_init(1) == _lib.ERR_NO_DEVICE            # the boot-time check says the same
    assert b"no CPU fallback" in lib.vfsx_last_error()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vfs_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the oracle", "").replace("CPU oracle", ""), (dirpath, f)


def _mk_hits(dist, row, ids, valid):
    h = np.zeros(len(dist), dtype=sharded.HIT_DTYPE)
    h["dist"], h["row"], h["id"], h["flags"] = dist, row, ids, valid
    return h


def test_merge_hits_host_order_and_ties():
    k, nq = 4, 2
    # two shards; ties on distance must resolve by global row; invalid entries lose
    a = np.zeros((nq, k), dtype=sharded.HIT_DTYPE)
    b = np.zeros((nq, k), dtype=sharded.HIT_DTYPE)
    a[0] = _mk_hits([0.5, 1.0, 1.0, 2.0], [3, 1, 7, 9], [13, 11, 17, 19], [1, 1, 1, 1])
    b[0] = _mk_hits([0.25, 1.0, 3.0, 0.0], [100, 104, 105, 0], [200, 204, 205, 0], [1, 1, 1, 0])
    a[1] = _mk_hits([-0.0, 0.0, np.nan, 0.0], [5, 6, 2, 0], [5, 6, 2, 0], [1, 1, 1, 0])
    b[1] = _mk_hits([0.0, 1e-9, 0.0, 0.0], [50, 51, 0, 0], [50, 51, 0, 0], [1, 1, 0, 0])
    out = sharded.merge_hits_host(np.stack([a, b]), k)
    assert out[0]["id"].tolist() == [200, 13, 11, 17]
    assert out[0]["row"].tolist() == [100, 3, 1, 7]
    # -0.0 == +0.0 -> row order 5, 6, 50; then 1e-9; NaN and invalid never beat them
    assert out[1]["row"].tolist() == [5, 6, 50, 51]
    assert (out["flags"] & 1).all()


def test_merge_hits_host_fewer_than_k():
    k = 3
    a = np.zeros((1, k), dtype=sharded.HIT_DTYPE)
    b = np.zeros((1, k), dtype=sharded.HIT_DTYPE)
    a[0] = _mk_hits([1.0, 0, 0], [2, 0, 0], [3, 0, 0], [1, 0, 0])
    out = sharded.merge_hits_host(np.stack([a, b]), k)
    assert (out[0]["flags"] & 1).tolist() == [1, 0, 0] and out[0]["id"][0] == 3


def test_split_rows_contiguous():
    assert sharded.split_rows(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sharded.split_rows(100_000_000, 8)[7] == (87_500_000, 100_000_000)
    assert sharded.split_rows(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


# ---- the Rust crate's extern "C" block must declare every entry point of include/vfsx.h with matching types ----

_C_TO_RUST = {
    "void": "()", "int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "double": "f64",
    "const char *": "*const c_char", "char *": "*mut c_char", "void *": "*mut c_void", "const void *": "*const c_void",
    "const float *": "*const f32", "float *": "*mut f32", "int8_t *": "*mut i8", "double *": "*mut f64",
    "const uint64_t *": "*const u64", "uint64_t *": "*mut u64", "uint32_t *": "*mut u32", "int *": "*mut c_int",
    "vfsx_corpus *": "*mut VfsxCorpus", "const vfsx_corpus *": "*const VfsxCorpus",
    "vfsx_comm *": "*mut VfsxComm", "const vfsx_comm *": "*const VfsxComm",
    "vfsx_hit *": "*mut VfsxHit", "const vfsx_hit *": "*const VfsxHit", "vfsx_log_info *": "*mut VfsxLogInfo",
}


def _c_type(decl: str) -> str:
    """'const float *queries' -> 'const float *' ; 'uint32_t nq' -> 'uint32_t' ; 'void' -> 'void'."""
    decl = " ".join(decl.split())
    if decl == "void":
        return "void"
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", decl)  # strip the parameter name
    t = m.group(1).strip() if m and m.group(1).strip() else decl
    return re.sub(r"\s*\*\s*", " *", t).strip()


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "vfsx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^([A-Za-z_][A-Za-z0-9_ \*]*?)\b(vfsx_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.M | re.S):
        ret = re.sub(r"\s*\*\s*", " *", " ".join(m.group(1).split())).strip()
        args = [_c_type(a) for a in m.group(3).split(",")]
        protos[m.group(2)] = (ret, [] if args == ["void"] else args)
    return protos


def rust_prototypes():
    src = open(os.path.join(ROOT, "rust", "vfsx", "src", "lib.rs")).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    block = re.sub(r"//[^\n]*", "", block)
    protos = {}
    for m in re.finditer(r"pub fn (vfsx_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
        protos[m.group(1)] = ((m.group(3) or "()").strip(), [" ".join(a.split()) for a in args])
    return protos


def test_rust_crate_declares_every_entry_point_with_matching_types():
    c, r = header_prototypes(), rust_prototypes()
    assert set(c) == set(_lib.SYMBOLS)  # the parser sees what the library exports
    assert set(r) == set(c), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, (ret, args) in c.items():
        want = (_C_TO_RUST[ret], [_C_TO_RUST[a] for a in args])
        assert r[name] == want, (name, r[name], want)


def test_cxx_mirror_and_ctypes_cover_the_sharded_entry_points():
    hpp = open(os.path.join(ROOT, "vfs_amd", "cxx", "vfsx_ranker.hpp")).read()
    for sym in ("vfsx_comm_unique_id", "vfsx_comm_init", "vfsx_comm_destroy", "vfsx_search_sharded_host"):
        assert sym in hpp, sym
