"""GPU: the library's host code under AddressSanitizer + UBSan while it drives the device (tests/cxx/host_san.cpp,
built by `make hostsan` in vfs_amd/csrc with the sanitizers on the host side only).  The program checks itself --
staged inserts and their commits, growth across storage segments, id bookkeeping, read-back, the log parser on the
reference's own records, exact path == batch path, the entry point that allocates nothing, a world-1 communicator --
and the sanitizers abort it on any finding."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_ubsan_on_the_gpu():
    exe = os.path.join(ROOT, "vfs_amd", "lib", "host_san")
    assert os.path.exists(exe), "run `make hostsan` in vfs_amd/csrc (or __graft_entry__.build())"
    env = dict(os.environ)
    # leak checking off: the HIP runtime keeps its own allocations until exit; link-order check off: the harness
    # preloads a library of its own ahead of every program
    env["ASAN_OPTIONS"] = "detect_leaks=0:verify_asan_link_order=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "vectors_head.dat")], capture_output=True,
                       text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " 0 failures" in r.stdout, out[-4000:]
