"""CPU: the host-only translation unit (vfs_amd/csrc/vfsx_host.cpp: log walker + bincode decoder, host merge,
error plumbing) built by g++ under AddressSanitizer + UndefinedBehaviorSanitizer, driven by tests/cxx/log_fuzz.cpp
with truncated / oversized / garbage / byte-mutated variants of the reference's own first records
(tests/golden/vectors_head.dat).  Expect VFSX_OK (a clean short count, serializer.rs:137-146) or
VFSX_ERR_SERIALIZATION / VFSX_ERR_IO and no sanitizer report (-fno-sanitize-recover=all aborts on one)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vfs_amd", "csrc")
DRIVER = os.path.join(ROOT, "vfs_amd", "lib", "log_fuzz_san")


def test_log_parser_survives_hostile_files_under_asan_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", CSRC, "san"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)  # ASan must be the first library of the process it instruments
    r = subprocess.run([DRIVER, os.path.join(ROOT, "tests", "golden", "vectors_head.dat"), str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["failed_expectations"] == 0 and out["cases_other"] == 0 and out["golden_records"] == 8
    assert out["cases_serialization_error"] > 1000 and out["cases_short"] > 1000 and out["cases_ok"] > 1000


def test_host_translation_unit_has_no_hip_dependency():
    """vfsx_host.cpp / vfsx_host.h must stay compilable by plain g++ (that is what makes the sanitizer build possible)."""
    for name in ("vfsx_host.cpp", "vfsx_host.h"):
        text = open(os.path.join(CSRC, name)).read()
        assert "hip_runtime" not in text and "#include <hip" not in text, name
