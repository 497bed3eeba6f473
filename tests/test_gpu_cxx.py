"""GPU: the C++ Ranker mirror (vfs_amd/cxx/vfsx_ranker.hpp) through the C ABI."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_ranker_selftest():
    exe = os.path.join(ROOT, "vfs_amd", "lib", "ranker_selftest")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("ok id=1 dist=3.74165")


def test_cxx_ranker_parity_against_oracle():
    """tests/cxx/ranker_parity.cpp: the C++ Ranker mirror vs the linked C oracle -- every method, dense
    and quantized stores, limits 1..100, ties, the integer-domain extension, the literal batch loop."""
    exe = os.path.join(ROOT, "tests", "cxx", "ranker_parity")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.strip().splitlines()[-1].startswith("ok ")


def test_cxx_sharded_search_end_to_end():
    """tests/cxx/sharded_parity.cpp: SURVEY 8(e) from a C++ host -- two shards on one GPU merged on the device, and
    the world-1 RCCL communicator (vfsx_comm_*, vfsx_search_sharded[_host], vfsx_comm_merge_hits) -- vs the oracle."""
    exe = os.path.join(ROOT, "tests", "cxx", "sharded_parity")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.strip().splitlines()[-1].startswith("ok ")


def test_wave_topk_primitives():
    """tests/cxx/wave_topk_check.hip: the lane exchanges, the 64-key bitonic sort and the register lists' bulk paths
    (offer_dense / merge_list) of vfsx_device.h against a host model and against one-at-a-time inserts."""
    exe = os.path.join(ROOT, "tests", "cxx", "wave_topk_check")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.strip().splitlines()[-1].startswith("ok ")
