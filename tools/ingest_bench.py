#!/usr/bin/env python3
"""Ingest throughput of the pinned, double-buffered H2D pipeline (vfsx_corpus_append*)
and of the log loader (vfsx_corpus_load_log).  Run on the GPU box: python tools/ingest_bench.py"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfs_amd
from vfs_amd import vfslog

n, d = 1_000_000, 768
rows = np.random.default_rng(0).uniform(-1, 1, (n, d)).astype(np.float32)
for label, quant, fn in (("append f32 -> f32 corpus", False, "append"),
                         ("append_quantize f32 -> i8 corpus (device-side quantize)", True, "append_f32")):
    c = vfs_amd.Corpus(d, quantize=quant, capacity_rows=n)
    getattr(c, fn)(rows[:1000])  # staging buffers, first-touch
    t0 = time.perf_counter()
    getattr(c, fn)(rows[1000:])
    dt = time.perf_counter() - t0
    gb = rows[1000:].nbytes / 1e9
    print(f"{label}: {gb:.2f} GB of host rows in {dt*1e3:.0f} ms = {gb/dt:.1f} GB/s, {(n-1000)/dt/1e6:.2f} M rows/s",
          flush=True)
    c.close()

# the reference's log format: 100k dense records of d=768 (3.1 KB each)
m = 100_000
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "vectors.dat")
    ts = vfslog.utc_now_rfc3339()
    with open(path, "wb") as f:
        for i in range(m):
            f.write(vfslog.encode_record(i + 1, rows[i], f"Vec_{i}", ["benchmark"], ts))
    size = os.path.getsize(path)
    t0 = time.perf_counter()
    c = vfslog.open_corpus(path)
    dt = time.perf_counter() - t0
    print(f"load_log: {m} records, {size/1e6:.0f} MB in {dt*1e3:.0f} ms = {size/dt/1e9:.2f} GB/s, {m/dt/1e6:.2f} M records/s",
          flush=True)
    c.close()
