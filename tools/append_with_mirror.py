"""Online inserts with the matrix-core path on: ms per (append one row + 4-query batch) on a 1M x 768 corpus, f32
store (int8 mirror / bf16 mirror) and int8 store.  GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import vfs_amd
from vfs_amd import synth

n, d = 1_000_000, 768
for quantize, mirror in ((False, "int8"), (False, "bf16"), (True, "auto")):
    c = vfs_amd.Corpus(d, quantize=quantize, capacity_rows=n + 4096)
    c.fill_synthetic(synth.CORPUS_SEED, n, row0=0, id0=1)
    c.enable_mfma(True, mirror=mirror)
    q = synth.rows_f32(synth.QUERY_SEED, 0, 4, d)
    row = synth.rows_f32(synth.QUERY_SEED, 100, 1, d)
    c.search_batch(q, 10, 1)
    t0 = time.perf_counter()
    for _ in range(50):
        c.search_batch(q, 10, 1)
    base = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        (c.append_f32 if quantize else c.append)(row)
        c.search_batch(q, 10, 1)
    both = (time.perf_counter() - t0) / 50
    print(f"{'int8 store' if quantize else 'f32 store '} operand {c.mfma_mirror()[0]:10s}: batch alone {base * 1e3:.3f} ms, "
          f"append + batch {both * 1e3:.3f} ms", flush=True)
    c.close()
