"""Randomized check of the single-query route through the mirror (f32 stores of >= 2.5e8 elements, k <= 32): synthetic
corpora of random shape plus appended near-duplicates / scaled copies of the queries and duplicate rows, every query
asked one per call with the path on and compared bit for bit with the same call with the path off.  GPU box.
python tools/fuzz_mirror_single.py [n_cases]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import vfs_amd
from vfs_amd import synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(2026)
t0 = time.time()
total_q = total_fb = 0
for case in range(n_cases):
    dim = int(rng.choice([48, 100, 128, 257, 768, 1024]))
    rows = int(rng.uniform(2.6e8, 5e8) / dim)
    mirror = str(rng.choice(["auto", "int8", "bf16"]))
    if mirror == "bf16":
        rows = int(rng.uniform(4.1e8, 6e8) / dim)
    c = vfs_amd.Corpus(dim, capacity_rows=rows + 4096)
    c.fill_synthetic(int(rng.integers(1, 1 << 30)), rows, row0=0, id0=1)
    queries = synth.rows_f32(int(rng.integers(1, 1 << 30)), 0, 24, dim)
    extra = []
    for q in queries[:8]:  # near-duplicates at several distances, scaled copies, exact duplicates of each other
        extra.append(q + rng.normal(0, 1e-3, dim).astype(np.float32))
        extra.append((q * rng.uniform(0.1, 10)).astype(np.float32))
        extra.append(q + rng.normal(0, 3e-2, dim).astype(np.float32))
    extra += [extra[0], extra[0], extra[4]]
    c.append(np.stack(extra).astype(np.float32))
    for metric in (1, 0):
        for k in (1, 10, 32):
            c.enable_mfma(False)
            ref = [c.search_batch(queries[i:i + 1], k, metric) for i in range(len(queries))]
            c.enable_mfma(True, mirror=mirror)
            c.mfma_stats(reset=True)
            for i in range(len(queries)):
                ids, dist, cnt = c.search_batch(queries[i:i + 1], k, metric)
                assert (ids == ref[i][0]).all(), (case, dim, rows, mirror, metric, k, i)
                assert (dist.view(np.uint32) == ref[i][1].view(np.uint32)).all(), (case, dim, rows, mirror, metric, k, i)
                assert (cnt == ref[i][2]).all()
            mq, fb = c.mfma_stats()
            total_q += mq
            total_fb += fb
    print(f"case {case}: {rows} x {dim}, mirror {mirror} -> {c.mfma_mirror()[0]}: ok ({time.time() - t0:.0f} s)", flush=True)
    c.close()
print(f"{n_cases} cases ok: {total_q} queries through the path, {total_fb} of them re-run by the exact scan")
