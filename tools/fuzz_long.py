"""Long randomized parity run (not part of the pytest suite): the cases of tests/test_gpu_fuzz.py with many
more seeds, GPU results vs the oracle bit for bit.  python tools/fuzz_long.py [first_seed n_seeds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gpu_fuzz as tf
from oracle import c_oracle

first, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) == 3 else (100, 100)
t0 = time.time()
for seed in range(first, first + n):
    tf.test_random_cases_match_the_oracle(c_oracle, seed)
    if seed % 4 == 0:
        tf.test_random_mfma_batches_equal_the_exact_scan(seed)
    if seed % 10 == 0:
        print(f"seed {seed} ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{n} seeds x 12 cases ok")


def big_case(rng, tag):
    """larger shapes than the pytest sweep: up to 400k rows (wide int8 kernel, balanced grids), dims to 2048,
    k to 1024, batches to 40 queries"""
    import vfs_amd
    d = int(rng.choice([4, 16, 33, 128, 257, 768, 1024, 2048]))
    n = int(rng.integers(1, max(2, min(400_000, 40_000_000 // d))))
    quantize = bool(rng.integers(0, 2))
    rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    if n > 100:
        rows[rng.integers(0, n, n // 50)] = rows[rng.integers(0, n, n // 50)]
    c = vfs_amd.Corpus(d, quantize=quantize)
    (c.append_f32 if quantize else c.append)(rows)
    ref_rows = c_oracle.quantize(rows, 127.0) if quantize else rows
    nq = int(rng.choice([1, 1, 2, 5, 8, 17, 40]))
    queries = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    k = int(rng.choice([1, 10, 10, 64, 100, 1024]))
    metric = int(rng.choice([0, 1, 2, 3] + ([4] if quantize else [])))
    if metric in (1, 3):
        zero = (np.abs(ref_rows.astype(np.int32)).sum(axis=1) == 0) if quantize else (np.abs(ref_rows).sum(axis=1) == 0)
        if zero.any():
            c.close()
            return
    mfma = bool(rng.integers(0, 2))
    c.enable_mfma(mfma)
    gi, gd, gc = c.search_batch(queries, k, metric)
    for qi in range(0, nq, max(1, nq // 4)):
        oi, od, _ = c_oracle.topk_scan(ref_rows, queries[qi], metric, k, scale=127.0)
        m = int(gc[qi])
        what = (tag, n, d, quantize, metric, k, nq, qi, mfma)
        assert m == len(oi) == min(k, n), what
        assert (gi[qi, :m] == oi).all(), what
        assert (tf._bits(gd[qi, :m]) == tf._bits(od)).all(), what
    c.close()


rng = np.random.default_rng(first * 7919 + 1)
n_big = max(10, n // 2)
for i in range(n_big):
    big_case(rng, i)
    if i % 10 == 0:
        print(f"big case {i} ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{n_big} big cases ok")
