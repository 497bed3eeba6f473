"""Fallback rate of the matrix-core batch path by operand kind (GPU box): 512 random queries against synthetic
corpora -- uniform rows, rows with norms over three decades, Gaussian rows -- through every operand an f32 store can
have (int8 mirror, bf16 mirror, the library's automatic choice) and through an int8 store.  Prints the queries sent
back to the exact scan; results are identical either way (tests/test_gpu_mfma.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfs_amd

rng = np.random.default_rng(1)
shapes = [(40000, 128, "uniform"), (200000, 768, "uniform"), (60000, 96, "uniform, norms over 3 decades"),
          (1000000, 256, "uniform"), (1000000, 256, "uniform, norms over 3 decades"), (300000, 512, "gaussian")]
for n, d, what in shapes:
    rows = (rng.normal(0, 1, (n, d)) if what == "gaussian" else rng.uniform(-1, 1, (n, d))).astype(np.float32)
    if "decades" in what:
        rows = (rows * np.exp(rng.uniform(np.log(0.01), np.log(10.0), (n, 1)))).astype(np.float32)
    q = (rng.normal(0, 1, (512, d)) if what == "gaussian" else rng.uniform(-1, 1, (512, d))).astype(np.float32)
    for quantize, mirror in ((False, "int8"), (False, "bf16"), (False, "auto"), (True, "auto")):
        c = vfs_amd.Corpus(d, quantize=quantize)
        if quantize:
            c.append_f32((rows / np.abs(rows).max()).astype(np.float32))  # quantized at ingest (scale 127)
        else:
            c.append(rows)
        c.enable_mfma(True, mirror=mirror)
        kind, rx = c.mfma_mirror()
        out = []
        for metric in (1, 0):
            c.mfma_stats(reset=True)
            c.search_batch(q, 10, metric)
            out.append(c.mfma_stats())
        print(f"{n}x{d} {what:30s} {'int8 store' if quantize else 'f32 store '} mirror {mirror:4s} -> {kind:10s} R_x {rx:.4f}  "
              f"cosine (queries, fallbacks) {out[0]}  euclidean {out[1]}  operand afterwards: {c.mfma_mirror()[0]}", flush=True)
        c.close()
