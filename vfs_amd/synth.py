"""Host twin of the device synthetic generator (vfsx_corpus_fill_synthetic): the
counter-based stream element(i, j) = splitmix64(seed + i*dim + j) -> top 24 bits ->
m * 2^-23 - 1, uniform in [-1, 1) and exact in f32.  Benchmark/test utility; not part
of the reference."""
from __future__ import annotations

import numpy as np

CORPUS_SEED = 0x5EED0001
QUERY_SEED = 0x5EED0002


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rows_f32(seed: int, row0: int, n: int, dim: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        ctr = (np.uint64(seed) + (np.uint64(row0) + np.arange(n, dtype=np.uint64))[:, None] * np.uint64(dim)
               + np.arange(dim, dtype=np.uint64)[None, :])
    m = (_splitmix64(ctr) >> np.uint64(40)).astype(np.uint32).astype(np.float32)
    return m * np.float32(1.0 / 8388608.0) - np.float32(1.0)
