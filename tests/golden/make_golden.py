#!/usr/bin/env python3
"""Generates tests/golden/ranker_golden.json.

The reference holds no golden vectors for this path and cannot be run
here (Rust nightly; no toolchain), so these fixtures are NOT outputs of the reference:
they are outputs of the two independent CPU restatements (oracle/vfs_oracle.c and
oracle/vfs_oracle_np.py), committed only when both agree bit-for-bit, plus KATs derived
by hand from the cited reference lines.  Parity stays "unpinned" in the task's sense.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co  # noqa: E402
from oracle import vfs_oracle_np as on  # noqa: E402


def hexf(a):
    return [format(int(x), "08x") for x in np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()]


CORPORA = {}


def corpus_key(rows, quantized):
    data = np.asarray(rows).astype(int).ravel().tolist() if quantized else hexf(rows)
    for key, val in CORPORA.items():
        if val["data"] == data and val["dim"] == int(rows.shape[1]):
            return key
    key = f"c{len(CORPORA)}"
    CORPORA[key] = {"quantized": quantized, "dim": int(rows.shape[1]), "n_rows": int(rows.shape[0]), "data": data}
    return key


def case(name, rows, query, metric, limit, batch, quantized=False, ids=None):
    rows_f = co.dequantize(rows, 127.0) if quantized else np.asarray(rows, np.float32)
    i1, d1 = co.exact_search(rows_f, query, metric, batch, limit, ids=ids)
    i2, d2 = on.exact_search(rows_f, query, metric, batch, limit, ids=ids)
    i3, d3, _ = co.topk_scan(rows if quantized else rows_f, query, metric, limit, ids=ids)
    assert (i1 == i2).all() and d1.tobytes() == d2.tobytes(), name
    assert (i1 == i3).all() and d1.tobytes() == d3.tobytes(), name
    return {
        "name": name, "metric": metric, "limit": limit, "batch": batch,
        "corpus": corpus_key(rows, quantized), "query": hexf(query),
        "ids": None if ids is None else [int(x) for x in ids],
        "expect_ids": [int(x) for x in i1], "expect_dist": hexf(d1),
    }


def main():
    rng = np.random.default_rng(20250418)
    cases = []
    # KAT 1 (tests.sh:81-104 data): quantize([1,2,3,4]) saturates to 127 -> dequant [1,1,1,1];
    # Euclidean vs [1,2,3,4] = sqrt(0+1+4+9) = sqrt(14)
    r = co.quantize(np.array([[1, 2, 3, 4]], np.float32), 127.0)
    assert r.tolist() == [[127, 127, 127, 127]]
    c = case("kat_tests_sh", r, np.array([1, 2, 3, 4], np.float32), 0, 6, 2, quantized=True)
    assert c["expect_dist"] == hexf(np.sqrt(np.float32(14.0)))
    cases.append(c)
    # ties: equal distances come back in scan order; k > N returns N
    rows = np.array([[1, 0], [0, 1], [-1, 0], [0, -1], [1, 0], [2, 2]], np.float32)
    cases.append(case("ties_scan_order", rows, np.zeros(2, np.float32), 0, 5, 3))
    cases.append(case("k_gt_n", rows, np.array([0.5, 0.25], np.float32), 0, 50, 4))
    cases.append(case("explicit_ids", rows, np.array([0.5, 0.25], np.float32), 1, 3, 1,
                      ids=np.array([90, 80, 70, 60, 50, 40], np.uint64)))
    # cosine of identical vectors: 1 - dot/(n*n) in sequential-order bits
    v = rng.uniform(-1, 1, (1, 37)).astype(np.float32)
    cases.append(case("cosine_identical", v, v[0], 1, 5, 1))
    # Simd* == scalar on the widths the reference supports
    for d in (2, 4, 8, 16, 32, 64):
        rows = rng.uniform(-1, 1, (12, d)).astype(np.float32)
        q = rng.uniform(-1, 1, d).astype(np.float32)
        a = case(f"simd_euclid_d{d}", rows, q, 2, 5, 5)
        b = case(f"euclid_d{d}", rows, q, 0, 5, 5)
        assert a["expect_ids"] == b["expect_ids"] and a["expect_dist"] == b["expect_dist"]
        cases.append(a)
        a = case(f"simd_cosine_d{d}", rows, q, 3, 5, 5)
        b = case(f"cosine_d{d}", rows, q, 1, 5, 5)
        assert a["expect_ids"] == b["expect_ids"] and a["expect_dist"] == b["expect_dist"]
        cases.append(a)
    # batch-size independence (A4) + random dense / quantized, ragged dims
    for d, n in ((3, 70), (16, 80), (100, 24), (128, 20)):
        rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
        q = rng.uniform(-1, 1, d).astype(np.float32)
        for metric in (0, 1):
            ref = None
            for batch in (1, 5, n):
                c = case(f"dense_d{d}_m{metric}_b{batch}", rows, q, metric, 10, batch)
                if ref is None:
                    ref = c
                    cases.append(c)
                assert c["expect_ids"] == ref["expect_ids"] and c["expect_dist"] == ref["expect_dist"]
    # saturated quantized corpus like the reference's benchmark data (benchmarks.py:33-34)
    for d, n in ((16, 400), (24, 130)):
        rows_f = rng.uniform(-10, 10, (n, d)).round(4).astype(np.float32)
        rows_f[::7] = rng.uniform(-1, 1, (len(rows_f[::7]), d)).astype(np.float32)
        rows_q = co.quantize(rows_f, 127.0)
        assert (rows_q == on.quantize(rows_f, 127.0)).all()
        q = rng.uniform(-1, 1, d).astype(np.float32)
        for metric in (0, 1):
            cases.append(case(f"quant_sat_d{d}_m{metric}", rows_q, q, metric, 10, 5, quantized=True))
            cases.append(case(f"quant_sat_d{d}_m{metric}_k100", rows_q, q, metric, 100, 5, quantized=True))
    quant_kat = {
        "in": hexf(np.array([0.5, -0.5, np.nan, -2.0, 2.0, 63.5 / 127, -63.5 / 127, 1.0, 0.0039370079], np.float32)),
        "some127": co.quantize(np.array([0.5, -0.5, np.nan, -2.0, 2.0, 63.5 / 127, -63.5 / 127, 1.0, 0.0039370079],
                                        np.float32), 127.0).astype(int).tolist(),
        "none": co.quantize(np.array([0.5, -0.5, np.nan, -2.0, 2.0, 63.5 / 127, -63.5 / 127, 1.0, 0.0039370079],
                                     np.float32), None).astype(int).tolist(),
        "dequant_all_127": hexf(co.dequantize(np.arange(-128, 128).astype(np.int8), 127.0)),
    }
    assert quant_kat["some127"][:5] == [64, -64, 0, -128, 127]  # half away from zero; NaN->0; saturation
    assert quant_kat["dequant_all_127"] == hexf(on.dequantize(np.arange(-128, 128).astype(np.int8), 127.0))
    out = {"note": "outputs of the build's two CPU restatements (not of the reference); see make_golden.py",
           "corpora": CORPORA, "cases": cases, "quantize": quant_kat}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ranker_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
