"""Loader for tests/golden/ranker_golden.json (see tests/golden/make_golden.py)."""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ranker_golden.json")


def unhex(lst):
    return np.array([int(x, 16) for x in lst], dtype=np.uint32).view(np.float32)


def load():
    with open(_PATH) as f:
        g = json.load(f)
    corpora = {}
    for key, c in g["corpora"].items():
        if c["quantized"]:
            rows = np.array(c["data"], dtype=np.int8).reshape(c["n_rows"], c["dim"])
        else:
            rows = unhex(c["data"]).reshape(c["n_rows"], c["dim"])
        corpora[key] = (rows, c["quantized"])
    cases = []
    for c in g["cases"]:
        rows, quantized = corpora[c["corpus"]]
        cases.append({
            "name": c["name"], "metric": c["metric"], "limit": c["limit"], "batch": c["batch"],
            "rows": rows, "quantized": quantized, "query": unhex(c["query"]),
            "ids": None if c["ids"] is None else np.array(c["ids"], dtype=np.uint64),
            "expect_ids": np.array(c["expect_ids"], dtype=np.uint64), "expect_dist": unhex(c["expect_dist"]),
        })
    return cases, g["quantize"]
