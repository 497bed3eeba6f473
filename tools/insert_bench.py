#!/usr/bin/env python3
"""Online-insert rate of the F3 path: what one POST /vectors costs on the engine side.
ServerState.register = one-row vfsx_corpus_append[_quantize] (pinned stage -> H2D -> rb64 transpose / quantize
kernel -> row norm -> stream sync) + the memtable / flush bookkeeping of storage_manager.rs:139-157,306-335
(every 10th insert appends 10 records to data/vectors.dat).  The reference's own figure for the same route is
~1-2 ms per POST with an ~80-95 ms spike on every 10th.  Run on the GPU box."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd import server

rng = np.random.default_rng(0)
for d, quant in ((16, True), (768, False), (768, True)):
    with tempfile.TemporaryDirectory() as tmp:
        st = server.ServerState(d, quant, "bench", state_dir=os.path.join(tmp, "state"), data_dir=os.path.join(tmp, "data"))
        os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
        rows = rng.uniform(-1, 1, (2100, d)).astype(np.float32)
        for v in rows[:100]:
            st.register(v, "w", ["warm"])
        lat = []
        t0 = time.perf_counter()
        for i, v in enumerate(rows[100:]):
            t = time.perf_counter()
            st.register(v, f"Vec_{i}", ["benchmark"])
            lat.append(time.perf_counter() - t)
        dt = time.perf_counter() - t0
        lat = np.array(lat) * 1e6
        flush = lat[9::10]  # every 10th insert flushes the memtable to the log
        plain = np.delete(lat, np.arange(9, len(lat), 10))
        print(f"d={d} {'int8 (device quantize)' if quant else 'f32'} store: {len(lat) / dt:.0f} inserts/s; "
              f"median {np.median(plain):.0f} us per insert, {np.median(flush):.0f} us on the flushing 10th "
              f"(searchable at once; {len(st.corpus)} rows resident)", flush=True)
        st.corpus.close()
