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
for d, quant in ((16, True), (16, True), (768, False), (768, True)):  # (the first case warms the process up: not printed)
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
        first_case = "warm" not in globals()
        warm = True
        if not first_case:
            print(f"d={d} {'int8 (device quantize)' if quant else 'f32'} store: {len(lat) / dt:.0f} inserts/s; "
                  f"median {np.median(plain):.0f} us per insert, {np.median(flush):.0f} us on the flushing 10th, "
                  f"p99 {np.percentile(lat, 99):.0f} us, max {lat.max():.0f} us "
                  f"(searchable at once; {len(st.corpus)} rows resident)", flush=True)
        st.corpus.close()

# the engine side alone: one-row vfsx_corpus_append[_quantize] calls through ctypes, every append committed at once
# (round 2's behaviour) against appends staged on the host and committed 4096 at a time (vfsx_corpus_set_insert_batch)
import vfs_amd
for d, quant in ((16, True), (768, False), (768, True)):
    rows = rng.uniform(-1, 1, (20000, d)).astype(np.float32)
    for batch in (0, 4096):
        c = vfs_amd.Corpus(d, quantize=quant)
        c.set_insert_batch(batch)
        for v in rows[:200]:
            c.append_f32(v[None, :])
        t0 = time.perf_counter()
        for v in rows[200:]:
            c.append_f32(v[None, :])
        c.commit()
        dt = time.perf_counter() - t0
        ids, dist, cnt = c.search_batch(rows[-1] if not quant else rows[-1], 1, 0)
        print(f"C ABI, d={d} {'int8' if quant else 'f32'}, insert batch {batch}: {(len(rows) - 200) / dt:.0f} one-row appends/s "
              f"({len(c)} rows resident, last row found: id {int(ids[0, 0])})", flush=True)
        c.close()
