"""Diagnostic (GPU box): the 256 x 256 prefilter's own launch time (HIP events around the kernel) for a build
selected with VFSX_LIB_PATH / VFSX_MFMA_PF -- also for timing-only experiment builds whose results are wrong
(every query then falls back to the exact scan, which the events do not cover).
usage: python tools/pf_time.py [rows] [dim] [nq] [int8|bf16|auto]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfs_amd
from vfs_amd import synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
kind = sys.argv[4] if len(sys.argv) > 4 else "bf16"
c = vfs_amd.Corpus(dim, quantize=(kind == "int8"), capacity_rows=rows)
c.fill_synthetic(synth.CORPUS_SEED, rows)
c.enable_mfma(True, mirror="auto" if kind == "int8" else kind)
q = synth.rows_f32(synth.QUERY_SEED, 0, nq, dim)
c.set_profiling(True)
times = []
for rep in range(int(os.environ.get("PF_REPS", "6"))):
    c.search_batch(q, 10, 1)
    ms, n = c.scan_time_ms()
    if n:
        times.append(ms / n)
mq, mf = c.mfma_stats()
flop = 2.0 * nq * rows * dim
best = min(times)
print(f"{os.environ.get('VFSX_LIB_PATH', 'default')} pf={os.environ.get('VFSX_MFMA_PF', 'lds')} {kind} {rows}x{dim} nq={nq}: "
      f"prefilter ms {['%.3f' % t for t in times]}  best {best:.3f} ms = {flop / best / 1e9:.0f} T(FL)OP/s  "
      f"fallbacks {mf}/{mq}", flush=True)
c.close()
