"""Diagnostic (GPU box): per-K-step cycle split of mfma_prefilter256_kernel from a -DVFSX_EXP_STAMPS build.
usage: VFSX_LIB_PATH=exp/stamps/libvfsx.so [VFSX_MFMA_PF=rv] python tools/exp_stamps.py [rows] [dim] [nq] [int8]
(the rows-in-registers kernel mfma_prefilter_rv_kernel carries the same stamps; the last figure is the in-kernel clock:
s_memtime cycles per s_memrealtime tick of 10 ns)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vfs_amd
from vfs_amd import synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
i8 = len(sys.argv) > 4 and sys.argv[4] == "int8"
lib = vfs_amd.load()
c = vfs_amd.Corpus(dim, quantize=i8, capacity_rows=rows)
c.fill_synthetic(synth.CORPUS_SEED, rows)
c.enable_mfma(True)
q = synth.rows_f32(synth.QUERY_SEED, 0, nq, dim)
out = (ctypes.c_ulonglong * 8)()
for rep in range(3):
    c.search_batch(q, 10, 1)
    lib.vfsx_exp_stamps.restype = ctypes.c_int
    lib.vfsx_exp_stamps(out, 1)
    vm, bar, body, fold, n, total, waves = [int(x) for x in out[:7]]
    rt = int(out[7])
    print(f"rep {rep}: waves {waves} ksteps/wave {n / waves:.0f}  per K-step cycles: vmcnt-wait {vm / n:.0f}  barrier {bar / n:.0f}  "
          f"body {body / n:.0f}  fold(per K-step avg) {fold / n:.0f}  wave total/kstep {total / n:.0f}  "
          f"clock {total / max(rt, 1) * 0.1:.3f} GHz", flush=True)
c.close()
