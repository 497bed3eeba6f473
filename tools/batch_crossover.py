"""Where does the matrix-core batch path overtake the exact 8-queries-per-pass scan?  Times
vfsx_search_device for batches of 2..512 queries with the MFMA path off and forced on
(VFSX_MFMA_MIN_BATCH=1 in the environment), on synthetic corpora resident in HBM.
Run on the GPU box:  VFSX_MFMA_MIN_BATCH=1 python tools/batch_crossover.py [rows dim]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vfs_amd
from vfs_amd import synth

shapes = [(1_000_000, 768), (10_000_000, 128), (4_000_000, 1024)]
if len(sys.argv) == 3:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
dev = torch.device("cuda:0")
for rows, dim in shapes:
    for quantize in (False, True):
        c = vfs_amd.Corpus(dim, quantize=quantize, capacity_rows=rows, device=0)
        c.fill_synthetic(synth.CORPUS_SEED, rows, row0=0, id0=1)
        q = torch.from_numpy(synth.rows_f32(synth.QUERY_SEED, 0, 512, dim)).to(dev)
        out = torch.zeros((512, 10, 6), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for metric in (1, 0):
            for nq in (2, 4, 8, 16, 32, 64, 128, 256, 512):
                ms = []
                for on in (False, True):
                    c.enable_mfma(on)
                    c.mfma_stats(reset=True)
                    for _ in range(2):
                        c.search_device(q.data_ptr(), nq, 10, metric, out.data_ptr(), stream)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        c.search_device(q.data_ptr(), nq, 10, metric, out.data_ptr(), stream)
                    e1.record()
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1) / 3)
                    st = c.mfma_stats()
                print(f"{rows}x{dim} {'i8' if quantize else 'f32'} metric {metric} nq {nq:4d}: exact {ms[0]:8.3f} ms  mfma {ms[1]:8.3f} ms"
                      f"  (mfma queries/fallbacks {st})", flush=True)
        c.close()
