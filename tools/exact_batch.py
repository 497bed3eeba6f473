"""Exact multi-query scan (matrix-core path off): ms per batch of 2 / 4 / 8 / 16 queries on synthetic corpora, k = 10
and 64.  usage: exact_batch.py [rows dim]   (GPU box; VFSX_LIB_PATH picks an A/B build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vfs_amd
from vfs_amd import synth

shapes = [(100_000, 128), (1_000_000, 768)]
if len(sys.argv) == 3:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
dev = torch.device("cuda:0")
for rows, dim in shapes:
    for quantize in (False, True):
        c = vfs_amd.Corpus(dim, quantize=quantize, capacity_rows=rows, device=0)
        c.fill_synthetic(synth.CORPUS_SEED, rows, row0=0, id0=1)
        q = torch.from_numpy(synth.rows_f32(synth.QUERY_SEED, 0, 64, dim)).to(dev)
        out = torch.zeros((64, 64, 6), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for metric in (1, 0):
            for k in (10, 64):
                line = []
                for nq in (2, 4, 8, 16):
                    for _ in range(3):
                        c.search_device(q.data_ptr(), nq, k, metric, out.data_ptr(), stream)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        c.search_device(q.data_ptr(), nq, k, metric, out.data_ptr(), stream)
                    e1.record()
                    torch.cuda.synchronize()
                    line.append(f"nq {nq}: {e0.elapsed_time(e1) / 20:7.4f}")
                print(f"{rows}x{dim} {'i8 ' if quantize else 'f32'} metric {metric} k {k:3d}  " + "  ".join(line) + "  ms", flush=True)
        c.close()
