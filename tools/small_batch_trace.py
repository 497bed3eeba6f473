"""One small batch through the matrix-core path, repeated, for a rocprofv3 --kernel-trace run:
usage: small_batch_trace.py rows dim nq quantize(0|1) metric"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vfs_amd
from vfs_amd import synth

rows, dim, nq, quantize, metric = (int(x) for x in sys.argv[1:6])
dev = torch.device("cuda:0")
c = vfs_amd.Corpus(dim, quantize=bool(quantize), capacity_rows=rows, device=0)
c.fill_synthetic(synth.CORPUS_SEED, rows, row0=0, id0=1)
c.enable_mfma(True)
q = torch.from_numpy(synth.rows_f32(synth.QUERY_SEED, 0, nq, dim)).to(dev)
out = torch.zeros((nq, 10, 6), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    c.search_device(q.data_ptr(), nq, 10, metric, out.data_ptr(), stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    c.search_device(q.data_ptr(), nq, 10, metric, out.data_ptr(), stream)
e1.record()
torch.cuda.synchronize()
print(f"{rows}x{dim} quantize={quantize} metric={metric} nq={nq}: {e0.elapsed_time(e1) / 50:.4f} ms per batch", c.mfma_stats())
