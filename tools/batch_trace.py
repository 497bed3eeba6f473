"""One batch size through the matrix-core path under rocprofv3 --kernel-trace: python tools/batch_trace.py <nq> [quantize]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vfs_amd
from vfs_amd import synth

nq = int(sys.argv[1])
K = int(os.environ.get("BT_K", "10"))
quantize = len(sys.argv) > 2 and sys.argv[2] == "1"
rows, dim = int(os.environ.get("BT_ROWS", "1000000")), int(os.environ.get("BT_DIM", "768"))
dev = torch.device("cuda:0")
c = vfs_amd.Corpus(dim, quantize=quantize, capacity_rows=rows, device=0)
c.fill_synthetic(synth.CORPUS_SEED, rows, row0=0, id0=1)
q = torch.from_numpy(synth.rows_f32(synth.QUERY_SEED, 0, nq, dim)).to(dev)
out = torch.zeros((nq, K, 6), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
c.enable_mfma(True)
for _ in range(12):
    c.search_device(q.data_ptr(), nq, K, 1, out.data_ptr(), stream)
torch.cuda.synchronize()
print(c.mfma_stats())
