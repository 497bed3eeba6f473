"""One query per call on f32 stores of growing size: the exact scan against the matrix-core path over the int8 / bf16
mirror (VFSX_MFMA_MIN_BATCH=1 in the environment forces the path).  ms per query, k = 10.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vfs_amd
from vfs_amd import synth

dev = torch.device("cuda:0")
for dim in (128, 768):
    for elems in (5e7, 1e8, 1.5e8, 2e8, 3e8, 4e8, 6e8):
        rows = int(elems / dim)
        c = vfs_amd.Corpus(dim, capacity_rows=rows, device=0)
        c.fill_synthetic(synth.CORPUS_SEED, rows, row0=0, id0=1)
        q = torch.from_numpy(synth.rows_f32(synth.QUERY_SEED, 0, 64, dim)).to(dev)
        out = torch.zeros((1, 10, 6), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        res = []
        for mode in ("off", "int8", "bf16"):
            if mode == "off":
                c.enable_mfma(False)
            else:
                c.enable_mfma(True, mirror=mode)
            for metric in (1, 0):
                for j in range(8):
                    c.search_device(q[j].data_ptr(), 1, 10, metric, out.data_ptr(), stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for j in range(64):
                    c.search_device(q[j].data_ptr(), 1, 10, metric, out.data_ptr(), stream)
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 64)
        print(f"{rows:8d} x {dim:4d} ({elems:.1e} elements)  exact cos {res[0]:.3f} euc {res[1]:.3f}   int8 mirror "
              f"cos {res[2]:.3f} euc {res[3]:.3f}   bf16 mirror cos {res[4]:.3f} euc {res[5]:.3f}  ms", flush=True)
        c.close()
