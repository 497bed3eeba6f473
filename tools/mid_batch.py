"""Matrix-core path, mid-size batches (129 .. 2048 queries) on a C2-size store: ms per batch and q/s, with the
geometry's strip choice and with VFSX_MFMA_STRIP forced (A/B of the whole-rounds rule).  Run on the GPU box."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch

    import vfs_amd
    from vfs_amd import synth
    rows, dim = 1_000_000, 768
    dev = torch.device("cuda:0")
    for quantize in (False, True):
        c = vfs_amd.Corpus(dim, quantize=quantize, capacity_rows=rows, device=0)
        c.fill_synthetic(synth.CORPUS_SEED, rows, row0=0, id0=1)
        q = torch.from_numpy(synth.rows_f32(synth.QUERY_SEED, 0, 2048, dim)).to(dev)
        out = torch.zeros((2048, 10, 6), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        c.enable_mfma(True)
        for nq in (64, 128, 200, 256, 512, 1024, 2048):
            for _ in range(2):
                c.search_device(q.data_ptr(), nq, 10, 1, out.data_ptr(), stream)
            torch.cuda.synchronize()
            c.mfma_stats(reset=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                c.search_device(q.data_ptr(), nq, 10, 1, out.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            mq, fb = c.mfma_stats()
            print(f"  {'int8 store' if quantize else 'f32 store '} nq {nq:5d}: {ms:7.3f} ms per batch, {nq / ms * 1e3:9.0f} q/s, "
                  f"{fb} of {mq} re-run by the exact scan", flush=True)
        c.close()
else:
    for strip in ("", "3"):
        env = dict(os.environ)
        if strip:
            env["VFSX_MFMA_STRIP"] = strip
        print(f"VFSX_MFMA_STRIP={strip or '(geometry)'}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
