import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfs_amd
rng = np.random.default_rng(1)
for (n, d, scale_rows) in [(40000,128,False),(200000,768,False),(60000,96,True),(1000000,256,False),(1000000,256,True)]:
    rows = rng.uniform(-1,1,(n,d)).astype(np.float32)
    if scale_rows:
        rows = (rows*np.exp(rng.uniform(np.log(0.01),np.log(10.0),(n,1)))).astype(np.float32)
    q = rng.uniform(-1,1,(512,d)).astype(np.float32)
    c = vfs_amd.Corpus(d); c.append(rows); c.enable_mfma(True)
    for metric in (1,0):
        c.mfma_stats(reset=True)
        c.search_batch(q, 10, metric)
        print(n, d, scale_rows, "metric", metric, "mfma/fallback", c.mfma_stats(), flush=True)
    c.close()
