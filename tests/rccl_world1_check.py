"""Helper for tests/test_gpu_api.py (not collected by pytest): one rank, backend nccl (= RCCL), the
real collective + device merge of vfs_amd.sharded on cuda:0.  Prints OK on success."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

import vfs_amd
from vfs_amd import sharded

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rng = np.random.default_rng(3)
n, d, k, nq = 5000, 32, 10, 7
rows = rng.uniform(-1, 1, (n, d)).astype(np.float32)
q = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
sc = sharded.ShardedCorpus(d, device=0)
sc.local.append(rows)
sc.set_row_base(0)
dq = torch.from_numpy(q).cuda()
local = torch.empty((nq, k, sharded.HIT_WORDS), dtype=torch.int32, device="cuda")
sc.local.search_device(dq.data_ptr(), nq, k, 1, local.data_ptr(), torch.cuda.current_stream().cuda_stream)
merged = sharded.exchange_and_merge(local, k, device_index=0, force_collective=True)
dist.barrier()
torch.cuda.synchronize()
ids_m, dist_m, cnt_m = sharded.hits_to_arrays(merged)
ids_l, dist_l, cnt_l = sharded.hits_to_arrays(local)
assert (ids_m == ids_l).all() and (dist_m.view(np.uint32) == dist_l.view(np.uint32)).all() and (cnt_m == k).all()
ids_e, dist_e, _ = sc.local.search_batch(q, k, 1)
assert (ids_m == ids_e).all()
# ShardedCorpus under the nccl backend exchanges through the library's own communicator (vfsx_comm_*),
# bootstrapped over this process group
assert sc.comm is not None and sc.comm.world == 1
ids_c, dist_c, cnt_c = sharded.hits_to_arrays(sc.search(dq, k, 1))
assert (ids_c == ids_e).all() and (dist_c.view(np.uint32) == dist_e.view(np.uint32)).all()
sc.close()
dist.destroy_process_group()
print("OK rccl all_gather_into_tensor + vfsx_merge_hits_device + vfsx_search_sharded")
