"""One-rank RCCL check of dump1090_amd.distributed (the CUDA-tensor branch the gloo CPU tests cannot reach) and the
cost of the per-step size exchange:  python tools/check_gather_nccl.py   (37.6 us per empty gather on the MI355X box)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from dump1090_amd.distributed import gather_arrays, gather_records
from dump1090_amd import _native as N
dev = torch.device("cuda:0")
for n in (0, 5, 1000, 0, 7):
    a = np.arange(n, dtype=np.uint64)
    out = gather_arrays(a, 0, None, dev)
    assert out is not None and np.array_equal(out, a), n
recs = np.zeros(3, dtype=N.RECORD_DTYPE); recs["j"] = [1, 2, 3]
r, c = gather_records(recs, np.arange(4, dtype=np.uint64), dst=0, device=dev)
assert np.array_equal(r["j"], [1, 2, 3]) and np.array_equal(c, np.arange(4))
import time
t0 = time.perf_counter()
for _ in range(200):
    gather_arrays(np.zeros(0, dtype=np.uint64), 0, None, dev)
print("empty gather: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
dist.destroy_process_group()
print("nccl world=1 gather ok")
