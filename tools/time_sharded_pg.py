"""The sharded fit THROUGH a process group at world size 1 (RCCL all-gather of the records included), with the time
of the collective path split out.  python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1
--master-port 29533 tools/time_sharded_pg.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from misc3d_amd import capi, distributed, synth

local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
dev = torch.device("cuda", local)
N, H = 1_000_000, 10_000
c = capi.Cloud(synth.plane_cloud_c2(N, seed=2), device=local)
T = {}
orig = distributed._all_gather_records
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); T["all_gather"] = T.get("all_gather", 0.0) + time.perf_counter() - t; return r
distributed._all_gather_records = timed
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for n in ("make_sampler", "score_shard", "score_shard_packed", "score_range", "refine", "exact_error"):
    wrap(c, n)
for _ in range(10):
    distributed.fit_sharded(c, N, 0, 0.01, H, 1.0, 11, device=dev, copy=False)
T.clear()
R = 200
import gc
gc.collect(); gc.disable()   # as bench.py does: a full collection with torch imported costs ~100 steps
t0 = time.perf_counter()
for _ in range(R):
    r = distributed.fit_sharded(c, N, 0, 0.01, H, 1.0, 11, device=dev, copy=False)
tot = (time.perf_counter() - t0) / R
print("sharded through RCCL, world 1: %.3f ms/step" % (tot * 1e3), {k: round(v / R * 1e3, 3) for k, v in T.items()})
t0 = time.perf_counter()
for _ in range(R):
    g = c.fit(0, 0.01, H, 1.0, seed=11, copy=False)
print("direct: %.3f ms/step" % ((time.perf_counter() - t0) / R * 1e3))
dist.destroy_process_group()
