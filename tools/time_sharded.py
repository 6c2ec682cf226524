"""Where does a sharded fit spend its time at world size 1?  (python tools/time_sharded.py, on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, distributed, synth

N, H = 1_000_000, 10_000
pts = synth.plane_cloud_c2(N, seed=2)
c = capi.Cloud(pts)
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for n in ("make_sampler", "score_shard", "score_shard_packed", "score_range", "refine", "exact_error"):
    wrap(c, n)
for _ in range(3):
    distributed.fit_sharded(c, N, 0, 0.01, H, 1.0, 11, copy=False)
T.clear()
t0 = time.perf_counter(); R = 20
for _ in range(R):
    r = distributed.fit_sharded(c, N, 0, 0.01, H, 1.0, 11, copy=False)
tot = (time.perf_counter() - t0) / R
print("total ms/step %.3f" % (tot * 1e3), {k: round(v / R * 1e3, 3) for k, v in T.items()}, "other %.3f" % ((tot - sum(T.values()) / R) * 1e3))
t0 = time.perf_counter()
for _ in range(R):
    g = c.fit(0, 0.01, H, 1.0, seed=11, copy=False)
print("direct ms/step %.3f" % ((time.perf_counter() - t0) / R * 1e3), g.stats["ms_score"], g.stats["ms_refine"])
