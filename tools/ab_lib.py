"""A/B two builds of the library on the same box: M3D_AB_LIB=<path> python tools/ab_lib.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from misc3d_amd import capi
if os.environ.get("M3D_AB_LIB"):
    capi.LIB_PATH = os.environ["M3D_AB_LIB"]
import numpy as np
from misc3d_amd import synth
pts = synth.plane_cloud_c2(1_000_000, seed=2)
c = capi.Cloud(pts)
for _ in range(5): c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
ts = []
for rep in range(5):
    t = time.perf_counter()
    for _ in range(40): g = c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
    ts.append((time.perf_counter() - t) / 40 * 1e3)
print(os.environ.get("M3D_AB_LIB", "default"), "C2 ms/fit min %.4f med %.4f" % (min(ts), sorted(ts)[2]), {k: round(g.stats[k], 3) for k in ("ms_score", "ms_refine")})
c.close()
room = synth.room_cloud_c5(10_000_000, 6)
capi.segment_plane_iterative(room, 0.01, max_iteration=1000, min_ratio=0.05, seed=19)
ts = []
for rep in range(3):
    t = time.perf_counter(); capi.segment_plane_iterative(room, 0.01, max_iteration=1000, min_ratio=0.05, seed=19); ts.append((time.perf_counter() - t) * 1e3)
print("   C5 ms min %.1f" % min(ts))
