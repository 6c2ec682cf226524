import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from misc3d_amd import capi, synth
pts = synth.room_cloud_c5(10_000_000, 6)
ts=[]
for rep in range(6):
    t0 = time.perf_counter(); rc, planes, clusters = capi.segment_plane_iterative(pts, 0.01, 1000, 0.05, seed=19, copy=False); t1 = time.perf_counter()
    ts.append(1e3*(t1-t0))
print(os.environ.get("M3D_DBG_NOFUSE"), " ".join(f"{t:.1f}" for t in ts), "clusters", len(clusters))
