"""C5 (segment_plane_iterative on 10 M points): wall clock of six calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
pts = synth.room_cloud_c5(int(os.environ.get("M3D_C5_POINTS", "10000000")), 6)
ts = []
for rep in range(int(os.environ.get("M3D_C5_REPS", "6"))):
    t0 = time.perf_counter()
    rc, planes, clusters = capi.segment_plane_iterative(pts, 0.01, 1000, 0.05, seed=19)
    ts.append(1e3 * (time.perf_counter() - t0))
    br = capi.last_segment_ms()
print(" ".join(f"{t:.1f}" for t in ts), "ms; clusters", len(clusters), "points", int(sum(len(c) for c in clusters)))
print("last call inside the library:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in br.items()})
