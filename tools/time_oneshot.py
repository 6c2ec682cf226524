"""The reference-shaped call: fit_plane(points, 0.01, 1000) with the default probability 0.9999 from host arrays (create +
fit + destroy per call), 1 M points; prints the wall clock of ten calls and the library's own breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
n = int(os.environ.get("M3D_POINTS", "1000000"))
kind = int(os.environ.get("M3D_KIND", "0"))
if kind == 0:
    pts, nrm = synth.plane_cloud_c2(n, 2), None
elif kind == 1:
    pts, nrm = synth.sphere_cloud_c3(n, 4), None
else:
    pts, nrm = synth.cylinder_cloud_c3(n, 3)
for prob, iters in ((0.9999, 1000), (1.0, 10000)):
    ts = []
    for rep in range(12):
        t0 = time.perf_counter()
        g = capi.fit(kind, pts, nrm, 0.01, iters, prob, seed=11 + rep, copy=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"kind {kind} probability {prob} max_iteration {iters}: " + " ".join(f"{t:.3f}" for t in ts[2:]) +
          f" ms; iterations {g.stats['iterations']} scored {g.stats['hypotheses_scored']} inliers {len(g.inliers)} ms_total(lib fit) {g.stats['ms_total']:.3f}")
