#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<PY
import sys, os, time
sys.path.insert(0, os.getcwd())
from misc3d_amd import capi, synth
pts = synth.plane_cloud_c2(1_000_000, seed=2)
c = capi.Cloud(pts)
s = capi.draw_samples(len(pts), 0, 10000, 11)
c.time_score(0, 0.01, s, reps=3, mode=1)
ms, _ = c.time_score(0, 0.01, s, reps=20, mode=1)
capi.set_config(kernel_timing=0)
for _ in range(50): c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
t0 = time.perf_counter()
for _ in range(300): g = c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
dt = (time.perf_counter() - t0) / 300 * 1e3
print("cull_mask_k %.1f us   fit %.4f ms   best %d inliers %d" % (ms * 1e3, dt, g.stats["best_index"], g.stats["n_inliers"]))
PY
