"""Small fixed workload for the rocprofv3 passes: one cloud upload (aos_to_soa_k: a known byte count
used to calibrate FETCH_SIZE / WRITE_SIZE) and a few launches of every hot kernel of fit_plane /
fit_sphere / fit_cylinder on the C2/C3-sized clouds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from misc3d_amd import capi, synth  # noqa: E402

N = int(os.environ.get("M3D_PMC_POINTS", "1000000"))
H = int(os.environ.get("M3D_PMC_HYP", "10000"))
pts = synth.plane_cloud_c2(N, 2)
with capi.Cloud(pts) as c:
    s = capi.draw_samples(N, 0, H, 11)
    for mode, name in ((0, "score_mask_k"), (1, "cull_mask_k"), (2, "score_k dense")):
        print("plane", name, "ms, listed pairs:", c.time_score(0, 0.01, s, reps=3, mode=mode))
    g = c.fit(0, 0.01, H, 1.0, seed=11)
    print("plane fit:", g.stats)
sp = synth.sphere_cloud_c3(N, 4)
with capi.Cloud(sp) as c:
    s = capi.draw_samples(N, 1, H, 13)
    for mode, name in ((0, "score_mask_k"), (1, "cull_mask_k"), (2, "score_k dense")):
        print("sphere", name, "ms, listed pairs:", c.time_score(1, 0.01, s, reps=3, mode=mode))
cp, cn = synth.cylinder_cloud_c3(N, 3)
with capi.Cloud(cp, cn) as c:
    s = capi.draw_samples(N, 2, H, 13)
    for mode, name in ((0, "score_mask_k"), (1, "cull_mask_k"), (2, "score_k dense")):
        print("cylinder", name, "ms, listed pairs:", c.time_score(2, 0.01, s, reps=3, mode=mode))
