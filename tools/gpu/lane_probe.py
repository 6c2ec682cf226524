import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from misc3d_amd import capi, synth
capi.set_config(lanes=8, kernel_timing=0)
pts = synth.plane_cloud_c1(200_000, 100)
def probe(tag):
    out = {}
    for lane in range(8):
        c = capi.Cloud(pts, lane=lane)
        jobs = [(c, 0, 0.01, 1000, 0.9999, 7)] * 200
        capi.fit_batch(jobs[:20], want_inliers=False)
        t0 = time.perf_counter(); capi.fit_batch(jobs, want_inliers=False); dt = time.perf_counter() - t0
        out[lane] = round(200 / dt)
        c.close()
    print(tag, "fits/s per lane alone:", out)
    cl = [capi.Cloud(pts, lane=t) for t in range(4)]
    jobs = [(cl[t], 0, 0.01, 1000, 0.9999, 7) for _ in range(200) for t in range(4)]
    capi.fit_batch(jobs[:40], want_inliers=False)
    t0 = time.perf_counter(); capi.fit_batch(jobs, want_inliers=False); dt = time.perf_counter() - t0
    print(tag, "4 lanes together:", round(800 / dt))
    for c in cl: c.close()
probe("fresh process:")
which = sys.argv[1] if len(sys.argv) > 1 else "match"
d = synth.registration_pair_c4(200000, seed=5)
if which in ("match", "both"):
    i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    probe("after a sliced match:")
if which in ("ransac", "both"):
    i0, i1 = capi.match_mutual_nn(d["feat_src"][:20000], d["feat_dst"][:20000])
    capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=20000, edge_length_threshold=0.9, confidence=1.0, seed=17)
    probe("after a registration:")
