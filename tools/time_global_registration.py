"""One m3d_global_registration call after another on one fragment pair (default 50 000 points a side, M3D_N2_POINTS): wall clock and the
library's own breakdown (match / RANSAC / information matrix)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
n = int(os.environ.get("M3D_N2_POINTS", "50000"))
d = synth.registration_pair_c4(n, seed=5)
for rep in range(6):
    t0 = time.perf_counter()
    ok, T, info, st = capi.global_registration(d["src"], d["dst"], d["feat_src"], d["feat_dst"], voxel_size=0.03 / 1.4, seed=17, want_stats=True)
    print(f"{(time.perf_counter() - t0) * 1e3:.3f} ms  ok {ok}  match {st['ms_match']:.3f} ransac {st['ms_ransac']:.3f} info {st['ms_info']:.3f}  matches {st['n_matches']}", flush=True)
