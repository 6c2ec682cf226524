"""What would the scan cost if the database rows came ordered by their reverse-search thresholds?  The target descriptors are
reordered on the host by the distance to their nearest source row (what the thresholds estimate), the call repeated."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.spatial import cKDTree
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(200000, seed=5)
fs, fd = d["feat_src"], d["feat_dst"]
mode = int(os.environ.get("SORTED", "0"))
if mode:
    dist, _ = cKDTree(fs[:len(fs) // 8]).query(fd, workers=-1)      # (the sample the thresholds come from: the first eighth)
    if mode == 1:
        order = np.argsort(-dist, kind="stable")
    else:                                                             # within chunks of `mode` consecutive rows only
        order = np.concatenate([c0 + np.argsort(-dist[c0:c0 + mode], kind="stable") for c0 in range(0, len(fd), mode)])
    fd = np.ascontiguousarray(fd[order])
for rep in range(3):
    t0 = time.perf_counter()
    i0, i1 = capi.match_mutual_nn(fs, fd)
    print(f"{(time.perf_counter() - t0) * 1e3:.2f} ms  matches {len(i0)} fallbacks {capi.match_last_fallbacks()}", flush=True)
