"""C4 with the reference's default confidence (0.999): the call a registration user makes -- a handful of validations, so
the set-up (uploads, bounding boxes, grids, source sort) is what is timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(int(os.environ.get("M3D_C4_POINTS", "200000")), seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
for rep in range(5):
    t0 = time.perf_counter()
    T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=100_000, edge_length_threshold=0.9, confidence=0.999, seed=17)
    print(f"{(time.perf_counter() - t0) * 1e3:.3f} ms  iterations {st['iterations']} validations {st['validations']} ms_total {st['ms_total']:.3f}", flush=True)
