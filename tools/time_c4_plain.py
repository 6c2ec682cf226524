"""C4 registration (200 k <-> 200 k, 100 k hypotheses, confidence 1): wall clock of the RANSAC call, three runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
n = int(os.environ.get("M3D_C4_POINTS", "200000"))
d = synth.registration_pair_c4(n, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
for rep in range(3):
    t0 = time.perf_counter()
    T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=100_000, edge_length_threshold=0.9, confidence=1.0, seed=17)
    dt = time.perf_counter() - t0
    print(f"{dt*1e3:.1f} ms  validations {st['validations']} fitness {st['fitness']:.4f} rmse {st['inlier_rmse']:.9f} best {st['best_index']} "
          f"pose err {np.abs(T - d['T']).max():.2e}", flush=True)
