"""C4's forced run (100 000 iterations, confidence 1) with the validation's candidate cache off / on (m3d_config.reg_cache):
wall time, the stats that must not move, and how many (tile, hypothesis) pairs the cache answered.
Usage on the GPU box: python tools/time_c4_forced.py [iterations]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from misc3d_amd import capi, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n = int(os.environ.get("M3D_C4_POINTS", "200000"))
d = synth.registration_pair_c4(n, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
ref = None
for cc in (0, 1, 0, 1):
    old = capi.set_config(reg_cache=cc)
    try:
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=iters,
                                             edge_length_threshold=0.9, confidence=1.0, seed=17)
            ts.append((time.perf_counter() - t0) * 1e3)
    finally:
        capi.restore_config(old)
    key = (T.tobytes(), st["best_index"], st["validations"], st["est_k"], st["fitness"], st["inlier_rmse"])   # (ties: borderline prunes vary from run to run)
    if ref is None:
        ref = key
    a, w = st["lds_wave_hypotheses"], st["global_wave_hypotheses"]
    print(json.dumps({"reg_cache": cc, "ms": [round(t, 3) for t in ts], "identical_to_first": key == ref,
                      "validations": st["validations"], "ties": st["ties"], "best_index": st["best_index"],
                      "pairs_exact_from_cache": a, "pairs_left_as_bounds": w,
                      "cache_share": (a / (a + w)) if a + w else None}))
