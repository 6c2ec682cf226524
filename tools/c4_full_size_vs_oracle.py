"""BASELINE configs[3] (C4) against the ORACLE at its full size -- too slow for the test suite (the oracle's matcher is a
200 k x 200 k x 33 brute force in fp64: minutes on the GPU box's host cores), so it is run once per round on the GPU box and
its output committed (profiles/rNN_c4_full_size_vs_oracle.txt):
  * match_correspondence, 200 k x 200 k x 33: the mutual nearest-neighbour pairs, all of them;
  * compute_transformation_ransac on those correspondences with the reference's own confidence (0.999: a few dozen
    iterations; every validation is a 200 k x 200 k exact nearest-neighbour search in the oracle): T bit for bit, iterations,
    validations, est_k, fitness, inlier_rmse;
  * the first M3D_C4_ORACLE_ITERS (default 300) iterations of BASELINE's forced run (confidence 1.0) the same way."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from misc3d_amd import capi, synth  # noqa: E402

n = int(os.environ.get("M3D_C4_POINTS", "200000"))
d = synth.registration_pair_c4(n, seed=5)
t0 = time.time()
g0, g1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
t1 = time.time()
o0, o1 = oracle.match_mutual_nn(d["feat_src"], d["feat_dst"])
t2 = time.time()
same = np.array_equal(g0.astype(np.int64), o0) and np.array_equal(g1.astype(np.int64), o1)
print(f"match_correspondence {n} x {n} x 33: GPU {len(g0)} pairs in {(t1 - t0) * 1e3:.1f} ms (first call), oracle {len(o0)} pairs in "
      f"{t2 - t1:.1f} s on {oracle.usable_cpus()} CPUs: pairs identical = {same}", flush=True)
assert same
for conf, iters, label in ((0.999, 100_000, "the reference's confidence 0.999"),
                           (1.0, int(os.environ.get("M3D_C4_ORACLE_ITERS", "300")), "confidence 1.0 (BASELINE's forced run, first iterations)")):
    t0 = time.time()
    T, st = capi.registration_ransac(d["src"], d["dst"], g0, g1, threshold=0.03, max_iter=iters, edge_length_threshold=0.9,
                                     confidence=conf, seed=17)
    t1 = time.time()
    o = oracle.registration_ransac(d["src"], d["dst"], o0, o1, thr=0.03, max_iter=iters, edge_thr=0.9, confidence=conf, seed=17)
    t2 = time.time()
    ok = (np.array_equal(T.view(np.uint64), o.T.view(np.uint64)) and st["iterations"] == o.iterations and
          st["validations"] == o.validations and st["est_k"] == o.est_k and st["fitness"] == o.fitness and
          st["best_index"] == o.best_index)
    print(f"compute_transformation_ransac, {label}, max_iter {iters}: GPU {(t1 - t0) * 1e3:.1f} ms, oracle {t2 - t1:.1f} s; iterations "
          f"{st['iterations']} / {o.iterations}, validations {st['validations']} / {o.validations}, est_k {st['est_k']} / {o.est_k}, "
          f"fitness {st['fitness']:.6f} / {o.fitness:.6f}, rmse {st['inlier_rmse']:.9g} / {o.inlier_rmse:.9g}: T bit-identical and "
          f"counters equal = {ok}", flush=True)
    assert ok and abs(st["inlier_rmse"] - o.inlier_rmse) <= 1e-9 * max(1.0, o.inlier_rmse)
print("C4 full size vs oracle: all identical")
