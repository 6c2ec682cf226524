"""Randomised stress of the paths round 3 added, against the oracle (run on the GPU box: tests/test_gpu_soak.py runs a fixed
60 s budget of it with a fixed seed inside the suite; longer runs from the command line):
the one-scan mutual matcher (random sizes incl. tiny and ragged ones, clustered descriptors, duplicated rows), the
segmentation with the speculative RefineModel of rounds that run to max_iteration, one-shot fits on the dense path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from misc3d_amd import capi, synth  # noqa: E402



def run(budget=240.0, reg_budget=0.0, seed=1, log=print):
    """Random cases against the oracle for `budget` seconds (matcher, segmentation, one-shot fits in turn) and `reg_budget`
    seconds (registration); raises AssertionError with the case's parameters on the first difference.
    Returns {"match": ..., "segment": ..., "fit": ..., "registration": ...} = cases compared."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_match = n_seg = n_fit = 0
    while time.time() < t_end:
        # ---- matcher
        ns, nd = (int(v) for v in rng.integers(1, 6000, 2))
        if rng.random() < 0.2:
            ns = int(rng.integers(1, 40))
        fs = rng.uniform(0, 1, (ns, 33))
        fd = rng.uniform(0, 1, (nd, 33))
        k = min(ns, nd) // 2
        if k:
            fd[:k] = fs[rng.permutation(ns)[:k]] + rng.normal(0, 10.0 ** rng.integers(-6, -1), (k, 33))
        if rng.random() < 0.3 and nd > 10:          # clusters: many near-duplicates of a few rows
            c = int(rng.integers(2, 10))
            fd[rng.integers(0, nd, nd // 3)] = fd[rng.integers(0, nd, c)][rng.integers(0, c, nd // 3)] + rng.normal(0, 1e-7, (nd // 3, 33))
        if rng.random() < 0.2 and ns > 300:         # one exact duplicate block
            fs[100:100 + 280] = fs[100]
        # (round 5: a third of the cases with the matrices going up in slices under the scan, whatever their size -- and then now
        # and then a late slice that does not fit the scale chosen from the first ones)
        sliced = rng.random() < 0.35
        if sliced and rng.random() < 0.3 and ns > 600:
            fs[ns - 50:] *= float(rng.choice([1.9, 3.0, 6.0]))
        old_cfg = capi.set_config(match_pipeline=2) if sliced else None
        try:
            a, b = capi.match_mutual_nn(fs, fd)
        finally:
            if old_cfg is not None:
                capi.restore_config(old_cfg)
        oa, ob = oracle.match_mutual_nn(fs, fd)
        assert np.array_equal(a.astype(np.int64), oa) and np.array_equal(b.astype(np.int64), ob), ("match", ns, nd, sliced)
        n_match += 1
        # ---- segmentation (adaptive stop never bites in the clutter rounds: speculative RefineModel)
        n = int(rng.integers(20_000, 120_000))
        pts = synth.room_cloud_c5(n, int(rng.integers(0, 1000)))
        mi = int(rng.choice([50, 100, 300, 1000]))
        mr = float(rng.choice([0.03, 0.05, 0.1]))
        sd = int(rng.integers(0, 10_000))
        ro, po, co = oracle.segment_plane_iterative(pts, 0.01, max_iteration=mi, min_ratio=mr, seed=sd, lookahead=64)
        rg, pg, cg = capi.segment_plane_iterative(pts, 0.01, max_iteration=mi, min_ratio=mr, seed=sd)
        assert len(co) == len(cg) and all(np.array_equal(x, y) for x, y in zip(co, cg)), ("segment", n, mi, mr, sd)
        assert np.allclose(po, pg, rtol=0, atol=1e-9)
        n_seg += 1
        # ---- one-shot fits, few hypotheses (dense path) and many (sorted path)
        kind = int(rng.integers(0, 3))
        m = int(rng.integers(3_000, 40_000))
        if kind == 0:
            p, nr = synth.plane_cloud_c1(m, sd), None
        elif kind == 1:
            p, nr = synth.sphere_cloud_c3(m, sd), None
        else:
            p, nr = synth.cylinder_cloud_c3(m, sd)
        # (the fp32 screens work on tile-local offsets: a scene scaled and moved far from the origin probes their bounds)
        sc = float(10.0 ** rng.uniform(-2, 2))
        p = p * sc + rng.uniform(-1, 1, 3) * sc * float(10.0 ** rng.uniform(0, 3))
        thr = 0.01 * sc
        for it, prob in ((int(rng.integers(1, 1025)), float(rng.choice([0.9999, 1.0, 0.5]))), (int(rng.integers(1025, 3000)), 1.0)):
            g = capi.fit(kind, p, nr, thr, it, prob, seed=sd)
            o = oracle.fit(kind, p, nr, thr=thr, max_iter=it, prob=prob, seed=sd, lookahead=32)
            assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.ret, o.best_index, o.count, o.iterations), ("fit", kind, m, it, prob, sd)
            assert np.array_equal(g.inliers, o.inliers), ("fit inliers", kind, m, it, prob, sd, sc)
            mag = float(np.abs(p).max())
            assert np.allclose(g.params, o.params, rtol=1e-9, atol=1e-9 * mag), ("fit params", kind, m, it, prob, sd, sc, g.params, o.params)
            n_fit += 1
    log(f"stress ok: {n_match} matcher cases, {n_seg} segmentations, {n_fit} fits in {budget:.0f} s (seed {seed})")
    # ---- registration RANSAC: the validation's neighbour search screens on 8-byte quantised list entries (16-bit coordinates
    # over the cell block): scenes of random size, scale and offset, thresholds that move the grid's cell edge
    t_end = time.time() + reg_budget
    n_reg = 0
    while time.time() < t_end:
        n = int(rng.integers(800, 5000))
        d = synth.registration_pair_c4(n, seed=int(rng.integers(0, 10_000)))
        sc = float(10.0 ** rng.uniform(-2, 2))
        off = rng.uniform(-1, 1, 3) * sc * float(10.0 ** rng.uniform(0, 2.5))
        src, dst = d["src"] * sc + off, d["dst"] * sc + off
        i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
        if len(i0) < 3:
            continue
        thr = sc * float(rng.choice([0.01, 0.03, 0.07, 0.2]))
        mi = int(rng.choice([300, 1000, 2500]))
        conf = float(rng.choice([1.0, 0.999]))
        sd = int(rng.integers(0, 10_000))
        T, st = capi.registration_ransac(src, dst, i0, i1, threshold=thr, max_iter=mi, edge_length_threshold=0.9, confidence=conf, seed=sd)
        o = oracle.registration_ransac(src, dst, i0.astype(np.int64), i1.astype(np.int64), thr=thr, max_iter=mi, edge_thr=0.9, confidence=conf, seed=sd)
        assert np.array_equal(T.view(np.uint64), o.T.view(np.uint64)), ("reg T", n, sc, thr, mi, conf, sd)
        assert (st["iterations"], st["validations"], st["est_k"], st["best_index"]) == (o.iterations, o.validations, o.est_k, o.best_index), ("reg counters", n, sc, thr, mi, conf, sd)
        assert st["fitness"] == o.fitness
        n_reg += 1
    if n_reg:
        log(f"stress ok: {n_reg} registrations identical to the oracle (T bit for bit, iterations, validations, est_k) in {reg_budget:.0f} s")
    return {"match": n_match, "segment": n_seg, "fit": n_fit, "registration": n_reg}


if __name__ == "__main__":
    run(float(os.environ.get("M3D_STRESS_SECONDS", "240")), float(os.environ.get("M3D_STRESS_REG_SECONDS", "0")),
        int(os.environ.get("M3D_STRESS_SEED", "1")))
