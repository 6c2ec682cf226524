"""The `cpu_baseline` leg of tools/bench_configs.py: the reference's CPU path -- restated in oracle/ -- timed on THIS box's host
cores on a bounded sample of each configuration's workload ("the reference's CPU path timed beside every
BASELINE config").  Nothing in the product imports this file or the oracle; a figure here is a reported baseline, not a
target: a large GPU/CPU ratio says nothing about a kernel, its roofline fraction does.

Every function returns {"value", "unit", "cores", "kind": "port", "sample", "flags"[, "native": {...}]}.
  fits         orc_fit_omp_baseline(kind): `omp parallel for schedule(static)` over hypotheses, per-point distance with sqrt and
               divide, -O3 no -march (CMakeLists.txt:7,15-16), the loop of include/misc3d/common/ransac.h:561-654; + the same
               source -O3 -march=native
  matcher      orc_match_mutual_nn: exact brute force, OpenMP over queries (src/correspondence_matching.cpp:13-84 runs FLANN /
               Annoy; the exact answer is what is matched) on a slice of the queries
  validation   the nearest-neighbour pass under transform_estimation.cpp:154-161 with a kd-tree (scipy cKDTree for Open3D's
               nanoflann), all cores -- and the oracle's brute force on a slice of the source for scale
  segmentation orc_segment_plane_iterative_parallel (src/iterative_plane_segmentation.cpp:25-36 over ransac.h's loop): big
               rounds on the whole cloud, tail rounds on the clutter the big planes leave
"""
import time

import numpy as np


def _cores():
    import oracle
    return oracle.usable_cpus()


def fit_baseline(kind, pts, nrm, thr, seed, H_gpu, budget_s=6.0, label="fit"):
    import oracle
    hw = _cores()
    out = None
    for name, getter in (("port", lambda: (oracle.fit_omp_baseline, oracle.set_omp_threads)), ("native", oracle.native_baseline)):
        try:
            fit_fn, set_threads = getter()
        except Exception as e:       # noqa: BLE001  (the native build needs gcc on the box)
            if out is not None:
                out["native"] = {"error": f"{type(e).__name__}: {e}"}
            continue
        set_threads(hw)
        fit_fn(kind, pts, nrm, thr, hw, seed)                      # thread pool / pages
        t0 = time.perf_counter()
        fit_fn(kind, pts, nrm, thr, 2 * hw, seed)
        per_h = (time.perf_counter() - t0) / (2 * hw)
        share = budget_s * (0.65 if name == "port" else 0.35)
        H = int(max(2 * hw, min(H_gpu, share / max(per_h, 1e-9))))
        H = (H // hw) * hw or hw
        t0 = time.perf_counter()
        fit_fn(kind, pts, nrm, thr, H, seed)
        dt = time.perf_counter() - t0
        rec = {"value": H / dt, "unit": "hypotheses/s", "cores": hw,
               "sample": f"{label}: {len(pts)} points x {H} hypotheses of the same sampler stream (the GPU's fit has {H_gpu}), {dt:.1f} s, "
                         "omp parallel for schedule(static) over hypotheses",
               "flags": "-O3 -ffp-contract=off (no -march), as the reference's CMakeLists.txt" if name == "port"
                        else "-O3 -march=native -ffp-contract=off"}
        if name == "port":
            rec["kind"] = "port"
            out = rec
        else:
            out["native"] = rec
    return out


def match_baseline(feat_src, feat_dst, budget_s=6.0):
    import oracle
    hw = _cores()
    oracle.set_omp_threads(hw)
    n, dim = feat_src.shape
    q = 256
    t0 = time.perf_counter()
    oracle.match_mutual_nn(feat_src[:q], feat_dst)
    per_q = (time.perf_counter() - t0) / q
    q = int(max(256, min(n, budget_s / max(per_q, 1e-9))))
    t0 = time.perf_counter()
    oracle.match_mutual_nn(feat_src[:q], feat_dst)
    dt = time.perf_counter() - t0
    pairs = 2.0 * q * len(feat_dst)        # (both directions: q x n and n x q)
    return {"value": pairs / dt, "unit": "pair-distances/s", "cores": hw, "kind": "port",
            "sample": f"exact mutual nearest neighbours of the first {q} of {n} queries against all {len(feat_dst)} rows and back, "
                      f"{dim}-D, {dt:.1f} s (the whole call is {n / q:.0f} x that: {dt * n / q:.0f} s)",
            "flags": "-O3 -ffp-contract=off (no -march); brute force, omp parallel for over queries",
            "whole_call_s_extrapolated": dt * n / q}


def validation_baseline(src, dst, T, thr, budget_s=6.0):
    import oracle
    from scipy.spatial import cKDTree
    hw = _cores()
    t0 = time.perf_counter()
    tree = cKDTree(dst)
    t_build = time.perf_counter() - t0
    oracle.reg_validate_kdtree(src[:2000], dst, T, thr, workers=hw, tree=tree)
    reps, t_acc, cnt = 0, 0.0, 0
    while t_acc < budget_s * 0.6 and reps < 50:
        Tk = T.copy()
        Tk[:3, 3] += 1e-4 * reps
        t0 = time.perf_counter()
        cnt, _ = oracle.reg_validate_kdtree(src, dst, Tk, thr, workers=hw, tree=tree)
        t_acc += time.perf_counter() - t0
        reps += 1
    rate = reps * len(src) / t_acc
    # the oracle's own brute force on a slice, for scale (it is the parity chain's validation, not the reference's data structure)
    oracle.set_omp_threads(hw)
    m = 512
    t0 = time.perf_counter()
    oracle.reg_validate(src[:m], dst, T, thr)
    dt_b = time.perf_counter() - t0
    return {"value": rate, "unit": "queries/s (nearest target point within the threshold)", "cores": hw, "kind": "port",
            "sample": f"{reps} validations of all {len(src)} source points near the true pose, kd-tree over the {len(dst)} target points "
                      f"(built once: {t_build * 1e3:.0f} ms), {t_acc:.1f} s; last count {cnt}",
            "flags": "scipy cKDTree (C++), workers = cores -- Open3D's KDTreeFlann is nanoflann under an OpenMP loop",
            "oracle_brute_force_queries_per_s": m / dt_b, "ms_per_validation": t_acc / reps * 1e3}


def segmentation_baseline(pts, thr, max_iteration, min_ratio, seed, big_clusters, budget_s=12.0):
    """big_clusters: the index arrays of the GPU's first planes -- the tail sample is the cloud without them."""
    import oracle
    hw = _cores()
    oracle.set_omp_threads(hw)
    n = len(pts)
    t0 = time.perf_counter()
    rc, planes, clusters = oracle.segment_plane_iterative(pts, thr, max_iteration=max_iteration, min_ratio=min_ratio, seed=seed,
                                                          max_clusters=2, lookahead=8 * hw)
    t_big = time.perf_counter() - t0
    keep = np.ones(n, dtype=bool)
    for c in big_clusters:
        keep[c] = False
    rest = np.ascontiguousarray(pts[keep])
    k_tail = 10
    t0 = time.perf_counter()
    rc2, planes2, clusters2 = oracle.segment_plane_iterative(rest, thr, max_iteration=max_iteration, min_ratio=0.0, seed=seed,
                                                             max_clusters=k_tail, lookahead=8 * hw)
    t_tail = time.perf_counter() - t0
    return {"value": t_big / max(len(clusters), 1) * 1e3, "unit": "ms per big round (a plane of the whole cloud)", "cores": hw, "kind": "port",
            "sample": f"the first {len(clusters)} rounds on all {n} points ({t_big:.1f} s; inliers {[len(c) for c in clusters]}), and "
                      f"{len(clusters2)} tail rounds on the {len(rest)} points the GPU's {len(big_clusters)} big planes leave ({t_tail:.1f} s)",
            "flags": "-O3 -ffp-contract=off (no -march); the sequential loop with the records of the next 8 x cores hypotheses computed "
                     "ahead by an OpenMP team (identical output)",
            "tail_ms_per_round": t_tail / max(len(clusters2), 1) * 1e3}
