"""Run every BASELINE.json configuration once at full size on ONE MI355X and print a JSON line per
config (timings from the C ABI's own clocks + wall clock).  Not the headline bench (that is
bench.py); these lines feed DESIGN.md section 6 and check that the full sizes run at all.

  C1 fit_plane 50k pts, 100 iters (plumbing)            C2 fit_plane 1M pts, 10k hyp (= bench.py)
  C3 fit_cylinder + fit_sphere 1M pts, 50k hyp          C4 match + compute_transformation_ransac 200k<->200k, 100k hyp
  C5 segment_plane_iterative 10M pts (single GPU leg)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from misc3d_amd import capi, synth  # noqa: E402

NO_CPU = "--no-cpu-baseline" in sys.argv or bool(os.environ.get("M3D_NO_CPU_BASELINE"))   # (runs under the profiler)
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
which = set(_args) or {"C1", "C2", "C3", "C4", "C5"}
ALL = len(_args) == 0
sys.path.insert(0, os.path.join(ROOT, "tools"))


def cpu(fn, *a, **kw):
    """the cpu_baseline object of a line (tools/cpu_baselines.py: the oracle's restatement of the reference's CPU path on this
    box's cores, a bounded sample), or None under --no-cpu-baseline"""
    if NO_CPU:
        return None
    import cpu_baselines
    return getattr(cpu_baselines, fn)(*a, **kw)

HBM_PEAK_GBS = 8000.0
FP64_VALU_PEAK_TOPS = 39.3      # 256 CU x 4 SIMD x 16 fp64 lanes/clk x 2.4 GHz, non-FMA ops (bench.py)
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16 MFMA (MI355X_MICROARCH.md)
N4_COUNTS = (55576.0, 18203.0)   # boundary_k: (VALU, LDS) instructions per wave of 64 points (one point per lane) -- from profiles/r06_pmc_boundary.txt
VALU_OPS_FP64 = {0: 7, 1: 10, 2: 22}    # score_mask_k: fp64 VALU instructions per (point, hypothesis)
VALU_OPS_SCREEN = {0: 3.625, 1: 4.125, 2: 6.125}   # score_screen_k: packed-fp32 screen, 29 / 33 / 49 instructions per 8 points; bench.py has the count
capi.set_config(kernel_timing=1)  # the fit rooflines need m3d_stats.ms_score_kernel (HIP events around the scoring launches)


def fit_roofline(kind, st):
    """VALU-issue roofline of the scoring launches of ONE fit, from the library's own counters: (tile, hypothesis)
    pairs evaluated x 512 points x VALU instructions per (point, hypothesis) / time of the scoring launches (HIP events).
    Recomputable from profiles/r02_configs_kernel_stats.csv (same launches, AverageNs x Calls of score_screen_k<kind> /
    score_mask_k<2>)."""
    if not st["ms_score_kernel"]:
        return None
    screened = bool(capi.get_config().score_fp32_screen)
    ops = (VALU_OPS_SCREEN if screened else VALU_OPS_FP64)[kind]
    tops = st["pairs_timed"] * 512.0 * ops / (st["ms_score_kernel"] * 1e-3) / 1e12   # (pairs of the timed launches: m3d_stats.pairs_timed)
    # the peak priced per instruction class (bench.py: CLASS_CYCLES from tools/ubench/valu_rates.hip, the loops' instruction mixes)
    import bench as _b
    n_mix, cyc_mix, _ = _b.peak_model((_b.MIX_SCREEN if screened else _b.MIX_FP64)[kind])
    peak = _b.SIMDS * 64.0 * _b.CLOCK_HZ / (cyc_mix / n_mix) / 1e12
    frac = st["pairs_timed"] * cyc_mix / (st["ms_score_kernel"] * 1e-3 * _b.SIMDS * _b.CLOCK_HZ)
    return {"bound": "valu-issue", "kernel": f"m3d::score_{'screen' if screened else 'mask'}_k<{kind}>", "achieved": tops,
            "peak": peak, "unit": "T lane-instructions/s (VALU issue)", "frac": frac,
            "frac_if_every_instruction_took_4_cycles": tops / FP64_VALU_PEAK_TOPS,
            "ops_per_pair": ops, "pairs_recounted_in_fp64": st["pairs_exact"],
            "launches": st["score_launches"], "kernel_ms_total": st["ms_score_kernel"], "tile_hypothesis_pairs": st["pairs_timed"], "pairs_of_the_untimed_lead_pass": st["pairs_scored"] - st["pairs_timed"]}


def emit(name, **kw):
    print(json.dumps({"config": name, **kw}), flush=True)


def timed_fit(cloud, kind, thr, H, prob, seed, reps=3):
    """(seconds, Fit) of the fastest of `reps` x 5 fits after enough of them to bring the clocks up (>= 40 ms of fits: a cold
    GPU times its first fits 5-8 % long, bench.py PRIMING_FITS); Fit.stats["ms_plain"] = the same with m3d_config.kernel_timing =
    0, the library's default (no HIP events attached to the scoring launches)."""
    t_end = time.perf_counter() + 0.04
    cloud.fit(kind, thr, H, prob, seed=seed, copy=False)
    while time.perf_counter() < t_end:
        cloud.fit(kind, thr, H, prob, seed=seed, copy=False)
    best = None
    for _ in range(reps * 5):
        t0 = time.perf_counter()
        g = cloud.fit(kind, thr, H, prob, seed=seed, copy=False)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, g)
    capi.set_config(kernel_timing=0)
    plain = None
    for _ in range(reps * 5):
        t0 = time.perf_counter()
        cloud.fit(kind, thr, H, prob, seed=seed, copy=False)
        dt = time.perf_counter() - t0
        plain = dt if plain is None else min(plain, dt)
    capi.set_config(kernel_timing=1)
    best[1].stats["ms_plain"] = plain * 1e3
    return best


if "C1" in which:
    pts = synth.plane_cloud_c1(50_000, 1)
    with capi.Cloud(pts) as c:
        dt, g = timed_fit(c, 0, 0.01, 100, 0.9999, 7)
    emit("C1 fit_plane 50k x 100 iters p=0.9999", ms=dt * 1e3, iterations=g.stats["iterations"],
         n_inliers=g.stats["n_inliers"], params=g.params.tolist(), roofline=fit_roofline(0, g.stats),
         cpu_baseline=cpu("fit_baseline", 0, pts, None, 0.01, 7, 100, 2.0, "C1 fit_plane"))

if "C2" in which:
    pts = synth.plane_cloud_c2(1_000_000, 2)
    with capi.Cloud(pts) as c:
        dt, g = timed_fit(c, 0, 0.01, 10_000, 1.0, 11)
    emit("C2 fit_plane 1M x 10k hyp", ms=dt * 1e3, hyp_per_s=10_000 / dt, n_inliers=g.stats["n_inliers"],
         best_index=g.stats["best_index"], ms_without_timing_events=g.stats["ms_plain"],
         stats={k: g.stats[k] for k in ("ms_sample", "ms_score", "ms_refine")},
         roofline=fit_roofline(0, g.stats), cpu_baseline=cpu("fit_baseline", 0, pts, None, 0.01, 11, 10_000, 5.0, "C2 fit_plane"))

if "C3" in which:
    cp, cn = synth.cylinder_cloud_c3(1_000_000, 3)
    with capi.Cloud(cp, cn) as c:
        dt, g = timed_fit(c, 2, 0.01, 50_000, 1.0, 13, reps=2)
    emit("C3 fit_cylinder 1M x 50k hyp", ms=dt * 1e3, hyp_per_s=50_000 / dt, n_inliers=g.stats["n_inliers"],
         best_index=g.stats["best_index"], params=g.params.tolist(),
         ms_without_timing_events=g.stats["ms_plain"],
         stats={k: g.stats[k] for k in ("ms_sample", "ms_score", "ms_refine", "exact_rmse_evals")},
         roofline=fit_roofline(2, g.stats), cpu_baseline=cpu("fit_baseline", 2, cp, cn, 0.01, 13, 50_000, 6.0, "C3 fit_cylinder"))
    sp = synth.sphere_cloud_c3(1_000_000, 4)
    with capi.Cloud(sp) as c:
        dt, g = timed_fit(c, 1, 0.01, 50_000, 1.0, 13, reps=2)
    emit("C3 fit_sphere 1M x 50k hyp", ms=dt * 1e3, hyp_per_s=50_000 / dt, n_inliers=g.stats["n_inliers"],
         best_index=g.stats["best_index"], params=g.params.tolist(),
         ms_without_timing_events=g.stats["ms_plain"],
         stats={k: g.stats[k] for k in ("ms_sample", "ms_score", "ms_refine", "exact_rmse_evals")},
         roofline=fit_roofline(1, g.stats), cpu_baseline=cpu("fit_baseline", 1, sp, None, 0.01, 13, 50_000, 6.0, "C3 fit_sphere"))

if "C4" in which:
    n = int(os.environ.get("M3D_C4_POINTS", "200000"))
    d = synth.registration_pair_c4(n, seed=5)
    def timed_match():
        t0 = time.perf_counter()
        r = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
        return time.perf_counter() - t0, r
    t_match, (i0, i1) = timed_match()
    ts = sorted(timed_match()[0] for _ in range(7))     # (calls two to four of a process still get faster: pools, clocks)
    t_match2 = ts[len(ts) // 2]
    sliced = bool(capi.match_last_path() & 2)
    old_cfg = capi.set_config(match_pipeline=0)
    try:
        tw = sorted(timed_match()[0] for _ in range(5))
    finally:
        capi.restore_config(old_cfg)
    inv = np.empty(n, dtype=np.int64)
    inv[d["perm"]] = np.arange(n)
    true_frac = float(np.mean(inv[i0.astype(np.int64)] == i1.astype(np.int64)))
    # the screen is ONE fp16 contraction over all (query, row) pairs -- since round 3 a single scan serves both search directions,
    # since round 4 over the hi halves only (K = 48: 33 values, the norms' pieces, padding) -- plus the two warm-up passes (1/16
    # and 1/8 of it): n x n x 48 x 2 x (1 + 3/16) flop.  Priced with the WHOLE call's wall clock (106 MB of descriptors over
    # PCIe, packing, binning, the fp64 verifications): a lower bound of the kernel's own fraction.  pair_dist_per_s counts both
    # directions' distances.
    mm_flop = float(n) * n * 48 * 2 * (1.0 + 3.0 / 16.0)
    emit("C4 match_correspondence", n=n, dim=33, ms_first=t_match * 1e3, ms=t_match2 * 1e3, ms_best=ts[0] * 1e3,
         uploads_sliced_under_the_scan=sliced, ms_both_matrices_uploaded_first=tw[len(tw) // 2] * 1e3, matches=len(i0),
         true_fraction=true_frac, pair_dist_per_s=2.0 * n * n / t_match2,
         roofline={"bound": "mfma", "kernel": "m3d::nn16_scan_k<false, true> (one scan, both directions)",
                   "achieved": mm_flop / t_match2 / 1e12,
                   "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s (fp16 MFMA, whole call's wall clock)",
                   "frac": mm_flop / t_match2 / 1e12 / MFMA_F16_PEAK_TFLOPS, "flop": mm_flop,
                   "note": "K = 48 since round 4 (hi halves only; K = 112 before): 3/7 of the flop per pair, so the fraction on the "
                           "smaller count is lower although the call is faster.  The scan ALONE, in one launch (profiles/r06_match_scan_findings.txt): "
                                   "3.8 ms = 0.40 of the peak on the nominal 3.84 TFLOP; matrix pipe busy 0.535 of the SIMD cycles, the VALU port "
                                   "nearly full at that rate (DESIGN 4)"},
         cpu_baseline=cpu("match_baseline", d["feat_src"], d["feat_dst"], 6.0))
    dts = []
    for _ in range(2):      # (the first call of a process also sizes the device's block free list: ~10 ms)
        t0 = time.perf_counter()
        T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=100_000,
                                         edge_length_threshold=0.9, confidence=1.0, seed=17)
        dts.append(time.perf_counter() - t0)
    dt = dts[1]
    # every validation = one exact nearest-neighbour query per source point.  SURVEY.md 8(d)'s unit is 24 B per (hypothesis,
    # point); the kernel is bound by the L1's tag look-ups (a list-entry gather touches ~29 cache lines per instruction:
    # ~0.9 look-ups per cycle and CU) with VALU issue at ~60 % behind it, not by HBM (TA / TCP / SQ
    # counters of reg_validate_k: profiles/r02_pmc_reg_validate.txt)
    queries = float(st["validations"]) * n
    emit("C4 compute_transformation_ransac 200k<->200k x 100k hyp", ms=dt * 1e3, ms_first=dts[0] * 1e3, hyp_per_s=100_000 / dt,
         validations=st["validations"], fitness=st["fitness"], best_index=st["best_index"],
         pose_err=float(np.abs(T - d["T"]).max()),
         nn_fp32_screen=st["nn_fp32_screen"], nn_screen_fallbacks=st["nn_screen_fallbacks"],
         candidate_cache={"pairs_exact": st["lds_wave_hypotheses"], "pairs_left_as_bounds": st["global_wave_hypotheses"],
                          "note": "(256-point tile, hypothesis) pairs of the validation: every query answered from registers under a "
                                  "certificate / an upper bound of the count and a lower bound of the sum for the pruning, walked only if "
                                  "the hypothesis survives it (m3d_reg_cache.hip)"},
         roofline={"bound": "hbm (algorithmic unit of SURVEY 8(d): 24 B per (hypothesis, point) query); the kernels' own limits are VALU "
                            "issue (reg_validate_cached_k: ~315 instructions per 64 queries, no memory access) and the L1's gather rate "
                            "(reg_validate_pairs_k: the walk of the far poses)",
                   "kernel": "m3d::reg_validate_cached_k + m3d::reg_validate_pairs_k<true>", "achieved": queries * 24.0 / dt / 1e9,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": queries * 24.0 / dt / 1e9 / HBM_PEAK_GBS, "queries": queries, "queries_per_s": queries / dt,
                   "note": "whole call's wall clock (setup, grid, 173 MB of neighbour lists, cache builds, replay included); queries = "
                           "validations x source points, the nominal unit -- the pruning phases evaluate ~73 % of them"},
         cpu_baseline=cpu("validation_baseline", d["src"], d["dst"], d["T"], 0.03, 6.0))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        T2, st2 = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=100_000,
                                           edge_length_threshold=0.9, confidence=0.999, seed=17)
        ts.append((time.perf_counter() - t0) * 1e3)
    q2 = float(st2["validations"]) * n
    emit("C4 same with the reference's confidence 0.999", ms=sorted(ts)[1], ms_first=ts[0],
         iterations=st2["iterations"], validations=st2["validations"], pose_err=float(np.abs(T2 - d["T"]).max()),
         roofline={"bound": "latency: a handful of validations (the loop ends after ~24 iterations), each a dependent chain of short launches "
                            "and host round trips; on the 8(d) unit", "kernel": "m3d::reg_validate_k<false>",
                   "achieved": q2 * 24.0 / (sorted(ts)[1] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": q2 * 24.0 / (sorted(ts)[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, "queries": q2},
         cpu_baseline={"see": "the line above: the same kd-tree validation, ms_per_validation x validations + the sequential loop"})
    # the step the reference's examples chain next: point-to-point ICP on the RANSAC pose (max distance 0.02)
    ts = []
    for _ in range(3):      # one-call entry point: median of three (the first call also warms the block free list)
        t0 = time.perf_counter()
        T3, st3 = capi.registration_icp(d["src"], d["dst"], 0.02, T2)
        ts.append((time.perf_counter() - t0) * 1e3)
    q3 = float(st3["iterations"]) * n

    def icp_cpu():
        import oracle
        t0 = time.perf_counter()
        m = 4000      # (brute-force correspondences: a slice of the source)
        oracle.registration_icp(d["src"][:m], d["dst"], 0.02, T2, max_iter=3)
        dt_c = time.perf_counter() - t0
        return {"value": 3 * m / dt_c, "unit": "correspondence queries/s", "cores": oracle.usable_cpus(), "kind": "port",
                "sample": f"3 ICP iterations of the first {m} source points against all {n} target points, brute-force nearest neighbour "
                          f"(OpenMP), {dt_c:.1f} s", "flags": "-O3 -ffp-contract=off (no -march)"}
    emit("C4 registration_icp on that pose (point-to-point, 0.02, 30 it)", ms=sorted(ts)[1], ms_first=ts[0],
         iterations=st3["iterations"], fitness=st3["fitness"], rmse=st3["inlier_rmse"],
         pose_err=float(np.abs(T3 - d["T"]).max()),
         roofline={"bound": "hbm on the 8(d) unit (24 B per query); the call is a chain of short launches per iteration (grid search, sums, "
                            "3x3 solve on the host)", "kernel": "m3d::icp_nn_k + icp_sums_k", "achieved": q3 * 24.0 / (sorted(ts)[1] * 1e-3) / 1e9,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": q3 * 24.0 / (sorted(ts)[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, "queries": q3},
         cpu_baseline=None if NO_CPU else icp_cpu())

if "N2" in which or ALL:
    # ReconstructionPipeline::GlobalRegistration over many fragment pairs (src/pipeline.cpp:428-439: one
    # std::thread per pair) -- m3d_global_registration_batch on ONE device with 1 / 2 / 4 / 8 pairs in flight (lanes).  Fragments
    # of M3D_N2_POINTS points (default 50 000: a 3 m fragment at 1.4 cm voxels), 33-D descriptors, 12 fragments' worth of pairs
    # drawn from 6 distinct synthetic pairs, the reference's defaults (max_iter 100 000, confidence 0.999).
    nfrag = int(os.environ.get("M3D_N2_POINTS", "50000"))
    base = [synth.registration_pair_c4(nfrag, seed=50 + k) for k in range(6)]
    pairs = [(b["src"], b["dst"], b["feat_src"], b["feat_dst"]) for b in base] * 4       # 24 pairs
    seeds = [1000 + k for k in range(len(pairs))]
    vox = 0.03 / 1.4
    capi.global_registration_batch(pairs[:4], vox, seeds=seeds[:4], inflight=1)            # warm the lanes' pools
    old_lanes = capi.set_config(lanes=8)
    ref = None
    rows = {}
    for inflight in (1, 2, 4, 8):
        capi.global_registration_batch(pairs[:inflight * 2], vox, seeds=seeds[:inflight * 2], inflight=inflight)   # warm every lane
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            res = capi.global_registration_batch(pairs, vox, seeds=seeds, inflight=inflight, want_stats=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if ref is None:
            ref = res
        same = all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(ref, res))
        rows[inflight] = {"ms": best * 1e3, "pairs_per_s": len(pairs) / best, "identical_to_serial": bool(same),
                          "lanes_used": len({r[3]["lane"] for r in res})}
    # the same 24 pairs as the reference holds them: 12 fragments resident for the call, pairs by index
    frs = [b["src"] for b in base] + [b["dst"] for b in base]
    fes = [b["feat_src"] for b in base] + [b["feat_dst"] for b in base]
    ipairs = [(k, 6 + k) for k in range(6)] * 4
    resident = {}
    for inflight in (1, 4, 8):
        capi.register_fragment_pairs(frs, fes, ipairs[:inflight * 2], vox, seeds=seeds[:inflight * 2], inflight=inflight)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            res = capi.register_fragment_pairs(frs, fes, ipairs, vox, seeds=seeds, inflight=inflight, want_stats=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        same = all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(ref, res))
        resident[inflight] = {"ms": best * 1e3, "pairs_per_s": len(ipairs) / best, "identical_to_serial": bool(same)}
        if inflight == 1:   # a pair alone on resident fragments (the later ones of the call: their fragments are there already)
            late = [r[3] for r in res[12:]]
            resident[inflight]["per_pair_ms"] = {k: float(np.median([st[k] for st in late])) for k in ("ms_match", "ms_ransac", "ms_info", "ms_total")}
    capi.restore_config(old_lanes)
    st = {k: float(np.median([r[3][k] for r in ref])) if k != "n_matches" else ref[0][3][k] for k in ("ms_match", "ms_ransac", "ms_info", "ms_total", "n_matches")}
    def n2_cpu():
        import oracle
        m = 8000      # (brute-force matcher: a capped fragment)
        b0 = synth.registration_pair_c4(m, seed=50)
        oracle.set_omp_threads(oracle.usable_cpus())
        t0 = time.perf_counter()
        oracle.global_registration(b0["src"], b0["dst"], b0["feat_src"], b0["feat_dst"], vox, seed=1000)
        dt_c = time.perf_counter() - t0
        return {"value": 1.0 / dt_c, "unit": f"pairs/s at {m} points per fragment", "cores": oracle.usable_cpus(), "kind": "port",
                "sample": f"one pair of {m}-point fragments (the GPU's have {nfrag}: the matcher's work goes with the square), {dt_c:.1f} s: "
                          "brute-force mutual NN (OpenMP) + the sequential RANSAC loop + the information matrix",
                "flags": "-O3 -ffp-contract=off (no -march)"}
    emit(f"N2 global_registration_batch {len(pairs)} pairs of {nfrag} pts x 33-D, one device", accepted=int(sum(r[0] for r in ref)),
         per_pair_serial_ms={k: st[k] for k in ("ms_match", "ms_ransac", "ms_info", "ms_total")}, matches=st["n_matches"],
         in_flight=rows, speedup_over_serial={k: rows[1]["ms"] / v["ms"] for k, v in rows.items()},
         fragments_resident=resident, resident_speedup_over_serial_batch={k: rows[1]["ms"] / v["ms"] for k, v in resident.items()},
         note="pairs are independent (no collective); what overlaps is one pair's uploads and host-side steps (cross-check, "
              "RANSAC replay, grid set-up) with another pair's kernels",
         roofline={"bound": "mfma (the matcher is 2/3 of a pair) on the fp16 screen's flop", "kernel": "m3d::nn16_scan_k (per pair) + the pair's "
                            "registration", "achieved": 2.0 * nfrag * nfrag * 48 * (1 + 3.0 / 16.0) * rows[4]["pairs_per_s"] / 1e12,
                   "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s (fp16 MFMA, 4 pairs in flight)",
                   "frac": 2.0 * nfrag * nfrag * 48 * (1 + 3.0 / 16.0) * rows[4]["pairs_per_s"] / 1e12 / MFMA_F16_PEAK_TFLOPS},
         cpu_baseline=None if NO_CPU else n2_cpu())

if "LANES" in which or ALL:
    # lanes (m3d_driver.hpp): independent callers on ONE device -- T host threads, each with its own resident 200 000-point cloud,
    # each running fit_plane(thr 0.01, 1000 iterations, probability 0.9999) in a loop, and the same with the one-shot m3d_fit_plane
    # from host arrays (upload + fit + list back per call); fits per second of all threads together.
    import threading
    npts = 200_000
    clouds_xyz = [synth.plane_cloud_c1(npts, 100 + t) for t in range(8)]
    old_cfg = capi.set_config(lanes=8, kernel_timing=0)
    rows = {}
    for mode in ("resident", "one_shot"):
        rows[mode] = {}
        for T in (1, 2, 4, 8):
            clouds = [capi.Cloud(clouds_xyz[t]) for t in range(T)] if mode == "resident" else None
            reps = 300 if mode == "resident" else 100
            start = threading.Barrier(T + 1)

            def work(t):
                f = (lambda: clouds[t].fit(0, 0.01, 1000, 0.9999, seed=7, copy=False)) if mode == "resident" else \
                    (lambda: capi.fit(0, clouds_xyz[t], None, 0.01, 1000, 0.9999, seed=7, copy=False))
                for _ in range(10):
                    f()
                start.wait()
                for _ in range(reps):
                    f()
                start.wait()

            ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            for th in ths:
                th.start()
            start.wait()
            t0 = time.perf_counter()
            start.wait()
            dt = time.perf_counter() - t0
            for th in ths:
                th.join()
            rows[mode][T] = {"fits_per_s": T * reps / dt, "ms_per_fit_per_thread": dt / reps * 1e3}
            if clouds:
                for c in clouds:
                    c.close()
    # ... and the same fits handed over as ONE call from ONE python thread: m3d_cloud_fit_batch, the clouds on lanes of their own
    # (m3d_cloud_create_lane) -- the loop runs on threads of the library, the interpreter's lock is released once
    rows["batch"] = {}
    for T in (1, 2, 4, 8):
        clouds = [capi.Cloud(clouds_xyz[t], lane=t) for t in range(T)]
        per = 200
        jobs = [(clouds[t], 0, 0.01, 1000, 0.9999, 7) for _ in range(per) for t in range(T)]
        capi.fit_batch(jobs[: 10 * T], want_inliers=False)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            res = capi.fit_batch(jobs, want_inliers=False)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rows["batch"][T] = {"fits_per_s": len(jobs) / best, "ms_per_fit_per_lane": best / per * 1e3}
        for c in clouds:
            c.close()
    capi.restore_config(old_cfg)
    capi.set_config(kernel_timing=1)
    def lanes_cpu():
        import oracle
        oracle.set_omp_threads(oracle.usable_cpus())
        t0 = time.perf_counter()
        reps_c = 5
        for _ in range(reps_c):
            oc = oracle.fit(0, clouds_xyz[0], None, thr=0.01, max_iter=1000, prob=0.9999, seed=7, lookahead=8 * oracle.usable_cpus())
        dt_c = (time.perf_counter() - t0) / reps_c
        return {"value": 1.0 / dt_c, "unit": "fits/s", "cores": oracle.usable_cpus(), "kind": "port",
                "sample": f"the same fit ({npts} points, adaptive stop after {oc.iterations} iterations), {reps_c} times, the records of the next "
                          "hypotheses computed ahead by an OpenMP team", "flags": "-O3 -ffp-contract=off (no -march)"}
    emit(f"LANES fit_plane {npts} pts x 1000 iterations (adaptive stop), T threads on one device", **rows,
         speedup={m: {T: rows[m][T]["fits_per_s"] / rows[m][1]["fits_per_s"] for T in rows[m]} for m in rows},
         roofline={"bound": "latency: a fit of a few dozen hypotheses is a chain of ~8 short launches and one host wait (~80 us); on the "
                            "8(d) unit (24 B x points per scored hypothesis) it is far from any bandwidth",
                   "kernel": "the whole fit", "achieved": rows["resident"][4]["fits_per_s"] * npts * 24.0 * 38 / 1e9, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s (38 hypotheses per fit, 4 threads)", "frac": rows["resident"][4]["fits_per_s"] * npts * 24.0 * 38 / 1e9 / HBM_PEAK_GBS},
         cpu_baseline=None if NO_CPU else lanes_cpu())

if "N3" in which or ALL:
    # EstimateNormalsFromMap at the reference example's size (848 x 480, k = 3) and at 4 Mpixel
    def n3_cpu(xyz, w, h, k):
        import oracle
        t0 = time.perf_counter()
        oracle.normals_from_map(xyz, w, h, k)
        dt_c = time.perf_counter() - t0
        return {"value": w * h / dt_c / 1e6, "unit": "Mpixel/s", "cores": 1, "kind": "port",
                "sample": f"the whole {w} x {h} map once, {dt_c * 1e3:.0f} ms (src/normal_estimation.cpp:64-207 is a serial loop)",
                "flags": "-O3 -ffp-contract=off (no -march)"}
    for (w, h, k) in ((848, 480, 3), (2048, 2048, 5)):
        rng = np.random.default_rng(1)
        u, v = np.meshgrid(np.arange(w), np.arange(h))
        z = 1.0 + 0.001 * u + 0.002 * v + rng.normal(0, 1e-4, (h, w))
        xyz = np.stack([(u - w / 2) / 500.0 * z, (v - h / 2) / 500.0 * z, z], -1).reshape(-1, 3)
        capi.normals_from_map(xyz, w, h, k)
        t0 = time.perf_counter()
        nrm, ms_dev = capi.normals_from_map(xyz, w, h, k, want_ms=True)
        dt = time.perf_counter() - t0
        W, H = w + 2 * k, h + 2 * k
        bytes_alg = w * h * 24.0 * 2 + W * H * 8.0 * 10 * 3     # map in, normals out, moment images written+read, sums written
        emit(f"N3 estimate_normals {w}x{h} k={k}", ms_total=dt * 1e3, ms_device=ms_dev, mpixel_per_s=w * h / (ms_dev * 1e3),
             device_GBps=bytes_alg / (ms_dev * 1e-3) / 1e9,
             roofline={"bound": "hbm", "kernel": "m3d::nm_box_sum_k + nm_normals_k + nm_moments_k", "achieved": bytes_alg / (ms_dev * 1e-3) / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_alg / (ms_dev * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "device time of the three kernels (HIP events); the box sums keep the reference's summation order "
                               "(a serial recurrence along every row and column), which is what holds them below the bandwidth; "
                               "the whole call is bound by the host link (map in, normals out)"},
             cpu_baseline=None if NO_CPU else n3_cpu(xyz, w, h, k))

if "N4" in which or ALL:
    # DetectBoundaryPoints on a 500 k-point planar patch (the inliers of a fit_plane), Hybrid(0.02, 30)
    rng = np.random.default_rng(3)
    nb = 500_000
    uv = rng.uniform(0, 3.0, (nb, 2))
    pp = np.c_[uv[:, 0], uv[:, 1], 0.2 * uv[:, 0] + rng.normal(0, 1e-3, nb)]
    capi.detect_boundary_points(pp[:1000], None, 2, 0.02, 30, 90.0)
    ts = []
    for _ in range(3):      # one-call entry point: median of three (the first full-size call also sizes the block free list)
        t0 = time.perf_counter()
        bidx = capi.detect_boundary_points(pp, None, 2, 0.02, 30, 90.0)
        ts.append((time.perf_counter() - t0) * 1e3)
    # per point: a hybrid search (27 cells, ~60 candidates, sorted insertion of the 30 nearest), a 3x3 eigen-problem for the
    # normal, 30 atan2 and their sort -- latency- and LDS-bound per thread; no bandwidth or issue roofline applies cleanly
    def n4_cpu():
        import oracle
        m = 40_000     # (the oracle's search is a uniform grid too, single thread)
        t0 = time.perf_counter()
        oracle.detect_boundary_points(pp[:m] * [1, 1, 1], None, 2, 0.02, 30, 90.0)
        dt_c = time.perf_counter() - t0
        return {"value": m / dt_c, "unit": "points/s", "cores": 1, "kind": "port",
                "sample": f"the first {m} points as a cloud of their own (same density along one edge: the patch is uniform), {dt_c:.1f} s; "
                          "src/boundary_detection.cpp:68-113 runs its loop under OpenMP: x cores at best", "flags": "-O3 -ffp-contract=off (no -march)"}
    # The kernel's own bound, from one counter pass (tools/pmc_boundary.sh -> profiles/r06_pmc_boundary.txt): boundary_k issues
    # N4_VALU_PER_POINT VALU + N4_LDS_PER_POINT LDS instructions per wave of 64 points (one point per LANE: the 30-neighbour insertion
    # sort and the angle sort run in LDS); the issue roofline is those instructions x 4 cycles over the launch's SIMD cycles.
    N4_VALU_PER_POINT, N4_LDS_PER_POINT = N4_COUNTS
    ms4 = sorted(ts)[1]
    n4_issue = nb / 64.0 * (N4_VALU_PER_POINT + N4_LDS_PER_POINT) * 4.0 / (1024 * 2.4e9)     # seconds if every instruction issued back to back
    emit("N4 detect_boundary_points 500k pts Hybrid(0.02, 30), normals estimated", ms=ms4, ms_first=ts[0],
         boundary_points=len(bidx), points_per_s=nb / (ms4 * 1e-3),
         roofline={"bound": "instruction issue (VALU + LDS): per point a 27-cell search with a sorted insertion of the 30 nearest, a 3x3 "
                            "eigen-problem, 30 atan2 and their sort, one point per lane -- divergent loops, nothing streams",
                   "kernel": "m3d::boundary_k", "achieved": nb / 64.0 * (N4_VALU_PER_POINT + N4_LDS_PER_POINT) / (ms4 * 1e-3) / 1e12,
                   "peak": 1024 * 2.4e9 / 4.0 / 1e12, "unit": "T wave-instructions/s (VALU + LDS issue, whole call's wall clock)",
                   "frac": n4_issue / (ms4 * 1e-3), "valu_per_point": N4_VALU_PER_POINT, "lds_per_point": N4_LDS_PER_POINT,
                   "source": "profiles/r06_pmc_boundary.txt (SQ_INSTS_VALU, SQ_INSTS_LDS of boundary_k / points)"},
         cpu_baseline=None if NO_CPU else n4_cpu())

if "C5" in which:
    n = int(os.environ.get("M3D_C5_POINTS", "10000000"))
    pts = synth.room_cloud_c5(n, 6)
    # the binding's defaults: every cluster comes back as an array of its own; the library writes the index lists into a
    # page-locked scratch the binding keeps (first call: + its allocation).  copy=False (views of a pageable array the
    # library fills through staged copies) is the slower way round: 43 against 37 ms
    t0 = time.perf_counter()
    rc, planes, clusters = capi.segment_plane_iterative(pts, 0.01, max_iteration=1000, min_ratio=0.05, seed=19)
    dt = time.perf_counter() - t0
    dts = []
    for _ in range(3):
        t0 = time.perf_counter()
        rc, planes, clusters = capi.segment_plane_iterative(pts, 0.01, max_iteration=1000, min_ratio=0.05, seed=19)
        dts.append(time.perf_counter() - t0)
    dt2 = sorted(dts)[1]
    # HBM-bound by construction (a round = a few hundred hypotheses on what is left of the cloud, then compaction + removal
    # passes over it): algorithmic bytes = per round 24 B x remaining points x 4 (RefineModel's counting and writing pass,
    # the removal's read of both copies) + 24 B x kept points x 2 (both copies written) + the 240 MB upload and transpose
    br = capi.last_segment_ms()       # the library's own clock of the last call: create / round loop / its big rounds
    rem, alg = n, 24.0 * n * 3
    alg_big = alg_tail = 0.0
    for cidx in clusters:
        b = 24.0 * rem * 4 + 24.0 * (rem - len(cidx)) * 2
        alg += b
        if rem > n / 8:
            alg_big += b
        else:
            alg_tail += b
        rem -= len(cidx)
    n_tail = max(br["n_rounds"] - br["n_big_rounds"], 1)
    t_tail = (br["rounds"] - br["big_rounds"]) * 1e-3
    emit("C5 segment_plane_iterative 10M pts (1 GPU, incl. 240 MB upload)", ms_first=dt * 1e3, ms=dt2 * 1e3, rc=rc,
         clusters=[len(c) for c in clusters][:12] + ["... %d more" % max(0, len(clusters) - 12)], planes=len(planes),
         roofline={"bound": "hbm", "kernel": "compact_count_k / compact_write_k / cull_mask_k over the remaining cloud, per round",
                   "achieved": alg / dt2 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / dt2 / 1e9 / HBM_PEAK_GBS,
                   "algorithmic_bytes": alg, "rounds": len(clusters),
                   "phases": {
                       "inside_the_library_ms": br["total"], "binding_ms": dt2 * 1e3 - br["total"],
                       "cloud_create": {"ms": br["cloud_create"], "bytes": 24.0 * n * 3,
                                        "GBps": 24.0 * n * 3 / (br["cloud_create"] * 1e-3) / 1e9,
                                        "bound": "PCIe: 240 MB from the caller's pageable array at ~52 GB/s = 4.6 ms of it"},
                       "big_rounds": {"rounds": br["n_big_rounds"], "ms": br["big_rounds"], "algorithmic_bytes": alg_big,
                                      "GBps": alg_big / (br["big_rounds"] * 1e-3) / 1e9,
                                      "frac": alg_big / (br["big_rounds"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "note": "rounds on more than an eighth of the cloud (the six walls, floor, ceiling)"},
                       "tail_rounds": {"rounds": n_tail, "ms": t_tail * 1e3, "us_per_round": t_tail * 1e6 / n_tail,
                                       "algorithmic_bytes": alg_tail, "GBps": alg_tail / t_tail / 1e9,
                                       "frac": alg_tail / t_tail / 1e9 / HBM_PEAK_GBS,
                                       "note": "1000 hypotheses on ~1 M clutter points each, the round's kernels back to back: ~58 us of box tests and scoring (latency-bound launches, 8 % of the (tile, hypothesis) pairs survive), ~24 us of RefineModel's compaction + the partition in creation order at 3.6 TB/s, ~8 us for the previous round's tombstone pass riding in minimal_fit_k's launch; no host wait but the records' (profiles/r03_c5_round_timeline.txt)"}},
                   "note": "whole call (PCIe upload of 240 MB and 76 MB of index lists back included)"},
         cpu_baseline=cpu("segmentation_baseline", pts, 0.01, 1000, 0.05, 19, [c for c in clusters if len(c) > n / 8], 14.0))
