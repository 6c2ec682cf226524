"""BASELINE.json configurations C2, C3, C4 and C5 compared with the ORACLE at their FULL sizes ("Next round" 1).

The oracle's sequential driver with an OpenMP look-ahead (orc_fit_parallel: the records of the next hypotheses are
computed ahead by a thread team and consumed in index order -- outputs identical to orc_fit bit for bit,
tests/test_oracle_primitives.py::test_fit_lookahead_is_the_sequential_loop) does C2 in a few seconds and C3 in tens of
seconds on the GPU box's host cores.  Compared per configuration:

  * the fit: ret, best_index, number of valid hypotheses, iterations run, the FULL inlier index list (array_equal),
    refined parameters within 1e-9 (the GeneralFit sums are order-free on the GPU);
  * ALL per-hypothesis records of the dense path (m3d_cloud_score_range: no culling-based pruning): valid flags,
    counts, minimal models bit for bit;
  * the records of the production path (tile culling + bound-and-prune, m3d_cloud_score_shard): a record is either the
    oracle's count, or it was pruned (reported as 0) -- and then the oracle's count does not exceed the best count of the
    hypotheses before it, i.e. the pruned hypothesis could not have changed the replay.

C5: the 10 M-point room through segment_plane_iterative -- the whole call with the reference's default
max_iteration = 100, and the whole call with BASELINE's 1000 per round (171 clusters) -- cluster index lists bit-equal.
"""
import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(np.nan_to_num(np.asarray(a, dtype=np.float64), nan=-7.0)).view(np.uint64)


def _compare_fit_and_records(capi, orc, kind, pts, nrm, thr, H, seed, lookahead=512):
    n = len(pts)
    o = orc.fit(kind, pts, nrm, thr=thr, max_iter=H, prob=1.0, seed=seed, trace=True, lookahead=lookahead)
    assert o.iterations == H                                    # probability 1: the loop runs to the end
    with capi.Cloud(pts, nrm) as c:
        g = c.fit(kind, thr, H, 1.0, seed=seed)
        # --- the fit -------------------------------------------------------------------------------------------------
        assert g.ret == o.ret
        assert g.stats["best_index"] == o.best_index
        assert g.stats["count"] == o.count and g.stats["iterations"] == o.iterations
        assert g.stats["fitness"] == o.fitness
        assert len(g.inliers) == len(o.inliers) == round(o.fitness * n)
        assert np.array_equal(g.inliers, o.inliers)             # the full list, e.g. 502 332 indices on C2
        assert np.allclose(g.params, o.params, rtol=0, atol=1e-9)
        # --- every hypothesis' record, dense path ----------------------------------------------------------------------
        samples = capi.draw_samples(n, kind, H, seed)
        assert np.array_equal(samples.astype(np.uint64), o.trace["samples"][:, : samples.shape[1]])
        val, mod, cnt = c.score_range(kind, thr, samples, 0, H)
        assert np.array_equal(val.astype(np.int32), o.trace["valid"])
        assert np.array_equal(cnt.astype(np.uint64), o.trace["counts"])
        ok = o.trace["valid"].astype(bool)
        assert np.array_equal(_bits(mod[ok]), _bits(o.trace["models"][ok]))
        # --- the production path's records (culled, pruned) ------------------------------------------------------------
        sm = c.make_sampler(kind, seed)
        try:
            v2, c2 = c.score_shard(sm, thr, 0, H, H, 1, 0)
        finally:
            sm.close()
        assert len(c2) == H and np.array_equal(v2.astype(np.int32), o.trace["valid"])
        oc = o.trace["counts"].astype(np.int64)
        exact = c2.astype(np.int64) == oc
        best_before = np.concatenate([[0], np.maximum.accumulate(oc)[:-1]])
        pruned = ~exact
        assert np.all(c2[pruned] == 0) and np.all(oc[pruned] <= best_before[pruned])
        assert exact[o.best_index]
        # the serial error sum of the winner (the tie rule's input, ransac.h:637) is the oracle's, bit for bit
        _, err = c.exact_error(kind, thr, o.trace["models"][o.best_index])
        assert np.float64(err).view(np.uint64) == np.float64(o.trace["errors"][o.best_index]).view(np.uint64)
    return o, g, int(pruned.sum())


def test_c2_full_size_vs_oracle(capi, orc):
    """BASELINE configs[1]: fit_plane, 1 M points, 10 000 hypotheses, thr 0.01, probability 1, sampler seed 11."""
    pts = synth.plane_cloud_c2(1_000_000, 2)
    o, g, n_pruned = _compare_fit_and_records(capi, orc, 0, pts, None, 0.01, 10_000, 11)
    assert len(o.inliers) > 490_000 and n_pruned > 1000          # the pruning really was exercised


def test_c3_sphere_full_size_vs_oracle(capi, orc):
    """BASELINE configs[2], sphere: 1 M points, 50 000 hypotheses."""
    pts = synth.sphere_cloud_c3(1_000_000, 4)
    o, g, n_pruned = _compare_fit_and_records(capi, orc, 1, pts, None, 0.01, 50_000, 13)
    assert len(o.inliers) > 450_000 and n_pruned > 1000


def test_c3_cylinder_full_size_vs_oracle(capi, orc):
    """BASELINE configs[2], cylinder: 1 M points with normals, 50 000 hypotheses."""
    pts, nrm = synth.cylinder_cloud_c3(1_000_000, 3)
    o, g, n_pruned = _compare_fit_and_records(capi, orc, 2, pts, nrm, 0.01, 50_000, 13)
    assert len(o.inliers) > 400_000 and n_pruned > 1000


def test_c2_adaptive_stop_full_size_vs_oracle(capi, orc):
    """The same cloud through the reference's own defaults (probability 0.9999: the adaptive bound ends the loop after a
    few dozen hypotheses) -- the chunked pipeline's early stop against the sequential loop."""
    pts = synth.plane_cloud_c2(1_000_000, 2)
    with capi.Cloud(pts) as c:
        for seed in (11, 12, 13):
            o = orc.fit(0, pts, None, thr=0.01, max_iter=1000, prob=0.9999, seed=seed, lookahead=16)
            g = c.fit(0, 0.01, 1000, 0.9999, seed=seed)
            assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (
                o.ret, o.best_index, o.count, o.iterations)
            assert o.iterations < 1000
            assert np.array_equal(g.inliers, o.inliers) and np.allclose(g.params, o.params, rtol=0, atol=1e-9)


@pytest.fixture(scope="module")
def room():
    return synth.room_cloud_c5(10_000_000, 6)


def test_c5_default_iterations_full_size_vs_oracle(capi, orc, room):
    """BASELINE configs[4]'s 10 M-point scene through SegmentPlaneIterative with the reference's default
    max_iteration = 100 (iterative_plane_segmentation.h:25-28): EVERY cluster's index list equals the oracle's."""
    n = len(room)
    ro, po, co = orc.segment_plane_iterative(room, 0.01, max_iteration=100, min_ratio=0.05, seed=19, lookahead=100)
    rg, pg, cg = capi.segment_plane_iterative(room, 0.01, max_iteration=100, min_ratio=0.05, seed=19)
    assert ro == 0 and rg == 1 and len(co) == len(cg) >= 6
    for a, b in zip(co, cg):
        assert np.array_equal(a, b)
    assert np.allclose(po, pg, rtol=0, atol=1e-9)
    assert sum(len(x) for x in cg) >= int(0.95 * n)


def test_c5_baseline_iterations_full_size_vs_oracle(capi, orc, room):
    """BASELINE's own setting, 1000 hypotheses per round, the WHOLE call: the six big planes (10^10 pair evaluations each
    for the oracle's thread team) and the ~165 picks out of the clutter that follow -- every cluster's index list and
    plane against the oracle.  (These are the rounds whose RefineModel the library starts on the device's own pick, before
    the host has replayed the round.)"""
    ro, po, co = orc.segment_plane_iterative(room, 0.01, max_iteration=1000, min_ratio=0.05, seed=19, lookahead=250)
    rg, pg, cg = capi.segment_plane_iterative(room, 0.01, max_iteration=1000, min_ratio=0.05, seed=19)
    assert ro == 0 and rg == 1 and len(co) == len(cg) > 100
    for a, b in zip(co, cg):
        assert np.array_equal(a, b)
    assert np.allclose(po, pg, rtol=0, atol=1e-9)


def test_c4_full_size_vs_oracle(capi, orc):
    """BASELINE configs[3] at full size: the mutual matcher on 200 k x 200 k x 33 descriptors against the oracle's fp64
    brute force (ALL pairs; ~1 minute of the box's host cores), then compute_transformation_ransac on those correspondences
    with the reference's own confidence (0.999: 24 iterations, 5 validations -- each a 200 k x 200 k exact search in the
    oracle): T bit for bit, iterations, validations, est_k, fitness; then the first 100 iterations of BASELINE's FORCED run
    (confidence 1.0: ~25 validations, ~40 s of oracle) the same way.  (300 iterations / 76 validations:
    tools/c4_full_size_vs_oracle.py, profiles/r03_c4_full_size_vs_oracle.txt.)"""
    n = 200_000
    d = synth.registration_pair_c4(n, seed=5)
    g0, g1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    o0, o1 = orc.match_mutual_nn(d["feat_src"], d["feat_dst"])
    assert len(o0) > 0.25 * n
    assert np.array_equal(g0.astype(np.int64), o0) and np.array_equal(g1.astype(np.int64), o1)
    assert capi.match_last_fallbacks() < 0.001 * n
    kw = dict(threshold=0.03, max_iter=100_000, edge_length_threshold=0.9, confidence=0.999, seed=17)
    T, st = capi.registration_ransac(d["src"], d["dst"], g0, g1, **kw)
    o = orc.registration_ransac(d["src"], d["dst"], o0, o1, thr=0.03, max_iter=100_000, edge_thr=0.9, confidence=0.999, seed=17)
    assert np.array_equal(T.view(np.uint64), o.T.view(np.uint64))
    assert (st["iterations"], st["validations"], st["est_k"], st["best_index"]) == (o.iterations, o.validations, o.est_k, o.best_index)
    assert st["fitness"] == o.fitness and abs(st["inlier_rmse"] - o.inlier_rmse) <= 1e-12
    # the forced run: no early stop, every hypothesis that passes the checkers is validated
    T, st = capi.registration_ransac(d["src"], d["dst"], g0, g1, threshold=0.03, max_iter=100, edge_length_threshold=0.9,
                                     confidence=1.0, seed=17)
    o = orc.registration_ransac(d["src"], d["dst"], o0, o1, thr=0.03, max_iter=100, edge_thr=0.9, confidence=1.0, seed=17)
    assert o.iterations == 100 and o.validations >= 10
    assert np.array_equal(T.view(np.uint64), o.T.view(np.uint64))
    assert (st["iterations"], st["validations"], st["est_k"], st["best_index"]) == (o.iterations, o.validations, o.est_k, o.best_index)
    assert st["fitness"] == o.fitness
