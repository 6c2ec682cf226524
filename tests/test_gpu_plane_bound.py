"""The planes' histogram bound (m3d_bound.hip: tile_frames_k, plane_bound_k, bound_keep_k) against the oracle.

A hypothesis the bound drops must be one the sequential loop (ransac.h:571-613) could not have noticed: its record is 0
and the oracle's count for it does not exceed the best count of the hypotheses before it.  Everything the fit returns
must be the oracle's, with the bound on, off and forced -- and the bound must really have spared work.
"""
import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu


def _fit_all_modes(capi, orc, pts, thr, H, seed, lookahead=256):
    o = orc.fit(0, pts, None, thr=thr, max_iter=H, prob=1.0, seed=seed, trace=True, lookahead=lookahead)
    pairs = {}
    for mode in (0, 1, 2):
        old = capi.set_config(plane_bound=mode)
        try:
            with capi.Cloud(pts) as c:
                g = c.fit(0, thr, H, 1.0, seed=seed)
                sm = c.make_sampler(0, seed)
                try:
                    v2, c2 = c.score_shard(sm, thr, 0, H, H, 1, 0)
                finally:
                    sm.close()
        finally:
            capi.restore_config(old)
        assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.ret, o.best_index, o.count, o.iterations), mode
        assert np.array_equal(g.inliers, o.inliers), mode
        assert np.allclose(g.params, o.params, rtol=0, atol=1e-9 * max(1.0, float(np.abs(o.params).max()))), mode   # (order-free GeneralFit sums)
        oc = o.trace["counts"].astype(np.int64)
        exact = c2.astype(np.int64) == oc
        best_before = np.concatenate([[0], np.maximum.accumulate(oc)[:-1]])
        assert np.array_equal(v2.astype(np.int32), o.trace["valid"]), mode
        assert np.all(c2[~exact] == 0) and np.all(oc[~exact] <= best_before[~exact]), mode
        if o.best_index >= 0:
            assert exact[o.best_index], mode
        pairs[mode] = (g.stats["pairs_scored"], int((~exact).sum()))
    return o, pairs


def test_bound_prunes_and_matches_oracle_c2_shape(capi, orc):
    """C2's cloud at a size the oracle finishes in seconds: 150 000 points (293 tiles), 3000 hypotheses."""
    pts = synth.plane_cloud_c2(150_000, 2)
    o, pairs = _fit_all_modes(capi, orc, pts, 0.01, 3000, 11)
    assert len(o.inliers) > 70_000
    assert pairs[2][0] < 0.8 * pairs[0][0], pairs          # fewer (tile, hypothesis) pairs counted point by point
    assert pairs[2][1] > pairs[0][1], pairs                # ... because more hypotheses were pruned
    assert abs(pairs[1][0] - pairs[0][0]) < 0.02 * pairs[0][0], pairs   # (3000 hypotheses on 293 tiles: too little work for the default to engage)


@pytest.mark.parametrize("case", ["exact_plane", "nan_points", "two_planes", "duplicates", "tiny_threshold", "huge_offset"])
def test_bound_degenerate_clouds(capi, orc, case):
    rng = np.random.default_rng(3)
    n = 40_000
    if case == "exact_plane":        # zero thickness: the frame's scale is floored, every point in one bin
        xy = rng.uniform(-1, 1, size=(n, 2))
        pts = np.column_stack([xy, 0.25 * xy[:, 0] - 0.5 * xy[:, 1] + 0.125])
        pts[: n // 4] = rng.uniform(-1, 1, size=(n // 4, 3))
    elif case == "nan_points":
        pts = synth.plane_cloud_c1(n, 5)
        pts[rng.integers(0, n, 3000), rng.integers(0, 3, 3000)] = np.nan
        pts[rng.integers(0, n, 50), 0] = np.inf
    elif case == "two_planes":       # tiles on the intersection line hold two surfaces
        pts = synth.plane_cloud_c2(n, 9)
    elif case == "duplicates":       # whole tiles of one repeated point (extent 0: no frame)
        pts = synth.plane_cloud_c1(n, 6)
        pts[: n // 2] = pts[0]
    elif case == "tiny_threshold":
        pts = synth.plane_cloud_c1(n, 7)
    else:                            # coordinates far from the origin: the rounding terms of the bound matter
        pts = synth.plane_cloud_c1(n, 8) + np.array([4.0e5, -3.0e5, 2.0e5])
    thr = 1e-5 if case == "tiny_threshold" else 0.01
    _fit_all_modes(capi, orc, np.ascontiguousarray(pts), thr, 2500, 21, lookahead=128)


def test_bound_in_sharded_windows(capi, orc):
    """m3d_cloud_score_shard, three ranks' slices in turn: every slice prunes with the bound against the same incumbent."""
    pts = synth.plane_cloud_c2(150_000, 4)
    H = 3000
    o = orc.fit(0, pts, None, thr=0.01, max_iter=H, prob=1.0, seed=5, trace=True, lookahead=256)
    oc = o.trace["counts"].astype(np.int64)
    best_before = np.concatenate([[0], np.maximum.accumulate(oc)[:-1]])
    old = capi.set_config(plane_bound=2)
    try:
        with capi.Cloud(pts) as c:
            sl = 512
            for rank in range(3):
                sm = c.make_sampler(0, 5)
                try:
                    v2, c2 = c.score_shard(sm, 0.01, 0, H, sl, 3, rank)
                finally:
                    sm.close()
                mine = np.concatenate([np.arange(j * sl, min(H, (j + 1) * sl)) for j in range(rank, -(-H // sl), 3)])
                got = c2.astype(np.int64)
                assert len(got) == len(mine)
                exact = got == oc[mine]
                assert np.all(got[~exact] == 0) and np.all(oc[mine][~exact] <= best_before[mine][~exact]), rank
    finally:
        capi.restore_config(old)


def _cloud_for(case, n, rng):
    if case == "c2":
        return synth.plane_cloud_c2(n, 2)
    if case == "c1":
        return synth.plane_cloud_c1(n, 5)
    if case == "exact_plane":
        xy = rng.uniform(-1, 1, size=(n, 2))
        pts = np.column_stack([xy, 0.25 * xy[:, 0] - 0.5 * xy[:, 1] + 0.125])
        pts[: n // 4] = rng.uniform(-1, 1, size=(n // 4, 3))
        return pts
    if case == "nan_points":
        pts = synth.plane_cloud_c1(n, 5)
        pts[rng.integers(0, n, 3000), rng.integers(0, 3, 3000)] = np.nan
        return pts
    if case == "huge_offset":
        return synth.plane_cloud_c1(n, 8) + np.array([4.0e5, -3.0e5, 2.0e5])
    if case == "clutter":            # no structure at all: every frame is arbitrary
        return rng.uniform(-1, 1, size=(n, 3))
    if case == "curved":             # a sphere's shell: tiles are thin but not flat
        return synth.sphere_cloud_c3(n, 4)
    raise ValueError(case)


@pytest.mark.parametrize("case,thr", [("c2", 0.01), ("c2", 0.002), ("c2", 0.05), ("c1", 0.01), ("exact_plane", 0.01),
                                      ("nan_points", 0.01), ("huge_offset", 0.01), ("clutter", 0.02), ("curved", 0.01)])
def test_upper_bound_is_an_upper_bound(capi, case, thr):
    """plane_bound_k on its own (m3d_bench_plane_upper_bounds: every hypothesis on the list, nothing pruned) against the exact
    counts of the dense path: ub >= count for EVERY hypothesis -- and tight where it should be (the best hypothesis of a
    cloud with a dominant plane: within a fifth of its count)."""
    rng = np.random.default_rng(17)
    n, H = 120_000, 4096
    pts = np.ascontiguousarray(_cloud_for(case, n, rng))
    samples = capi.draw_samples(n, 0, H, 23)
    with capi.Cloud(pts) as c:
        ub = c.plane_upper_bounds(thr, samples).astype(np.int64)
        val, _, cnt = c.score_range(0, thr, samples, 0, H, want_models=False)
    cnt = cnt.astype(np.int64)
    ok = val.astype(bool)
    assert ok.sum() > 0.9 * H or case in ("nan_points",)
    bad = np.nonzero(ok & (ub < cnt))[0]
    assert len(bad) == 0, (case, thr, bad[:5], ub[bad[:5]], cnt[bad[:5]])
    if case in ("c2", "c1", "exact_plane") and thr == 0.01:
        b = int(np.argmax(np.where(ok, cnt, -1)))
        assert cnt[b] > 0.3 * n and ub[b] - cnt[b] < 0.2 * cnt[b], (case, int(cnt[b]), int(ub[b]))


@pytest.mark.parametrize("kind,case,thr", [(2, "cylinder_c3", 0.01), (2, "cylinder_c3", 0.003), (2, "cylinder_c3", 0.04), (2, "cylinder_far", 0.01),
                                           (2, "clutter", 0.02), (1, "sphere_c3", 0.01), (1, "sphere_c3", 0.002), (1, "sphere_far", 0.01),
                                           (1, "clutter", 0.02), (1, "cylinder_c3", 0.01), (2, "sphere_c3", 0.01)])
def test_shell_upper_bound_is_an_upper_bound(capi, kind, case, thr):
    """plane_bound_k<1> / <2> on their own (m3d_bench_upper_bounds: every hypothesis on the list, nothing pruned): over one tile a
    sphere's or a cylinder's shell is a slab up to a sagitta (m3d_bound_fp.hpp cyl_pair_ub).  Against the exact counts of the dense
    path: ub >= count for EVERY hypothesis -- scenes at the origin and 300 scene sizes away, thresholds from the noise's size to
    four times the baseline's, clouds of the other shape and clutter -- and tight where it should be (the best hypothesis of a
    cloud with a dominant shell: within 30 % / 40 % of its count)."""
    rng = np.random.default_rng(19 + kind)
    n, H = 150_000, 4096
    nrm = None
    if case.startswith("cylinder"):
        pts, nrm = synth.cylinder_cloud_c3(n, 3)
    elif case.startswith("sphere"):
        pts = synth.sphere_cloud_c3(n, 4)
    else:
        pts = rng.uniform(-1, 1, size=(n, 3))
    if kind == 2 and nrm is None:          # cylinder hypotheses need normals: random ones on a cloud of another shape
        nrm = rng.normal(size=(n, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    if case.endswith("far"):
        pts = pts + np.array([900.0, -700.0, 400.0])
    pts = np.ascontiguousarray(pts)
    samples = capi.draw_samples(n, kind, H, 29)
    with capi.Cloud(pts, nrm if kind == 2 else None) as c:
        ub = c.upper_bounds(kind, thr, samples).astype(np.int64)
        val, _, cnt = c.score_range(kind, thr, samples, 0, H, want_models=False)
    cnt = cnt.astype(np.int64)
    ok = val.astype(bool)
    bad = np.nonzero(ok & (ub < cnt))[0]
    assert len(bad) == 0, (kind, case, thr, bad[:5], ub[bad[:5]], cnt[bad[:5]])
    if (kind, case) in ((2, "cylinder_c3"), (1, "sphere_c3"), (2, "cylinder_far"), (1, "sphere_far")) and thr == 0.01:
        b = int(np.argmax(np.where(ok, cnt, -1)))
        # (at 150 000 points a tile on the shell is ~15 cm across on a radius of 25 cm -- cylinder -- or 50 cm -- sphere --: its own
        # sagitta is of the threshold's size, which the slab has to take in; at C3's million points a tile is 5 cm)
        assert cnt[b] > 0.3 * n and ub[b] - cnt[b] < (0.4 if kind == 2 else 0.3) * cnt[b], (kind, case, int(cnt[b]), int(ub[b]))


def test_bound_soak_scaled_and_shifted_scenes(capi, orc):
    """Fixed seed, fixed budget (20 s): plane clouds of random size, scaled by 10^-2 .. 10^2 and moved up to 10^3 scene sizes
    from the origin (the bound evaluates in fp32 beside an fp64 centre value: its margins), thresholds from a third of the
    noise to ten times it, the bound forced on -- the fit must be the oracle's and every upper bound an upper bound."""
    import time
    rng = np.random.default_rng(20260930)
    old = capi.set_config(plane_bound=2)
    t_end, n_cases = time.time() + 20.0, 0
    try:
        while time.time() < t_end:
            n = int(rng.integers(15_000, 150_000))
            sd = int(rng.integers(0, 10_000))
            pts = synth.plane_cloud_c2(n, sd) if rng.random() < 0.6 else synth.plane_cloud_c1(n, sd)
            sc = float(10.0 ** rng.uniform(-2, 2))
            pts = np.ascontiguousarray(pts * sc + rng.uniform(-1, 1, 3) * sc * float(10.0 ** rng.uniform(0, 3)))
            thr = float(rng.choice([0.001, 0.003, 0.01, 0.03])) * sc
            H = int(rng.integers(300, 3000))
            o = orc.fit(0, pts, None, thr=thr, max_iter=H, prob=1.0, seed=sd, lookahead=128)
            samples = capi.draw_samples(n, 0, min(H, 2048), sd)
            with capi.Cloud(pts) as c:
                g = c.fit(0, thr, H, 1.0, seed=sd)
                ub = c.plane_upper_bounds(thr, samples).astype(np.int64)
                val, _, cnt = c.score_range(0, thr, samples, 0, len(samples), want_models=False)
            case = (n, sd, sc, thr, H)
            assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.ret, o.best_index, o.count, o.iterations), case
            assert np.array_equal(g.inliers, o.inliers), case
            ok = val.astype(bool)
            assert np.all(ub[ok] >= cnt.astype(np.int64)[ok]), case
            n_cases += 1
    finally:
        capi.restore_config(old)
    assert n_cases >= 5, n_cases


def test_bound_over_several_chunks(capi, orc):
    """40 000 hypotheses = a short first chunk + two long ones (M3D_CHUNK_CAP 24 576): the later chunks prune against the
    incumbent of the earlier ones (keep_mask_k writes the survivor list, MinimalFit and the box tests run on the pre-stream)."""
    pts = synth.plane_cloud_c2(200_000, 6)
    o, pairs = _fit_all_modes(capi, orc, pts, 0.01, 40_000, 17, lookahead=512)
    assert len(o.inliers) > 95_000
    assert pairs[2][0] < 0.8 * pairs[0][0], pairs


def test_bound_after_a_longer_fit(capi, orc):
    """The bound's scratch (survivor list, tickets) is shared by the fits of a device: a window of 3000 hypotheses, then one of
    1500, then 700, then 3000 again -- each must be the oracle's (round 4 kept the tickets BEHIND the list: a shorter window found
    an older list's ids where it expected zeros, and a block's first workgroup applied the keep rule to a partial sum)."""
    pts = synth.plane_cloud_c2(150_000, 3)
    old = capi.set_config(plane_bound=2)
    try:
        with capi.Cloud(pts) as c:
            for H, seed in ((3000, 5), (1500, 6), (700, 7), (3000, 8), (1100, 9)):
                o = orc.fit(0, pts, None, thr=0.01, max_iter=H, prob=1.0, seed=seed, lookahead=128)
                g = c.fit(0, 0.01, H, 1.0, seed=seed)
                assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.ret, o.best_index, o.count, o.iterations), (H, seed)
                assert np.array_equal(g.inliers, o.inliers), (H, seed)
    finally:
        capi.restore_config(old)


def test_bound_on_a_cloud_that_shrinks(capi, orc):
    """A held cloud: fit (the frames are built), remove the winning minimal model's inliers (SelectByIndex(inliers, invert): the
    sorted copy is re-partitioned, its tiles change, the frames must go), fit again, remove, fit -- every fit the oracle's fit of
    the points that are left, index lists in terms of the cloud as created."""
    pts = synth.plane_cloud_c2(150_000, 12)
    old = capi.set_config(plane_bound=2)
    try:
        with capi.Cloud(pts) as c:
            keep = np.ones(len(pts), dtype=bool)
            for rnd, seed in enumerate((5, 6, 7)):
                rest = np.nonzero(keep)[0]
                o = orc.fit(0, pts[rest], None, thr=0.01, max_iter=2500, prob=1.0, seed=seed, lookahead=128)
                g = c.fit(0, 0.01, 2500, 1.0, seed=seed)
                assert (g.ret, g.stats["best_index"], g.stats["iterations"]) == (o.ret, o.best_index, o.iterations), rnd
                mine = g.inliers.astype(np.int64)
                assert np.array_equal(mine, rest[o.inliers.astype(np.int64)]), rnd
                # RefineModel's list IS the winning minimal model's inlier set (ransac.h:537-543): removing that model's inliers
                # removes exactly the list
                model = c.minimal_model(0, 0.01, capi.draw_samples(len(rest), 0, 2500, seed)[o.best_index])
                assert c.remove_inliers(0, 0.01, model) == len(mine), rnd
                keep[mine] = False
    finally:
        capi.restore_config(old)


@pytest.mark.parametrize("mode", [1, 2])
def test_bound_on_a_cloud_without_structure(capi, orc, mode):
    """Uniform clutter: the lead's best count is a few hundred, the old rule keeps every hypothesis -- in the default mode
    plane_bound_k discards a list that long (nothing worth pruning against), forced it bounds all of them and drops none that
    matters; the fit is the oracle's either way."""
    rng = np.random.default_rng(4)
    pts = np.ascontiguousarray(rng.uniform(-1, 1, size=(120_000, 3)))
    o = orc.fit(0, pts, None, thr=0.01, max_iter=2500, prob=1.0, seed=3, lookahead=128)
    old = capi.set_config(plane_bound=mode)
    try:
        with capi.Cloud(pts) as c:
            for _ in range(2):
                g = c.fit(0, 0.01, 2500, 1.0, seed=3)
                assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.ret, o.best_index, o.count, o.iterations)
                assert np.array_equal(g.inliers, o.inliers)
    finally:
        capi.restore_config(old)
