"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar: bit-exact for everything integer (valid flags, inlier counts, best index, iteration
counts, inlier index lists) AND for the minimal-fit models (same fp64 operation order, correctly
rounded sqrt/div on both sides); refined (GeneralFit) parameters within 1e-9 (the reference sums
serially, the GPU with a fixed tree: north_star tolerance is 1e-5)."""
import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu

PARAM_TOL = 1e-9


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# the scoring paths m3d_config selects between; they must be indistinguishable from outside
PATHS = {"culled": {}, "dense": {"dense_scoring": 1}, "no_early_pick": {"speculative_refine": 0},
         "small_blocks": {"score_groups_per_block": 1, "lead_hypotheses": 64},
         "fp64_only": {"score_fp32_screen": 0, "cull_fp32": 0},
         "single_launch": {"score_phases": 0},   # (default -1: cylinders in three phases with re-pruning in between, the others in one)
         "two_phases": {"score_phases": 2}, "three_phases": {"score_phases": 3},
         "plane_bound_always": {"plane_bound": 2},   # planes: the histogram bound (m3d_bound.hip) at every size (default: long fits of large clouds)
         "plane_bound_off": {"plane_bound": 0},
         "plane_bound_fp64_paths": {"plane_bound": 2, "score_fp32_screen": 0, "cull_fp32": 0}}   # (no fp32 box-test records: plane_bound_k reads the mask words)

# round 4's refuted variants: compiled with `make DEFS=-DM3D_EXPERIMENTAL` only (m3d_kernels.hpp) -- their rows join the matrix
# when the library under test is such a build
EXPERIMENTAL_PATHS = {"mfma": {"score_mfma": 1},   # planes: score_mfma_k, the screen on the matrix pipe
                      "four_wave_wgs": {"score_waves4": 1},
                      "one_pass_compaction": {"compact_one_pass": 1}}   # compact_write_k, ONE: counts published and awaited inside the launch
try:
    from misc3d_amd import capi as _capi_probe
    if _capi_probe.experimental():
        PATHS.update(EXPERIMENTAL_PATHS)
except Exception:      # noqa: BLE001  (no library: the GPU tests cannot run anyway)
    pass


@pytest.fixture(params=sorted(PATHS))
def scoring_path(request, capi):
    """Runs the test once per scoring path: the production culled path (cull_tiles_k + score_screen_k / score_mask_k
    with the device's early pick), the dense kernel (score_k: every tile x every hypothesis; the bench's reference
    point), the culled path without the speculative RefineModel, the culled path with one hypothesis group per
    workgroup, the culled path without any fp32 (fp64 box tests cull_tiles_k, score_mask_k for every model), and the
    culled path with the planes' screen on the matrix pipe (score_mfma_k)."""
    old = capi.set_config(**PATHS[request.param])
    yield request.param
    capi.restore_config(old)


def _clouds(kind, n, seed):
    if kind == 0:
        return synth.plane_cloud_c1(n, seed), None
    if kind == 1:
        return synth.sphere_cloud_c3(n, seed), None
    return synth.cylinder_cloud_c3(n, seed)


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n", [20000, 4097, 777])
def test_score_range_bit_exact(capi, orc, scoring_path, kind, n):
    """minimal_fit_k + score_k + reduce vs MinimalFit + EvaluateModel, hypothesis by hypothesis."""
    pts, nrm = _clouds(kind, n, seed=10 + kind)
    H = 300
    samples = capi.draw_samples(n, kind, H, seed=99)
    with capi.Cloud(pts, nrm) as c:
        valid, models, counts = c.score_range(kind, 0.01, samples)
    ovalid, omodels, ocounts, _ = orc.score_samples(kind, pts, nrm, 0.01, samples.astype(np.uint64))
    assert np.array_equal(valid.astype(bool), ovalid.astype(bool))
    v = ovalid.astype(bool)
    # NaN radii (sphere) compare equal bitwise only if both are NaN: compare with nan-equality
    assert np.array_equal(_bits(models[v]), _bits(omodels[v])) or np.array_equal(
        np.nan_to_num(models[v], nan=-7.0), np.nan_to_num(omodels[v], nan=-7.0))
    assert np.array_equal(counts.astype(np.uint64), ocounts)
    assert counts[v].max() > 0.2 * n  # the structure was found by some hypothesis


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_exact_error_serial_sum(capi, orc, kind):
    """compact(mode 1) + serial_sum_k reproduce EvaluateModel's count and serial error sum bitwise."""
    n = 30000
    pts, nrm = _clouds(kind, n, seed=20 + kind)
    samples = capi.draw_samples(n, kind, 40, seed=5)
    ovalid, omodels, ocounts, oerrs = orc.score_samples(kind, pts, nrm, 0.01, samples.astype(np.uint64))
    order = np.argsort(-ocounts.astype(np.int64))[:5]
    with capi.Cloud(pts, nrm) as c:
        for h in order:
            cnt, err = c.exact_error(kind, 0.01, omodels[h])
            assert cnt == int(ocounts[h])
            assert _bits([err])[0] == _bits([oerrs[h]])[0]


@pytest.mark.parametrize("kind,n,max_iter,prob,seed", [
    (0, 20000, 1000, 0.9999, 7),      # adaptive stop (default arguments of the python API)
    (0, 20000, 300, 1.0, 11),         # every hypothesis evaluated
    (0, 5000, 100, 0.9999, 3),        # C1-like plumbing
    (1, 20000, 400, 0.9999, 13),
    (1, 6000, 200, 1.0, 2),
    (2, 20000, 400, 0.9999, 13),
    (2, 6000, 300, 1.0, 4),
])
def test_fit_matches_oracle(capi, orc, scoring_path, kind, n, max_iter, prob, seed):
    pts, nrm = _clouds(kind, n, seed=30 + kind)
    o = orc.fit(kind, pts, nrm, thr=0.01, max_iter=max_iter, prob=prob, seed=seed)
    g = capi.fit(kind, pts, nrm, threshold=0.01, max_iteration=max_iter, probability=prob, seed=seed)
    assert g.ret == o.ret
    assert g.stats["best_index"] == o.best_index
    assert g.stats["count"] == o.count
    assert g.stats["iterations"] == o.iterations
    assert g.stats["fitness"] == o.fitness
    assert np.array_equal(g.inliers, o.inliers)          # bit-exact inlier index set
    assert np.allclose(g.params, o.params, rtol=0, atol=PARAM_TOL)
    if kind == 2:  # cylinder GeneralFit is a no-op: the parameters are the minimal model, bit for bit
        assert np.array_equal(_bits(g.params), _bits(o.params))


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_box_tests_by_hypothesis_and_by_tile_match_oracle(capi, orc, kind):
    """Windows of 128 groups and more take their fp32 box tests with a lane per HYPOTHESIS (cull_hyp32_k, round 6), smaller ones and
    m3d_config.cull_fp32 = 2 with a lane per tile (cull_tiles32_k): same words, same counters, and either way the oracle's fit.
    20 000 hypotheses: a first chunk of 2048, then one window of 281 groups."""
    pts, nrm = _clouds(kind, 60_000, seed=40 + kind)
    H, seed = 20_000, 21 + kind
    o = orc.fit(kind, pts, nrm, thr=0.01, max_iter=H, prob=1.0, seed=seed, lookahead=512)
    got = []
    # (plane_bound = 2: the histogram bound at this size too -- for cylinders behind a lane-per-hypothesis window it reads the box tests'
    #  per-hypothesis words, cyl_bound_words_k, otherwise it repeats the tests, plane_bound_k<2>)
    for mode, bound in ((1, 1), (2, 1), (1, 2), (2, 2)):
        old = capi.set_config(cull_fp32=mode, plane_bound=bound)
        try:
            g = capi.fit(kind, pts, nrm, threshold=0.01, max_iteration=H, probability=1.0, seed=seed)
        finally:
            capi.restore_config(old)
        assert g.ret == o.ret and g.stats["best_index"] == o.best_index and g.stats["count"] == o.count
        assert np.array_equal(g.inliers, o.inliers)
        assert np.allclose(g.params, o.params, rtol=0, atol=PARAM_TOL)
        got.append(g)
    for g in got[1:]:
        assert np.array_equal(_bits(got[0].params), _bits(g.params))


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("phases", [2, 3])
def test_phased_scoring_matches_oracle(capi, orc, kind, phases):
    """launch_score_phased needs >= 256 tiles and a window of >= 32 groups to engage: 150 000 points x 3000 hypotheses.  The fit
    must be the oracle's, and the phased run must really have evaluated fewer (tile, hypothesis) pairs than the one-launch run
    (hypotheses that leave after a quarter or a half of their tiles)."""
    pts, nrm = _clouds(kind, 150_000, seed=70 + kind)
    o = orc.fit(kind, pts, nrm, thr=0.01, max_iter=3000, prob=1.0, seed=5, lookahead=256)
    res = {}
    for ph in (0, phases):
        old = capi.set_config(score_phases=ph, plane_bound=0)   # (the planes' histogram bound -- where it engages -- prunes on its own account and steps aside for the phases)
        try:
            g = capi.fit(kind, pts, nrm, threshold=0.01, max_iteration=3000, probability=1.0, seed=5)
        finally:
            capi.restore_config(old)
        assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.ret, o.best_index, o.count, o.iterations)
        assert np.array_equal(g.inliers, o.inliers)
        assert np.allclose(g.params, o.params, rtol=0, atol=PARAM_TOL)
        res[ph] = g.stats["pairs_scored"]
    assert res[phases] < res[0], res


def test_c1_baseline_config_verbatim(capi, orc):
    """BASELINE.json configs[0] as SURVEY.md 8(d) spells it: the 50 000-point plumbing cloud (60 % on the plane, data seed 1),
    fit_plane with 100 iterations, threshold 0.01, the reference's default probability, sampler seed 7 -- HIP path against the
    oracle: the whole result."""
    pts = synth.plane_cloud_c1(50_000, 1)
    o = orc.fit(0, pts, None, thr=0.01, max_iter=100, prob=0.9999, seed=7)
    g = capi.fit(0, pts, None, threshold=0.01, max_iteration=100, probability=0.9999, seed=7)
    assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.ret, o.best_index, o.count, o.iterations)
    assert g.stats["fitness"] == o.fitness
    assert np.array_equal(g.inliers, o.inliers) and len(o.inliers) > 25_000
    assert np.allclose(g.params, o.params, rtol=0, atol=PARAM_TOL)


def test_fit_with_ties_uses_serial_rmse(capi, orc):
    """Tiny clouds produce many equal-fitness hypotheses; the tie rule (ransac.h:595-596) needs the
    serial-order error sum."""
    rng = np.random.default_rng(0)
    for trial in range(6):
        n = 60 + 7 * trial
        # near-planar points (noise << threshold) + far outliers: every all-inlier sample reaches the
        # same count n - 6, so the best model is decided by the rmse tie rule over and over
        pts = np.c_[rng.uniform(-1, 1, (n, 2)), rng.uniform(-0.002, 0.002, n)]
        pts[:6, 2] += 5.0
        o = orc.fit(0, pts, thr=0.01, max_iter=200, prob=1.0, seed=trial)
        g = capi.fit(0, pts, threshold=0.01, max_iteration=200, probability=1.0, seed=trial)
        assert g.stats["best_index"] == o.best_index and g.stats["count"] == o.count
        assert np.array_equal(g.inliers, o.inliers)
        assert g.stats["ties"] > 0
        if not np.isnan(g.stats["inlier_rmse"]):
            assert g.stats["inlier_rmse"] == o.inlier_rmse
    # duplicate hypotheses (6 points -> the same triple is drawn again and again): the order-free sums
    # are EQUAL, so the decision falls through to the serial-order sums
    pts = np.c_[rng.uniform(-1, 1, (8, 2)), rng.uniform(-0.002, 0.002, 8)]
    pts[:2, 2] += 5.0                      # two outliers keep the fitness below 1 (no immediate stop)
    o = orc.fit(0, pts, thr=0.01, max_iter=300, prob=1.0, seed=5)
    g = capi.fit(0, pts, threshold=0.01, max_iteration=300, probability=1.0, seed=5)
    assert g.stats["best_index"] == o.best_index and g.stats["count"] == o.count
    assert g.stats["exact_rmse_evals"] > 0 and g.stats["inlier_rmse"] == o.inlier_rmse


def test_edge_cases(capi, orc):
    rng = np.random.default_rng(1)
    # exactly 3 points
    pts = rng.normal(size=(3, 3))
    o = orc.fit(0, pts, thr=0.01, max_iter=20, prob=0.9999, seed=1)
    g = capi.fit(0, pts, threshold=0.01, max_iteration=20, probability=0.9999, seed=1)
    assert g.ret == o.ret and np.array_equal(g.inliers, o.inliers) and g.stats["count"] == o.count
    # exact plane: fitness 1 -> immediate stop (ransac.h:607-610)
    xy = rng.integers(-512, 512, size=(5000, 2)) / 256.0
    pts = np.c_[xy, np.full(5000, 0.25)]
    o = orc.fit(0, pts, thr=0.01, max_iter=50, prob=0.9999, seed=1)
    g = capi.fit(0, pts, threshold=0.01, max_iteration=50, probability=0.9999, seed=1)
    assert g.stats["count"] == o.count == 1 and g.stats["fitness"] == 1.0
    assert np.array_equal(g.inliers, np.arange(5000, dtype=np.uint64))
    # NaN / inf coordinates are never inliers but stay in the cloud
    pts = synth.plane_cloud_c1(3000, seed=2)
    pts[5] = np.nan
    pts[17, 1] = np.inf
    o = orc.fit(0, pts, thr=0.01, max_iter=100, prob=1.0, seed=2)
    g = capi.fit(0, pts, threshold=0.01, max_iteration=100, probability=1.0, seed=2)
    assert g.stats["best_index"] == o.best_index and np.array_equal(g.inliers, o.inliers)
    # vanishing threshold: no inliers at all -> no best model -> soft failure
    pts = synth.plane_cloud_c1(2000, seed=3)
    o = orc.fit(0, pts, thr=1e-300, max_iter=30, prob=1.0, seed=2)
    g = capi.fit(0, pts, threshold=1e-300, max_iteration=30, probability=1.0, seed=2)
    assert g.ret == o.ret and g.stats["best_index"] == o.best_index and len(g.inliers) == len(o.inliers)
    # max_iteration = 0
    g = capi.fit(0, pts, threshold=0.01, max_iteration=0, probability=0.9, seed=2)
    assert g.ret == 0 and g.stats["best_index"] == -1 and len(g.inliers) == 0


def test_threshold_boundary_decisions(capi, orc):
    """Points placed within a few ulp of the threshold on both sides: the exact cut-off must make the
    same decision as the reference's divide-then-compare for every one of them."""
    rng = np.random.default_rng(4)
    base = rng.uniform(-1, 1, (4000, 3))
    base[:, 2] = 0.0
    thr = 0.01
    offs = thr * (1.0 + rng.integers(-8, 9, size=4000) * 2.0 ** -52)
    base[:, 2] = np.where(rng.random(4000) < 0.5, offs, -offs)
    anchors = np.array([[0.0, 0, 0], [1.0, 0, 0], [0, 1.0, 0]])
    pts = np.concatenate([anchors, base])
    samples = np.array([[0, 1, 2]], dtype=np.uint32)
    with capi.Cloud(pts) as c:
        valid, models, counts = c.score_range(0, thr, samples)
        rc, params, inl = c.refine(0, thr, models[0])
        # m3d_cloud_refine_expect: the count known from scoring changes only the order of the work; a wrong
        # expectation is detected and the ordinary order taken
        for expected in (int(counts[0]), int(counts[0]) + 5, 0, len(pts) + 10):
            rc2, params2, inl2 = c.refine(0, thr, models[0], expected=expected)
            assert rc2 == rc and np.array_equal(params2, params) and np.array_equal(inl2, inl)
    ov, om, oc, _ = orc.score_samples(0, pts, None, thr, samples.astype(np.uint64))
    assert counts[0] == oc[0] and 3 < oc[0] < len(pts)
    d = np.array([orc.distance(0, p, om[0]) for p in pts])
    assert np.array_equal(inl, np.nonzero(d < thr)[0].astype(np.uint64))


def test_segmentation_matches_oracle(capi, orc):
    rng = np.random.default_rng(5)
    a = np.c_[rng.uniform(-1, 1, (6000, 2)), rng.normal(0, 2e-3, 6000)]
    b = np.c_[rng.normal(0, 2e-3, 4000) + 2.0, rng.uniform(-1, 1, (4000, 2))]
    cpts = np.c_[rng.uniform(-1, 1, 3000), rng.normal(0, 2e-3, 3000) - 1.5, rng.uniform(-1, 1, 3000)]
    noise = rng.uniform(-3, 3, (800, 3))
    pts = np.concatenate([a, b, cpts, noise])[rng.permutation(13800)]
    orc_rc, oplanes, oclusters = orc.segment_plane_iterative(pts, 0.01, max_iteration=100, min_ratio=0.1, seed=3)
    rc, planes, clusters = capi.segment_plane_iterative(pts, 0.01, max_iteration=100, min_ratio=0.1, seed=3)
    assert orc_rc == 0 and rc == 1
    assert len(planes) == len(oplanes) >= 3
    for k in range(len(planes)):
        assert np.array_equal(clusters[k], oclusters[k])
        assert np.allclose(planes[k], oplanes[k], rtol=0, atol=PARAM_TOL)


def test_segmentation_tombstones_vs_partition_vs_oracle(capi, orc):
    """Rounds that remove a sliver of the cloud kill their inliers in place in the sorted copy (x = NaN in the fp64 array and
    in the tiles' fp32 offsets: poison_plane_inliers_k; score_screen_k masks the dead lanes) instead of partitioning it; a
    real compaction follows when an eighth of the copy is dead.  A room with ~60 clutter rounds goes through several such
    cycles: clusters and planes must be the oracle's, and the partition-every-round path's (sorted_tombstones = 0), bit for
    bit -- with the speculative RefineModel (rounds that run to max_iteration) and without it (adaptive stop)."""
    pts = synth.room_cloud_c5(300_000, 17)
    pts[[5, 77, 4000, 250_000]] = np.nan      # non-finite input points stay in the cloud (and out of the sorted copy, whose
    pts[123, 1] = np.inf                        # only NaN are the tombstones)
    for max_it, min_ratio, seed in ((300, 0.02, 5), (1000, 0.03, 6)):
        ro, po, co = orc.segment_plane_iterative(pts, 0.01, max_iteration=max_it, min_ratio=min_ratio, seed=seed, lookahead=128)
        r1, p1, c1 = capi.segment_plane_iterative(pts, 0.01, max_iteration=max_it, min_ratio=min_ratio, seed=seed)
        old = capi.set_config(sorted_tombstones=0)
        try:
            r0, p0, c0 = capi.segment_plane_iterative(pts, 0.01, max_iteration=max_it, min_ratio=min_ratio, seed=seed)
        finally:
            capi.restore_config(old)
        assert len(co) == len(c1) == len(c0) > 30
        for a, b, c in zip(co, c1, c0):
            assert np.array_equal(a, b) and np.array_equal(a, c)
        assert np.array_equal(p1, p0) and np.allclose(po, p1, rtol=0, atol=PARAM_TOL)
    # the fp64-only scoring path (no screen: every pair through the exact code, where a dead point's NaN is never `< T`)
    old = capi.set_config(score_fp32_screen=0, cull_fp32=0)
    try:
        r2, p2, c2 = capi.segment_plane_iterative(pts, 0.01, max_iteration=1000, min_ratio=0.03, seed=6)
    finally:
        capi.restore_config(old)
    assert len(c2) == len(c1) and all(np.array_equal(a, b) for a, b in zip(c1, c2)) and np.array_equal(p1, p2)


def test_full_size_properties(capi):
    """BASELINE config C2 size (1M points): size-independent properties instead of the oracle."""
    pts = synth.plane_cloud_c2(1_000_000, seed=2)
    with capi.Cloud(pts) as c:
        g = c.fit(0, 0.01, 2000, 1.0, seed=11)
        assert g.ret == 1 and g.stats["iterations"] == 2000
        n_in = len(g.inliers)
        assert n_in == g.stats["n_inliers"] == round(g.stats["fitness"] * len(pts))
        assert 0.45 * len(pts) < n_in < 0.56 * len(pts)
        assert np.all(np.diff(g.inliers.astype(np.int64)) > 0)          # ascending, unique
        # plane A of the generator
        nA = np.array([0.2, -0.3, 0.93]) / np.linalg.norm([0.2, -0.3, 0.93])
        s = np.sign(g.params[:3] @ nA)
        assert np.allclose(s * g.params[:3], nA, atol=2e-3) and abs(s * g.params[3] + 0.5) < 2e-3
        # idempotence: same seed -> identical result; counts are consistent with refine()
        g2 = c.fit(0, 0.01, 2000, 1.0, seed=11)
        assert np.array_equal(g.inliers, g2.inliers) and np.array_equal(_bits(g.params), _bits(g2.params))
        # per-hypothesis counts are independent of how the range is split (shardable unit)
        samples = capi.draw_samples(len(pts), 0, 256, seed=11)
        _, m_all, c_all = c.score_range(0, 0.01, samples)
        _, m_a, c_a = c.score_range(0, 0.01, samples, 0, 100)
        _, m_b, c_b = c.score_range(0, 0.01, samples, 100, 256)
        assert np.array_equal(c_all, np.concatenate([c_a, c_b]))
        assert np.array_equal(_bits(m_all), _bits(np.concatenate([m_a, m_b])))
        best = int(np.argmax(c_all))
        cnt, err = c.exact_error(0, 0.01, m_all[best])
        assert cnt == int(c_all[best]) and 0 < err < 0.01 * cnt


def test_sharded_driver_on_gpu_equals_single_call(capi):
    """distributed.fit_sharded with the product scorer (capi.Cloud) at world size 1 reproduces
    m3d_cloud_fit: same hypothesis, same inliers (both adaptive and exhaustive modes)."""
    from misc3d_amd import distributed
    pts = synth.plane_cloud_c1(50_000, seed=1)
    with capi.Cloud(pts) as c:
        for prob, H, seed in ((1.0, 3000, 11), (0.9999, 1000, 7)):
            g = c.fit(0, 0.01, H, prob, seed=seed)
            s = distributed.fit_sharded(c, len(pts), 0, 0.01, H, prob, seed)
            assert s.best_index == g.stats["best_index"] and s.count == g.stats["count"]
            assert s.iterations == g.stats["iterations"]
            assert np.array_equal(s.inliers, g.inliers)
            # (this python driver refines through m3d_cloud_refine -- GeneralFit sums over the inlier list --, the
            # one-call fit through the moments fused into the compaction: same least-squares problem, different trees)
            assert np.allclose(s.params, g.params, rtol=0, atol=1e-12)


def test_cloud_remove_inliers_and_sharded_segmentation(capi, orc):
    """m3d_cloud_remove_inliers (SelectByIndex(inliers, invert) on the resident cloud) and the sharded
    SegmentPlaneIterative driver at world size 1 with the product scorer: same planes and clusters as
    the one-call m3d_segment_plane_iterative and as the oracle; index lists refer to the cloud as created;
    non-finite points stay in the cloud."""
    from misc3d_amd import distributed
    pts = synth.room_cloud_c5(60_000, seed=6)
    pts[[5, 77, 4000]] = np.nan
    pts[123, 1] = np.inf
    rc, planes, clusters = capi.segment_plane_iterative(pts, 0.02, max_iteration=150, min_ratio=0.1, seed=19)
    orc_rc, oplanes, oclusters = orc.segment_plane_iterative(pts, 0.02, max_iteration=150, min_ratio=0.1, seed=19)
    assert rc == 1 and orc_rc == 0 and len(planes) == len(oplanes) >= 4
    with capi.Cloud(pts) as c:
        r = distributed.segment_plane_iterative_sharded(c, 0.02, 150, 0.1, seed=19)
        assert r.ret == 1 and len(r.planes) == len(planes)
        removed_total = sum(len(x) for x in r.clusters[:-1])
        assert c.n == len(pts) - removed_total and c.n_created == len(pts)
        for k in range(len(planes)):
            assert np.array_equal(r.clusters[k], clusters[k].astype(np.int64))
            assert np.array_equal(r.clusters[k], oclusters[k].astype(np.int64))
            assert np.allclose(r.planes[k], planes[k], rtol=0, atol=1e-12)
            assert np.allclose(r.planes[k], oplanes[k], rtol=0, atol=PARAM_TOL)
        # the shrunk cloud still answers every entry point: a fit on it equals a fit on a fresh upload of
        # the remaining points (indices mapped through the removal)
        keep = np.ones(len(pts), dtype=bool)
        for x in r.clusters[:-1]:
            keep[x] = False
        rest = np.nonzero(keep)[0]
        g = c.fit(0, 0.02, 200, 1.0, seed=5)
        with capi.Cloud(pts[rest]) as c2:
            g2 = c2.fit(0, 0.02, 200, 1.0, seed=5)
        assert g.stats["best_index"] == g2.stats["best_index"] and np.array_equal(g.params, g2.params)
        assert np.array_equal(g.inliers.astype(np.int64), rest[g2.inliers.astype(np.int64)])


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_culled_scoring_robustness(capi, orc, scoring_path, kind):
    """The box tests of cull_k must never drop an inlier: clouds far from the origin, tiny and huge
    scales, degenerate extents, tile-boundary sizes -- counts stay bit-identical to the oracle."""
    rng = np.random.default_rng(100 + kind)
    cases = []
    base, nrm = _clouds(kind, 6000, seed=40 + kind)
    cases.append((base + np.array([1.0e5, -2.0e5, 3.0e4]), nrm, 0.01))          # far from the origin
    cases.append((base * 1e-3, nrm, 1e-5))                                        # millimetre-scale scene
    cases.append((base * 1e3 + 7.0, nrm, 10.0))                                   # kilometre-scale scene
    for n in (511, 512, 513, 1024, 2049):                                         # tile boundaries
        p, q = _clouds(kind, n, seed=50 + kind)
        cases.append((p, q, 0.01))
    flat = base.copy()
    flat[:, 2] = 0.25                                                             # zero extent along z
    cases.append((flat, nrm, 0.01))
    same = np.tile(base[:1], (700, 1))                                            # all points identical
    cases.append((same, None if nrm is None else np.tile(nrm[:1], (700, 1)), 0.01))
    for pts, nn, thr in cases:
        n = len(pts)
        samples = capi.draw_samples(n, kind, 200, seed=3)
        with capi.Cloud(pts, nn) as c:
            valid, models, counts = c.score_range(kind, thr, samples)
        ovalid, omodels, ocounts, _ = orc.score_samples(kind, pts, nn, thr, samples.astype(np.uint64))
        assert np.array_equal(valid.astype(bool), ovalid.astype(bool))
        assert np.array_equal(counts.astype(np.uint64), ocounts), (kind, n, thr)


def test_pinned_output_buffer_same_result(capi):
    """m3d_host_alloc output buffers only change WHEN the inlier list is copied (early, under the GeneralFit sums),
    not what is copied; views handed out with copy=False stay valid after the cloud is closed."""
    from misc3d_amd import synth
    pts = synth.plane_cloud_c2(200_000, seed=2)
    plain = capi.fit(0, pts, threshold=0.01, max_iteration=500, probability=1.0, seed=4)   # pageable numpy buffer
    c = capi.Cloud(pts)
    g = c.fit(0, 0.01, 500, 1.0, seed=4, copy=False)
    assert isinstance(c._inl_buf.base, capi._PinnedU64)
    view = g.inliers
    c.close()
    assert np.array_equal(view, plain.inliers) and np.array_equal(g.params, plain.params)
    ptr = capi.lib().m3d_host_alloc(64)
    assert ptr
    capi.lib().m3d_host_free(ptr)


def test_early_pick_loses_rmse_tie(capi, orc):
    """Probability-1 fits start RefineModel's compaction on the device's own pick (most inliers, lowest index).
    Here a later hypothesis has the SAME inlier count and a lower rmse, so the sequential rule prefers it: the
    replay overrules the pick (m3d_stats.early_pick_redone == 1) and the result is the oracle's."""
    rng = np.random.default_rng(12)
    n = 4000
    xy = rng.uniform(-1, 1, (n, 2))
    # two parallel sheets with the same number of points each; the upper one is flatter (lower rmse)
    z = np.where(np.arange(n) % 2 == 0, 0.0 + rng.uniform(-4e-3, 4e-3, n), 0.5 + rng.uniform(-1e-3, 1e-3, n))
    pts = np.ascontiguousarray(np.c_[xy, z])
    # three exact points per sheet make hypothesis planes z = 0 and z = 0.5 whose slabs (thr 0.01) hold a whole sheet
    pts[0], pts[2], pts[4] = (-1, -1, 0.0), (1, -1, 0.0), (0, 1, 0.0)
    pts[1], pts[3], pts[5] = (-1, -1, 0.5), (1, -1, 0.5), (0, 1, 0.5)
    hit = 0
    for seed in range(40):
        o = orc.fit(0, pts, None, thr=0.01, max_iter=1500, prob=1.0, seed=seed)
        g = capi.fit(0, pts, None, 0.01, 1500, 1.0, seed=seed)
        assert (g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.best_index, o.count, o.iterations)
        assert np.array_equal(g.inliers.astype(np.uint64), o.inliers.astype(np.uint64))
        hit += int(g.stats["early_pick_redone"])
    assert hit > 0      # (39 of the 40 seeds on MI355X)


@pytest.mark.parametrize("max_iter", [300, 2000, 40000])
def test_perfect_fit_stops_probability_one_loop(capi, orc, max_iter):
    """fitness == 1 is the one thing that stops a probability-1 loop (ransac.h:607-609): every point lies on the plane,
    so the first valid hypothesis ends it -- inside the first chunk, long before the last one, or in a single-chunk
    fit where the device's early pick has already been acted on.  Iteration count and model as the oracle's."""
    rng = np.random.default_rng(21)
    xy = rng.integers(-512, 512, (6000, 2)).astype(np.float64) / 256.0
    pts = np.ascontiguousarray(np.c_[xy, np.full(len(xy), 0.25)])          # exactly representable: distances are 0
    for seed in (1, 2, 3):
        o = orc.fit(0, pts, None, thr=0.01, max_iter=max_iter, prob=1.0, seed=seed)
        g = capi.fit(0, pts, None, 0.01, max_iter, 1.0, seed=seed)
        assert o.count <= 3 and len(o.inliers) == len(pts)
        assert (g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (o.best_index, o.count, o.iterations)
        assert np.array_equal(g.inliers.astype(np.uint64), o.inliers.astype(np.uint64))
        assert np.allclose(g.params, o.params, rtol=0, atol=1e-9)


def _screen_case(capi, orc, kind, pts, thr, samples, nrm=None):
    """counts of the screened scoring == oracle == unscreened scoring"""
    ov, om, oc, _ = orc.score_samples(kind, pts, nrm, thr, samples.astype(np.uint64))
    with capi.Cloud(pts, nrm) as c:
        valid, models, counts = c.score_range(kind, thr, samples)
        old = capi.set_config(score_fp32_screen=0)
        try:
            valid0, models0, counts0 = c.score_range(kind, thr, samples)
        finally:
            capi.restore_config(old)
    assert np.array_equal(valid.astype(bool), ov.astype(bool))
    assert np.array_equal(counts.astype(np.uint64), oc), (kind, thr)
    assert np.array_equal(counts0, counts)


def test_fp32_screen_points_at_the_cut_off(capi, orc):
    """score_screen_k decides in fp32 only what its rounding bound allows: points a few fp64 ulps, a few fp32 ulps and
    a few bounds away from the threshold, on both sides, must all be counted as the fp64 test counts them."""
    rng = np.random.default_rng(14)
    thr = 0.01
    n = 6000
    base = rng.uniform(-1, 1, (n, 3))
    rel = np.concatenate([rng.integers(-8, 9, n // 3) * 2.0 ** -52,          # fp64 ulps
                          rng.integers(-64, 65, n // 3) * 2.0 ** -24,        # fp32 ulps: inside the screen's band
                          rng.uniform(-3e-5, 3e-5, n - 2 * (n // 3))])       # around the edge of the band
    base[:, 2] = np.where(rng.random(n) < 0.5, 1.0, -1.0) * thr * (1.0 + rel)
    anchors = np.array([[0.0, 0, 0], [1.0, 0, 0], [0, 1.0, 0]])
    pts = np.concatenate([anchors, base])
    samples = np.concatenate([np.array([[0, 1, 2]], dtype=np.uint32), capi.draw_samples(len(pts), 0, 63, seed=8)])
    _screen_case(capi, orc, 0, pts, thr, samples)
    # the same cloud tilted and moved away from the origin (the bound grows with the coordinates)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    for shift in (0.0, 30.0, 4.0e4):
        _screen_case(capi, orc, 0, pts @ q.T + shift, thr, samples)
    # sphere: radial offsets around r +- thr
    r = 0.5
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    rad = r + np.where(rng.random(n) < 0.5, 1.0, -1.0) * thr * (1.0 + rel)
    sph = np.array([0.3, -0.2, 1.0]) + d * rad[:, None]
    on = np.array([0.3, -0.2, 1.0]) + r * np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [-1.0, 0, 0]])
    pts = np.concatenate([on, sph])
    samples = np.concatenate([np.array([[0, 1, 2, 3]], dtype=np.uint32), capi.draw_samples(len(pts), 1, 63, seed=9)])
    for shift in (0.0, 25.0):
        _screen_case(capi, orc, 1, pts + shift, thr, samples)
    # cylinder: radial offsets around r +- thr from the axis through a along dirn; the first two points + normals
    # give MinimalFit exactly that axis (ransac.h:354-417: the normals of two surface points meet on it)
    r = 0.25
    a = np.array([0.1, 0.2, 0.3])
    dirn = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    e1 = np.cross(dirn, [0.0, 0.0, 1.0])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(dirn, e1)
    ang = rng.uniform(0, 2 * np.pi, n)
    along = rng.uniform(-1, 1, n)
    radial = np.cos(ang)[:, None] * e1 + np.sin(ang)[:, None] * e2
    rad = r + np.where(rng.random(n) < 0.5, 1.0, -1.0) * thr * (1.0 + rel)
    cyl = a + along[:, None] * dirn + rad[:, None] * radial
    first = a + np.array([[-0.5], [0.7]]) * dirn + r * np.stack([e1, e2])
    pts = np.concatenate([first, cyl])
    nrm = np.concatenate([np.stack([e1, e2]), radial])
    samples = np.concatenate([np.array([[0, 1]], dtype=np.uint32), capi.draw_samples(len(pts), 2, 63, seed=10)])
    for shift in (0.0, 25.0):
        _screen_case(capi, orc, 2, pts + shift, thr, samples, nrm)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_fp32_screen_unscreenable_inputs(capi, orc, kind):
    """Inputs the fp32 pass cannot represent go to the exact code, pair by pair or tile by tile: non-finite points,
    coordinates beyond the fp32 range, thresholds below the rounding bound, degenerate thresholds."""
    pts, nrm = _clouds(kind, 5000, seed=70 + kind)
    samples = capi.draw_samples(len(pts), kind, 128, seed=4)
    bad = pts.copy()
    bad[17] = [np.nan, 0.0, 0.0]
    bad[600] = [np.inf, 1.0, 1.0]
    bad[601] = [0.0, -np.inf, 1.0]
    bad[1300] = [1.0e39, 0.0, 0.0]        # finite, beyond fp32
    bad[2900] = [0.0, 0.0, -3.0e300]
    ok_rows = np.ones(len(pts), bool)
    ok_rows[[17, 600, 601, 1300, 2900]] = False
    smp = samples[np.all(ok_rows[samples], axis=1)]
    _screen_case(capi, orc, kind, bad, 0.01, smp, nrm)
    _screen_case(capi, orc, kind, pts, 1e-9, samples, nrm)        # threshold far below the fp32 bound
    _screen_case(capi, orc, kind, pts, 1e-300, samples, nrm)
    _screen_case(capi, orc, kind, pts, 1e6, samples, nrm)         # everything is an inlier
    _screen_case(capi, orc, kind, pts * 1e-25, 1e-27, samples, nrm)   # fp32 denormal territory for the squares
    _screen_case(capi, orc, kind, pts * 1e20, 1e18, samples, nrm)     # squares overflow fp32


def test_fp32_screen_recount_rate(capi):
    """The screen is worth having only if the exact recount is rare: on the C2-like cloud a fraction of a percent of the
    (tile, hypothesis) pairs."""
    pts = synth.plane_cloud_c2(200_000, seed=2)
    g = capi.fit(0, pts, threshold=0.01, max_iteration=2000, probability=1.0, seed=1)
    st = g.stats
    assert st["pairs_scored"] > 0
    assert st["pairs_exact"] <= 0.005 * st["pairs_scored"], st
    old = capi.set_config(score_fp32_screen=0)
    try:
        g0 = capi.fit(0, pts, threshold=0.01, max_iteration=2000, probability=1.0, seed=1)
    finally:
        capi.restore_config(old)
    assert g0.stats["pairs_exact"] == 0
    assert g0.stats["best_index"] == st["best_index"] and g0.stats["count"] == st["count"]
    assert np.array_equal(g0.inliers, g.inliers)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_lead_pass_inside_the_box_test_launch(capi, kind):
    """cull_lead_k: on one GPU with the fp32 paths on, the leading hypotheses of a fit's first chunk are counted inside the
    launch that runs the chunk's box tests.  Same fit as with the separate launches (fp64 box tests switch the fusion
    off), same evaluated pairs, and the pairs of the lead pass are not among those of the timed launches."""
    pts = synth.plane_cloud_c2(150_000, seed=4) if kind == 0 else None
    nrm = None
    if pts is None:
        rng = np.random.default_rng(4)
        if kind == 1:
            d = rng.normal(size=(150_000, 3))
            d /= np.linalg.norm(d, axis=1)[:, None]
            pts = np.r_[0.3 * d[:90_000] + rng.normal(0, 1e-3, (90_000, 3)), rng.uniform(-1, 1, (60_000, 3))]
        else:
            t = rng.uniform(0, 2 * np.pi, 150_000)
            z = rng.uniform(-1, 1, 150_000)
            pts = np.c_[0.2 * np.cos(t), 0.2 * np.sin(t), z] + rng.normal(0, 1e-3, (150_000, 3))
            nrm = np.c_[np.cos(t), np.sin(t), np.zeros_like(t)]
            pts[100_000:] = rng.uniform(-1, 1, (50_000, 3))
    kw = dict(threshold=0.01, max_iteration=3000, probability=1.0, seed=3)
    old = capi.set_config(kernel_timing=1)
    try:
        g = capi.fit(kind, pts, nrm, **kw)
        capi.set_config(kernel_timing=1, cull_fp32=0)
        g0 = capi.fit(kind, pts, nrm, **kw)
    finally:
        capi.restore_config(old)
    st, st0 = g.stats, g0.stats
    assert st["best_index"] == st0["best_index"] and st["count"] == st0["count"] and np.array_equal(g.inliers, g0.inliers)
    assert st["pairs_scored"] > 0 and st0["pairs_scored"] > 0
    assert 0 < st["pairs_timed"] < st["pairs_scored"]          # the lead pass ran inside cull_lead_k, untimed
    assert st0["pairs_timed"] == st0["pairs_scored"]           # separate launches: everything is timed
    assert st["score_launches"] + 1 == st0["score_launches"]   # one timed launch less (the lead's)
