"""Known-answer tests that pin the CPU oracle (oracle/misc3d_oracle.c).

The reference has no tests or golden vectors for this path, so the oracle is pinned by
analytic cases, by libstdc++'s own std::mt19937 / uniform_int_distribution compiled in this image,
and by numpy cross-checks.  Every quirk of the reference marked in SURVEY.md 8a has a test here.
"""
import subprocess

import numpy as np
import pytest


# ---------------------------------------------------------------------------------------------
# RNG / sampler (include/misc3d/utils.h:71-97)
# ---------------------------------------------------------------------------------------------
def test_mt19937_known_answers(orc):
    g = orc.MT19937(5489)  # default seed of std::mt19937
    first = [g.next() for _ in range(5)]
    assert first == [3499211612, 581869302, 3890346734, 3586334585, 545404204]
    for _ in range(10000 - 5 - 1):
        g.next()
    assert g.next() == 4123659995  # 10000th output required by the C++ standard [rand.predef]


@pytest.mark.parametrize("seed", [0, 1, 42, 2**32 - 1, 2**32 + 7])
def test_rng_matches_libstdcxx(orc, seed):
    out = subprocess.run([orc.RNG_CHECK_PATH, str(seed), "700", "1000003", "12344"], capture_output=True,
                         text=True, check=True).stdout.splitlines()
    raw = [int(v) for v in out[0].split()]
    mod = [int(v) for v in out[1].split()]
    uni = [int(v) for v in out[2].split()]
    g = orc.MT19937(seed)
    assert [g.next() for _ in range(700)] == raw
    assert [r % 1000003 for r in raw] == mod
    g = orc.MT19937(seed)
    assert [g.uniform_int(12344) for _ in range(700)] == uni


def test_sampler_rejects_duplicates_and_uses_modulo(orc):
    g = orc.MT19937(7)
    raw = orc.MT19937(7)
    for _ in range(200):
        s = g.sample(5, 3)  # tiny population: duplicates are frequent
        assert len(set(s.tolist())) == 3
        expect = []
        while len(expect) < 3:
            v = raw.next() % 5
            if v not in expect:
                expect.append(v)
        assert s.tolist() == expect
    a = orc.draw_samples(1000, 4, 50, 99)
    g = orc.MT19937(99)
    assert np.array_equal(a, np.array([g.sample(1000, 4) for _ in range(50)]))


# ---------------------------------------------------------------------------------------------
# plane (ransac.h:134-221)
# ---------------------------------------------------------------------------------------------
def test_plane_minimal_fit_exact(orc):
    ok, m = orc.plane_minimal_fit([0, 0, 1, 1, 0, 1, 0, 1, 1])
    assert ok and m.tolist() == [0.0, 0.0, 1.0, -1.0]
    ok, m = orc.plane_minimal_fit([0, 0, 1, 0, 1, 1, 1, 0, 1])  # opposite winding flips the sign
    assert ok and m.tolist() == [-0.0, -0.0, -1.0, 1.0] or m.tolist() == [0.0, 0.0, -1.0, 1.0]
    ok, _ = orc.plane_minimal_fit([0, 0, 0, 1, 1, 1, 2, 2, 2])  # collinear -> norm < 1e-8
    assert not ok
    ok, _ = orc.plane_minimal_fit([0, 0, 0, 1e-5, 0, 0, 0, 1e-5, 0])  # |cross| = 1e-10 < EPS
    assert not ok
    ok, _ = orc.plane_minimal_fit([0, 0, 0, 1e-3, 0, 0, 0, 1e-3, 0])  # |cross| = 1e-6 >= EPS
    assert ok


def test_plane_distance_and_order(orc):
    assert orc.distance(orc.PLANE, [0.5, 0.5, 3.0], [0, 0, 1, -1]) == 2.0
    # distance divides by the norm of (a,b,c) even when it is not 1 (ransac.h:218-219)
    assert orc.distance(orc.PLANE, [0, 0, 3.0], [0, 0, 2, -2]) == 2.0
    # 4-element association (a*x + c*z) + (b*y + d): choose values where the order matters
    a, b, c, d = 1e16, 1.0, -1e16, 1.0
    x = y = z = 1.0
    assert orc.distance(orc.PLANE, [x, y, z], [a, b, c, d]) == abs((a * x + c * z) + (b * y + d)) / np.sqrt(
        (a * a + b * b) + c * c)


def test_plane_general_fit(orc):
    rng = np.random.default_rng(0)
    xy = rng.integers(-64, 64, size=(500, 2)) / 64.0  # exact binary fractions
    pts = np.c_[xy, np.full(500, 0.5)]
    ok, m = orc.plane_general_fit(pts)
    assert ok and np.allclose(np.abs(m), [0, 0, 1, 0.5], atol=1e-15)
    assert m[2] * m[3] < 0
    ok, _ = orc.plane_general_fit(pts[:2])  # MinimalCheck
    assert not ok
    ok, _ = orc.plane_general_fit(np.zeros((10, 3)))  # degenerate: norm < EPS
    assert not ok
    # tilted noisy plane vs an SVD fit
    n = np.array([0.2, -0.3, 0.93])
    n /= np.linalg.norm(n)
    p = rng.uniform(-1, 1, (2000, 3))
    p -= np.outer(p @ n + 0.5, n)
    p += rng.normal(0, 1e-3, (2000, 1)) * n
    ok, m = orc.plane_general_fit(p)
    c = p.mean(0)
    _, _, vt = np.linalg.svd(p - c)
    nn = vt[2] * np.sign(vt[2] @ m[:3])
    assert ok and np.allclose(m[:3], nn, atol=1e-6) and abs(m[3] + nn @ c) < 1e-6


# ---------------------------------------------------------------------------------------------
# sphere (ransac.h:223-344)
# ---------------------------------------------------------------------------------------------
def test_sphere_minimal_fit(orc):
    ok, m = orc.sphere_minimal_fit([1, 0, 0, 0, 1, 0, 0, 0, 1, -1, 0, 0])
    assert ok and np.allclose(m, [0, 0, 0, 1], atol=1e-15)
    c, r = np.array([0.3, -0.2, 1.0]), 0.5
    rng = np.random.default_rng(1)
    u = rng.normal(size=(4, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    ok, m = orc.sphere_minimal_fit((c + r * u).ravel())
    assert ok and np.allclose(m, [*c, r], atol=1e-12)
    ok, _ = orc.sphere_minimal_fit([0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0])  # 4 coplanar points
    assert not ok
    ok, _ = orc.sphere_minimal_fit([0, 0, 0, 1, 0, 0, 2, 0, 0, 0, 1, 5])  # first 3 collinear
    assert not ok


def test_sphere_distance(orc):
    m = [0, 0, 0, 2.0]
    assert orc.distance(orc.SPHERE, [3, 0, 0], m) == 1.0
    assert orc.distance(orc.SPHERE, [0, 0.5, 0], m) == 1.5
    assert orc.distance(orc.SPHERE, [0, 0, 0], m) == 2.0
    assert np.isnan(orc.distance(orc.SPHERE, [1, 0, 0], [0, 0, 0, np.nan]))  # NaN radius is never an inlier


def test_sphere_general_fit_vs_lstsq(orc):
    rng = np.random.default_rng(2)
    c, r = np.array([0.3, -0.2, 1.0]), 0.5
    u = rng.normal(size=(3000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    p = c + (r + rng.normal(0, 2e-3, (3000, 1))) * u
    ok, m = orc.sphere_general_fit(p)
    A = np.c_[2 * p, np.ones(len(p))]
    b = (p ** 2).sum(1)
    w = np.linalg.lstsq(A, b, rcond=None)[0]
    ref = [w[0], w[1], w[2], np.sqrt(w[0] ** 2 + w[1] ** 2 + w[2] ** 2 + w[3])]
    assert ok and np.allclose(m, ref, rtol=0, atol=1e-11)
    ok, _ = orc.sphere_general_fit(p[:3])
    assert not ok


# ---------------------------------------------------------------------------------------------
# cylinder (ransac.h:350-446, utils.h:313-322)
# ---------------------------------------------------------------------------------------------
def _p2l(q, p1, p2):
    a, b, c = q - p1, q - p2, p2 - p1
    return np.linalg.norm(np.cross(a, b)) / np.linalg.norm(c)


def test_cylinder_minimal_fit_geometry_and_radius_quirk(orc):
    # perfect cylinder around the z axis, r = 1, outward normals
    p = np.array([[1.0, 0, 0], [0, 1.0, 0.5]])
    n = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    ok, m = orc.cylinder_minimal_fit(p.ravel(), n.ravel())
    assert ok
    line_pt, line_dir = m[:3], m[3:6]
    # PCL-style construction (ransac.h:376-404): both points lie on the axis
    assert np.allclose(line_pt[:2], [0, 0], atol=1e-12) or abs(np.cross(line_pt, line_dir)[2]) < 1e-9
    assert np.allclose(np.abs(line_dir), [0, 0, 1], atol=1e-12)
    # ransac.h:413-414 quirk: the DIRECTION is used as the second point of the line
    assert m[6] == pytest.approx(_p2l(p[0], line_pt, line_dir), rel=1e-15)


def test_cylinder_degeneracy_quirk(orc):
    n = [1.0, 0, 0, 0, 1.0, 0]
    # ransac.h:367-374: (p0x - p1x <= DBL_EPS) && |dy| <= FLT_EPS && |dz| <= FLT_EPS  -> invalid,
    # i.e. two DISTINCT points differing only by +x are rejected when p0x < p1x ...
    ok, _ = orc.cylinder_minimal_fit([0, 0, 0, 1, 0, 0], n)
    assert not ok
    # ... and accepted in the opposite order
    ok, _ = orc.cylinder_minimal_fit([1, 0, 0, 0, 0, 0], n)
    assert ok
    ok, _ = orc.cylinder_minimal_fit([0, 0, 0, 1, 1e-6, 0], n)  # |dy| > FLT_EPS
    assert ok


def test_cylinder_parallel_normals_branch(orc):
    # denominator < 1e-8 branch (ransac.h:391-394)
    ok, m = orc.cylinder_minimal_fit([1, 0, 0, 1, 0.5, 1.0], [1.0, 0, 0, 1.0, 0, 0])
    assert ok and np.all(np.isfinite(m[:6]))


def test_cylinder_distance(orc):
    w = [0, 0, 0, 0, 0, 1.0, 1.0]
    assert orc.distance(orc.CYLINDER, [2, 0, 5], w) == 1.0
    assert orc.distance(orc.CYLINDER, [0.25, 0, -3], w) == 0.75
    # direction of length 2: ref = centre + dir, the formula normalises by |ref - centre|
    assert orc.distance(orc.CYLINDER, [2, 0, 5], [0, 0, 0, 0, 0, 2.0, 1.0]) == 1.0


# ---------------------------------------------------------------------------------------------
# driver (ransac.h:506-654)
# ---------------------------------------------------------------------------------------------
def test_evaluate_model_serial_sum(orc):
    pts = np.array([[0, 0, 1.001], [0, 0, 0.995], [0, 0, 2.0], [1, 1, 1.0]])
    cnt, err = orc.evaluate_model(orc.PLANE, pts, 0.01, [0, 0, 1, -1])
    d = [abs((0 * p[0] + 1 * p[2]) + (0 * p[1] - 1)) / 1.0 for p in pts]
    assert cnt == 3 and err == (d[0] + d[1]) + d[3]


def test_fit_exact_plane_stops_after_first_valid(orc):
    rng = np.random.default_rng(3)
    xy = rng.integers(-512, 512, size=(400, 2)) / 256.0
    pts = np.c_[xy, np.full(400, 0.25)]
    r = orc.fit(orc.PLANE, pts, thr=0.01, max_iter=50, prob=0.9999, seed=1)
    assert r.ret == 1 and r.fitness == 1.0 and r.count == 1          # ransac.h:607-610
    assert np.array_equal(r.inliers, np.arange(400))
    assert np.allclose(np.abs(r.params), [0, 0, 1, 0.25], atol=1e-15)


def test_fit_errors_and_soft_failures(orc):
    pts = np.random.default_rng(4).normal(size=(50, 3))
    assert orc.fit(orc.PLANE, pts, prob=0.0, seed=1).ret == -1
    assert orc.fit(orc.PLANE, pts, prob=1.0001, seed=1).ret == -1
    assert orc.fit(orc.PLANE, pts[:2], seed=1).ret == -2
    assert orc.fit(orc.CYLINDER, pts, normals=None, seed=1).ret == -3
    # max_iter = 0: no hypothesis, zero model -> plane distance NaN -> no inliers -> GeneralFit false
    r = orc.fit(orc.PLANE, pts, max_iter=0, seed=1)
    assert r.ret == 0 and len(r.inliers) == 0 and r.best_index == -1


def test_fit_adaptive_stop_and_prob_one(orc):
    from misc3d_amd import synth
    pts = synth.plane_cloud_c1(5000, seed=1)
    r = orc.fit(orc.PLANE, pts, thr=0.01, max_iter=1000, prob=0.9999, seed=7)
    assert r.ret == 1 and 0.55 < r.fitness < 0.65 and r.count < 200 and r.iterations < 1000
    assert abs(abs(r.params[2]) - 1) < 1e-3 and abs(abs(r.params[3]) - 1) < 1e-3
    r1 = orc.fit(orc.PLANE, pts, thr=0.01, max_iter=150, prob=1.0, seed=7)
    assert r1.iterations == 150  # log(0) -> +inf -> min(., max_iter): every hypothesis runs
    # the hypotheses visited before the stop are the same in both runs
    assert r1.fitness >= r.fitness


def test_double_to_size_t_x86(orc):
    f = orc.lib().orc_double_to_size_t_x86
    assert f(0.0) == 0 and f(69.9) == 69 and f(1e3) == 1000
    assert f(float("nan")) == 2 ** 63 and f(float("-inf")) == 2 ** 63
    assert f(-5.5) == 2 ** 64 - 5
    assert f(2.0 ** 63) == 2 ** 63 and f(float("inf")) == 0


def test_segmentation_two_planes(orc):
    rng = np.random.default_rng(5)
    a = np.c_[rng.uniform(-1, 1, (1500, 2)), rng.normal(0, 1e-3, 1500)]
    b = np.c_[rng.normal(0, 1e-3, 1000) + 2.0, rng.uniform(-1, 1, (1000, 2))]
    noise = rng.uniform(-3, 3, (200, 3))
    pts = np.concatenate([a, b, noise])[rng.permutation(2700)]
    rc, planes, clusters = orc.segment_plane_iterative(pts, 0.01, max_iteration=100, min_ratio=0.1, seed=3)
    assert rc == 0 and len(planes) >= 2
    assert abs(abs(planes[0][2]) - 1) < 1e-3 and abs(abs(planes[1][0]) - 1) < 1e-3
    allidx = np.concatenate(clusters)
    assert len(np.unique(allidx)) == len(allidx)               # clusters are disjoint
    assert len(allidx) >= int((1 - 0.1) * len(pts))            # loop condition :28-29
    for c in clusters:
        assert np.all(np.diff(c.astype(np.int64)) > 0)         # SelectByIndex keeps order
    rc, planes, clusters = orc.segment_plane_iterative(pts[:2], 0.01)
    assert rc == 1 and len(planes) == 0                        # :13-17


def test_oracle_normals_from_map_against_numpy(orc):
    """EstimateNormalsFromMap restatement vs a direct per-pixel window PCA in numpy (np.cov + eigh): same normals
    up to the cancellation noise of the E[x^2]-E[x]^2 form; NaN for invalid pixels; J3x3 vs numpy.linalg.eigh."""
    rng = np.random.default_rng(0)
    for _ in range(300):
        M = rng.normal(size=(3, 3))
        A = M @ M.T * rng.uniform(1e-6, 1)
        n = orc.j3x3_smallest_eigvec(A)
        wv, v = np.linalg.eigh(A)
        if (wv[1] - wv[0]) / wv[2] > 1e-6:
            assert min(np.abs(n - v[:, 0]).max(), np.abs(n + v[:, 0]).max()) < 1e-9
    w, h, k = 60, 44, 2
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    z = 1.0 + 0.003 * u - 0.002 * v + rng.normal(0, 1e-4, (h, w))
    xyz = np.stack([(u - 30) / 100.0 * z, (v - 22) / 100.0 * z, z], -1).reshape(-1, 3)
    hole = rng.random(w * h) < 0.05
    xyz[hole] = np.nan
    N = orc.normals_from_map(xyz, w, h, k)
    assert np.isnan(N[hole]).all()
    P = xyz.reshape(h, w, 3)
    worst = 0.0
    for r in range(h):
        for c in range(w):
            if np.isnan(P[r, c, 2]):
                continue
            win = P[max(0, r - k): r + k + 1, max(0, c - k): c + k + 1].reshape(-1, 3)
            win = win[~np.isnan(win[:, 2])]
            wv, vv = np.linalg.eigh(np.cov(win.T, bias=True))
            nn = vv[:, 0] if np.dot(-P[r, c], vv[:, 0]) >= 0 else -vv[:, 0]
            worst = max(worst, np.abs(N[r * w + c] - nn).max())
    assert worst < 1e-7


def test_oracle_boundary_detection_properties(orc):
    """DetectBoundaryPoints restatement: on a noisy planar patch with a hole every point near the outer edge or the
    hole's rim is flagged, deep interior points are not; given normals and estimated normals agree (the angular
    gaps do not depend on the in-plane basis); a larger angle threshold flags fewer points."""
    rng = np.random.default_rng(2)
    uv = rng.uniform(0, 1, (4000, 2))
    uv = uv[np.hypot(uv[:, 0] - 0.5, uv[:, 1] - 0.5) > 0.2]
    pts = np.c_[uv[:, 0], uv[:, 1], 0.1 * uv[:, 0] + rng.normal(0, 1e-4, len(uv))]
    nrm = np.tile(np.array([-0.1, 0.0, 1.0]) / np.linalg.norm([-0.1, 0.0, 1.0]), (len(uv), 1))
    edge = np.minimum.reduce([uv[:, 0], 1 - uv[:, 0], uv[:, 1], 1 - uv[:, 1], np.abs(np.hypot(uv[:, 0] - 0.5, uv[:, 1] - 0.5) - 0.2)])
    b = orc.detect_boundary_points(pts, nrm, 2, 0.06, 30, 90.0)
    flagged = np.zeros(len(pts), dtype=bool)
    flagged[b] = True
    assert flagged[edge < 0.004].mean() > 0.95 and flagged[edge > 0.07].mean() < 0.02   # (random 90-degree gaps happen)
    b2 = orc.detect_boundary_points(pts, None, 2, 0.06, 30, 90.0)
    assert len(set(b.tolist()) ^ set(b2.tolist())) <= 2
    assert len(orc.detect_boundary_points(pts, nrm, 2, 0.06, 30, 150.0)) < len(b)
    assert np.array_equal(b, np.sort(b))


# ------------------------------------------------------------------------------------------------
# the look-ahead (OpenMP) form of the sequential driver the full-size GPU tests use
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("prob,lookahead", [(0.9999, 7), (0.9999, 64), (1.0, 33), (0.5, 16)])
def test_fit_lookahead_is_the_sequential_loop(orc, kind, prob, lookahead):
    """orc_fit_parallel computes records ahead with an OpenMP team and consumes them in index order: every output of
    orc_fit -- best index, valid-hypothesis count, iterations run (early stop), rmse from the SERIAL error sum, inlier
    list, refined parameters and the per-hypothesis trace -- must come out bit for bit."""
    from misc3d_amd import synth
    if kind == 0:
        pts, nrm = synth.plane_cloud_c1(6000, 1), None
    elif kind == 1:
        pts, nrm = synth.sphere_cloud_c3(6000, 4), None
    else:
        pts, nrm = synth.cylinder_cloud_c3(6000, 3)
    a = orc.fit(kind, pts, nrm, thr=0.01, max_iter=300, prob=prob, seed=5, trace=True)
    b = orc.fit(kind, pts, nrm, thr=0.01, max_iter=300, prob=prob, seed=5, trace=True, lookahead=lookahead)
    assert (a.ret, a.best_index, a.count, a.iterations) == (b.ret, b.best_index, b.count, b.iterations)
    assert a.fitness == b.fitness and a.inlier_rmse == b.inlier_rmse
    assert np.array_equal(a.inliers, b.inliers)
    assert np.array_equal(a.params.view(np.uint64), b.params.view(np.uint64))
    k = a.iterations
    for key in ("samples", "valid", "counts"):
        assert np.array_equal(a.trace[key][:k], b.trace[key][:k]), key
    assert np.array_equal(a.trace["errors"][:k].view(np.uint64), b.trace["errors"][:k].view(np.uint64))
    assert np.array_equal(np.nan_to_num(a.trace["models"][:k], nan=-7.0).view(np.uint64),
                          np.nan_to_num(b.trace["models"][:k], nan=-7.0).view(np.uint64))


def test_segmentation_lookahead_is_the_sequential_loop(orc):
    from misc3d_amd import synth
    pts = synth.room_cloud_c5(30000, 6)
    ra, pa, ca = orc.segment_plane_iterative(pts, 0.01, max_iteration=60, min_ratio=0.05, seed=3)
    rb, pb, cb = orc.segment_plane_iterative(pts, 0.01, max_iteration=60, min_ratio=0.05, seed=3, lookahead=24)
    assert ra == rb and len(ca) == len(cb) >= 6
    assert np.array_equal(pa.view(np.uint64), pb.view(np.uint64))
    assert all(np.array_equal(x, y) for x, y in zip(ca, cb))
    # max_clusters cuts the same run short
    rc, pc, cc = orc.segment_plane_iterative(pts, 0.01, max_iteration=60, min_ratio=0.05, seed=3, max_clusters=3, lookahead=24)
    assert len(cc) == 3 and all(np.array_equal(x, y) for x, y in zip(ca[:3], cc))


def test_reg_validate_kdtree_equals_brute_force():
    """tools/cpu_baselines.py times the validation with a kd-tree (what the reference runs: KDTreeFlann); it must count what the
    oracle's brute force counts."""
    import oracle as orc
    from misc3d_amd import synth
    d = synth.registration_pair_c4(4000, seed=3, dim=8)
    T = d["T"].copy()
    T[0, 3] += 0.004
    c0, e0 = orc.reg_validate(d["src"], d["dst"], T, 0.01)
    c1, e1 = orc.reg_validate_kdtree(d["src"], d["dst"], T, 0.01, workers=1)
    assert c0 == c1 > 100 and abs(e0 - e1) <= 1e-12 * e0
