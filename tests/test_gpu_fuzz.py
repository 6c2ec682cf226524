"""Randomised parity sweep: many small, odd-shaped problems per entry point, GPU (through the C ABI) against
the oracle.  Sizes around tile / wave / chunk boundaries, thresholds from tiny to huge, clouds far from the
origin, duplicated and non-finite points.  Index work must be bit-exact, parameters within 1e-5 (north_star)."""
import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu
PARAM_TOL = 1e-5


def _cloud(rng, kind, n):
    if kind == 0:
        pts, nrm = synth.plane_cloud_c1(n, int(rng.integers(1 << 30))), None
    elif kind == 1:
        pts, nrm = synth.sphere_cloud_c3(n, int(rng.integers(1 << 30))), None
    else:
        pts, nrm = synth.cylinder_cloud_c3(n, int(rng.integers(1 << 30)))
    pts = np.ascontiguousarray(pts)
    mode = rng.integers(0, 5)
    if mode == 1:                                   # far from the origin
        pts = pts + rng.uniform(-1e4, 1e4, 3)
    elif mode == 2 and n > 20:                      # duplicates
        pts[rng.integers(0, n, n // 10)] = pts[rng.integers(0, n)]
    elif mode == 3 and n > 20:                      # non-finite entries
        pts[rng.integers(0, n, 3)] = np.nan
        pts[rng.integers(0, n), 1] = np.inf
    return pts, (np.ascontiguousarray(nrm) if nrm is not None else None)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_fuzz_fits(capi, orc, kind):
    rng = np.random.default_rng(1000 + kind)
    sizes = [4, 5, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4100, 7000]
    for it in range(26):
        n = int(sizes[it % len(sizes)])
        pts, nrm = _cloud(rng, kind, n)
        thr = float(10.0 ** rng.uniform(-3.3, -0.7))
        max_iter = int(rng.choice([1, 3, 64, 127, 129, 300, 700, 1100, 2500]))   # > 1024 with prob 1: lead pass inside the chunk
        prob = float(rng.choice([1.0, 0.9999, 0.99, 0.5]))
        seed = int(rng.integers(1 << 31))
        o = orc.fit(kind, pts, nrm, thr=thr, max_iter=max_iter, prob=prob, seed=seed)
        g = capi.fit(kind, pts, nrm, thr, max_iter, prob, seed=seed)
        tag = (kind, it, n, thr, max_iter, prob, seed)
        assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (
            o.ret, o.best_index, o.count, o.iterations), tag
        assert np.array_equal(g.inliers.astype(np.uint64), o.inliers.astype(np.uint64)), tag
        if o.ret == 1 and np.isfinite(o.params).all():
            scale = max(1.0, float(np.abs(o.params).max()))
            assert np.allclose(g.params, o.params, rtol=0, atol=PARAM_TOL * scale), tag


def test_fuzz_segmentation(capi, orc):
    rng = np.random.default_rng(77)
    for it in range(10):
        n = int(rng.choice([300, 1000, 2049, 5000]))
        pts = np.ascontiguousarray(synth.room_cloud_c5(n, int(rng.integers(1 << 30))))
        if it % 3 == 1:
            pts[rng.integers(0, n, 4)] = np.nan
        thr = float(rng.choice([0.005, 0.02, 0.08]))
        mi = int(rng.choice([20, 100, 333]))
        mr = float(rng.choice([0.05, 0.3, 0.6]))
        seed = int(rng.integers(1 << 31))
        orc_rc, oplanes, oclusters = orc.segment_plane_iterative(pts, thr, mi, mr, seed=seed)
        rc, planes, clusters = capi.segment_plane_iterative(pts, thr, max_iteration=mi, min_ratio=mr, seed=seed)
        assert (rc, orc_rc) in ((1, 0), (2, 2)), (it, rc, orc_rc)
        assert len(planes) == len(oplanes)
        for a, b in zip(clusters, oclusters):
            assert np.array_equal(a.astype(np.uint64), b.astype(np.uint64)), it
        assert np.allclose(planes, oplanes, rtol=0, atol=PARAM_TOL)


def test_fuzz_registration_matcher_icp(capi, orc):
    rng = np.random.default_rng(5)
    for it in range(8):
        n = int(rng.choice([200, 500, 900, 1300]))
        dim = int(rng.choice([33, 33, 8, 5]))
        d = synth.registration_pair_c4(n, seed=int(rng.integers(1 << 30)), dim=dim, true_fraction=float(rng.uniform(0.3, 0.8)),
                                       sigma=float(rng.choice([5e-4, 2e-3])))
        a, b = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
        oa, ob = orc.match_mutual_nn(d["feat_src"], d["feat_dst"])
        assert np.array_equal(a.astype(np.int64), oa) and np.array_equal(b.astype(np.int64), ob), it
        if len(oa) < 3:
            continue
        thr = float(rng.choice([0.01, 0.03, 0.1]))
        mi = int(rng.choice([50, 300, 600]))
        conf = float(rng.choice([1.0, 0.999]))
        seed = int(rng.integers(1 << 31))
        o = orc.registration_ransac(d["src"], d["dst"], oa, ob, thr=thr, max_iter=mi, edge_thr=0.9, confidence=conf, seed=seed)
        T, st = capi.registration_ransac(d["src"], d["dst"], a, b, threshold=thr, max_iter=mi, edge_length_threshold=0.9,
                                         confidence=conf, seed=seed)
        tag = (it, n, dim, thr, mi, conf, seed)
        assert (st["best_index"], st["iterations"], st["validations"], st["est_k"]) == (
            o.best_index, o.iterations, o.validations, o.est_k), tag
        assert np.array_equal(T.view(np.uint64), o.T.view(np.uint64)) and st["fitness"] == o.fitness, tag
        if o.best_index >= 0:
            Ti, sti, corr = capi.registration_icp(d["src"], d["dst"], thr, T, max_iteration=3, want_correspondences=True)
            oT, ofit, orm, oit, ocorr = orc.registration_icp(d["src"], d["dst"], thr, o.T, max_iter=3)
            assert sti["iterations"] == oit and np.array_equal(corr, ocorr) and sti["fitness"] == ofit, tag
            assert np.allclose(Ti, oT, rtol=0, atol=1e-9), tag


def test_fuzz_normals(capi, orc):
    rng = np.random.default_rng(9)
    for it in range(16):
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        k = int(rng.integers(0, 9))
        u, v = np.meshgrid(np.arange(w), np.arange(h))
        z = 1.0 + rng.uniform(-0.01, 0.01) * u + rng.uniform(-0.01, 0.01) * v + rng.normal(0, 1e-3, (h, w))
        xyz = np.stack([(u - w / 2) / 90.0 * z, (v - h / 2) / 90.0 * z, z], -1).reshape(-1, 3)
        xyz[rng.random(w * h) < rng.choice([0.0, 0.05, 0.5])] = np.nan
        vp = rng.uniform(-1, 1, 3)
        got, ref = capi.normals_from_map(xyz, w, h, k, vp), orc.normals_from_map(xyz, w, h, k, vp)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), (it, w, h, k)
        assert np.array_equal(np.nan_to_num(got).view(np.uint64), np.nan_to_num(ref).view(np.uint64)), (it, w, h, k)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_lead_pass_and_many_chunks_against_oracle(capi, orc, kind):
    """probability 1 with more than 1024 hypotheses: the first chunk counts its leading hypotheses on their own and
    prunes the rest with their best count; 40 000 hypotheses span three chunks.  Same best index, count and inlier
    list as the sequential oracle."""
    rng = np.random.default_rng(500 + kind)
    for n, H in ((3000, 1030), (20000, 5000), (9000, 40000)):
        pts, nrm = _cloud(rng, kind, n)
        seed = int(rng.integers(1 << 31))
        o = orc.fit(kind, pts, nrm, thr=0.01, max_iter=H, prob=1.0, seed=seed)
        g = capi.fit(kind, pts, nrm, 0.01, H, 1.0, seed=seed)
        tag = (kind, n, H, seed)
        assert (g.ret, g.stats["best_index"], g.stats["count"], g.stats["iterations"]) == (
            o.ret, o.best_index, o.count, o.iterations), tag
        assert np.array_equal(g.inliers.astype(np.uint64), o.inliers.astype(np.uint64)), tag
