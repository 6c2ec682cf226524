"""DetectBoundaryPoints on the GPU against the oracle: identical index sets, with given
normals and with normals estimated from the neighbourhood, for Hybrid and Radius searches; API checks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _patch(n, seed, holes=True):
    rng = np.random.default_rng(seed)
    uv = rng.uniform(0, 1, (n, 2))
    if holes:                                            # a disc cut out of the patch: inner boundary too
        uv = uv[np.hypot(uv[:, 0] - 0.5, uv[:, 1] - 0.5) > 0.18]
    pts = np.c_[uv[:, 0], uv[:, 1], 0.3 * uv[:, 0] - 0.2 * uv[:, 1] + rng.normal(0, 1e-4, len(uv))]
    nrm = np.tile(np.array([-0.3, 0.2, 1.0]) / np.linalg.norm([-0.3, 0.2, 1.0]), (len(uv), 1))
    return np.ascontiguousarray(pts), nrm, uv


@pytest.mark.parametrize("search,radius,max_nn", [(2, 0.05, 30), (2, 0.08, 12), (2, 0.03, 128), (1, 0.04, 0),
                                                  (0, 0.0, 30), (0, 0.0, 7), (0, 0.0, 128)])
@pytest.mark.parametrize("with_normals", [True, False])
def test_boundary_matches_oracle(capi, orc, search, radius, max_nn, with_normals):
    pts, nrm, uv = _patch(3500, seed=int(radius * 1000) + max_nn)
    pts[[5, 99]] = pts[7]                                # coincident points (skipped as neighbours, :36-38)
    n_in = nrm if with_normals else None
    got = capi.detect_boundary_points(pts, n_in, search, radius, max_nn, 90.0)
    ref = orc.detect_boundary_points(pts, n_in, search, radius, max_nn, 90.0)
    assert np.array_equal(got.astype(np.int64), ref)
    assert len(ref) > 50
    if radius >= 0.05 and max_nn >= 30:   # dense enough neighbourhoods: only real edges are flagged
        edge = np.minimum.reduce([uv[:, 0], 1 - uv[:, 0], uv[:, 1], 1 - uv[:, 1],
                                  np.abs(np.hypot(uv[:, 0] - 0.5, uv[:, 1] - 0.5) - 0.18)])
        assert len(ref) < len(pts) // 2 and np.median(edge[ref]) < radius    # outer edge or the hole's rim


def test_boundary_threshold_and_errors(capi, orc):
    pts, nrm, _ = _patch(1500, seed=3, holes=False)
    for thr in (30.0, 120.0, 200.0):
        got = capi.detect_boundary_points(pts, nrm, 2, 0.07, 30, thr)
        assert np.array_equal(got.astype(np.int64), orc.detect_boundary_points(pts, nrm, 2, 0.07, 30, thr))
    far = pts * 50.0                                     # nobody has 3 neighbours within the radius
    assert len(capi.detect_boundary_points(far, None, 2, 0.05, 30, 90.0)) == 0
    with pytest.raises(capi.M3DError):
        capi.detect_boundary_points(pts[:0], None, 2, 0.05, 30, 90.0)      # "No PointCloud data."
    with pytest.raises(capi.M3DError):
        capi.detect_boundary_points(pts, None, 1, 0.5, 0, 90.0)            # radius search with > 128 neighbours
    with pytest.raises(capi.M3DError):
        capi.detect_boundary_points(pts, None, 0, 0.05, 200, 90.0)         # at most 128 neighbours
    few = pts[:20]                                                          # fewer points than k: all of them are used
    assert np.array_equal(capi.detect_boundary_points(few, None, 0, 0.0, 30, 90.0).astype(np.int64),
                          orc.detect_boundary_points(few, None, 0, 0.0, 30, 90.0))
    blob = np.ascontiguousarray(np.random.default_rng(4).normal(size=(2500, 3)))   # volume-filling cloud, KNN
    assert np.array_equal(capi.detect_boundary_points(blob, None, 0, 0.0, 20, 90.0).astype(np.int64),
                          orc.detect_boundary_points(blob, None, 0, 0.0, 20, 90.0))


def test_python_api_detect_boundary_points(capi):
    import misc3d_amd as m3d
    pts, nrm, _ = _patch(1200, seed=9)
    ref = capi.detect_boundary_points(pts, nrm, 2, 0.06, 30, 90.0).tolist()
    assert m3d.features.detect_boundary_points((pts, nrm), ("hybrid", 0.06, 30)) == ref

    class Param:                                          # duck-typed open3d.geometry.KDTreeSearchParamHybrid
        radius, max_nn = 0.06, 30

    class Cloud:
        points, normals = pts, nrm

    assert m3d.features.detect_boundary_points(Cloud(), Param()) == ref

    class Knn:                                            # duck-typed open3d.geometry.KDTreeSearchParamKNN
        knn = 25

    assert m3d.features.detect_boundary_points(Cloud(), Knn()) == capi.detect_boundary_points(pts, nrm, 0, 0.0, 25).tolist()
    assert m3d.features.detect_boundary_points(pts, ("knn", 25)) == capi.detect_boundary_points(pts, None, 0, 0.0, 25).tolist()
