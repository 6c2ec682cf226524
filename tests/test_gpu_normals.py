"""EstimateNormalsFromMap on the GPU against the oracle: the box sums replicate the
reference's summation order and both sides run the same J3x3 eigen-solver, so the normals are compared
BIT FOR BIT; plus properties at the example's full size (848 x 480, k = 3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _depth_map(w, h, seed, holes=0.03):
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    z = 1.0 + 0.001 * u + 0.002 * v + 0.05 * np.exp(-((u - w / 2) ** 2 + (v - h / 2) ** 2) / (0.02 * w * w))
    z = z + rng.normal(0, 1e-4, (h, w))
    xyz = np.stack([(u - w / 2) / 200.0 * z, (v - h / 2) / 200.0 * z, z], -1).reshape(-1, 3)
    xyz[rng.random(w * h) < holes] = np.nan
    return xyz


@pytest.mark.parametrize("w,h,k", [(160, 120, 3), (97, 61, 5), (64, 48, 1), (50, 40, 0), (40, 30, 9), (7, 5, 2)])
def test_normals_match_oracle_bitwise(capi, orc, w, h, k):
    xyz = _depth_map(w, h, seed=w + k)
    vp = (0.1, -0.2, -0.5)
    got = capi.normals_from_map(xyz, w, h, k, vp)
    ref = orc.normals_from_map(xyz, w, h, k, vp)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref[:, 0])
    assert np.array_equal(got[ok].view(np.uint64), ref[ok].view(np.uint64))
    assert np.isnan(got[np.isnan(xyz[:, 2])]).all()


def test_normals_edge_cases(capi, orc):
    xyz = _depth_map(32, 24, seed=5, holes=0.0)
    xyz[:] = np.nan                                         # nothing valid
    assert np.isnan(capi.normals_from_map(xyz, 32, 24, 3)).all()
    one = _depth_map(32, 24, seed=6, holes=0.0)
    keep = one[100].copy()
    one[:] = np.nan
    one[100] = keep                                         # a single valid pixel: zero covariance
    got, ref = capi.normals_from_map(one, 32, 24, 3), orc.normals_from_map(one, 32, 24, 3)
    assert np.array_equal(got[100].view(np.uint64), ref[100].view(np.uint64))
    with pytest.raises(capi.M3DError):
        capi.normals_from_map(one, 31, 24, 3)               # size mismatch (normal_estimation.cpp:187-191)


def test_normals_full_size_properties(capi):
    """the reference example's size (examples/cpp/normal_estimation.cpp:32: 848 x 480, k = 3)"""
    w, h, k = 848, 480, 3
    xyz = _depth_map(w, h, seed=1)
    n, ms = capi.normals_from_map(xyz, w, h, k, want_ms=True)
    ok = ~np.isnan(xyz[:, 2])
    assert np.isnan(n[~ok]).all() and not np.isnan(n[ok]).any()
    assert np.abs(np.linalg.norm(n[ok], axis=1) - 1).max() < 1e-12
    assert (np.einsum("ij,ij->i", -xyz[ok], n[ok]) >= 0).all()          # oriented towards the view point (origin)
    # a gently sloped surface seen from the origin: normals point back along -z
    assert np.median(n[ok][:, 2]) < -0.9
    assert ms > 0


def test_python_api_estimate_normals(capi):
    """common.estimate_normals(pc, shape, k=5, view_point=[0,0,0]) (python/py_common.cpp:79-89): ndarray in ->
    normals out; duck-typed cloud in -> its .normals set and the object returned; size mismatch raises."""
    import misc3d_amd as m3d
    w, h = 64, 40
    xyz = _depth_map(w, h, seed=3)
    n = m3d.common.estimate_normals(xyz, (w, h), 3)
    assert np.array_equal(np.nan_to_num(n), np.nan_to_num(capi.normals_from_map(xyz, w, h, 3)))

    class Cloud:
        def __init__(self, p):
            self.points, self.normals = p, None

    c = Cloud(xyz)
    assert m3d.common.estimate_normals(c, (w, h)) is c
    assert np.array_equal(np.nan_to_num(np.asarray(c.normals)), np.nan_to_num(capi.normals_from_map(xyz, w, h, 5)))
    with pytest.raises(RuntimeError, match="not equal to given point map size"):
        m3d.common.estimate_normals(xyz, (w + 1, h), 3)
