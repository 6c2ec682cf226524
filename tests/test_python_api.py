"""The reference's python API surface (python/py_common.cpp, py_registration.cpp, py_segmentation.cpp)
on the pybind host module.  CPU part: names, defaults, enum values, error convention (RuntimeError
with the reference's messages), no silent fallback.  GPU part: results equal the oracle's."""
import inspect

import numpy as np
import pytest

from misc3d_amd import synth


@pytest.fixture(scope="module")
def m3d():
    import misc3d_amd
    misc3d_amd.set_verbosity_level(misc3d_amd.VerbosityLevel.Error)
    return misc3d_amd


def test_module_layout_and_signatures(m3d):
    for sub, names in (("common", ["fit_plane", "fit_sphere", "fit_cylinder"]),
                       ("segmentation", ["segment_plane_iterative"]),
                       ("registration", ["compute_transformation_least_square", "compute_transformation_ransac",
                                         "compute_transformation_teaser", "match_correspondence", "MatchMethod"])):
        for n in names:
            assert hasattr(getattr(m3d, sub), n), (sub, n)
    import re

    def has_default(fn, name, value):
        return re.search(rf"{name}:[^,)]*= {re.escape(value)}[,)]", fn.__doc__) is not None

    for fn in (m3d.common.fit_plane, m3d.common.fit_sphere, m3d.common.fit_cylinder):   # py_common.cpp:70-78
        assert has_default(fn, "threshold", "0.01") and has_default(fn, "max_iteration", "1000")
        assert has_default(fn, "probability", "0.9999") and fn.__doc__.index("pc:") < fn.__doc__.index("threshold:")
    fn = m3d.registration.compute_transformation_ransac                                    # py_registration.cpp:55-67
    assert has_default(fn, "threshold", "0.01") and has_default(fn, "max_iter", "100000")
    assert has_default(fn, "edge_length_threshold", "0.9")
    fn = m3d.segmentation.segment_plane_iterative                                          # py_segmentation.cpp:87-96
    assert has_default(fn, "max_iteration", "100") and has_default(fn, "min_ratio", "0.05")
    fn = m3d.registration.match_correspondence                                             # py_registration.cpp:73-106
    assert has_default(fn, "n_trees", "4") and "ANNOY" in fn.__doc__
    assert has_default(m3d.registration.compute_transformation_least_square, "scaling", "False")
    MM = m3d.registration.MatchMethod
    assert int(MM.FLANN) == 0 and int(MM.ANNOY) == 1 and m3d.registration.ANNOY == MM.ANNOY
    assert [int(v) for v in (m3d.Error, m3d.Warning, m3d.Info, m3d.Debug)] == [0, 1, 2, 3]
    m3d.set_verbosity_level(m3d.VerbosityLevel.Warning)
    assert m3d.get_verbosity_level() == m3d.VerbosityLevel.Warning
    m3d.set_verbosity_level(m3d.VerbosityLevel.Error)


def test_error_convention_without_gpu(m3d):
    pts = np.random.default_rng(0).normal(size=(50, 3))
    with pytest.raises(RuntimeError, match=r"\[Misc3D Error\] Probability must be > 0 or <= 1.0"):
        m3d.common.fit_plane(pts, 0.01, 100, 0.0)                                   # ransac.h:483-485
    with pytest.raises(RuntimeError, match=r"\[Misc3D Error\] Fit cylinder requires normals."):
        m3d.common.fit_cylinder(pts, 0.01, 100)                                     # py_common.cpp:50-52
    with pytest.raises(RuntimeError, match=r"\[Misc3D Error\] Can not fit model due to lack of points"):
        m3d.common.fit_plane(pts[:2], 0.01, 100)                                    # ransac.h:510-513
    with pytest.raises(RuntimeError, match=r"less than 3"):
        m3d.registration.compute_transformation_least_square(pts[:2], pts[:2])     # transform_estimation.cpp:17-19
    with pytest.raises(RuntimeError, match=r"not equal"):
        m3d.registration.compute_transformation_least_square(pts[:10], pts[:12])   # transform_estimation.cpp:20-22
    with pytest.raises(RuntimeError, match=r"less than 3"):
        m3d.registration.compute_transformation_ransac(pts[:2], pts, ([0, 1, 1], [0, 1, 2]))
    with pytest.raises(RuntimeError):
        m3d.registration.compute_transformation_teaser(pts, pts)
    assert m3d.segmentation.segment_plane_iterative(pts[:2], 0.01) == []            # :13-17: warning + empty list
    with pytest.raises((ValueError, TypeError)):
        m3d.common.fit_plane(np.zeros((5, 2)))
    # correspondence indices as arrays: integers only, no negative values (they would wrap to 2^64 - 1)
    with pytest.raises(TypeError, match="integers"):
        m3d.registration.compute_transformation_ransac(pts, pts, (np.array([0.0, 1.0, 2.0]), np.array([0, 1, 2])))
    with pytest.raises(ValueError, match="negative"):
        m3d.registration.compute_transformation_ransac(pts, pts, (np.array([0, -1, 2]), np.array([0, 1, 2])))
    m3d.release_host_scratch()   # (nothing pinned without a device: a no-op that must not fail)


def test_no_silent_fallback_without_gpu(m3d):
    if m3d.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="HIP device"):
        m3d.common.fit_plane(np.random.default_rng(0).normal(size=(100, 3)), 0.01, 10)


class _DuckCloud:
    """Stands in for open3d.geometry.PointCloud: exposes .points / .normals convertible by np.asarray."""

    def __init__(self, points, normals=None):
        self.points = points
        self.normals = normals if normals is not None else np.zeros((0, 3))


@pytest.mark.gpu
def test_fit_functions_match_oracle(m3d, orc):
    pts = synth.plane_cloud_c1(20000, 1)
    w, idx = m3d.common.fit_plane(_DuckCloud(pts), 0.01, 100, seed=7)
    o = orc.fit(orc.PLANE, pts, thr=0.01, max_iter=100, prob=0.9999, seed=7)
    assert isinstance(w, np.ndarray) and w.shape == (4,) and isinstance(idx, list)
    assert idx == o.inliers.tolist() and np.allclose(w, o.params, atol=1e-9)
    sp = synth.sphere_cloud_c3(20000, 4)
    w, idx = m3d.common.fit_sphere(sp, 0.01, 200, seed=3)
    o = orc.fit(orc.SPHERE, sp, thr=0.01, max_iter=200, prob=0.9999, seed=3)
    assert idx == o.inliers.tolist() and np.allclose(w, o.params, atol=1e-9)
    cp, cn = synth.cylinder_cloud_c3(20000, 3)
    w, idx = m3d.common.fit_cylinder(_DuckCloud(cp, cn), 0.01, 200, seed=5)
    o = orc.fit(orc.CYLINDER, cp, cn, thr=0.01, max_iter=200, prob=0.9999, seed=5)
    assert w.shape == (7,) and idx == o.inliers.tolist() and np.array_equal(w, o.params)
    w2, idx2 = m3d.common.fit_cylinder((cp, cn), 0.01, 200, seed=5)
    assert np.array_equal(w, w2) and idx == idx2
    # keyword-only extra: the indices as an int64 array instead of the reference's list[int]; many hypotheses take the
    # Hilbert-sorted path, few the dense one (same results)
    for it in (100, 3000):
        wa, ia = m3d.common.fit_plane(pts, 0.01, it, 1.0, seed=7, as_arrays=True)
        wl, il = m3d.common.fit_plane(pts, 0.01, it, 1.0, seed=7)
        ob = orc.fit(orc.PLANE, pts, thr=0.01, max_iter=it, prob=1.0, seed=7)
        assert isinstance(ia, np.ndarray) and ia.dtype == np.int64 and ia.tolist() == il == ob.inliers.tolist()
        assert np.array_equal(wa, wl) and np.allclose(wa, ob.params, atol=1e-9)
    # unseeded call (reference behaviour: std::random_device) still finds the plane
    w, idx = m3d.common.fit_plane(pts)
    assert abs(abs(w[2]) - 1) < 1e-2 and len(idx) > 0.5 * len(pts)


@pytest.mark.gpu
def test_soft_failure_zeroes_params(m3d):
    # max_iteration = 0 -> no model -> GeneralFit fails -> params zeroed, size 4 even for the cylinder
    cp, cn = synth.cylinder_cloud_c3(2000, 3)
    w, idx = m3d.common.fit_cylinder((cp, cn), 0.01, 0, seed=1)
    # cylinder GeneralFit is a no-op returning true (ransac.h:427-433): ret is true, 7 zero params
    assert w.shape == (7,) and np.all(w == 0)
    w, idx = m3d.common.fit_plane(cp, 0.01, 0, seed=1)
    assert w.shape == (4,) and np.all(w == 0) and idx == []


@pytest.mark.gpu
def test_segmentation_and_registration_api(m3d, orc):
    rng = np.random.default_rng(5)
    a = np.c_[rng.uniform(-1, 1, (6000, 2)), rng.normal(0, 2e-3, 6000)]
    b = np.c_[rng.normal(0, 2e-3, 4000) + 2.0, rng.uniform(-1, 1, (4000, 2))]
    pts = np.concatenate([a, b, rng.uniform(-3, 3, (500, 3))])[rng.permutation(10500)]
    res = m3d.segmentation.segment_plane_iterative(pts, 0.01, 100, 0.1, seed=3, return_indices=True)
    rc, oplanes, oclusters = orc.segment_plane_iterative(pts, 0.01, 100, 0.1, seed=3)
    assert len(res) == len(oplanes) >= 2
    for (plane, cloud, idx), op, oc in zip(res, oplanes, oclusters):
        assert np.allclose(plane, op, atol=1e-9)
        assert idx.dtype == np.int64 and np.array_equal(idx, oc.astype(np.int64))     # (rows of one index array)
        # the cluster's points are gathered on the device (m3d_segment_plane_iterative_clouds): SelectByIndex, bit for bit
        assert np.array_equal(np.asarray(cloud), pts[oc.astype(np.int64)])
    res2 = m3d.segmentation.segment_plane_iterative(pts, 0.01, 100, 0.1, seed=3)
    assert len(res2[0]) == 2
    # registration
    d = synth.registration_pair_c4(4000, seed=5)
    i0, i1 = m3d.registration.match_correspondence(d["feat_src"].T, d["feat_dst"].T)   # (dim, N) like Eigen
    oa, ob = orc.match_mutual_nn(d["feat_src"], d["feat_dst"])
    assert i0 == oa.tolist() and i1 == ob.tolist()
    T = m3d.registration.compute_transformation_ransac(d["src"], d["dst"], (i0, i1), 0.03, 3000, seed=17,
                                                       confidence=1.0)
    o = orc.registration_ransac(d["src"], d["dst"], oa, ob, thr=0.03, max_iter=3000, confidence=1.0, seed=17)
    assert T.shape == (4, 4) and np.array_equal(T, o.T) and np.allclose(T, d["T"], atol=0.01)
    # the same through integer arrays (keyword-only extra as_arrays; the lists of the reference's API cost an element-by-
    # element conversion each way): identical pairs, identical pose; int32 and uint64 arrays are accepted as well
    a0, a1 = m3d.registration.match_correspondence(d["feat_src"].T, d["feat_dst"].T, as_arrays=True)
    assert isinstance(a0, np.ndarray) and a0.dtype == np.int64 and a0.tolist() == i0 and a1.tolist() == i1
    for conv in (lambda v: v, lambda v: v.astype(np.int32), lambda v: v.astype(np.uint64)):
        T2 = m3d.registration.compute_transformation_ransac(d["src"], d["dst"], (conv(a0), conv(a1)), 0.03, 3000, seed=17,
                                                            confidence=1.0)
        assert np.array_equal(T, T2)
    with pytest.raises(ValueError):
        m3d.registration.compute_transformation_ransac(d["src"], d["dst"], (i0, i1, i1), 0.03, 3000)
    inv = np.empty(4000, dtype=np.int64)
    inv[d["perm"]] = np.arange(4000)
    Tl = m3d.registration.compute_transformation_least_square(d["src"], d["dst"][inv])
    assert np.allclose(Tl, orc.umeyama(d["src"], d["dst"][inv]), atol=1e-9) and np.allclose(Tl, d["T"], atol=1e-3)


@pytest.mark.gpu
def test_cpp_host_api_executable():
    """include/misc3d/** used from plain C++ (the shape of the reference's examples/cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "_build", "test_host_api")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_host_minimal_fit_matches_oracle(orc, kind):
    """m3d_minimal_fit (host-side MinimalFit used by the sharded driver for the winning hypothesis) is
    bit-identical to the oracle's MinimalFit, degenerate samples included.  Pure host code: no GPU needed."""
    import ctypes as C
    from misc3d_amd import capi, synth
    if kind == 2:
        pts, nrm = synth.cylinder_cloud_c3(3000, 3)
    else:
        pts, nrm = (synth.sphere_cloud_c3(3000, 4) if kind == 1 else synth.plane_cloud_c1(3000, 1)), None
    m = capi.MINIMAL_SAMPLE[kind]
    samples = capi.draw_samples(len(pts), kind, 400, 5)
    samples[7] = samples[7][0]              # degenerate: repeated point
    v, models, _, _ = orc.score_samples(kind, pts, nrm, 0.01, samples.astype(np.uint64))
    for h in range(len(samples)):
        p = np.ascontiguousarray(pts[samples[h]])
        n = np.ascontiguousarray(nrm[samples[h]]) if nrm is not None else None
        out = np.zeros(8)
        ok = C.c_uint8(9)
        rc = capi.lib().m3d_minimal_fit(kind, p.ctypes.data_as(C.c_void_p),
                                        n.ctypes.data_as(C.c_void_p) if n is not None else None,
                                        out.ctypes.data_as(C.c_void_p), C.cast(C.byref(ok), C.c_void_p))
        assert rc == 1 and ok.value == v[h]
        if v[h]:
            k = capi.NUM_PARAMS[kind]
            assert np.array_equal(out[:k].view(np.uint64), models[h][:k].view(np.uint64)), h
    assert m in (2, 3, 4)


def test_ply_reader_writer_roundtrip(tmp_path):
    """misc3d_amd.io: the Open3D-free PLY reader for the reference's example clouds (binary little-endian doubles,
    ascii, float/uchar properties, extra elements ignored)."""
    from misc3d_amd import io
    rng = np.random.default_rng(0)
    pts, nrm = rng.normal(size=(257, 3)), rng.normal(size=(257, 3))
    for binary in (True, False):
        p = str(tmp_path / f"c{int(binary)}.ply")
        io.write_ply(p, pts, nrm, binary=binary)
        d = io.read_ply(p)
        assert np.array_equal(d["points"], pts) and np.array_equal(d["normals"], nrm) and d["colors"] is None
    # float xyz + uchar colours + a face element behind the vertices
    p = str(tmp_path / "mixed.ply")
    rec = np.zeros(5, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = np.arange(5), np.arange(5) * 2, np.arange(5) * 3
    rec["red"], rec["green"], rec["blue"] = 255, 0, 51
    with open(p, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\ncomment made by a test\nelement vertex 5\nproperty float x\n"
                b"property float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                b"element face 1\nproperty list uchar int vertex_indices\nend_header\n")
        f.write(rec.tobytes())
        f.write(bytes([3]) + np.array([0, 1, 2], dtype="<i4").tobytes())
    d = io.read_ply(p)
    assert np.array_equal(d["points"][:, 1], np.arange(5) * 2.0) and d["normals"] is None
    assert np.allclose(d["colors"][0], [1.0, 0.0, 0.2])
    with pytest.raises(ValueError):
        (tmp_path / "bad.ply").write_bytes(b"plx\n")
        io.read_ply(str(tmp_path / "bad.ply"))
