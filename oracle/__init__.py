"""CPU oracle for the Misc3D RANSAC hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``misc3d_amd``) never does.  PARITY UNPINNED: see the header of
``misc3d_oracle.c`` and DESIGN.md.

The arithmetic lives in plain C (``misc3d_oracle.c``, ``misc3d_oracle_reg.c``, ``misc3d_oracle_normals.c``); this module is a
ctypes shim that builds the shared object on demand with ``make -C oracle``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# M3D_FP_ORDER=1|2: the oracle built for the same alternative floating-point association as the product library
FP_ORDER = int(os.environ.get("M3D_FP_ORDER", "0") or 0)
_LIB_PATH = os.path.join(_HERE, "_build", *([f"order{FP_ORDER}"] if FP_ORDER else []), "libmisc3d_oracle.so")
RNG_CHECK_PATH = os.path.join(_HERE, "_build", "std_rng_check")

PLANE, SPHERE, CYLINDER = 0, 1, 2
_M = {PLANE: 3, SPHERE: 4, CYLINDER: 2}
_NP = {PLANE: 4, SPHERE: 4, CYLINDER: 7}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("misc3d_oracle.c", "misc3d_oracle_reg.c", "misc3d_oracle_normals.c", "misc3d_oracle_boundary.c", "std_rng_check.cpp", "Makefile")]
    srcs = srcs  # (one make builds the default and both alternative associations)
    stale = force or not os.path.exists(_LIB_PATH) or not os.path.exists(RNG_CHECK_PATH)
    if not stale:
        t = min(os.path.getmtime(_LIB_PATH), os.path.getmtime(RNG_CHECK_PATH))
        stale = any(os.path.getmtime(s) > t for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib = None


class _Stats(C.Structure):
    _fields_ = [("fitness", C.c_double), ("inlier_rmse", C.c_double), ("count", C.c_uint64),
                ("iterations", C.c_uint64), ("best_index", C.c_int64), ("general_fit_ok", C.c_int)]


class _Trace(C.Structure):
    _fields_ = [("samples", C.c_void_p), ("valid", C.c_void_p), ("models", C.c_void_p),
                ("counts", C.c_void_p), ("errors", C.c_void_p)]


class _RegStats(C.Structure):
    _fields_ = [("fitness", C.c_double), ("inlier_rmse", C.c_double), ("validations", C.c_uint64),
                ("iterations", C.c_int64), ("best_index", C.c_int64), ("est_k", C.c_int64)]


class _RegTrace(C.Structure):
    _fields_ = [("triples", C.c_void_p), ("T", C.c_void_p), ("passed", C.c_void_p),
                ("counts", C.c_void_p), ("err2", C.c_void_p)]


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a container can see 128 CPUs
    and own 16 of them; an OpenMP team of 128 on a quota of 16 spends its time in barriers: the look-ahead fit took
    9.9 s instead of 0.1 s on the GPU box)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        if "OMP_NUM_THREADS" not in os.environ:      # (an explicit setting wins)
            _lib.orc_set_omp_threads(C.c_int(usable_cpus()))
        _lib.orc_plane_distance.restype = C.c_double
        _lib.orc_sphere_distance.restype = C.c_double
        _lib.orc_cylinder_distance.restype = C.c_double
        _lib.orc_point2line.restype = C.c_double
        _lib.orc_distance.restype = C.c_double
        _lib.orc_mt_next.restype = C.c_uint32
        _lib.orc_uniform_int.restype = C.c_uint32
        _lib.orc_double_to_size_t_x86.restype = C.c_uint64
        _lib.orc_double_to_size_t_x86.argtypes = [C.c_double]
        _lib.orc_reg_corr_inlier_ratio.restype = C.c_double
        _lib.orc_match_mutual_nn.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


# ------------------------------------------------------------------------------------------------
# RNG
# ------------------------------------------------------------------------------------------------
class MT19937:
    def __init__(self, seed: int):
        self._buf = C.create_string_buffer(624 * 4 + 8)
        lib().orc_mt_seed(self._buf, C.c_uint64(seed))

    def next(self) -> int:
        return int(lib().orc_mt_next(self._buf))

    def uniform_int(self, range_incl: int) -> int:
        return int(lib().orc_uniform_int(self._buf, C.c_uint32(range_incl)))

    def sample(self, size: int, m: int):
        out = np.zeros(m, dtype=np.uint64)
        lib().orc_sample(self._buf, C.c_size_t(size), C.c_int(m), _p(out))
        return out


def draw_samples(n: int, m: int, H: int, seed: int):
    out = np.zeros((H, m), dtype=np.uint64)
    lib().orc_draw_samples(C.c_size_t(n), C.c_int(m), C.c_size_t(H), C.c_uint64(seed), _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# estimators
# ------------------------------------------------------------------------------------------------
def plane_minimal_fit(p):
    p = _f64(p, (9,))
    out = np.zeros(4)
    ok = lib().orc_plane_minimal_fit(_p(p), _p(out))
    return bool(ok), out


def sphere_minimal_fit(p):
    p = _f64(p, (12,))
    out = np.zeros(4)
    ok = lib().orc_sphere_minimal_fit(_p(p), _p(out))
    return bool(ok), out


def cylinder_minimal_fit(p, nrm):
    p = _f64(p, (6,))
    nrm = _f64(nrm, (6,))
    out = np.zeros(7)
    ok = lib().orc_cylinder_minimal_fit(_p(p), _p(nrm), _p(out))
    return bool(ok), out


def distance(kind, q, model):
    q = _f64(q, (3,))
    model = _f64(model)
    return float(lib().orc_distance(C.c_int(kind), _p(q), _p(model)))


def plane_general_fit(pts):
    pts = _f64(pts).reshape(-1, 3)
    out = np.zeros(4)
    ok = lib().orc_plane_general_fit(_p(pts), C.c_size_t(len(pts)), _p(out))
    return bool(ok), out


def sphere_general_fit(pts):
    pts = _f64(pts).reshape(-1, 3)
    out = np.zeros(4)
    ok = lib().orc_sphere_general_fit(_p(pts), C.c_size_t(len(pts)), _p(out))
    return bool(ok), out


def evaluate_model(kind, xyz, thr, model):
    xyz = _f64(xyz).reshape(-1, 3)
    model = _f64(model)
    cnt = C.c_uint64(0)
    err = C.c_double(0)
    lib().orc_evaluate_model(C.c_int(kind), _p(xyz), C.c_size_t(len(xyz)), C.c_double(thr), _p(model),
                             C.byref(cnt), C.byref(err))
    return int(cnt.value), float(err.value)


@dataclass
class FitResult:
    ret: int
    params: np.ndarray
    inliers: np.ndarray
    fitness: float
    inlier_rmse: float
    count: int
    iterations: int
    best_index: int
    trace: dict | None = None


def fit(kind, xyz, normals=None, thr=0.01, max_iter=1000, prob=0.9999, seed=0, trace=False, lookahead=1) -> FitResult:
    """Sequential seeded restatement of RANSAC::FitModel (ransac.h:506-624).  lookahead > 1: the same loop with the
    records of the next `lookahead` hypotheses computed ahead by an OpenMP team (orc_fit_parallel; identical outputs,
    for the full-size BASELINE configurations)."""
    xyz = _f64(xyz).reshape(-1, 3)
    n = len(xyz)
    nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
    params = np.zeros(_NP[kind])
    inl = np.zeros(max(n, 1), dtype=np.uint64)
    ni = C.c_size_t(0)
    st = _Stats()
    tr = None
    tr_arrays = None
    if trace:
        m = _M[kind]
        tr_arrays = dict(samples=np.zeros((max_iter, m), dtype=np.uint64), valid=np.zeros(max_iter, dtype=np.int32),
                         models=np.zeros((max_iter, _NP[kind])), counts=np.zeros(max_iter, dtype=np.uint64),
                         errors=np.zeros(max_iter))
        tr = _Trace(*[_p(tr_arrays[k]) for k in ("samples", "valid", "models", "counts", "errors")])
    args = (C.c_int(kind), _p(xyz), _p(nrm), C.c_size_t(n), C.c_double(thr), C.c_size_t(max_iter),
            C.c_double(prob), C.c_uint64(seed), _p(params), _p(inl), C.byref(ni), C.byref(st),
            C.byref(tr) if tr is not None else None)
    if lookahead > 1:
        ret = lib().orc_fit_parallel(*args, C.c_size_t(lookahead))
    else:
        ret = lib().orc_fit(*args)
    return FitResult(ret, params, inl[: ni.value].copy(), st.fitness, st.inlier_rmse, int(st.count),
                     int(st.iterations), int(st.best_index), tr_arrays)


def refine(kind, xyz, thr, model):
    """RefineModel (ransac.h:534-549): (ret, refined params, inlier indices)."""
    xyz = _f64(xyz).reshape(-1, 3)
    m = np.zeros(_NP[kind])
    m[:] = _f64(model)[: _NP[kind]]
    inl = np.zeros(max(len(xyz), 1), dtype=np.uint64)
    ni = C.c_size_t(0)
    ok = lib().orc_refine(C.c_int(kind), _p(xyz), C.c_size_t(len(xyz)), C.c_double(thr), _p(m), _p(inl),
                          C.byref(ni))
    return int(ok), m, inl[: ni.value].copy()


def score_samples(kind, xyz, normals, thr, samples):
    xyz = _f64(xyz).reshape(-1, 3)
    nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
    samples = np.ascontiguousarray(samples, dtype=np.uint64)
    H = len(samples)
    valid = np.zeros(H, dtype=np.int32)
    models = np.zeros((H, _NP[kind]))
    counts = np.zeros(H, dtype=np.uint64)
    errors = np.zeros(H)
    lib().orc_score_samples(C.c_int(kind), _p(xyz), _p(nrm), C.c_size_t(len(xyz)), C.c_double(thr), _p(samples),
                            C.c_size_t(H), _p(valid), _p(models), _p(counts), _p(errors))
    return valid, models, counts, errors


def fit_omp_baseline(kind, xyz, normals, thr, H, seed):
    xyz = _f64(xyz).reshape(-1, 3)
    nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
    model = np.zeros(_NP[kind])
    cnt = C.c_uint64(0)
    bi = C.c_int64(-1)
    lib().orc_fit_omp_baseline(C.c_int(kind), _p(xyz), _p(nrm), C.c_size_t(len(xyz)), C.c_double(thr),
                               C.c_size_t(H), C.c_uint64(seed), _p(model), C.byref(cnt), C.byref(bi))
    return model, int(cnt.value), int(bi.value)


def omp_threads() -> int:
    return int(lib().orc_omp_threads())


def set_omp_threads(n: int) -> None:
    lib().orc_set_omp_threads(C.c_int(int(n)))


_native = None


def native_baseline():
    """(fit_omp_baseline, set_omp_threads) of the SAME source built `-O3 -march=native` (SURVEY.md 8(d) asks for that
    number beside the reference-shaped one).  Compiled where it runs -- a binary tuned for this container's CPU need not
    run on the GPU box's -- into a temporary directory (`make -C oracle native NATIVE_DIR=...`)."""
    global _native
    if _native is None:
        import tempfile
        d = tempfile.mkdtemp(prefix="m3d_oracle_native_")
        subprocess.run(["make", "-C", _HERE, "-s", "native", f"NATIVE_DIR={d}"], check=True)
        L = C.CDLL(os.path.join(d, "libmisc3d_oracle_native.so"))

        def fit(kind, xyz, normals, thr, H, seed):
            xyz = _f64(xyz).reshape(-1, 3)
            nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
            model = np.zeros(_NP[kind])
            cnt = C.c_uint64(0)
            bi = C.c_int64(-1)
            L.orc_fit_omp_baseline(C.c_int(kind), _p(xyz), _p(nrm), C.c_size_t(len(xyz)), C.c_double(thr),
                                   C.c_size_t(H), C.c_uint64(seed), _p(model), C.byref(cnt), C.byref(bi))
            return model, int(cnt.value), int(bi.value)

        _native = (fit, lambda n: L.orc_set_omp_threads(C.c_int(int(n))))
    return _native


def segment_plane_iterative(xyz, thr, max_iteration=100, min_ratio=0.05, seed=0, max_clusters=4096, lookahead=1):
    xyz = _f64(xyz).reshape(-1, 3)
    n = len(xyz)
    planes = np.zeros((max_clusters, 4))
    offs = np.zeros(max_clusters + 1, dtype=np.uint64)
    idx = np.zeros(max(n, 1), dtype=np.uint64)
    k = C.c_size_t(0)
    args = (_p(xyz), C.c_size_t(n), C.c_double(thr), C.c_int(max_iteration), C.c_double(min_ratio), C.c_uint64(seed),
            C.c_size_t(max_clusters), _p(planes), _p(offs), _p(idx), C.byref(k))
    if lookahead > 1:
        rc = lib().orc_segment_plane_iterative_parallel(*args, C.c_size_t(lookahead))
    else:
        rc = lib().orc_segment_plane_iterative(*args)
    k = k.value
    return rc, planes[:k].copy(), [idx[int(offs[i]): int(offs[i + 1])].copy() for i in range(k)]


# ------------------------------------------------------------------------------------------------
# registration
# ------------------------------------------------------------------------------------------------
def k3x3_rotation(sigma):
    sigma = _f64(sigma, (9,))
    R = np.zeros(9)
    sv = np.zeros(3)
    lib().orc_k3x3_rotation(_p(sigma), _p(R), _p(sv))
    return R.reshape(3, 3), sv


def umeyama(src, dst, with_scaling=False):
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    T = np.zeros(16)
    lib().orc_umeyama(_p(src), _p(dst), C.c_size_t(len(src)), C.c_int(int(with_scaling)), _p(T))
    return T.reshape(4, 4)


def reg_checkers(ps, pd, T, edge_thr, dist_thr):
    """orc_reg_checkers on the 3 sampled correspondences given as points (3 x 3 each) -> bool"""
    ps = _f64(ps, (9,))
    pd = _f64(pd, (9,))
    T = _f64(T, (16,))
    ids = np.arange(3, dtype=np.int64)
    return bool(lib().orc_reg_checkers(_p(ps), _p(pd), _p(ids), _p(ids), _p(T), C.c_double(edge_thr), C.c_double(dist_thr)))


def reg_validate(src, dst, T, thr):
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    T = _f64(T, (16,))
    cnt = C.c_uint64(0)
    e2 = C.c_double(0)
    lib().orc_reg_validate(_p(src), C.c_size_t(len(src)), _p(dst), C.c_size_t(len(dst)), _p(T), C.c_double(thr),
                           C.byref(cnt), C.byref(e2))
    return int(cnt.value), float(e2.value)


@dataclass
class RegResult:
    ret: int
    T: np.ndarray
    fitness: float
    inlier_rmse: float
    validations: int
    iterations: int
    best_index: int
    est_k: int
    trace: dict | None = None


def registration_ransac(src, dst, corr_src, corr_dst, thr=0.01, max_iter=100000, edge_thr=0.9, confidence=0.999,
                        seed=0, trace=False) -> RegResult:
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    cs = np.ascontiguousarray(corr_src, dtype=np.int64)
    cd = np.ascontiguousarray(corr_dst, dtype=np.int64)
    T = np.zeros(16)
    st = _RegStats()
    tr = None
    arrs = None
    if trace:
        arrs = dict(triples=np.zeros((max_iter, 3), dtype=np.int64), T=np.zeros((max_iter, 16)),
                    passed=np.zeros(max_iter, dtype=np.int32), counts=np.zeros(max_iter, dtype=np.uint64),
                    err2=np.zeros(max_iter))
        tr = _RegTrace(*[_p(arrs[k]) for k in ("triples", "T", "passed", "counts", "err2")])
    ret = lib().orc_registration_ransac(_p(src), C.c_size_t(len(src)), _p(dst), C.c_size_t(len(dst)), _p(cs), _p(cd),
                                        C.c_size_t(len(cs)), C.c_double(thr), C.c_int(max_iter), C.c_double(edge_thr),
                                        C.c_double(confidence), C.c_uint64(seed), _p(T), C.byref(st),
                                        C.byref(tr) if tr is not None else None)
    return RegResult(ret, T.reshape(4, 4), st.fitness, st.inlier_rmse, int(st.validations), int(st.iterations),
                     int(st.best_index), int(st.est_k), arrs)


def information_matrix(src, dst, max_dist, T):
    """GetInformationMatrixFromPointClouds -> 6 x 6"""
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    Tm = _f64(T).reshape(16).copy()
    info = np.zeros(36)
    lib().orc_information_matrix(_p(src), C.c_size_t(len(src)), _p(dst), C.c_size_t(len(dst)), C.c_double(max_dist),
                                 _p(Tm), _p(info))
    return info.reshape(6, 6)


def global_registration(src, dst, feat_src, feat_dst, voxel_size, max_iter=100000, edge_thr=0.9, confidence=0.999, seed=0):
    """ReconstructionPipeline::GlobalRegistration, Ransac method (src/pipeline.cpp:790-828) -> (success, T 4x4, info 6x6,
    number of mutual matches); raises ValueError where the solver throws (fewer than 3 points)."""
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    fs = _f64(feat_src)
    fd = _f64(feat_dst)
    if len(fs) != len(src) or len(fd) != len(dst) or fs.shape[1] != fd.shape[1]:
        raise ValueError("one descriptor per point, equal widths")
    T = np.zeros(16)
    info = np.zeros(36)
    nm = C.c_uint64(0)
    f = lib().orc_global_registration
    f.restype = C.c_int
    r = f(_p(src), C.c_size_t(len(src)), _p(dst), C.c_size_t(len(dst)), _p(fs), _p(fd), C.c_int(fs.shape[1]),
          C.c_double(voxel_size), C.c_int(max_iter), C.c_double(edge_thr), C.c_double(confidence), C.c_uint64(seed), _p(T),
          _p(info), C.byref(nm))
    if r < 0:
        raise ValueError("The number of points pair is less than 3.")
    return bool(r), T.reshape(4, 4), info.reshape(6, 6), int(nm.value)


def registration_icp(src, dst, max_dist, T_init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    """-> (T 4x4, fitness, inlier_rmse, iterations, correspondences (ns,) int64 with -1 = none)"""
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    Ti = _f64(T_init if T_init is not None else np.eye(4)).reshape(16).copy()
    T = np.zeros(16)
    fit, rm = C.c_double(0), C.c_double(0)
    nc = C.c_uint64(0)
    corr = np.zeros(max(len(src), 1), dtype=np.int64)
    lib().orc_registration_icp.restype = C.c_int
    it = lib().orc_registration_icp(_p(src), C.c_size_t(len(src)), _p(dst), C.c_size_t(len(dst)), C.c_double(max_dist),
                                    _p(Ti), C.c_int(max_iter), C.c_double(rel_fitness), C.c_double(rel_rmse), _p(T),
                                    C.byref(fit), C.byref(rm), C.byref(nc), _p(corr))
    return T.reshape(4, 4), float(fit.value), float(rm.value), int(it), corr[: len(src)]


def detect_boundary_points(xyz, normals=None, search=2, radius=0.01, max_nn=30, angle_threshold=90.0):
    """DetectBoundaryPoints (src/boundary_detection.cpp:68-113); search 1 = Radius, 2 = Hybrid -> ascending indices"""
    xyz = _f64(xyz).reshape(-1, 3)
    nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
    out = np.zeros(max(len(xyz), 1), dtype=np.int64)
    lib().orc_detect_boundary_points.restype = C.c_size_t
    k = lib().orc_detect_boundary_points(_p(xyz), _p(nrm), C.c_size_t(len(xyz)), C.c_int(search), C.c_double(radius),
                                         C.c_int(max_nn), C.c_double(angle_threshold), _p(out))
    return out[:k].copy()


def normals_from_map(xyz, w, h, k=5, view_point=(0.0, 0.0, 0.0)):
    """EstimateNormalsFromMap (src/normal_estimation.cpp:64-207): xyz (h*w, 3) point map -> (h*w, 3) normals."""
    xyz = _f64(xyz).reshape(-1, 3)
    assert len(xyz) == w * h
    vp = _f64(view_point, (3,))
    out = np.zeros((w * h, 3))
    lib().orc_normals_from_map(_p(xyz), C.c_uint(w), C.c_uint(h), C.c_uint(k), _p(vp), _p(out))
    return out


def j3x3_smallest_eigvec(A):
    A = _f64(A, (9,))
    n = np.zeros(3)
    lib().orc_j3x3_smallest_eigvec(_p(A), _p(n))
    return n


def match_mutual_nn(feat_src, feat_dst):
    """feat_*: (N, dim) row-major == Eigen dim x N column-major (correspondence_matching.h:39-41)."""
    fs = _f64(feat_src)
    fd = _f64(feat_dst)
    ns, dim = fs.shape
    nd = fd.shape[0]
    o0 = np.zeros(max(ns, 1), dtype=np.int64)
    o1 = np.zeros(max(ns, 1), dtype=np.int64)
    k = lib().orc_match_mutual_nn(_p(fs), C.c_size_t(ns), _p(fd), C.c_size_t(nd), C.c_int(dim), _p(o0), _p(o1))
    return o0[:k].copy(), o1[:k].copy()


def reg_validate_kdtree(src, dst, T, thr, workers=-1, tree=None):
    """GetRegistrationResultAndCorrespondences with a kd-tree, as the reference runs it (Open3D's KDTreeFlann::SearchHybrid(p, thr,
    1) under transform_estimation.cpp:154-161): scipy's cKDTree stands in for nanoflann.  Same (count, err2) as orc_reg_validate's
    brute force (tests/test_oracle_primitives.py holds them against each other) at O(n log n) -- the CPU baseline of the
    registration's validation in tools/cpu_baselines.py, NOT part of the parity chain (scipy's tie order is its own)."""
    from scipy.spatial import cKDTree
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    T = _f64(T).reshape(4, 4)
    if tree is None:
        tree = cKDTree(dst)
    p = src @ T[:3, :3].T + T[:3, 3]
    d, _ = tree.query(p, k=1, distance_upper_bound=thr, workers=workers)
    ok = np.isfinite(d) & (d * d < thr * thr)
    return int(ok.sum()), float((d[ok] ** 2).sum())
