"""misc3d_amd -- MI355X (gfx950) implementation of the Misc3D RANSAC hot path.

Drop-in for the reference's python API on this path (module layout of python/py_misc3d.cpp:25-62):

    import misc3d_amd as m3d
    w, index = m3d.common.fit_plane(pcd, 0.01, 100)
    w, index = m3d.common.fit_sphere(pcd, 0.01, 100)
    w, index = m3d.common.fit_cylinder(pcd, 0.01, 100)
    results  = m3d.segmentation.segment_plane_iterative(pcd, 0.01, 100, 0.1)
    idx      = m3d.registration.match_correspondence(fpfh_src, fpfh_dst)
    T        = m3d.registration.compute_transformation_ransac(src, dst, idx, 0.03, 100000)
    T        = m3d.registration.compute_transformation_least_square(src, dst)
    T, info  = m3d.registration_icp(src, dst, 0.02, T)      # the Open3D call the reference's examples chain next
    index    = m3d.features.detect_boundary_points(plane, ("hybrid", 0.02, 30))
    normals  = m3d.common.estimate_normals(pcd, (848, 480), 3)
    ok, T, info = m3d.reconstruction.global_registration(frag_s, frag_t, fpfh_s, fpfh_t, voxel_size)   # pipeline.cpp:790-828
    results  = m3d.reconstruction.register_fragment_pairs(fragments, fpfhs, voxel_size=voxel_size)     # pipeline.cpp:428-439

Layout (only what the path needs):
  csrc/      HIP kernels, host driver, C ABI            -> lib/libmisc3d_amd.so
  host/      pybind11 module over include/misc3d/**      -> _py_misc3d*.so  (this API)
  capi.py    ctypes binding of include/misc3d_amd.h      (tests, bench, distributed driver)
  distributed.py  hypothesis sharding over torch.distributed (RCCL)
  synth.py   seeded synthetic clouds of the BASELINE.json configurations

There is no CPU fallback: the native libraries must be built (python __graft_entry__.py) and every
compute call needs a HIP device.
"""
__version__ = "0.1.0"

try:
    from . import _py_misc3d as _ext
except ImportError as e:  # fail loudly: no pure-python stand-in exists
    raise ImportError(
        "misc3d_amd: the native host module misc3d_amd/_py_misc3d*.so (or lib/libmisc3d_amd.so) is missing or "
        f"failed to load ({e}). Build with `python __graft_entry__.py`.") from e

common = _ext.common
registration = _ext.registration
segmentation = _ext.segmentation
VerbosityLevel = _ext.VerbosityLevel
set_verbosity_level = _ext.set_verbosity_level
release_host_scratch = _ext.release_host_scratch   # frees the page-locked blocks kept between calls ("Page-locked memory")
get_verbosity_level = _ext.get_verbosity_level
device_count = _ext.device_count
Error, Warning, Info, Debug = (VerbosityLevel.Error, VerbosityLevel.Warning, VerbosityLevel.Info,
                               VerbosityLevel.Debug)



def registration_icp(source, target, max_correspondence_distance, init=None, max_iteration=30,
                     relative_fitness=1e-6, relative_rmse=1e-6, device=0):
    """Point-to-point ICP = open3d.pipelines.registration.registration_icp(source, target,
    max_correspondence_distance, init, TransformationEstimationPointToPoint(), ICPConvergenceCriteria(...)),
    which the reference's examples run on the pose of compute_transformation_ransac
    (examples/cpp/transform_estimation.cpp:82-86).  source / target: (N, 3) arrays or objects with `.points`.
    Returns (4x4 pose, dict(fitness, inlier_rmse, correspondences, iterations, converged))."""
    import numpy as _np

    from . import capi as _capi
    pts = [_np.asarray(getattr(c, "points", c), dtype=_np.float64).reshape(-1, 3) for c in (source, target)]
    return _capi.registration_icp(pts[0], pts[1], max_correspondence_distance, init, max_iteration,
                                  relative_fitness, relative_rmse, device)


class _Features:
    """misc3d.features (python/py_features.cpp): detect_boundary_points"""

    @staticmethod
    def detect_boundary_points(pc, param=("hybrid", 0.01, 30), angle_threshold=90.0, device=0):
        """DetectBoundaryPoints (src/boundary_detection.cpp:68-113).  pc: (N, 3) array, (points, normals) tuple or
        an object with .points / .normals.  param: an open3d KDTreeSearchParamHybrid / KDTreeSearchParamRadius /
        KDTreeSearchParamKNN (anything with .radius and optionally .max_nn, or with .knn), or a tuple
        ("hybrid", radius, max_nn) / ("radius", radius) / ("knn", k).
        Returns the list of boundary point indices (ascending)."""
        import numpy as _np

        from . import capi as _capi
        if isinstance(pc, tuple) and len(pc) == 2:
            pts, nrm = pc
        else:
            pts, nrm = getattr(pc, "points", pc), getattr(pc, "normals", None)
        pts = _np.asarray(pts, dtype=_np.float64).reshape(-1, 3)
        if nrm is not None:
            nrm = _np.asarray(nrm, dtype=_np.float64).reshape(-1, 3)
            if len(nrm) != len(pts) or len(nrm) == 0:
                nrm = None
        if isinstance(param, tuple):
            kind = str(param[0]).lower()
            if kind == "knn":
                radius, max_nn = 0.0, int(param[1])
            else:
                radius = float(param[1])
                max_nn = int(param[2]) if len(param) > 2 else 0
        elif hasattr(param, "knn") and not hasattr(param, "radius"):
            kind, radius, max_nn = "knn", 0.0, int(param.knn)
        else:
            radius = float(param.radius)
            max_nn = int(getattr(param, "max_nn", 0))
            kind = "hybrid" if hasattr(param, "max_nn") else "radius"
        if kind not in ("hybrid", "radius", "knn"):
            raise RuntimeError("[Misc3D Error] param: KDTreeSearchParamHybrid / KDTreeSearchParamRadius / KDTreeSearchParamKNN")
        search = {"hybrid": _capi.SEARCH_HYBRID, "radius": _capi.SEARCH_RADIUS, "knn": _capi.SEARCH_KNN}[kind]
        try:
            idx = _capi.detect_boundary_points(pts, nrm, search, radius, max_nn, angle_threshold, device)
        except _capi.M3DError as e:
            raise RuntimeError(str(e)) from e
        return [int(i) for i in idx]


features = _Features()


def _xyz(c):
    import numpy as _np
    return _np.ascontiguousarray(_np.asarray(getattr(c, "points", c), dtype=_np.float64).reshape(-1, 3))


def _feat(f, n):
    """open3d Feature (.data: dim x N) or an ndarray, (N, dim) or (dim, N) -> (N, dim) C-contiguous"""
    import numpy as _np
    is_feature = not isinstance(f, _np.ndarray) and hasattr(f, "data")      # (an ndarray has a .data of its own: its buffer)
    a = _np.asarray(f.data if is_feature else f, dtype=_np.float64)
    if is_feature or (a.ndim == 2 and a.shape[0] != n and a.shape[1] == n):
        a = a.T          # Eigen dim x N column-major == (N, dim) row-major: a transposed VIEW of the same memory
    return _np.ascontiguousarray(a)


class _Reconstruction:
    """misc3d.reconstruction, the loop-closure half of ReconstructionPipeline (src/pipeline.cpp): GlobalRegistration
    (:790-828) and the loop over fragment pairs that calls it (:428-439).  Calls release the GIL: Python threads that call
    global_registration / fit_* / match_correspondence side by side run side by side on the device (lanes)."""

    @staticmethod
    def global_registration(source, target, feature_source, feature_target, voxel_size, max_iter=100000,
                            edge_length_threshold=0.9, confidence=0.999, *, seed=None, device=0):
        """ReconstructionPipeline::GlobalRegistration with the Ransac method: match_correspondence ->
        compute_transformation_ransac(1.4 voxel_size) -> information matrix -> accepted unless info[5, 5] / min(Ns, Nt) < 0.3.
        Returns (success, 4x4 pose, 6x6 information)."""
        from . import capi as _capi
        src, dst = _xyz(source), _xyz(target)
        try:
            return _capi.global_registration(src, dst, _feat(feature_source, len(src)), _feat(feature_target, len(dst)),
                                             voxel_size, max_iter, edge_length_threshold, confidence, seed, device)
        except _capi.M3DError as e:
            raise RuntimeError(str(e)) from e

    @staticmethod
    def register_fragment_pairs(fragments, features, pairs=None, voxel_size=0.01, max_iter=100000,
                                edge_length_threshold=0.9, confidence=0.999, *, seeds=None, devices=(0,), inflight=0):
        """BuildPoseGraphForScene's loop closures: every (s, t) of `pairs` through global_registration, dealt to `devices`,
        `inflight` pairs at a time per device.  Default pairs: all s < t with t > s + 1 -- the reference sends ADJACENT fragments
        (t == s + 1) to the multi-scale ICP odometry seeded from the fragment pose graph, never to GlobalRegistration
        (src/pipeline.cpp:752-764); pass `pairs` explicitly to register those here as well.
        Returns [(s, t, success, pose, information), ...]."""
        from . import capi as _capi
        pts = [_xyz(f) for f in fragments]
        fts = [_feat(f, len(p)) for f, p in zip(features, pts)]
        if pairs is None:
            pairs = [(s, t) for s in range(len(pts)) for t in range(s + 2, len(pts))]
        try:   # (every fragment is uploaded once per device and stays resident for the call)
            res = _capi.register_fragment_pairs(pts, fts, pairs, voxel_size, max_iter, edge_length_threshold, confidence,
                                                seeds, devices, inflight)
        except _capi.M3DError as e:
            raise RuntimeError(str(e)) from e
        return [(s, t) + r for (s, t), r in zip(pairs, res)]


reconstruction = _Reconstruction()


def registration_session(*args, **kwargs):
    """capi.RegSession: compute_transformation_ransac cut into begin_chunk / validate / replay, the unit
    misc3d_amd.distributed.registration_ransac_sharded shards over ranks."""
    from . import capi as _capi
    return _capi.RegSession(*args, **kwargs)

__all__ = ["common", "registration", "segmentation", "features", "reconstruction", "registration_icp", "registration_session", "VerbosityLevel", "set_verbosity_level",
           "get_verbosity_level", "device_count"]
