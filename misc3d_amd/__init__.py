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
get_verbosity_level = _ext.get_verbosity_level
device_count = _ext.device_count
Error, Warning, Info, Debug = (VerbosityLevel.Error, VerbosityLevel.Warning, VerbosityLevel.Info,
                               VerbosityLevel.Debug)

__all__ = ["common", "registration", "segmentation", "VerbosityLevel", "set_verbosity_level",
           "get_verbosity_level", "device_count"]
