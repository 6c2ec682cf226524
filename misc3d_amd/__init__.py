"""misc3d_amd -- MI355X (gfx950) implementation of the Misc3D RANSAC hot path.

Layout (only what the path needs):
  csrc/      HIP kernels, host driver, C ABI  -> lib/libmisc3d_amd.so
  capi.py    ctypes binding of include/misc3d_amd.h
  common / registration / segmentation: the reference's python API (python/py_*.cpp) on top of it
  synth.py   seeded synthetic clouds of the BASELINE.json configurations
"""
__version__ = "0.1.0"
