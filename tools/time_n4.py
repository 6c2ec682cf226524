import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi
rng = np.random.default_rng(3)
nb = 500_000
uv = rng.uniform(0, 3.0, (nb, 2))
pp = np.c_[uv[:, 0], uv[:, 1], 0.2 * uv[:, 0] + rng.normal(0, 1e-3, nb)]
for _ in range(5):
    t0 = time.perf_counter()
    b = capi.detect_boundary_points(pp, None, 2, 0.02, 30, 90.0)
    print("boundary %.1f ms  n=%d" % ((time.perf_counter() - t0) * 1e3, len(b)))
