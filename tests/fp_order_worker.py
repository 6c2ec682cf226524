"""Worker of tests/test_fp_orders.py, run with M3D_FP_ORDER=<k> in the environment: the product's HOST MinimalFit
(m3d_minimal_fit: the same m3d_fp.hpp code the kernels run, compiled for the host) against the oracle built for the
same association, bit for bit, on seeded random samples.  Needs no GPU.  Prints a digest of the models so that the
parent can tell the associations apart."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from misc3d_amd import capi  # noqa: E402

order = int(os.environ.get("M3D_FP_ORDER", "0"))
assert capi.FP_ORDER == order and oracle.FP_ORDER == order
assert int(oracle.lib().orc_fp_order()) == order, "oracle library built for another association"
assert f"order{order}" in capi.LIB_PATH or order == 0
rng = np.random.default_rng(5)
digest = {}
for kind, m, fit in ((0, 3, oracle.plane_minimal_fit), (1, 4, oracle.sphere_minimal_fit), (2, 2, None)):
    h = hashlib.sha256()
    bad = 0
    for t in range(400):
        scale = 10.0 ** rng.integers(-3, 4)
        pts = rng.normal(size=(m, 3)) * scale + rng.normal(size=3) * scale
        nrm = rng.normal(size=(m, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        model = np.zeros(8)
        ok = __import__("ctypes").c_uint8(0)
        import ctypes as C
        rc = capi.lib().m3d_minimal_fit(kind, pts.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p),
                                        model.ctypes.data_as(C.c_void_p), C.cast(C.byref(ok), C.c_void_p))
        assert rc == 1
        if kind == 2:
            o_ok, o_model = oracle.cylinder_minimal_fit(pts, nrm)
        else:
            o_ok, o_model = fit(pts)
        npar = 7 if kind == 2 else 4
        if bool(ok.value) != bool(o_ok):
            bad += 1
        elif o_ok:
            a = np.nan_to_num(model[:npar], nan=-7.0).view(np.uint64)
            b = np.nan_to_num(np.asarray(o_model, dtype=np.float64)[:npar], nan=-7.0).view(np.uint64)
            if not np.array_equal(a, b):
                bad += 1
            h.update(a.tobytes())
    assert bad == 0, (kind, bad)
    digest[kind] = h.hexdigest()[:16]
# the registration checkers (m3d_reg_fp.hpp reg_checkers <-> orc_reg_checkers): Eigen Vector3d norms behind an
# edge-length ratio test.  Triples whose edge ratio sits ON the threshold (dt = ds / thr up to a few ulps), so that the
# last bit of a norm decides: product == oracle under this association, and the decisions depend on the association
import ctypes as C  # noqa: E402
h = hashlib.sha256()
bad = 0
thr = 0.9
for t in range(6000):
    ps = rng.normal(size=(3, 3))
    pd = ps.copy()
    # stretch ONE edge of the target triangle so that |e_dst| * thr ~= |e_src| within an ulp or two
    e = ps[1] - ps[0]
    pd[1] = pd[0] + e / thr * (1.0 + rng.integers(-3, 4) * 2.0 ** -52)
    T = np.eye(4)
    g = capi.lib().m3d_bench_reg_checkers(ps.ctypes.data_as(C.c_void_p), pd.ctypes.data_as(C.c_void_p),
                                          T.ctypes.data_as(C.c_void_p), thr, 1e9)
    o = oracle.reg_checkers(ps, pd, T, thr, 1e9)
    assert g in (0, 1)
    bad += int(bool(g) != bool(o))
    h.update(bytes([g]))
assert bad == 0, ("reg_checkers", bad)
digest[3] = h.hexdigest()[:16]
print("FP_ORDER_OK", order, digest[0], digest[1], digest[2], digest[3])
