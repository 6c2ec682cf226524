"""The MFMA screen of the plane scoring (score_mfma_k, m3d_config.score_mfma) against exact arithmetic.

m3d_bench_mfma_probe runs ONE tile of 512 points and a set of plane records through the production kernel's own operand
builders and the matrix pipe and returns the pipe's two values per (hypothesis, point), the band on their product and the
bound E_p the record claims.  The values are held against T -+ S~ evaluated with exact rational arithmetic on the fp32
offsets the kernel reports: the error must stay below E_p with room to spare, and no DECIDED product may have the wrong sign.
The end-to-end proof -- identical counts for every hypothesis -- is tests/test_gpu_parity.py's `mfma` scoring path.
"""
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_experimental_build(capi):
    if not capi.experimental():
        pytest.skip("score_mfma_k is compiled with -DM3D_EXPERIMENTAL only (misc3d_amd/csrc/m3d_kernels.hpp)")


def _tile(rng, extent, centre, flat=None):
    pts = centre + (rng.random((512, 3)) - 0.5) * 2 * np.asarray(extent)
    if flat is not None:
        n, d, sigma = flat
        n = n / np.linalg.norm(n)
        pts = pts - np.outer(pts @ n - d, n) + np.outer(rng.normal(0, sigma, 512), n)
    lo, hi = pts.min(0), pts.max(0)
    c = 0.5 * lo + 0.5 * hi
    return pts, np.concatenate([c, np.maximum(hi - c, c - lo) * (1 + 1e-12) + 1e-300])


@pytest.mark.parametrize("extent,flat,scale,offset", [
    ((0.06, 0.06, 0.06), True, 1.0, 0.0),      # a tile of the winning plane on the C2 cloud
    ((0.3, 0.3, 0.3), False, 1.0, 0.0),        # a fat tile of outliers
    ((0.06, 0.02, 0.11), True, 1e-3, 0.0),     # millimetre scene
    ((0.1, 0.1, 0.1), False, 1e3, 0.0),
    ((0.06, 0.06, 0.06), True, 1.0, 4.0e4),    # far from the origin: the offsets are what the pipe sees
])
def test_pipe_values_within_the_bound(capi, extent, flat, scale, offset):
    rng = np.random.default_rng(17)
    T = 0.01 * scale
    n_true = np.array([0.2, -0.3, 0.93])
    n_true /= np.linalg.norm(n_true)
    centre = np.array([0.4, -0.2, 0.5]) * scale + offset
    centre = centre - n_true * (centre @ n_true - (0.5 * scale + offset * n_true.sum()))
    pts, box = _tile(rng, np.asarray(extent) * scale, centre,
                     (n_true, centre @ n_true, 0.003 * scale) if flat else None)
    recs = []
    for i in range(70):   # (more than one batch of 64: both sub-batches, a ragged tail)
        n = n_true + rng.normal(0, 0.02 if i % 3 else 0.5, 3)
        n /= np.linalg.norm(n)
        recs.append((n[0], n[1], n[2], -(n @ centre) + rng.normal(0, 0.004 * scale), T))
    max_abs = float(np.abs(pts).max())
    u, h, sg, ep, off = capi.mfma_probe(pts, box, max_abs, recs)
    assert np.isfinite(h).all(), "every record of this scene is screenable"
    worst = 0.0
    decided = wrong = 0
    for k, (a, b, c, d, Tk) in enumerate(recs):
        D = float(Fraction(a) * Fraction(box[0]) + Fraction(b) * Fraction(box[1]) + Fraction(c) * Fraction(box[2]) + Fraction(d))
        for i in range(0, 512, 3):
            S = (Fraction(a) * Fraction(float(off[i, 0])) + Fraction(b) * Fraction(float(off[i, 1])) +
                 Fraction(c) * Fraction(float(off[i, 2])) + Fraction(D))
            worst = max(worst, abs(u[k, i, 0] - float(Fraction(Tk) - S)) / ep[k], abs(u[k, i, 1] - float(Fraction(Tk) + S)) / ep[k])
            t = u[k, i, 0] * u[k, i, 1]
            if abs(t) >= h[k]:
                decided += 1
                wrong += (t > 0) != (abs(S) < Tk)
    assert worst < 0.5, f"the pipe's values are within {worst:.3f} E_p of exact arithmetic; the bound must keep a factor 2"
    assert wrong == 0
    assert decided > 0.98 * 70 * 171


def test_unscreenable_records_say_so(capi):
    rng = np.random.default_rng(3)
    pts, box = _tile(rng, (0.05, 0.05, 0.05), np.zeros(3))
    recs = [(0.0, 0.0, 1.0, 0.0, 0.01),            # fine
            (0.0, 0.0, 1.0, 0.0, 1e-12),           # threshold below the rounding bound
            (0.0, 0.0, 1.0, 1e6, 0.01),            # a plane a million tile sizes away: its constant leaves fp16's range
            (40.0, 0.0, 0.0, 0.0, 0.01),           # a normal that is not a unit vector
            (float("nan"), 0.0, 1.0, 0.0, 0.01)]
    u, h, sg, ep, off = capi.mfma_probe(pts, box, 1.0, recs)
    assert np.isfinite(h[0]) and not np.isfinite(h[1:]).any()
    assert np.isfinite(u[1:]).all(), "an unscreened record's column stays finite (it is zero)"
