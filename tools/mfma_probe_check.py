"""The MFMA screen's values on hand-made tiles against exact arithmetic (run on the GPU box).

T -+ S~, S~ = alpha . x~ + D, is evaluated with python Fractions on the fp32 offsets the kernel reports (so the comparison
isolates the matrix pipe and the operand pieces); prints the worst |u_i - (T -+ S~)| in units of the record's bound E_p, the
fraction of (point, hypothesis) products inside the band, and wrong verdicts among the decided ones (must be 0).
"""
import sys
from fractions import Fraction

import numpy as np

sys.path.insert(0, ".")
from misc3d_amd import capi  # noqa: E402


def tile(rng, extent, centre, flat=None):
    pts = centre + (rng.random((512, 3)) - 0.5) * 2 * np.asarray(extent)
    if flat is not None:   # points near a plane n . x = d: the inlier tiles of C2
        n, d, sigma = flat
        n = n / np.linalg.norm(n)
        pts = pts - np.outer(pts @ n - d, n) + np.outer(rng.normal(0, sigma, 512), n)
    lo, hi = pts.min(0), pts.max(0)
    c = 0.5 * lo + 0.5 * hi
    half = np.maximum(hi - c, c - lo) * (1 + 1e-12) + 1e-300
    return pts, np.concatenate([c, half])


def run(name, pts, box, recs, max_abs):
    u, h, sg, ep, off = capi.mfma_probe(pts, box, max_abs, recs)
    worst = 0.0
    inband = decided_wrong = n = 0
    for k, (a, b, c, d, T) in enumerate(recs):
        if not np.isfinite(h[k]):
            print(f"  record {k}: not screened")
            continue
        D = float(Fraction(a) * Fraction(box[0]) + Fraction(b) * Fraction(box[1]) + Fraction(c) * Fraction(box[2]) + Fraction(d))
        # (the kernel's D: the fp64 evaluation of the same; the difference is part of E_in)
        for i in range(0, 512, 5):
            S = Fraction(a) * Fraction(float(off[i, 0])) + Fraction(b) * Fraction(float(off[i, 1])) + Fraction(c) * Fraction(float(off[i, 2])) + Fraction(D)
            e1 = abs(u[k, i, 0] - float(Fraction(T) - S))
            e2 = abs(u[k, i, 1] - float(Fraction(T) + S))
            worst = max(worst, e1 / ep[k], e2 / ep[k])
            t = u[k, i, 0] * u[k, i, 1]
            if abs(t) < h[k]:
                inband += 1
            elif (t > 0) != (abs(S) < T):
                decided_wrong += 1
            n += 1
    T = recs[0][4]
    print(f"{name}: worst |u - (T -+ S~)| = {worst:.4f} E_p   in-band {inband}/{n} = {inband / max(n, 1):.4%}   wrong verdicts {decided_wrong}"
          f"   median band in distance h / 2T = {np.nanmedian(h) / (2 * T):.3e} (T = {T})")


def main():
    rng = np.random.default_rng(5)
    T = 0.01
    n_true = np.array([0.2, -0.3, 0.93])
    n_true /= np.linalg.norm(n_true)
    for name, extent, flat in (("inlier tile 0.06", (0.06, 0.06, 0.06), (n_true, 0.5, 0.003)),
                               ("mixed tile 0.1", (0.1, 0.1, 0.1), None),
                               ("fat tile 0.3", (0.3, 0.3, 0.3), None)):
        centre = np.array([0.4, -0.2, 0.5])
        centre = centre - n_true * (centre @ n_true - 0.5)   # on the plane
        pts, box = tile(rng, extent, centre, flat)
        recs = []
        for _ in range(96):
            n = n_true + rng.normal(0, 0.02, 3)
            n /= np.linalg.norm(n)
            d = -(n @ centre) + rng.normal(0, 0.004)
            recs.append((n[0], n[1], n[2], d, T))
        run(name, pts, box, recs, 3.0)


if __name__ == "__main__":
    main()
