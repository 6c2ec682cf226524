"""Seeded synthetic clouds for the BASELINE.json configurations.

Pure numpy data generation used by ``bench.py`` and the tests; no arithmetic of the hot path.
All generators are deterministic functions of (seed, n) through numpy's PCG64.
"""
from __future__ import annotations

import numpy as np


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def _plane_points(rng, n, normal, d, sigma, extent):
    """n points on the plane normal.x + d = 0 (|normal| = 1) inside a box of half-size extent."""
    normal = _unit(normal)
    a = _unit(np.cross(normal, [1.0, 0.0, 0.0] if abs(normal[0]) < 0.9 else [0.0, 1.0, 0.0]))
    b = np.cross(normal, a)
    uv = rng.uniform(-extent, extent, size=(n, 2))
    p0 = -d * normal
    return p0 + uv[:, :1] * a + uv[:, 1:] * b + rng.normal(0.0, sigma, size=(n, 1)) * normal


def plane_cloud_c1(n=50_000, seed=1):
    """C1 plumbing cloud: 60 % plane z = 1 (sigma 2 mm), 40 % uniform in [-1,1]^3."""
    rng = np.random.default_rng(seed)
    n_in = int(0.6 * n)
    xy = rng.uniform(-1, 1, size=(n_in, 2))
    z = 1.0 + rng.normal(0, 0.002, size=(n_in, 1))
    pts = np.concatenate([np.hstack([xy, z]), rng.uniform(-1, 1, size=(n - n_in, 3))])
    return np.ascontiguousarray(pts[rng.permutation(n)])


def plane_cloud_c2(n=1_000_000, seed=2):
    """C2: 50 % plane A, 20 % plane B (sigma 3 mm), 30 % uniform in [-2,2]^3."""
    rng = np.random.default_rng(seed)
    na, nb = int(0.5 * n), int(0.2 * n)
    pa = _plane_points(rng, na, [0.2, -0.3, 0.93], -0.5, 0.003, 2.0)
    pb = _plane_points(rng, nb, [1.0, 0.1, 0.05], 0.8, 0.003, 2.0)
    po = rng.uniform(-2, 2, size=(n - na - nb, 3))
    pts = np.concatenate([pa, pb, po])
    return np.ascontiguousarray(pts[rng.permutation(n)])


def _random_unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def cylinder_cloud_c3(n=1_000_000, seed=3):
    """C3 cylinder: axis point (0.1,0.2,0.3), dir (1,2,3)/|.|, r = 0.25, length 2, radial sigma 2 mm,
    normals = radial direction perturbed by ~1 degree; 50 % inliers + 50 % uniform outliers with
    random unit normals.  Returns (points, normals)."""
    rng = np.random.default_rng(seed)
    n_in = n // 2
    axis = _unit([1.0, 2.0, 3.0])
    p0 = np.array([0.1, 0.2, 0.3])
    a = _unit(np.cross(axis, [1.0, 0.0, 0.0]))
    b = np.cross(axis, a)
    t = rng.uniform(-1, 1, size=(n_in, 1))
    phi = rng.uniform(0, 2 * np.pi, size=(n_in, 1))
    radial = np.cos(phi) * a + np.sin(phi) * b
    r = 0.25 + rng.normal(0, 0.002, size=(n_in, 1))
    pts_in = p0 + t * axis + r * radial
    nrm_in = radial + rng.normal(0, np.deg2rad(1.0), size=(n_in, 3))
    nrm_in /= np.linalg.norm(nrm_in, axis=1, keepdims=True)
    pts_out = rng.uniform(-1.5, 1.5, size=(n - n_in, 3)) + p0
    nrm_out = _random_unit(rng, n - n_in)
    perm = rng.permutation(n)
    pts = np.concatenate([pts_in, pts_out])[perm]
    nrm = np.concatenate([nrm_in, nrm_out])[perm]
    return np.ascontiguousarray(pts), np.ascontiguousarray(nrm)


def sphere_cloud_c3(n=1_000_000, seed=4):
    """C3 sphere: centre (0.3,-0.2,1.0), r = 0.5, sigma 2 mm, 50 % inliers + 50 % uniform outliers."""
    rng = np.random.default_rng(seed)
    n_in = n // 2
    c = np.array([0.3, -0.2, 1.0])
    u = _random_unit(rng, n_in)
    r = 0.5 + rng.normal(0, 0.002, size=(n_in, 1))
    pts = np.concatenate([c + r * u, c + rng.uniform(-1.5, 1.5, size=(n - n_in, 3))])
    return np.ascontiguousarray(pts[rng.permutation(n)])


def room_cloud_c5(n=10_000_000, seed=6):
    """C5 room: floor 25 %, ceiling 15 %, walls 15/15/10/10 %, clutter 10 %; sigma 3 mm."""
    rng = np.random.default_rng(seed)
    fr = [0.25, 0.15, 0.15, 0.15, 0.10, 0.10]
    planes = [([0, 0, 1], 0.0), ([0, 0, 1], -2.5), ([1, 0, 0], 3.0), ([1, 0, 0], -3.0), ([0, 1, 0], 2.0),
              ([0, 1, 0], -2.0)]
    parts = []
    used = 0
    for f, (nrm, d) in zip(fr, planes):
        k = int(f * n)
        parts.append(_plane_points(rng, k, nrm, d, 0.003, 3.0))
        used += k
    parts.append(rng.uniform(-3, 3, size=(n - used, 3)))
    pts = np.concatenate(parts)
    return np.ascontiguousarray(pts[rng.permutation(n)])


def rigid_transform(angle_deg=40.0, axis=(1, 1, 1), t=(0.3, -0.1, 0.2)):
    axis = _unit(axis)
    th = np.deg2rad(angle_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def registration_pair_c4(n=200_000, seed=5, dim=33, true_fraction=0.3, sigma=0.002):
    """C4: src = 6 random planar/spherical patches in [-1,1]^3; dst = R src + t + noise, permuted;
    synthetic non-negative `dim`-D descriptors: dst descriptor = src descriptor + small noise for a
    `true_fraction` of the points, random otherwise.  Returns dict(src, dst, feat_src, feat_dst, T, perm)."""
    rng = np.random.default_rng(seed)
    per = n // 6
    parts = []
    for k in range(6):
        m = per if k < 5 else n - 5 * per
        centre = rng.uniform(-0.7, 0.7, size=3)
        if k % 2 == 0:
            parts.append(_plane_points(rng, m, _random_unit(rng, 1)[0], 0.0, 0.0, 0.3) + centre)
        else:
            parts.append(centre + 0.25 * _random_unit(rng, m))
    src = np.concatenate(parts)
    src = src[rng.permutation(n)]
    T = rigid_transform()
    dst_full = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, sigma, size=(n, 3))
    perm = rng.permutation(n)          # dst[j] = dst_full[perm[j]]
    dst = dst_full[perm]
    feat_src = rng.uniform(0, 1, size=(n, dim))
    feat_dst_full = rng.uniform(0, 1, size=(n, dim))
    good = rng.random(n) < true_fraction
    feat_dst_full[good] = np.abs(feat_src[good] + rng.normal(0, 0.01, size=(int(good.sum()), dim)))
    feat_dst = feat_dst_full[perm]
    return dict(src=np.ascontiguousarray(src), dst=np.ascontiguousarray(dst),
                feat_src=np.ascontiguousarray(feat_src), feat_dst=np.ascontiguousarray(feat_dst), T=T, perm=perm,
                good=good)
