// m3d_bound_fp.hpp -- the arithmetic of the planes' histogram bound (m3d_bound.hip), shared by the device kernels and the
// host-side check tests/cpp/test_plane_bound.cpp (no GPU, no library: the same expressions compiled by g++).
//
// A tile's frame record (kFrameStride doubles, written by tile_frames_k):
//   [0..2] c   [3..5] e   [6..8] u   [9..11] v   [12] U >= |u . (p - c)|   [13] V >= |v . (p - c)|   [14] R >= |res_p|_inf
//   [15] wlo   [16] invd (bins per unit of w)   [17] W >= |w_p|   [18] finite points   [19] valid
// with w_p = e . (p - c), u_p, v_p the numbers tile_frames_k computed and res_p = (p - c) - w_p e - u_p u - v_p v, and its
// cumulative histogram cum[k] = points with bound_bin(w_p) < k, k = 0 .. kBoundBins + 2.
//
// For a plane S(p) = a x + b y + c z + d with cut-off T (exact test |s64| < T, s64 the fp64 evaluation: within 1e-15 M_g of S):
//   S(p) = S(c) + (n . e) w_p + (n . u) u_p + (n . v) v_p + n . res_p,
// so |S(p)| < T  =>  |S(c) + g w_p| < T + a,  g = n . e,  a >= |n . u| U + |n . v| V + |n|_1 R + roundings:
// the points that can be inliers have w_p between L = (-T - a - S(c)) / g and H = (T + a - S(c)) / g (g > 0 after a sign flip)
// and bound_bin is monotone in w, so their number is at most cum[bin(H) + 1] - cum[bin(L)].
//
// Evaluation (plane_pair_ub): S(c) in fp64 (it cancels: |n . c| and |d| are of the cloud's size), everything after it in fp32,
// each number pushed in the safe direction by far more than its rounding:
//   * g, n . u, n . v from operands rounded to fp32, two fused multiply-adds and a product each: off by at most 5 x 2^-24 |n|_1;
//     e32 = 5e-7 |n|_1 enters the slack with W, U, V -- the computed g is then THE coefficient of the inequality, exactly;
//   * U, V, W, R are rounded UP when converted (frame_to_f32: x (1 + 1e-6)), T and 1e-14 M_g likewise;
//   * the slack is inflated by 1e-5 of itself and of (T + a + |S(c)|) -- against ~1e-6 for its dozen fp32 operations and the
//     conversion of S(c) --, the interval's ends by 1e-5 of their size (the reciprocal -- the device's v_rcp_f32 is within
//     1 ulp, the host divides -- and the product);
//   * the bins are taken one further out on both sides: the fp32 bin coordinate (x - wlo) invd is within 1e-4 bins of
//     bound_bin's fp64 one wherever it lies inside the clamp.
// Records or frames with numbers outside [1e-12, 1e12] are not bounded (an fp32 product could leave the normal range): the
// pair then counts kTilePoints.
// M3D_BOUND_NO_SLACK (tests only): a = 0 and no outward bins -- the mutation the host check must catch.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#ifndef M3D_HD
#define M3D_HD __host__ __device__ __forceinline__
#endif
#else
#ifndef M3D_HD
#define M3D_HD inline
#endif
#endif

namespace m3d {

constexpr int kFrameStride = 20;      // doubles per tile frame (above)
constexpr int kBoundBins = 126;       // interior bins of a tile's histogram (bin 0 / kBoundBins + 1: below / above its range)
constexpr int kCumStride = 132;       // uint16 per tile: cum[0 .. kBoundBins + 2] (+ padding to 264 bytes)
constexpr int kBoundTilePoints = 512; // (= kTilePoints)

// bin of a coordinate w along the tile's thin direction: 0 = below the histogram's range, 1 .. kBoundBins inside,
// kBoundBins + 1 = above.  MONOTONE in w (a subtraction, a multiplication by a positive number, floor, clamp):
// w1 <= w2 => bound_bin(w1) <= bound_bin(w2), which is all the bound needs.
M3D_HD int bound_bin(double w, double wlo, double invd) {
    const double t = floor((w - wlo) * invd);
    return (int)fmin(fmax(t, -1.0), (double)kBoundBins) + 1;
}

// frame slot k as plane_pair_ub reads it: fp32, the extents rounded up
M3D_HD float frame_to_f32(double v, uint32_t k) {
    const bool up = k == 12u || k == 13u || k == 14u || k == 17u;   // U, V, R, W
    return (float)(up ? v * (1.0 + 1e-6) : v);
}

// what plane_pair_ub needs of a hypothesis: its scoring record (a, b, c, d, T) and a few derived fp32 numbers
struct PlaneBoundRec {
    double a, b, c, d;
    float af, bf, cf, n1, e32, Tf, mgf;
    bool ok;
};
M3D_HD PlaneBoundRec plane_bound_record(const double* rec /* a, b, c, d, T */, double max_abs) {
    PlaneBoundRec r;
    r.a = rec[0]; r.b = rec[1]; r.c = rec[2]; r.d = rec[3];
    const double T = rec[4];
    const double n1d = (fabs(r.a) + fabs(r.b)) + fabs(r.c);
    const double Mg = n1d * max_abs + fabs(r.d);
    r.ok = (T > 1e-12) && (T < 1e12) && (n1d > 1e-12) && (n1d < 1e12) && (Mg < 1e12);
    r.af = (float)r.a; r.bf = (float)r.b; r.cf = (float)r.c;
    r.n1 = (float)(n1d * (1.0 + 1e-6));
    r.e32 = 5e-7f * r.n1;
    r.Tf = (float)(T * (1.0 + 1e-6));
    r.mgf = (float)(1e-14 * Mg * (1.0 + 1e-6));
    return r;
}

M3D_HD float bound_rcp(float g) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(g);
#else
    return 1.0f / g;
#endif
}

// upper bound of the inliers of hypothesis `r` among the points of one tile: c3 = the frame's centre (fp64), f = its slots as
// frame_to_f32 gives them (f[k] for k = 3 .. 19), cm = its cumulative histogram
template <class CumPtr>
M3D_HD uint32_t plane_pair_ub(const PlaneBoundRec& r, const double* c3, const float* f, CumPtr cm) {
    const double sc0 = ((r.a * c3[0] + r.b * c3[1]) + r.c * c3[2]) + r.d;
    const float g0 = __builtin_fmaf(r.af, f[3], __builtin_fmaf(r.bf, f[4], r.cf * f[5]));
    const float nu = __builtin_fmaf(r.af, f[6], __builtin_fmaf(r.bf, f[7], r.cf * f[8]));
    const float nv = __builtin_fmaf(r.af, f[9], __builtin_fmaf(r.bf, f[10], r.cf * f[11]));
    const float U = f[12], V = f[13], R = f[14], W = f[17];
    const float scf = (float)sc0;
    float aa = __builtin_fmaf(__builtin_fabsf(nu) + r.e32, U,
                              __builtin_fmaf(__builtin_fabsf(nv) + r.e32, V, __builtin_fmaf(r.n1, R, __builtin_fmaf(r.e32, W, r.mgf))));
    aa = __builtin_fmaf(aa, 1.00001f, 1e-5f * ((r.Tf + aa) + __builtin_fabsf(scf)));
#ifdef M3D_BOUND_NO_SLACK
    aa = 0.0f;
#endif
    const float g = __builtin_fabsf(g0), sc = g0 < 0.0f ? -scf : scf;
    const float ig = bound_rcp(g);   // (g = 0: inf; with a zero numerator NaN -- caught below)
    float L = ((-r.Tf - aa) - sc) * ig, H = ((r.Tf + aa) - sc) * ig;
    L -= 1e-5f * __builtin_fabsf(L);
    H += 1e-5f * __builtin_fabsf(H);
    const bool framed = f[19] != 0.0f && r.ok;
    const float wlo = f[15], invd = f[16];
#ifdef M3D_BOUND_NO_SLACK
    const float out = 0.0f;
#else
    const float out = 1.0f;   // one bin further out on both sides
#endif
    const float tL = __builtin_floorf((L - wlo) * invd) - out, tH = __builtin_floorf((H - wlo) * invd) + out;
    const bool whole = !(tL == tL) || !(tH == tH);   // (0 x inf, inf - inf: the direction says nothing -- every finite point of the tile)
    const int bl = (int)__builtin_fminf(__builtin_fmaxf(whole ? 0.0f : tL, -1.0f), (float)kBoundBins) + 1;
    const int bh = (int)__builtin_fminf(__builtin_fmaxf(whole ? 0.0f : tH, -1.0f), (float)kBoundBins) + 1;
    const int lo_c = (int)cm[bl], hi_c = (int)cm[bh + 1];
    uint32_t u_t = (uint32_t)(hi_c > lo_c ? hi_c - lo_c : 0);
    u_t = whole ? (uint32_t)f[18] : u_t;
    return framed ? u_t : (uint32_t)kBoundTilePoints;
}

// ---- cylinders (round 5) ----------------------------------------------------------------------------------------------------
// The same histograms bound a CYLINDER hypothesis: over one tile its shell is a slab, up to a sagitta.  Scoring record
// (p1, p2, t_lo, t_hi): inlier  <=>  t_lo <= t(q) <= t_hi,  t = |(q - p1) x (q - p2)|^2 = |L|^2 dist(q, axis)^2,  L = p2 - p1
// (m3d_fp.hpp cylinder_cutoffs: the reference's fabs(dist - r) < threshold, ransac.h:435-445, on the pre-sqrt quantity).  With
// d = L / |L|, m = the part of (c - p1) orthogonal to d, f = |m|, n = m / f (the radial direction at the tile's centre c) and
// Delta = p - c:   dist(p)^2 = (f + n . Delta)^2 + tau^2,  tau = the part of Delta orthogonal to d and n,  |tau| <= |Delta| <= rho.
// For a tile that lies on ONE side of the axis (f > rho):
//   dist <= d_hi  =>  n . Delta <= d_hi - f;      dist >= d_lo > rho  =>  n . Delta >= sqrt(d_lo^2 - rho^2) - f
// (d_lo <= rho: no lower end) -- a slab of width (d_hi - d_lo) + rho^2 / (2 d_lo) along n, and from there on plane_pair_ub's
// argument with n as the normal:  n . Delta = g w_p + (n . u) u_p + (n . v) v_p + n . res_p,  g = n . e.
// d_lo, d_hi carry the exact test's own rounding (E_t = 1e-12 |L| W^3, W = 2 max |coordinate| + |p1|_1 + |p2|_1: twenty times the
// bound of m3d_fp.hpp's cylinder_screen_record on the fp64 evaluation of t); c - p1 in fp64, everything behind it in fp32 with its
// roundings in the slack (cyl_pair_ub) and plane_pair_ub's margins.  rho = sqrt(W^2 + U^2 + V^2) (1 + 1e-4) + 1.75 R (cyl_tile_rho).
struct CylBoundRec {
    double p1[3];
    float df[3], dlof, dhif, mg;   // the unit direction, d_lo rounded down, d_hi rounded up
    bool ok;
};
M3D_HD CylBoundRec cyl_bound_record(const double* rec /* p1 (3), p2 (3), t_lo, t_hi */, double max_abs) {
    CylBoundRec r;
    const double L[3] = {rec[3] - rec[0], rec[4] - rec[1], rec[5] - rec[2]};
    const double Ln = sqrt((L[0] * L[0] + L[1] * L[1]) + L[2] * L[2]);
    const double Wm = 2.0 * max_abs + ((fabs(rec[0]) + fabs(rec[1])) + fabs(rec[2])) + ((fabs(rec[3]) + fabs(rec[4])) + fabs(rec[5]));
    const double Et = 1e-12 * Ln * ((Wm * Wm) * Wm);
    const double tlo = rec[6], thi = rec[7];
    r.ok = (tlo == tlo) && (thi == thi) && (thi >= 0.0) && (Ln > 1e-12) && (Ln < 1e12) && (Wm < 1e12) && (thi < 1e60);
    for (int k = 0; k < 3; ++k) {
        r.p1[k] = rec[k];
        r.df[k] = (float)(L[k] / Ln);
    }
    const double dhi = sqrt(thi + Et) / Ln * (1.0 + 1e-12);
    const double tl = tlo - Et;
    const double dlo = tl > 0.0 ? sqrt(tl) / Ln * (1.0 - 1e-12) : 0.0;
    r.dhif = (float)(dhi * (1.0 + 1e-6));
    r.dlof = (float)(dlo * (1.0 - 1e-6));
    r.mg = (float)(1e-13 * Wm * (1.0 + 1e-6));   // (c - p1 in fp64)
    r.ok = r.ok && (dhi < 1e12);
    return r;
}
// A SPHERE is the same with no axis to project out: record (centre, s_lo, s_hi), inlier <=> s_lo <= |q - centre|^2 <= s_hi
// (m3d_fp.hpp sphere_cutoffs, ransac.h:332-343); m = c - centre, d = 0.
M3D_HD CylBoundRec sphere_bound_record(const double* rec /* centre (3), s_lo, s_hi */, double max_abs) {
    CylBoundRec r;
    const double Wm = 2.0 * max_abs + ((fabs(rec[0]) + fabs(rec[1])) + fabs(rec[2]));
    const double Es = 1e-12 * (Wm * Wm);   // (the fp64 evaluation of s: three squares of differences <= W, two additions)
    const double slo = rec[3], shi = rec[4];
    r.ok = (slo == slo) && (shi == shi) && (shi >= 0.0) && (Wm < 1e12) && (shi < 1e30);
    for (int k = 0; k < 3; ++k) {
        r.p1[k] = rec[k];
        r.df[k] = 0.0f;
    }
    const double dhi = sqrt(shi + Es) * (1.0 + 1e-12);
    const double sl = slo - Es;
    const double dlo = sl > 0.0 ? sqrt(sl) * (1.0 - 1e-12) : 0.0;
    r.dhif = (float)(dhi * (1.0 + 1e-6));
    r.dlof = (float)(dlo * (1.0 - 1e-6));
    r.mg = (float)(1e-13 * Wm * (1.0 + 1e-6));
    return r;
}
M3D_HD float bound_sqrt(float x) {   // (the device's v_sqrt_f32: within 1 ulp for normal arguments, the margins below take 3e-7)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
// rho >= |p - c| for every point of a tile: |w e + u_p u + v_p v| <= sqrt(W^2 + U^2 + V^2) (1 + 1e-4) (e, u, v orthonormal to ~1e-6:
// tile_frames_k) + the residual's length.  f: the frame's slots as frame_to_f32 gives them.
M3D_HD float cyl_tile_rho(const float* f) {
    const float U = f[12], V = f[13], R = f[14], W = f[17];
    return __builtin_fmaf(bound_sqrt(__builtin_fmaf(W, W, __builtin_fmaf(U, U, V * V))), 1.0001f, 1.75f * R);
}
// Evaluation: c - p1 in fp64 (both of the cloud's size), then fp32: every step from there to f rounds a number of size <= |c - p1|
// to 2^-24 of itself -- the dot product with d, m, its square, the square root (one instruction on the device, within 1 ulp), the
// reciprocal -- so f, the interval's ends and n . Delta (|Delta| <= rho < f) are off by less than 1e-6 |c - p1|_1 together: the
// slack carries 2e-6 |c - p1|_1 (and d, d_lo, d_hi enter rounded to fp32, outwards).
template <class CumPtr>
M3D_HD uint32_t cyl_pair_ub(const CylBoundRec& r, const double* c3, const float* f, float rho, CumPtr cm) {
    const float U = f[12], V = f[13], R = f[14], W = f[17];
    const bool framed = f[19] != 0.0f && r.ok;
    const float v0 = (float)(c3[0] - r.p1[0]), v1 = (float)(c3[1] - r.p1[1]), v2 = (float)(c3[2] - r.p1[2]);
    const float v1n = (__builtin_fabsf(v0) + __builtin_fabsf(v1)) + __builtin_fabsf(v2);
    const float k = __builtin_fmaf(v0, r.df[0], __builtin_fmaf(v1, r.df[1], v2 * r.df[2]));
    const float m0 = __builtin_fmaf(-k, r.df[0], v0), m1 = __builtin_fmaf(-k, r.df[1], v1), m2 = __builtin_fmaf(-k, r.df[2], v2);
    const float f2 = __builtin_fmaf(m0, m0, __builtin_fmaf(m1, m1, m2 * m2));
    const float fc = bound_sqrt(f2);
    const float eabs = 2e-6f * v1n;   // (what f, the ends and n . Delta are off by, above)
    const bool one_side = fc - eabs > rho * 1.001f && f2 < 1e30f && f2 > 1e-30f;   // (else the axis may pass through the tile: every finite point counts)
    const float inv = bound_rcp(fc);
    const float nx = m0 * inv, ny = m1 * inv, nz = m2 * inv;
    const float n1 = 1.7321f, e32 = 5e-7f * 1.7321f;   // |n|_1 <= sqrt 3; the fp32 rounding of the three dot products below
    const float g0 = __builtin_fmaf(nx, f[3], __builtin_fmaf(ny, f[4], nz * f[5]));
    const float nu = __builtin_fmaf(nx, f[6], __builtin_fmaf(ny, f[7], nz * f[8]));
    const float nv = __builtin_fmaf(nx, f[9], __builtin_fmaf(ny, f[10], nz * f[11]));
    const float hif = r.dhif - fc;
    const float rho1 = rho * 1.0001f;
    // d_lo^2 - rho^2, less the roundings of its two squares (it is a difference of nearly equal numbers when the tile just fits)
    const float q2 = __builtin_fmaf(r.dlof, r.dlof, -(rho1 * rho1)) - 3e-7f * (r.dlof * r.dlof);
    const bool has_lo = q2 > 1e-30f;
    const float lof = has_lo ? bound_sqrt(q2) * (1.0f - 3e-7f) - fc : 0.0f;
    float aa = __builtin_fmaf(__builtin_fabsf(nu) + e32, U,
                              __builtin_fmaf(__builtin_fabsf(nv) + e32, V, __builtin_fmaf(n1, R, __builtin_fmaf(e32, W, r.mg + eabs))));
    aa = __builtin_fmaf(aa, 1.00001f, 1e-5f * ((__builtin_fabsf(hif) + __builtin_fabsf(lof)) + aa));
#ifdef M3D_BOUND_NO_SLACK
    aa = 0.0f;
#endif
    const float ninf = -__builtin_huge_valf();
    // the slab in the direction of growing w: [lo, hi] for g0 > 0, [-hi, -lo] otherwise
    const float g = __builtin_fabsf(g0);
    const float a_lo = g0 < 0.0f ? -hif : (has_lo ? lof : ninf), a_hi = g0 < 0.0f ? (has_lo ? -lof : -ninf) : hif;
    const float ig = bound_rcp(g);
    float Lw = (a_lo - aa) * ig, Hw = (a_hi + aa) * ig;
    Lw -= 1e-5f * __builtin_fabsf(Lw);
    Hw += 1e-5f * __builtin_fabsf(Hw);
    const float wlo = f[15], invd = f[16];
#ifdef M3D_BOUND_NO_SLACK
    const float out = 0.0f;
#else
    const float out = 1.0f;
#endif
    const float tL = __builtin_floorf((Lw - wlo) * invd) - out, tH = __builtin_floorf((Hw - wlo) * invd) + out;
    const bool whole = !(tL == tL) || !(tH == tH) || !one_side;
    const int bl = (int)__builtin_fminf(__builtin_fmaxf(whole ? 0.0f : tL, -1.0f), (float)kBoundBins) + 1;
    const int bh = (int)__builtin_fminf(__builtin_fmaxf(whole ? 0.0f : tH, -1.0f), (float)kBoundBins) + 1;
    const int lo_c = (int)cm[bl], hi_c = (int)cm[bh + 1];
    uint32_t u_t = (uint32_t)(hi_c > lo_c ? hi_c - lo_c : 0);
    u_t = whole ? (uint32_t)f[18] : u_t;
    return framed ? u_t : (uint32_t)kBoundTilePoints;
}

}  // namespace m3d
