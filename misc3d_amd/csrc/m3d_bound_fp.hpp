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

}  // namespace m3d
