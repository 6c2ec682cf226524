// m3d_fp.hpp -- floating-point operation order of the Misc3D hot path, written once for the host
// driver and the gfx950 kernels.
//
// The reference (include/misc3d/common/ransac.h) is built with plain -O3 on x86-64 (no FMA,
// CMakeLists.txt:16) and evaluates its small dot products through Eigen's packet reductions, so
// every product and every sum below is individually rounded and the association is fixed:
//   4-element sums (e0+e2)+(e1+e3), 3-element sums (e0+e1)+e2   
// This file and everything that includes it MUST be compiled with -ffp-contract=off.
//
// M3D_FP_ORDER (compile time; oracle/misc3d_oracle.c has the same switch as ORC_FP_ORDER) selects which Eigen the
// association is restated for -- the reference cannot be built here (no Eigen / Open3D in the image), so the
// association is a [RECALL] and this is the ONE place to flip it once tools/pin_reference shows which one the real
// build has:
//   0  Eigen >= 3.3, SSE2 packets, unaligned fixed-size vectorisation (default):
//        3-element reductions (e0 + e1) + e2 -- one packet (e0, e1) reduced, then the scalar tail;
//        4x4 determinant = Eigen 3.3 determinant_impl<.,4> (six 2x2 minors, bruteforce_det4_helper)
//   1  Eigen 3.2 (Vector3d is not vectorised: redux_novec_unroller splits 3 = 1 + 2):
//        3-element reductions e0 + (e1 + e2); determinant as in 0
//   2  Eigen 3.4: reductions as in 0; 4x4 determinant = 3.4's determinant_impl<.,4> (det2 / det3 cofactors, pmadd
//        without FMA = a * b + c)
// 4-element reductions of aligned Vector4d are (e0 + e2) + (e1 + e3) in all three (two SSE2 packets added, then
// predux).  The build puts the variants side by side: lib/libmisc3d_amd.so (0), lib/order1/, lib/order2/.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define M3D_HD __host__ __device__ __forceinline__
#else
#define M3D_HD inline
#endif

#pragma clang fp contract(off)

namespace m3d {

#ifndef M3D_FP_ORDER
#define M3D_FP_ORDER 0
#endif
#if M3D_FP_ORDER < 0 || M3D_FP_ORDER > 2
#error "M3D_FP_ORDER must be 0 (Eigen >= 3.3), 1 (Eigen 3.2) or 2 (Eigen 3.4)"
#endif
constexpr int kFpOrder = M3D_FP_ORDER;

constexpr double kEps = 1.0e-8;  // ransac.h:14 EPS

// the 3-element reduction every dot product / squaredNorm of a Vector3d goes through
M3D_HD double sum3(double e0, double e1, double e2) {
#if M3D_FP_ORDER == 1
    return e0 + (e1 + e2);
#else
    return (e0 + e1) + e2;
#endif
}
M3D_HD double dot3(double ax, double ay, double az, double bx, double by, double bz) {
    return sum3(ax * bx, ay * by, az * bz);
}
M3D_HD double norm3(double x, double y, double z) { return sqrt(dot3(x, y, z, x, y, z)); }
M3D_HD double dot4(double a0, double a1, double a2, double a3, double b0, double b1, double b2,
                   double b3) {
    return (a0 * b0 + a2 * b2) + (a1 * b1 + a3 * b3);
}

M3D_HD uint64_t f2u(double v) {
    uint64_t u;
    memcpy(&u, &v, 8);
    return u;
}
M3D_HD double u2f(uint64_t u) {
    double v;
    memcpy(&v, &u, 8);
    return v;
}
constexpr uint64_t kInfBits = 0x7FF0000000000000ull;

// Non-negative doubles are ordered like their bit patterns.  `pred` must be monotone on
// [lo, hi]: true on a (possibly empty) prefix, false afterwards.  Precondition: pred(lo) true,
// pred(hi) false.  Returns the bits of the FIRST value for which pred is false.
template <class F>
M3D_HD uint64_t first_false(uint64_t lo, uint64_t hi, F pred) {
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (pred(u2f(mid)))
            lo = mid;
        else
            hi = mid;
    }
    return hi;
}
// `pred` false on a prefix, true afterwards.  Precondition: pred(lo) false, pred(hi) true.
// Returns the bits of the FIRST value for which pred is true.
template <class F>
M3D_HD uint64_t first_true(uint64_t lo, uint64_t hi, F pred) {
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (pred(u2f(mid)))
            hi = mid;
        else
            lo = mid;
    }
    return hi;
}

// The same searches with a GUESS: the cut-offs sit within a few ulps of a value the caller can compute (r -+ thr, its
// square, ...), so the bracket is first narrowed to 16 ulps around the guess when the predicate confirms it (two
// evaluations) and bisected from there (four more) instead of over the whole range (62 dependent sqrt / divide
// sequences each: the sphere's and the cylinder's four searches made minimal_fit_k 28-32 us per 12 500 hypotheses).
// Any bracket with pred(lo) != pred(hi) gives the same answer by monotonicity; a guess that is not confirmed leaves the
// wide bracket (or half of it) in place.  NaN / out-of-range guesses are ignored.
template <class F>
M3D_HD uint64_t first_false_guided(uint64_t lo, uint64_t hi, double guess, F pred) {
    if (guess >= 0.0 && guess < INFINITY) {
        const uint64_t b = f2u(guess);
        if (b > lo && b < hi) {
            constexpr uint64_t W = 16;
            if (pred(u2f(b))) {
                lo = b;
                if (hi - b > W && !pred(u2f(b + W))) hi = b + W;
            } else {
                hi = b;
                if (b - lo > W && pred(u2f(b - W))) lo = b - W;
            }
        }
    }
    return first_false(lo, hi, pred);
}
template <class F>
M3D_HD uint64_t first_true_guided(uint64_t lo, uint64_t hi, double guess, F pred) {
    if (guess >= 0.0 && guess < INFINITY) {
        const uint64_t b = f2u(guess);
        if (b > lo && b < hi) {
            constexpr uint64_t W = 16;
            if (pred(u2f(b))) {
                hi = b;
                if (b - lo > W && !pred(u2f(b - W))) lo = b - W;
            } else {
                lo = b;
                if (hi - b > W && pred(u2f(b + W))) hi = b + W;
            }
        }
    }
    return first_true(lo, hi, pred);
}

// ------------------------------------------------------------------------------------------------
// Plane: ransac.h:134-221
// ------------------------------------------------------------------------------------------------
// MinimalFit, ransac.h:138-162.  p0,p1,p2 -> (a,b,c,d).  false = collinear sample.
M3D_HD bool plane_minimal_fit(const double* p0, const double* p1, const double* p2, double* out) {
    const double e0x = p1[0] - p0[0], e0y = p1[1] - p0[1], e0z = p1[2] - p0[2];
    const double e1x = p2[0] - p0[0], e1y = p2[1] - p0[1], e1z = p2[2] - p0[2];
    double ax = e0y * e1z - e0z * e1y;
    double ay = e0z * e1x - e0x * e1z;
    double az = e0x * e1y - e0y * e1x;
    const double norm = norm3(ax, ay, az);
    if (norm < kEps) return false;
    const double n2 = norm3(ax, ay, az);  // ransac.h:154 recomputes abc.norm()
    ax /= n2;
    ay /= n2;
    az /= n2;
    out[0] = ax;
    out[1] = ay;
    out[2] = az;
    out[3] = -dot3(ax, ay, az, p0[0], p0[1], p0[2]);
    return true;
}
// numerator and denominator of CalcPointToModelDistance, ransac.h:215-220
M3D_HD double plane_num(double a, double b, double c, double d, double x, double y, double z) {
    return fabs((a * x + c * z) + (b * y + d));  // d * 1 == d exactly
}
M3D_HD double plane_distance(const double* m, double x, double y, double z) {
    return plane_num(m[0], m[1], m[2], m[3], x, y, z) / norm3(m[0], m[1], m[2]);
}
// Exact cut-off: for a fixed model, RN(num / nrm) < thr  <=>  num < T.  RN(num/nrm) is monotone in
// num, so T is the first double for which the reference's own test fails.
// first_false with a starting guess: the answer is almost always within a few ulps of `guess` (the cut-off of
// RN(num / nrm) < thr sits next to thr * nrm), so walk from there -- by monotonicity the first false value found that
// way IS the first false value -- and fall back to the bisection over the whole range when the walk does not settle.
// 63 dependent divisions per hypothesis become 3 or 4 (minimal_fit_k: 9.7 -> 4 us for 10 000 planes).
template <class F>
M3D_HD uint64_t first_false_near(double guess, F pred) {
    if (guess >= 0.0 && guess < INFINITY) {   // (also rejects NaN)
        uint64_t b = f2u(guess);
        if (pred(u2f(b))) {
            for (int k = 0; k < 8 && b + 1 < kInfBits; ++k) {
                ++b;
                if (!pred(u2f(b))) return b;
            }
        } else {
            for (int k = 0; k < 8 && b > 0; ++k) {
                if (pred(u2f(b - 1))) return b;
                --b;
            }
            if (b == 0) return 0;
        }
    }
    return first_false(0, kInfBits, pred);   // precondition of the caller: pred(0) true, pred(inf) false
}
M3D_HD double plane_cutoff(const double* m, double thr) {
    const double nrm = norm3(m[0], m[1], m[2]);
    auto inl = [=](double num) { return num / nrm < thr; };
    if (!inl(0.0)) return 0.0;  // also NaN threshold / NaN model: nothing is an inlier
    return u2f(first_false_near(thr * nrm, inl));
}

// ------------------------------------------------------------------------------------------------
// Sphere: ransac.h:223-344
// ------------------------------------------------------------------------------------------------
M3D_HD double det4_helper(const double (*m)[4], int j, int k, int a, int b) {
    return (m[j][0] * m[k][1] - m[k][0] * m[j][1]) * (m[a][2] * m[b][3] - m[b][2] * m[a][3]);
}
#if M3D_FP_ORDER == 2
// Eigen 3.4 determinant_impl<.,4> ([RECALL]): 2x2 minors of columns 0-1, 3x3 cofactors along column 2, expansion
// along column 3; pmadd(a, b, c) = a * b + c (no FMA in the reference build)
M3D_HD double det4_d2(const double (*m)[4], int i0, int i1) { return m[i0][0] * m[i1][1] - m[i1][0] * m[i0][1]; }
M3D_HD double det4_d3(const double (*m)[4], int i0, double d0, int i1, double d1, int i2, double d2) {
    return m[i0][2] * d0 + ((-m[i1][2]) * d1 + m[i2][2] * d2);
}
M3D_HD double det4(const double (*m)[4]) {
    const double d01 = det4_d2(m, 0, 1), d02 = det4_d2(m, 0, 2), d03 = det4_d2(m, 0, 3);
    const double d12 = det4_d2(m, 1, 2), d13 = det4_d2(m, 1, 3), d23 = det4_d2(m, 2, 3);
    const double c0 = det4_d3(m, 1, d23, 2, d13, 3, d12);
    const double c1 = det4_d3(m, 0, d23, 2, d03, 3, d02);
    const double c2 = det4_d3(m, 0, d13, 1, d03, 3, d01);
    const double c3 = det4_d3(m, 0, d12, 1, d02, 2, d01);
    return ((-m[0][3]) * c0 + m[1][3] * c1) + ((-m[2][3]) * c2 + m[3][3] * c3);
}
#else
// Eigen 3.3 determinant_impl<.,4> (Costabel's 30-multiply form)
M3D_HD double det4(const double (*m)[4]) {
    return det4_helper(m, 0, 1, 2, 3) - det4_helper(m, 0, 2, 1, 3) + det4_helper(m, 0, 3, 1, 2) +
           det4_helper(m, 1, 2, 0, 3) - det4_helper(m, 1, 3, 0, 2) + det4_helper(m, 2, 3, 0, 1);
}
#endif
// ValidationCheck + MinimalFit, ransac.h:225-234,239-294.  p = 4 points (12 doubles).
M3D_HD bool sphere_minimal_fit(const double* p, double* out) {
    double plane[4];
    if (!plane_minimal_fit(p, p + 3, p + 6, plane)) return false;
    if (plane_distance(plane, p[9], p[10], p[11]) < kEps) return false;
    double sq[4];
    for (int i = 0; i < 4; ++i)
        sq[i] = dot3(p[3 * i], p[3 * i + 1], p[3 * i + 2], p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    double m[4][4];
    for (int i = 0; i < 4; ++i) {
        m[i][0] = p[3 * i];
        m[i][1] = p[3 * i + 1];
        m[i][2] = p[3 * i + 2];
        m[i][3] = 1.0;
    }
    const double M11 = det4(m);
    for (int i = 0; i < 4; ++i) {
        m[i][0] = sq[i];
        m[i][1] = p[3 * i + 1];
        m[i][2] = p[3 * i + 2];
    }
    const double M12 = det4(m);
    for (int i = 0; i < 4; ++i) {
        m[i][0] = sq[i];
        m[i][1] = p[3 * i];
        m[i][2] = p[3 * i + 2];
    }
    const double M13 = det4(m);
    for (int i = 0; i < 4; ++i) {
        m[i][0] = sq[i];
        m[i][1] = p[3 * i];
        m[i][2] = p[3 * i + 1];
    }
    const double M14 = det4(m);
    for (int i = 0; i < 4; ++i) {
        m[i][0] = sq[i];
        m[i][1] = p[3 * i];
        m[i][2] = p[3 * i + 1];
        m[i][3] = p[3 * i + 2];
    }
    const double M15 = det4(m);
    const double cx = 0.5 * (M12 / M11), cy = -0.5 * (M13 / M11), cz = 0.5 * (M14 / M11);
    out[0] = cx;
    out[1] = cy;
    out[2] = cz;
    out[3] = sqrt(dot3(cx, cy, cz, cx, cy, cz) - (M15 / M11));
    return true;
}
// squared distance to the centre in the reference's order (the argument of .norm(), ransac.h:336)
M3D_HD double sphere_s(double cx, double cy, double cz, double x, double y, double z) {
    const double dx = x - cx, dy = y - cy, dz = z - cz;
    return sum3(dx * dx, dy * dy, dz * dz);
}
M3D_HD double sphere_dist_from_d(double d, double r) { return d <= r ? r - d : d - r; }
M3D_HD double sphere_distance(const double* m, double x, double y, double z) {
    return sphere_dist_from_d(sqrt(sphere_s(m[0], m[1], m[2], x, y, z)), m[3]);
}

// For |d - r| style distances: the set {d >= 0 : dist(d) < thr} is an interval [dA, dB] around r
// (dist is monotone on both sides of r).  Returns false when it is empty.
// gA, gB: where the ends are expected (r - thr, r + thr; NaN: no guess).
template <class F>
M3D_HD bool radial_interval(double r, F inl_d, double* dA, double* dB, double gA = __builtin_nan(""),
                            double gB = __builtin_nan("")) {
    if (!(r == r) || fabs(r) == INFINITY) return false;
    const double d0 = r > 0.0 ? r : 0.0;
    if (!inl_d(d0)) return false;
    const uint64_t b0 = f2u(d0);
    if (b0 == 0 || inl_d(0.0))
        *dA = 0.0;
    else
        *dA = u2f(first_true_guided(0, b0, gA, inl_d));
    *dB = u2f(first_false_guided(b0, kInfBits, gB, inl_d) - 1);  // inl_d(inf) is always false
    return true;
}
// Monotone map g(t) (t >= 0): interval {t : dA <= g(t) <= dB} as [lo, hi]; false when empty.
// g_lo, g_hi: where the ends are expected (NaN: no guess).
template <class G>
M3D_HD bool preimage_interval(double dA, double dB, G g, double* lo, double* hi, double g_lo = __builtin_nan(""),
                              double g_hi = __builtin_nan("")) {
    auto ge = [=](double t) { return g(t) >= dA; };  // false..false true..true
    auto le = [=](double t) { return g(t) <= dB; };  // true..true false..false
    uint64_t blo;
    if (ge(0.0))
        blo = 0;
    else if (!ge(u2f(kInfBits - 1)))
        return false;
    else
        blo = first_true_guided(0, kInfBits - 1, g_lo, ge);
    if (!le(u2f(blo))) return false;
    uint64_t bhi;
    if (le(u2f(kInfBits - 1)))
        bhi = kInfBits - 1;
    else
        bhi = first_false_guided(blo, kInfBits - 1, g_hi, le) - 1;
    *lo = u2f(blo);
    *hi = u2f(bhi);
    return true;
}
// Exact cut-offs: dist(q) < thr  <=>  s_lo <= s(q) <= s_hi.  Empty -> (NaN, NaN).
M3D_HD void sphere_cutoffs(const double* m, double thr, double* s_lo, double* s_hi) {
    const double r = m[3];
    double dA, dB;
    const double nan = u2f(0x7FF8000000000000ull);
    *s_lo = nan;
    *s_hi = nan;
    if (!radial_interval(r, [=](double d) { return sphere_dist_from_d(d, r) < thr; }, &dA, &dB, r - thr, r + thr))
        return;
    double lo, hi;
    if (!preimage_interval(dA, dB, [](double s) { return sqrt(s); }, &lo, &hi, dA * dA, dB * dB)) return;
    *s_lo = lo;
    *s_hi = hi;
}

// ------------------------------------------------------------------------------------------------
// Cylinder: ransac.h:350-446, utils.h:313-322
// ------------------------------------------------------------------------------------------------
// CalcPoint2LineDistance pieces: t = |a x b|^2 with a = q - p1, b = q - p2
M3D_HD double line_t(double p1x, double p1y, double p1z, double p2x, double p2y, double p2z,
                     double x, double y, double z) {
    const double ax = x - p1x, ay = y - p1y, az = z - p1z;
    const double bx = x - p2x, by = y - p2y, bz = z - p2z;
    const double cx = ay * bz - az * by;
    const double cy = az * bx - ax * bz;
    const double cz = ax * by - ay * bx;
    return sum3(cx * cx, cy * cy, cz * cz);
}
M3D_HD double point2line(const double* q, const double* p1, const double* p2) {
    const double t = line_t(p1[0], p1[1], p1[2], p2[0], p2[1], p2[2], q[0], q[1], q[2]);
    const double L = norm3(p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]);
    return sqrt(t) / L;
}
// MinimalFit, ransac.h:354-417.  p = 2 points, n = 2 normals -> 7 params.
M3D_HD bool cylinder_minimal_fit(const double* p, const double* n, double* out) {
    // ransac.h:367-374 (comparisons sit INSIDE fabs() in the reference)
    if ((p[0] - p[3] <= 2.220446049250313e-16) && (fabs(p[1] - p[4]) <= 1.1920928955078125e-07) &&
        (fabs(p[2] - p[5]) <= 1.1920928955078125e-07))
        return false;
    const double p1[4] = {p[0], p[1], p[2], 0.0}, p2[4] = {p[3], p[4], p[5], 0.0};
    const double n1[4] = {n[0], n[1], n[2], 0.0}, n2[4] = {n[3], n[4], n[5], 0.0};
    double w[4];
    for (int k = 0; k < 4; ++k) w[k] = (n1[k] + p1[k]) - p2[k];
    const double a = dot4(n1[0], n1[1], n1[2], n1[3], n1[0], n1[1], n1[2], n1[3]);
    const double b = dot4(n1[0], n1[1], n1[2], n1[3], n2[0], n2[1], n2[2], n2[3]);
    const double c = dot4(n2[0], n2[1], n2[2], n2[3], n2[0], n2[1], n2[2], n2[3]);
    const double d = dot4(n1[0], n1[1], n1[2], n1[3], w[0], w[1], w[2], w[3]);
    const double e = dot4(n2[0], n2[1], n2[2], n2[3], w[0], w[1], w[2], w[3]);
    const double den = a * c - b * b;
    double sc, tc;
    if (den < 1e-8) {
        sc = 0.0;
        tc = (b > c ? d / b : e / c);
    } else {
        sc = (b * e - c * d) / den;
        tc = (a * e - b * d) / den;
    }
    double lp[4], ld[4];
    for (int k = 0; k < 4; ++k) lp[k] = (p1[k] + n1[k]) + sc * n1[k];
    for (int k = 0; k < 4; ++k) ld[k] = (p2[k] + tc * n2[k]) - lp[k];
    const double z = dot4(ld[0], ld[1], ld[2], ld[3], ld[0], ld[1], ld[2], ld[3]);
    if (z > 0.0) {  // Eigen >= 3.3 normalize()
        const double s = sqrt(z);
        for (int k = 0; k < 4; ++k) ld[k] /= s;
    }
    out[0] = lp[0];
    out[1] = lp[1];
    out[2] = lp[2];
    out[3] = ld[0];
    out[4] = ld[1];
    out[5] = ld[2];
    out[6] = point2line(p, lp, ld);  // ransac.h:413-414: direction passed as 2nd point
    return true;
}
// CalcPointToModelDistance, ransac.h:435-445: ref = centre + direction (rounded), L = |ref - centre|
M3D_HD void cylinder_ref(const double* w, double* ref, double* L) {
    ref[0] = w[0] + w[3];
    ref[1] = w[1] + w[4];
    ref[2] = w[2] + w[5];
    *L = norm3(ref[0] - w[0], ref[1] - w[1], ref[2] - w[2]);
}
M3D_HD double cylinder_distance(const double* w, double x, double y, double z) {
    double ref[3], L;
    cylinder_ref(w, ref, &L);
    const double t = line_t(w[0], w[1], w[2], ref[0], ref[1], ref[2], x, y, z);
    return fabs(sqrt(t) / L - w[6]);
}
// dist(q) < thr  <=>  t_lo <= t(q) <= t_hi.  Empty -> (NaN, NaN).
M3D_HD void cylinder_cutoffs(const double* w, double thr, double* t_lo, double* t_hi) {
    double ref[3], L;
    cylinder_ref(w, ref, &L);
    const double r = w[6];
    const double nan = u2f(0x7FF8000000000000ull);
    *t_lo = nan;
    *t_hi = nan;
    double dA, dB;
    if (!radial_interval(r, [=](double d) { return fabs(d - r) < thr; }, &dA, &dB, r - thr, r + thr)) return;
    double lo, hi;
    if (!preimage_interval(dA, dB, [=](double t) { return sqrt(t) / L; }, &lo, &hi, (dA * L) * (dA * L), (dB * L) * (dB * L)))
        return;
    *t_lo = lo;
    *t_hi = hi;
}

// ------------------------------------------------------------------------------------------------
// fp32 SCREENING records of the culled scoring kernel (score_screen_k, m3d_cull_kernels.hip).
//
// The inlier decision of the reference is an fp64 one and stays one: the packed-fp32 pass only sorts the points of
// a tile into "certainly inside", "certainly outside" and "too close to call" for one hypothesis, with a rounding
// bound that covers the fp32 evaluation, the fp32 rounding of its inputs AND the fp64 rounding of the value the exact
// test looks at; a (tile, hypothesis) pair with a single point of the third kind is counted again by the exact fp64
// code (tile_count).  Counts therefore equal the fp64 counts bit for bit; the fp32 pass only decides how often the
// fp64 code runs.
//
// Everything is evaluated RELATIVE TO THE TILE'S BOX CENTRE o: the kernel keeps xr = fl32(fl64(x - o)) (|x - o| <= the
// box's half extent e, boxes are small), the record of a (tile, hypothesis) pair moves the model there in fp64.  The
// bound then scales with the tile's extent and the model's value at the tile, not with the cloud's coordinates: 30x
// fewer recounts on the 1M-point bench cloud than with absolute coordinates.
//
// Notation: u = 2^-24 (fp32 unit round-off), A >= |coordinate| of every finite point of the cloud.
//
// Plane.  Exact test |s64| < T, s64 = the fp64 value of plane_num's argument; S = a x + b y + c z + d in real
// arithmetic = a (x - ox) + b (y - oy) + c (z - oz) + S_o.  s32 = fma(a~, xr, fma(b~, yr, fma(c~, zr, D))) with
// D = fl32(s_o), s_o = the fp64 value of the model at o: each of the four terms carries at most five factors
// (1 + delta), |delta| <= u (two input roundings, three fmas), so |s32 - (sum a (x - o) + s_o)| <= gamma_5 M_l with
// M_l = |a| ex + |b| ey + |c| ez + |s_o|; the fp64 roundings (x - o, s_o against S_o, s64 against S) stay below
// 7 * 2^-53 M_g, M_g = (|a| + |b| + |c|) A + |d|.  E = 8 u M_l + 1e-15 M_g (+ an absolute term for flushed fp32
// denormals) bounds |s32 - s64| with room to spare.  The kernel forms q = fma(s32, s32, -mid2), mid2 = fp32(T^2):
// sign(q) is the sign of s32^2 - mid2 (one rounding of an exact value); |s64| < T is certain when s32^2 < (T - E)^2
// and impossible when s32^2 >= (T + E)^2, i.e. whenever |q| >= h with h >= (2 T E + E^2 + |mid2 - T^2|) / (1 - u).
//
// Sphere.  Exact test lo <= sv64 <= hi  <=>  |sv64 - mid| <= half.  With C = fl32(fl64(c - o)) the kernel forms
// d~ = fl32(xr - C), within 2.01 u W_l (+ 3 * 2^-53 W_g) of x - c, W_l = max_k (e_k + |c_k - o_k|), W_g = A + max |c_k|;
// sum d~_k^2 is then within 12.2 u W_l^2 of the real sum, the three fmas of t = fma(dz, dz, fma(dy, dy,
// fma(dx, dx, -mid32))) add at most 3.01 u (3 W_l^2 + mid), the rounding of mid u mid; the fp64 roundings (c - o:
// 2^-53 * 2 W_g per component, times 2 |x - c| <= 2 W_l, three components; sv64 itself: 6 * 2^-53 W_l^2) stay below
// 2e-15 W_l (W_g + W_l): E_t = 24 u W_l^2 + 5 u mid + 1e-14 W_l (W_g + W_l).  The kernel forms v = |t| - half32; the
// pair is certain when |v| >= h, h >= (E_t + u half) / (1 - u).
//
// A record that cannot be screened (non-finite or huge values, a cut-off below the rounding bound) carries h = NaN:
// `!(m >= h)` then sends every pair of that hypothesis to the exact code.
// ------------------------------------------------------------------------------------------------
constexpr double kU32 = 5.9604644775390625e-8;   // 2^-24
#ifndef M3D_SCREEN_BOUND_DIVISOR   // tests/test_screen_bounds.py builds the host check with 16: it must then FAIL
#define M3D_SCREEN_BOUND_DIVISOR 1.0
#endif
M3D_HD float f32_round_up_pos(double v) {   // smallest float >= v, v > 0 finite and far below the fp32 overflow
    float f = (float)v;
    if ((double)f < v) {
        uint32_t b;
        __builtin_memcpy(&b, &f, 4);
        ++b;
        __builtin_memcpy(&f, &b, 4);
    }
    return f;
}
M3D_HD float f32_nan() {
    const uint32_t b = 0x7FC00000u;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}
// rec = the plane's scoring record (a, b, c, d, T); box = (centre xyz, half extents xyz).
// out = (a, b, c, mid2, D, h, -, -)
M3D_HD void plane_screen_record(const double* rec, const double* box, double max_abs, float* out) {
    const double a = rec[0], b = rec[1], c = rec[2], d = rec[3], T = rec[4];
    const double s_o = ((a * box[0] + b * box[1]) + c * box[2]) + d;
    const double sa = (fabs(a) + fabs(b)) + fabs(c);
    const double Ml = ((fabs(a) * box[3] + fabs(b) * box[4]) + fabs(c) * box[5]) + fabs(s_o);
    const double Mg = sa * max_abs + fabs(d);
    const double E = ((8.0 * kU32 * Ml + 1e-15 * Mg) + 4e-38 * ((3.0 * max_abs + sa) + 4.0)) / M3D_SCREEN_BOUND_DIVISOR;
    const bool ok = (Mg < 1e18) && (sa < 1e18) && (max_abs < 1e18) && (T > E) && (T > 1e-15) && (T < 1e18);
    const double h = ((2.0 * T * E + E * E) + 2.0 * kU32 * (T * T)) * 1.001;
    out[0] = ok ? (float)a : 0.0f;
    out[1] = ok ? (float)b : 0.0f;
    out[2] = ok ? (float)c : 0.0f;
    out[3] = ok ? (float)(T * T) : 0.0f;
    out[4] = ok ? (float)s_o : 0.0f;
    out[5] = ok ? f32_round_up_pos(h > 1e-30 ? h : 1e-30) : f32_nan();
    out[6] = out[7] = 0.0f;
}
// rec = the sphere's scoring record (cx, cy, cz, lo, hi).  out = (K, half^2, -, -, -2 Cx, -2 Cy, -2 Cz, h)
// EXPANDED form.  x~ = the tile's fp32 offsets from the box centre, C = the sphere's centre relative to the box centre
// ROUNDED to fp32 (so that the expansion is an identity of the numbers the kernel holds):
//     |x~ - C|^2 - mid = w + C~ . x~ + K,      w = |x~|^2 (once per point and tile),  C~ = -2 C,  K = |C|^2 - mid.
// The kernel forms u = fma(C~z, z~, fma(C~y, y~, fma(C~x, x~, w + K))) -- 4 packed instructions per two points where
// (x~ - C)^2 summed took 6 -- and q = fma(u, u, -half^2): inside <=> lo <= t <= hi <=> |t - mid| <= half <=> q <= 0 (the
// plane's trick; `|u| - half` per point cost two unpacked subtractions).  33 VALU instructions per hypothesis, 45 before.
// Error of u against t - mid, t the exact code's |q - c|^2:
//   * inputs: x~_k = x_k (1 + d), C_k = c_k (1 + d'): (x~_k - C_k) is within u W_k of x_k - c_k, W_k = h_k + |c_k| >=
//     |x_k - c_k|: <= 2.01 u Ws on the sum of squares, Ws = sum W_k^2;
//   * w = fl(fl(fl(x~^2) + y~^2) + z~^2) <= gamma_3 |x~|^2 <= 3.1 u Hs, Hs = sum h_k^2;  K32 = K (1 + d): u |K|;
//   * the chain: four roundings, the earliest term passing all four: <= gamma_4 (w + |K| + sum |C~_k x~_k|)
//     <= 4.1 u (Hs + |K| + 2 P), P = sum h_k |c_k|.
//   E = 2.1 u Ws + 7.5 u Hs + 5.5 u |K| + 8.5 u P (+ the fp64 side: the exact code's own rounding on coordinates of
//   size Wg, the box centre's subtraction, mid / half / |C|^2; + flushed denormals).
// For a tile near the sphere's surface (|c| ~ r >> h) that is ~2 u r^2: the direct form's bound was 24 u (r + h)^2.
// q against q* = (t - mid)^2 - half^2 as for the plane: |q32| >= h = 2 half E + E^2 + 2 u half^2 => same sign.
M3D_HD void sphere_screen_record(const double* rec, const double* box, double max_abs, float* out) {
    const double lo = rec[3], hi = rec[4];
    const double cx = rec[0] - box[0], cy = rec[1] - box[1], cz = rec[2] - box[2];
    const float c32[3] = {(float)cx, (float)cy, (float)cz};
    const double Cx = (double)c32[0], Cy = (double)c32[1], Cz = (double)c32[2];
    const double mid = 0.5 * lo + 0.5 * hi, half = 0.5 * hi - 0.5 * lo;
    const double Cs = (Cx * Cx + Cy * Cy) + Cz * Cz;
    const double K = Cs - mid;
    const double hx = box[3], hy = box[4], hz = box[5];
    const double wx = hx + fabs(cx), wy = hy + fabs(cy), wz = hz + fabs(cz);
    const double Ws = (wx * wx + wy * wy) + wz * wz;
    const double Hs = (hx * hx + hy * hy) + hz * hz;
    const double P = (hx * fabs(cx) + hy * fabs(cy)) + hz * fabs(cz);
    const double Wl = fmax(fmax(wx, wy), wz);
    const double Wg = max_abs + fmax(fmax(fabs(rec[0]), fabs(rec[1])), fabs(rec[2]));
    const double E = ((((2.1 * kU32 * Ws + 7.5 * kU32 * Hs) + (5.5 * kU32 * fabs(K) + 8.5 * kU32 * P)) +
                       (1e-14 * (Wl * (Wg + Wl)) + 1e-15 * ((mid + half) + Cs))) +
                      (1e-30 * (Wl + 1.0) + 1e-37 * (4.0 + 4.0 * ((wx + wy) + wz)))) / M3D_SCREEN_BOUND_DIVISOR;
    const bool ok = (lo <= hi) && (lo >= 0.0) && (hi < 1e17) && (Wg < 1e18) && (Ws < 1e17) && (half > E) && (half > 1e-15) &&
                    (hx >= 0.0);
    const double h = ((2.0 * half * E + E * E) + 2.0 * kU32 * (half * half)) * 1.001;
    out[0] = ok ? (float)K : 0.0f;
    out[1] = ok ? (float)(half * half) : 0.0f;
    out[2] = out[3] = 0.0f;
    out[4] = ok ? -2.0f * c32[0] : 0.0f;
    out[5] = ok ? -2.0f * c32[1] : 0.0f;
    out[6] = ok ? -2.0f * c32[2] : 0.0f;
    out[7] = ok ? f32_round_up_pos(h > 1e-30 ? h : 1e-30) : f32_nan();
}

// rec = the cylinder's scoring record (p1 = centre (3), p2 = ref (3), t_lo, t_hi).  The exact code forms
// t = |(q - p1) x (q - p2)|^2 = |L|^2 dist(q, axis)^2, L = p2 - p1.  With two vectors E1, E2 of length |L|, orthogonal to
// L and to each other, t = (E1 . (q - c'))^2 + (E2 . (q - c'))^2 for ANY point c' of the axis: two plane evaluations.
// The record takes c' = the axis point nearest to the box centre o and moves everything there in fp64:
// d_i = E_i . (q - o) + D_i, D_i = E_i . (o - c').  The kernel forms d_i with three v_pk_fma each (as for a plane: at most
// five factors (1 + delta) per term, |d_i~ - d_i| <= gamma_5 M_i, M_i = |E_ix| ex + |E_iy| ey + |E_iz| ez + |D_i|) and
// t' = fma(d2, d2, fma(d1, d1, -mid32)): within 10.1 u (M_1^2 + M_2^2) + 2.02 u (2 M^2 + mid) + u mid of t - mid:
// E_t = 32 u M^2 + 4 u mid, M = max(M_1, M_2).  fp64 roundings: E_1, E_2 are orthonormal-times-|L| to 4e-16, which lets
// 4e-16 of the along-axis part of q - c' (<= W_l) into d_i: <= 1.6e-15 (|L| W_l)^2 on t; the exact code's own t (a cross
// product of two LONG differences: components within 8 * 2^-53 W_m^2, W_m = max(A + |p1|, A + |p2|), of real ones of
// size <= K = 2 |L| W_l) and the moved point (within 12 * 2^-53 W_m of the axis): <= 2.2e-14 K W_m^2.
// E_t += 1e-13 (|L| W_l)^2 + 5e-14 K W_m^2 + 1e-27 W_m^4.
// (First version: t = |L x (q - c')|^2 from the rounded differences, 15 packed instructions per two points and a bound
// of 200 u (|L| W_l)^2 -- 2.8 % of the pairs of the C3 fit went back to the exact code; this form: 11 and a sixth of it.)
// out = (E1x, E1y, E1z, D1, E2x, E2y, E2z, D2, mid, half^2, h, -)
// L = p2 - p1, Ln = |L|, E1 = |L| (L x a) / |L x a| with a = the coordinate axis L leans on least, E2 = L x E1 / |L|;
// *vn = |L x a| (0: no axis)
M3D_HD void cylinder_frame(const double* rec, double* L, double* Ln, double* E1, double* E2, double* vn) {
    L[0] = rec[3] - rec[0];
    L[1] = rec[4] - rec[1];
    L[2] = rec[5] - rec[2];
    const double L2 = (L[0] * L[0] + L[1] * L[1]) + L[2] * L[2];
    *Ln = sqrt(L2);
    const double ax = fabs(L[0]), ay = fabs(L[1]), az = fabs(L[2]);
    const int i0 = (ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2);
    const double a[3] = {i0 == 0 ? 1.0 : 0.0, i0 == 1 ? 1.0 : 0.0, i0 == 2 ? 1.0 : 0.0};
    const double v[3] = {L[1] * a[2] - L[2] * a[1], L[2] * a[0] - L[0] * a[2], L[0] * a[1] - L[1] * a[0]};
    *vn = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    for (int k = 0; k < 3; ++k) E1[k] = v[k] / *vn * *Ln;
    E2[0] = (L[1] * E1[2] - L[2] * E1[1]) / *Ln;
    E2[1] = (L[2] * E1[0] - L[0] * E1[2]) / *Ln;
    E2[2] = (L[0] * E1[1] - L[1] * E1[0]) / *Ln;
}
M3D_HD void cylinder_screen_record(const double* rec, const double* box, double max_abs, float* out) {
    const double lo = rec[6], hi = rec[7];
    double L[3], Ln, E1[3], E2[3], vn;
    cylinder_frame(rec, L, &Ln, E1, E2, &vn);
    const double L2 = Ln * Ln;
    // c' = p1 + ((o - p1) . L / |L|^2) L, relative to o
    const double s = (((box[0] - rec[0]) * L[0] + (box[1] - rec[1]) * L[1]) + (box[2] - rec[2]) * L[2]) / L2;
    const double c[3] = {(rec[0] + s * L[0]) - box[0], (rec[1] + s * L[1]) - box[1], (rec[2] + s * L[2]) - box[2]};
    const double D1 = -((E1[0] * c[0] + E1[1] * c[1]) + E1[2] * c[2]);
    const double D2 = -((E2[0] * c[0] + E2[1] * c[1]) + E2[2] * c[2]);
    const double M1 = ((fabs(E1[0]) * box[3] + fabs(E1[1]) * box[4]) + fabs(E1[2]) * box[5]) + fabs(D1);
    const double M2 = ((fabs(E2[0]) * box[3] + fabs(E2[1]) * box[4]) + fabs(E2[2]) * box[5]) + fabs(D2);
    const double M = fmax(M1, M2);
    const double Wl = fmax(fmax(box[3] + fabs(c[0]), box[4] + fabs(c[1])), box[5] + fabs(c[2]));
    const double Wa = max_abs + fmax(fmax(fabs(rec[0]), fabs(rec[1])), fabs(rec[2]));
    const double Wb = max_abs + fmax(fmax(fabs(rec[3]), fabs(rec[4])), fabs(rec[5]));
    const double Wm = fmax(Wa, Wb);
    const double G = Wm * Wm;
    const double LW = Ln * Wl;
    const double mid = 0.5 * lo + 0.5 * hi, half = 0.5 * hi - 0.5 * lo;
    bool ok = (lo <= hi) && (lo >= 0.0) && (hi < 1e17) && (L2 > 0.0) && (vn > 0.0) && (M < 1e8) && (LW < 1e9) &&
              (G < 1e15) && (fabs(s) < 1e15) && (half > 1e-15);
    // u = t' (within Et of t - mid, above), q = fma(u, u, -half^2): inside <=> q <= 0, decided when |q| >= h -- as for the
    // plane (one packed instruction for two points where `|u| - half` took two unpacked ones)
    const double Et = ((((32.0 * kU32 * (M * M) + 4.0 * kU32 * mid) + (1e-13 * (LW * LW) + 5e-14 * (2.0 * LW) * G + 1e-27 * (G * G))) +
                        1e-30 * (M + 1.0)) + 1e-15 * (mid + half)) / M3D_SCREEN_BOUND_DIVISOR;
    const double h = ((2.0 * half * Et + Et * Et) + 2.0 * kU32 * (half * half)) * 1.001;
    ok = ok && (half > Et);
    out[0] = ok ? (float)E1[0] : 0.0f;
    out[1] = ok ? (float)E1[1] : 0.0f;
    out[2] = ok ? (float)E1[2] : 0.0f;
    out[3] = ok ? (float)D1 : 0.0f;
    out[4] = ok ? (float)E2[0] : 0.0f;
    out[5] = ok ? (float)E2[1] : 0.0f;
    out[6] = ok ? (float)E2[2] : 0.0f;
    out[7] = ok ? (float)D2 : 0.0f;
    out[8] = ok ? (float)mid : 0.0f;
    out[9] = ok ? (float)(half * half) : 0.0f;
    out[10] = ok ? f32_round_up_pos(h > 1e-30 ? h : 1e-30) : f32_nan();
    out[11] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Records of the MFMA screen (score_mfma_k, m3d_score_mfma.hip).
//
// tools/ubench/valu_rates.hip (profiles/r04_ubench_valu_rates.txt) settled what an instruction costs on gfx950: every VOP3
// instruction -- packed fp32, fp64, v_alignbit, v_min3 -- occupies its SIMD for ~4.15 cycles per wave, a VOP2 v_mul_f32 2.3,
// and a v_mfma_f32_32x32x16_f16 issued between them ~10 while the matrix pipe works in the VALU's shadow.  The packed screen
// pays 3.6 VALU instructions per (point, hypothesis).  Here the matrix pipe delivers, per point and hypothesis, the two
// numbers u1 = T - S and u2 = T + S (S = alpha . x~ + D on the tile's fp32 offsets x~, D = the model's value at the box
// centre): inside <=> both positive <=> t = u1 u2 > 0 -- the VALU is left with one v_mul_f32, the shift that collects t's sign
// and half a v_min3 for "was any |t| too small to call".
//
// Operands.  fp16 carries 11 bits, so every coordinate X = x~ sigma and coefficient b is cut in two pieces (X = Xh + Xl,
// |Xl| <= 2^-11 |X|; the same for b) and X b ~ Xh bh + Xh bl + Xl bh: three K-slots per coordinate, nine + three for the
// constant (2048 against three pieces of (T -+ D) Sigma / 2048: 33 bits) = 12 of K = 16, ONE MFMA per column and 32 x 32
// block.  Products of fp16 numbers are exact in the pipe's fp32 accumulation.  Scaling by powers of two (exact): tile,
// sigma = 2^p with e_max sigma in (2^10, 2^11]; coefficients, 1024 alpha (unit normals: below 2^10.1; the pieces of alpha are
// then the hypothesis' own, whatever the tile); the pipe's unit is Sigma = 1024 sigma, the same for every hypothesis of a
// tile, and the constant's pieces are cut from (T -+ D) sigma / 2.  An fp16 piece below 2^-14 is subnormal and the pipe may
// flush it (tools/ubench/mfma_f16_error.hip): it meets a partner of at most 2^11.1 -- below 0.3 per K-slot, 4 in all.
// (First form of the round, in the history: q = T^2 - S^2 as ONE quadratic form over ten monomials, K = 32 -- 1.5 VALU
// instructions per pair instead of 2.5, but the expanded form carries terms of size (r + |D|)^2 for a result that matters at
// size T^2: its band was 35 x the packed screen's and 23 % of the C2 pairs went back to the exact code.  A linear form's error
// scales with r + |D|, like the packed screen's.)
//
// Bound.  u = 2^-24; r = sum |alpha_i| e_i; G = r + T + |D| >= the sum of |terms| of either column over the box.
//   * coefficients: 1024 alpha_i is rounded to fp32 before it is cut: u |X b| (1 u);
//   * pieces: X b - (Xh bh + Xh bl + Xl bh) = Xl bl + (rounding of Xl) b + X (rounding of bl): 3 * 2^-22 |X b| = 12 u |X b| (12.5 u);
//   * the pipe's accumulation of 16 exact products: measured <= 5.5 u sum|terms| for operands in the normal range, 9.2 with
//     subnormal pieces in a chain of two (profiles/r04_ubench_mfma_f16_error.txt); budgeted kMfmaAcc = 12 u;
//   E_p = 25.5 u G + 4 / Sigma bounds |u_i / Sigma - (T -+ S~)|, S~ = alpha . x~ + D in real arithmetic on the offsets the
//   kernel holds; S~ is within E_in = u r + 1e-15 M_g of the exact code's fp64 value s64 (the offsets' rounding; the fp64
//   roundings of the centre, D and s64 itself, as for plane_screen_record).  E = E_p + E_in.
//   t >= h > 0: the factors have one sign, and u1 + u2 = 2 T +- 2 E_p > 0 makes it the positive one; were |u1| <= E, |u2| would
//   be <= 2 T + 3 E and t <= E (2 T + 3 E): with h = E (2 T + 3 E) 1.001 both |u_i| > E, i.e. T - s64 > 0 and T + s64 > 0: inside.
//   t <= -h: opposite signs, and by the same argument the negative factor is below -E: |s64| > T.
//   (t itself is one fp32 product of the two: relative 2^-24, inside the 1.001.)
// A pair with min |t| < h over its 512 points is recounted by the exact code; a record that cannot be screened carries h = NaN.
// tests/test_gpu_mfma_screen.py measures |u_i / Sigma - (T -+ S~)| against E_p on the GPU (m3d_bench_mfma_probe).
// ------------------------------------------------------------------------------------------------
constexpr double kMfmaAcc = 12.0;      // accumulation error of one v_mfma_f32_32x32x16_f16, in u x sum |terms|
constexpr double kMfmaConstA = 2048.0; // A-side value of the constant's three K-slots
constexpr double kMfmaCoef = 1024.0;   // B-side scale of the plane's normal
constexpr int kMfmaNoTile = -100000;   // mfma_tile_exp: the tile cannot be screened
// p with e_max 2^p in (2^10, 2^11]; kMfmaNoTile for an empty / non-finite / astronomically large or small box
M3D_HD int mfma_tile_exp(const double* box) {
    const double e = fmax(fmax(box[3], box[4]), box[5]);
    if (!(e > 1e-30) || !(e < 1e30)) return kMfmaNoTile;
    int ex;
    (void)__builtin_frexp(e, &ex);   // e = f 2^ex, f in [0.5, 1)
    return 11 - ex;
}
// rec = the plane's scoring record (a, b, c, d, T), box = (centre, half extents), p = mfma_tile_exp(box).
// k[0] = (T - D) sigma / 2, k[1] = (T + D) sigma / 2: what the constants' pieces are cut from (column u1 = T - S takes -1024 alpha,
// column u2 = T + S takes +1024 alpha); *hs = the band on t = u1 u2 in the pipe's units, h Sigma^2 (NaN: not screened).
M3D_HD void plane_mfma_record(const double* rec, const double* box, double max_abs, int p, double* k, double* hs,
                              double* e_p = nullptr /* E_p, unscaled */) {
    const double a = rec[0], bb = rec[1], c = rec[2], d = rec[3], T = rec[4];
    const double D = ((a * box[0] + bb * box[1]) + c * box[2]) + d;
    const double sa = (fabs(a) + fabs(bb)) + fabs(c);
    const double r = (fabs(a) * box[3] + fabs(bb) * box[4]) + fabs(c) * box[5];
    const double Mg = sa * max_abs + fabs(d);
    const int pe = p == kMfmaNoTile ? 0 : p;
    const double sig = __builtin_ldexp(1.0, pe), Sg = kMfmaCoef * sig, iSg = __builtin_ldexp(1.0 / kMfmaCoef, -pe);
    const double G = (r + T) + fabs(D);
    const double Ep = ((13.5 + kMfmaAcc) * kU32 * G + 4.0 * iSg) / M3D_SCREEN_BOUND_DIVISOR;
    const double E = Ep + (kU32 * r + 1e-15 * Mg);
    const double h = (E * (2.0 * T + 3.0 * E)) * 1.001;
    const double hsc = (h * Sg) * Sg;
    const double k1 = (T - D) * (0.5 * sig), k2 = (T + D) * (0.5 * sig);
    // |1024 alpha| and the constants' first pieces must be fp16 numbers (65504); T > 2 E: something is left to decide
    const bool ok = (p != kMfmaNoTile) && (sa < 30.0) && (Mg < 1e18) && (max_abs < 1e18) && (T > 1e-15) && (T < 1e18) &&
                    (fabs(k1) < 6.0e4) && (fabs(k2) < 6.0e4) && (T > 2.0 * E) && (hsc < 1e37) && (hsc > 0.0);
    k[0] = ok ? k1 : 0.0;
    k[1] = ok ? k2 : 0.0;
    *hs = ok ? (double)f32_round_up_pos(hsc > 1e-30 ? hsc : 1e-30) : (double)f32_nan();
    if (e_p) *e_p = Ep;
}

// ------------------------------------------------------------------------------------------------
// fp32 records of the BOX tests (cull_tiles32_k).  The box tests only have to be conservative -- a tile is dropped
// when it provably holds no inlier of the exact fp64 test -- so they can run in fp32 (two hypotheses per packed
// instruction, no fp64 sqrt / divide for the cylinder) as long as every rounding goes into the margin.  Coordinates are
// taken relative to the centre O of the cloud's bounding box (R = largest |coordinate - O|): tile_boxes_k stores each
// box as fp32 (centre, half extents rounded OUTWARDS so that the fp32 box contains the fp64 one).
// A tile is dropped when the test value is NEGATIVE (its sign bit is what the kernel collects):
//   plane     (r + K) - |s|,  s = a bx + b by + c bz + d' (d' = the model's value at O), r = |a| hx + |b| hy + |c| hz,
//             K = the fp64 box test's cut-off + 10 u M' (M' = (|a| + |b| + |c|) R + |d'|: five roundings on s, four on r)
//   sphere    dmax2 - loM  or  hiM - dmin2,  loM = lo - E, hiM = hi + E, E = 40 u W^2, W = max |c - O| + R
//   cylinder  (sHiM + Rt) - dist  or  (dist + Rt) - sLoM,  dist = sqrt(d1^2 + d2^2) with the two plane values of
//             cylinder_screen_record taken at the box centre, Rt = |L| x the box's bounding radius, sHiM / sLoM =
//             sqrt(t_hi + slack) + E_d / sqrt(t_lo - slack) - E_d, E_d = 24 u M', slack = the exact code's own fp64 rounding.
// "No inlier" records drop every tile (K = -inf ...); records with non-finite or huge values keep every tile.
// Records are stored PAIRWISE interleaved (hypotheses 2 j and 2 j + 1: v0 v0' v1 v1' ...) so that one scalar load
// delivers the operand pairs of the packed instructions.
// ------------------------------------------------------------------------------------------------
M3D_HD float f32_inf() {
    const uint32_t b = 0x7F800000u;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}
M3D_HD float f32_round_down(double v) {   // largest float <= v (finite v)
    float f = (float)v;
    if ((double)f > v) {
        uint32_t b;
        __builtin_memcpy(&b, &f, 4);
        if (f > 0.0f) --b;
        else if (f < 0.0f) ++b;
        else b = 0x80000001u;   // below +0: the smallest negative denormal
        __builtin_memcpy(&f, &b, 4);
    }
    return f;
}
M3D_HD float f32_round_up(double v) { return -f32_round_down(-v); }
// plane: (a, b, c, d', |a|, |b|, |c|, K)
M3D_HD void plane_cull32_record(const double* rec, bool valid, const double* o, double radius, double max_abs, float* out) {
    for (int k = 0; k < 12; ++k) out[k] = 0.0f;
    const double a = rec[0], b = rec[1], c = rec[2], d = rec[3], T = rec[4], cutB = rec[5];
    if (!valid || !(T > 0.0)) {
        out[7] = -f32_inf();
        return;
    }
    const double dp = ((a * o[0] + b * o[1]) + c * o[2]) + d;
    const double sa = (fabs(a) + fabs(b)) + fabs(c);
    const double Mp = sa * radius + fabs(dp), Mg = sa * max_abs + fabs(d);
    const double K = cutB + ((10.0 * kU32 * Mp + 1e-15 * Mg) + 1e-36 * (sa + 1.0));
    if (!(Mp < 1e18) || !(sa < 1e18) || !(radius < 1e18) || !(K < 1e30)) {   // (also NaN)
        out[7] = f32_inf();
        return;
    }
    out[0] = (float)a;
    out[1] = (float)b;
    out[2] = (float)c;
    out[3] = (float)dp;
    out[4] = fabsf(out[0]);
    out[5] = fabsf(out[1]);
    out[6] = fabsf(out[2]);
    out[7] = f32_round_up(K);
}
// sphere: (c'x, c'y, c'z, loM, hiM)
M3D_HD void sphere_cull32_record(const double* rec, bool valid, const double* o, double radius, double max_abs, float* out) {
    for (int k = 0; k < 12; ++k) out[k] = 0.0f;
    const double lo = rec[3], hi = rec[4];
    if (!valid || !(lo <= hi)) {
        out[3] = f32_inf();
        out[4] = -f32_inf();
        return;
    }
    const double c[3] = {rec[0] - o[0], rec[1] - o[1], rec[2] - o[2]};
    const double W = fmax(fmax(fabs(c[0]), fabs(c[1])), fabs(c[2])) + radius;
    const double Wg = max_abs + fmax(fmax(fabs(rec[0]), fabs(rec[1])), fabs(rec[2]));
    const double E = 40.0 * kU32 * (W * W) + 1e-14 * (W * Wg);
    if (!(W < 1e18) || !(Wg < 1e18) || !(hi < 1e36)) {
        out[3] = -f32_inf();
        out[4] = f32_inf();
        return;
    }
    out[0] = (float)c[0];
    out[1] = (float)c[1];
    out[2] = (float)c[2];
    out[3] = f32_round_down(lo - E);
    out[4] = f32_round_up(hi + E);
}
// cylinder: (E1x, E1y, E1z, D1', E2x, E2y, E2z, D2', Ln, sHiM, sLoM)
M3D_HD void cylinder_cull32_record(const double* rec, bool valid, const double* o, double radius, double max_abs, float* out) {
    for (int k = 0; k < 12; ++k) out[k] = 0.0f;
    const double lo = rec[6], hi = rec[7];
    if (!valid || !(lo <= hi)) {
        out[9] = -f32_inf();
        out[10] = -f32_inf();
        return;
    }
    double L[3], Ln, E1[3], E2[3], vn;
    cylinder_frame(rec, L, &Ln, E1, E2, &vn);
    const double q[3] = {o[0] - rec[0], o[1] - rec[1], o[2] - rec[2]};
    const double D1 = (E1[0] * q[0] + E1[1] * q[1]) + E1[2] * q[2];
    const double D2 = (E2[0] * q[0] + E2[1] * q[1]) + E2[2] * q[2];
    const double M1 = ((fabs(E1[0]) + fabs(E1[1])) + fabs(E1[2])) * radius + fabs(D1);
    const double M2 = ((fabs(E2[0]) + fabs(E2[1])) + fabs(E2[2])) * radius + fabs(D2);
    const double M = fmax(M1, M2);
    const double Wa = max_abs + fmax(fmax(fabs(rec[0]), fabs(rec[1])), fabs(rec[2]));
    const double Wb = max_abs + fmax(fmax(fabs(rec[3]), fabs(rec[4])), fabs(rec[5]));
    const double Wm = fmax(Wa, Wb), G = Wm * Wm;
    const double slack = 5e-14 * (3.0 * M) * G + 1e-27 * (G * G);
    const double Ed = 24.0 * kU32 * M + 1e-14 * (Ln * Wm);
    const bool ok = (Ln > 0.0) && (vn > 0.0) && (M < 1e15) && (G < 1e15) && (hi < 1e30) && (radius < 1e18) && (Ln < 1e15);
    if (!ok) {
        out[9] = f32_inf();
        out[10] = -f32_inf();
        return;
    }
    const double tlo = lo - slack > 0.0 ? lo - slack : 0.0;
    for (int k = 0; k < 3; ++k) {
        out[k] = (float)E1[k];
        out[4 + k] = (float)E2[k];
    }
    out[3] = (float)D1;
    out[7] = (float)D2;
    out[8] = f32_round_up(Ln * (1.0 + 1e-6));
    out[9] = f32_round_up(sqrt(hi + slack) * (1.0 + 1e-12) + Ed);
    out[10] = f32_round_down(sqrt(tlo) * (1.0 - 1e-12) - Ed);
}

}  // namespace m3d
