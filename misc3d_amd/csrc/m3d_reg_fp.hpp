// m3d_reg_fp.hpp -- arithmetic of the correspondence-registration path, host + device.
//
// What it replaces (third-party code behind src/transform_estimation.cpp:59-66,124-164):
//   Eigen::umeyama                         -> umeyama3 / rotation_from_covariance ("K3x3")
//   Open3D CorrespondenceCheckerBasedOnEdgeLength / ...BasedOnDistance -> reg_checkers
//   Open3D PointCloud::Transform           -> transform_point
// Operation order is fixed (sums in index order, 3-element reductions (e0+e1)+e2, no FMA) and is
// the same specification the CPU oracle follows, so hypotheses agree bit for bit.
// Compile with -ffp-contract=off.
#pragma once
#include "m3d_fp.hpp"

#pragma clang fp contract(off)

namespace m3d {

// "K3x3": rotation of umeyama from the 3x3 covariance sigma (row-major).
//   One-sided (Hestenes) Jacobi on the columns of A = sigma, V accumulates the rotations;
//   pairs (0,1),(0,2),(1,2); <= 30 sweeps; a pair is skipped when gamma == 0 or
//   |gamma| <= 2^-52 sqrt(alpha beta).  Singular values = column norms sorted descending (stable),
//   u1,u2 = columns / sigma, u3 = u1 x u2, R = u1 v1^T + u2 v2^T + sign(det V) u3 v3^T
//   (= U diag(1,1,sign(det U det V)) V^T of Eigen::umeyama, also for rank-2 input).
//   sv[2] carries the sign S of umeyama's S vector (used by the scaling factor).
M3D_HD void rotation_from_covariance(const double* sigma, double* R, double* sv) {
    double a[3][3], v[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            a[r][c] = sigma[3 * r + c];
            v[r][c] = (r == c) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = (k == 2) ? 1 : 0;
            const int q = (k == 0) ? 1 : 2;
            const double alpha = (a[0][p] * a[0][p] + a[1][p] * a[1][p]) + a[2][p] * a[2][p];
            const double beta = (a[0][q] * a[0][q] + a[1][q] * a[1][q]) + a[2][q] * a[2][q];
            const double gamma = (a[0][p] * a[0][q] + a[1][p] * a[1][q]) + a[2][p] * a[2][q];
            if (gamma == 0.0) continue;
            if (fabs(gamma) <= 2.220446049250313e-16 * sqrt(alpha * beta)) continue;
            rotated = true;
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t);
            const double s = c * t;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double ap = a[r][p], aq = a[r][q];
                a[r][p] = c * ap - s * aq;
                a[r][q] = s * ap + c * aq;
                const double vp = v[r][p], vq = v[r][q];
                v[r][p] = c * vp - s * vq;
                v[r][q] = s * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    double sg[3];
    for (int c = 0; c < 3; ++c) sg[c] = sqrt((a[0][c] * a[0][c] + a[1][c] * a[1][c]) + a[2][c] * a[2][c]);
    int o0 = 0, o1 = 1, o2 = 2;  // stable insertion sort, descending
    if (sg[o1] > sg[o0]) {
        const int t = o0;
        o0 = o1;
        o1 = t;
    }
    if (sg[o2] > sg[o1]) {
        const int t = o1;
        o1 = o2;
        o2 = t;
        if (sg[o1] > sg[o0]) {
            const int t2 = o0;
            o0 = o1;
            o1 = t2;
        }
    }
    // column extraction without dynamic register indexing
    auto col = [&](const double (*m)[3], int c, double* out) {
        for (int r = 0; r < 3; ++r) out[r] = c == 0 ? m[r][0] : (c == 1 ? m[r][1] : m[r][2]);
    };
    const double s1 = o0 == 0 ? sg[0] : (o0 == 1 ? sg[1] : sg[2]);
    const double s2 = o1 == 0 ? sg[0] : (o1 == 1 ? sg[1] : sg[2]);
    if (!(s1 > 0.0)) {  // sigma == 0: rotation undefined -> identity
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[3 * r + c] = (r == c) ? 1.0 : 0.0;
        sv[0] = sv[1] = sv[2] = 0.0;
        return;
    }
    double a1[3], a2[3], a3[3], v1[3], v2[3], v3[3];
    col(a, o0, a1);
    col(a, o1, a2);
    col(a, o2, a3);
    col(v, o0, v1);
    col(v, o1, v2);
    col(v, o2, v3);
    double u1[3], u2[3], u3[3];
    for (int r = 0; r < 3; ++r) u1[r] = a1[r] / s1;
    if (s2 > 1e-300 && s2 > 2.220446049250313e-16 * s1) {
        for (int r = 0; r < 3; ++r) u2[r] = a2[r] / s2;
    } else {  // rank 1: deterministic completion
        int kmin = 0;
        if (fabs(u1[1]) < fabs(u1[kmin])) kmin = 1;
        if (fabs(u1[2]) < fabs(u1[kmin])) kmin = 2;
        const double pr = kmin == 0 ? u1[0] : (kmin == 1 ? u1[1] : u1[2]);
        double w[3];
        for (int r = 0; r < 3; ++r) w[r] = (r == kmin ? 1.0 : 0.0) - pr * u1[r];
        const double nw = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
        for (int r = 0; r < 3; ++r) u2[r] = w[r] / nw;
    }
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    // determinant of the SORTED V = [v1 v2 v3] (columns)
    const double detv = (v1[0] * (v2[1] * v3[2] - v3[1] * v2[2]) - v2[0] * (v1[1] * v3[2] - v3[1] * v1[2])) +
                        v3[0] * (v1[1] * v2[2] - v2[1] * v1[2]);
    const double sgn = detv < 0.0 ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = (u1[r] * v1[c] + u2[r] * v2[c]) + (sgn * u3[r]) * v3[c];
    sv[0] = s1;
    sv[1] = s2;
    sv[2] = sgn * ((u3[0] * a3[0] + u3[1] * a3[1]) + u3[2] * a3[2]);
}

// T (row-major 4x4) from means, covariance (already scaled by 1/n) and source variance.
M3D_HD void umeyama_assemble(const double* ms, const double* md, const double* sigma, double src_var,
                             bool with_scaling, double* T) {
    double R[9], sv[3];
    rotation_from_covariance(sigma, R, sv);
    double c = 1.0;
    if (with_scaling) c = 1.0 / src_var * ((sv[0] + sv[1]) + sv[2]);
    for (int r = 0; r < 3; ++r) {
        const double rm = (R[3 * r] * ms[0] + R[3 * r + 1] * ms[1]) + R[3 * r + 2] * ms[2];
        T[4 * r + 3] = md[r] - c * rm;
        for (int cc = 0; cc < 3; ++cc) T[4 * r + cc] = c * R[3 * r + cc];
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

// Eigen::umeyama on three correspondences, sums in index order (Open3D
// TransformationEstimationPointToPoint::ComputeTransformation with ransac_n = 3).
M3D_HD void umeyama3(const double* ps, const double* pd, double* T) {
    const double one_over_n = 1.0 / 3.0;
    double ms[3], md[3];
    for (int k = 0; k < 3; ++k) {
        ms[k] = ((0.0 + ps[k]) + ps[3 + k]) + ps[6 + k];
        md[k] = ((0.0 + pd[k]) + pd[3 + k]) + pd[6 + k];
        ms[k] *= one_over_n;
        md[k] *= one_over_n;
    }
    double sig[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double var[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        double s[3], d[3];
        for (int k = 0; k < 3; ++k) {
            s[k] = ps[3 * i + k] - ms[k];
            d[k] = pd[3 * i + k] - md[k];
        }
        for (int r = 0; r < 3; ++r) {
            var[r] += s[r] * s[r];
            for (int c = 0; c < 3; ++c) sig[3 * r + c] += d[r] * s[c];
        }
    }
    for (int k = 0; k < 9; ++k) sig[k] *= one_over_n;
    const double src_var = ((var[0] + var[1]) + var[2]) * one_over_n;
    umeyama_assemble(ms, md, sig, src_var, false, T);
}

// (T * (x,y,z,1)).head<3>(): column-by-column accumulation
M3D_HD void transform_point(const double* T, double x, double y, double z, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
}

// EdgeLength (all pairs) then Distance checker on the 3 sampled correspondences
M3D_HD bool reg_checkers(const double* ps, const double* pd, const double* T, double edge_thr,
                         double dist_thr) {
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) {
            const double ds = norm3(ps[3 * i] - ps[3 * j], ps[3 * i + 1] - ps[3 * j + 1],
                                    ps[3 * i + 2] - ps[3 * j + 2]);
            const double dt = norm3(pd[3 * i] - pd[3 * j], pd[3 * i + 1] - pd[3 * j + 1],
                                    pd[3 * i + 2] - pd[3 * j + 2]);
            if (ds < dt * edge_thr || dt < ds * edge_thr) return false;
        }
    for (int i = 0; i < 3; ++i) {
        double pt[3];
        transform_point(T, ps[3 * i], ps[3 * i + 1], ps[3 * i + 2], pt);
        if (norm3(pd[3 * i] - pt[0], pd[3 * i + 1] - pt[1], pd[3 * i + 2] - pt[2]) > dist_thr) return false;
    }
    return true;
}

}  // namespace m3d
