/* misc3d_oracle_normals.c -- CPU restatement of misc3d::common::EstimateNormalsFromMap
 * (src/normal_estimation.cpp:36-207; python: common.estimate_normals, python/py_common.cpp:79-89).
 *
 * TEST INFRASTRUCTURE ONLY (see misc3d_oracle.c): the product never links or calls this.
 * PARITY UNPINNED: the reference holds no tests or vectors for this function and cannot be built here
 * (Eigen / Open3D absent).  What is restated exactly: the padded moment images, the order of the
 * sliding-window box sums (SumDense, :36-62: first window row by row, then per column the running value
 * plus the (2k+1) per-row differences, in that order), the covariance expression (:143-154) and the
 * orientation rule (:165-171).  What cannot be restated bit for bit: Eigen::SelfAdjointEigenSolver::compute
 * (iterative QL on the tridiagonalised matrix); it is replaced on BOTH sides (oracle and product) by one
 * fully specified algorithm "J3x3" -- cyclic Jacobi, pairs (0,1),(0,2),(1,2), at most 24 sweeps, stop when
 * all three off-diagonal entries are exactly zero; smallest eigenvalue (lowest index on ties); eigenvector
 * normalised; the reference leaves the normals of invalid (z != z) pixels uninitialised -- here they are NaN. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* A: symmetric 3x3 (row-major, 9 entries).  n: unit eigenvector of the smallest eigenvalue. */
void orc_j3x3_smallest_eigvec(const double *Ain, double *n) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(A, Ain, sizeof(A));
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 24; ++sweep) {
        if (A[1] == 0.0 && A[2] == 0.0 && A[5] == 0.0) break;
        for (int e = 0; e < 3; ++e) {
            const int p = P[e], q = Q[e];
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            /* A <- J^T A J with J = rotation in the (p,q) plane */
            for (int k = 0; k < 3; ++k) {   /* columns p, q */
                const double akp = A[3 * k + p], akq = A[3 * k + q];
                A[3 * k + p] = c * akp - s * akq;
                A[3 * k + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {   /* rows p, q */
                const double apk = A[3 * p + k], aqk = A[3 * q + k];
                A[3 * p + k] = c * apk - s * aqk;
                A[3 * q + k] = s * apk + c * aqk;
            }
            A[3 * p + q] = 0.0;
            A[3 * q + p] = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                V[3 * k + p] = c * vkp - s * vkq;
                V[3 * k + q] = s * vkp + c * vkq;
            }
        }
    }
    int m = 0;
    if (A[4] < A[3 * m + m]) m = 1;
    if (A[8] < A[3 * m + m]) m = 2;
    const double x = V[m], y = V[3 + m], z = V[6 + m];
    const double nrm = sqrt((x * x + y * y) + z * z);
    n[0] = x / nrm;
    n[1] = y / nrm;
    n[2] = z / nrm;
}

/* sliding-window sums of one padded image (W x H, window (2k+1)^2), exactly in SumDense's order */
static void sum_dense(const double *data, size_t W, size_t H, size_t k, double *dst) {
    for (size_t r = k; r < H - k; ++r) {
        double *ptr = dst + r * W + k;
        double acc = 0.0;
        for (size_t r0 = r - k; r0 <= r + k; ++r0)
            for (size_t c0 = 0; c0 <= 2 * k; ++c0) acc += data[r0 * W + c0];
        *ptr = acc;
        for (size_t c = k + 1; c < W - k; ++c) {
            ++ptr;
            acc = *(ptr - 1);
            for (size_t r0 = r - k; r0 <= r + k; ++r0) acc += data[r0 * W + c + k] - data[r0 * W + c - k - 1];
            *ptr = acc;
        }
    }
}

/* xyz: h x w x 3 (row-major point map), normals: h x w x 3 out.  Returns 0. */
int orc_normals_from_map(const double *xyz, unsigned w, unsigned h, unsigned k, const double *view_point,
                         double *normals) {
    const size_t W = (size_t)w + 2 * k, H = (size_t)h + 2 * k, WH = W * H;
    double *img = (double *)calloc(WH * 10, sizeof(double));   /* x y z xx xy xz yy yz zz mask */
    double *sum = (double *)calloc(WH * 10, sizeof(double));
    for (size_t r = 0; r < h; ++r)
        for (size_t c = 0; c < w; ++c) {
            const double *p = xyz + (r * w + c) * 3;
            const size_t idx = (r + k) * W + (c + k);
            if (p[2] == p[2]) {
                img[9 * WH + idx] = 1.0;
                img[idx] = p[0];
                img[WH + idx] = p[1];
                img[2 * WH + idx] = p[2];
                img[3 * WH + idx] = p[0] * p[0];
                img[4 * WH + idx] = p[0] * p[1];
                img[5 * WH + idx] = p[0] * p[2];
                img[6 * WH + idx] = p[1] * p[1];
                img[7 * WH + idx] = p[1] * p[2];
                img[8 * WH + idx] = p[2] * p[2];
            }
        }
    for (int a = 0; a < 10; ++a) sum_dense(img + a * WH, W, H, k, sum + a * WH);   /* mask sums are exact integers */
    const double nan = NAN;
    for (size_t r = 0; r < h; ++r)
        for (size_t c = 0; c < w; ++c) {
            const size_t idx = (r + k) * W + (c + k);
            double *out = normals + (r * w + c) * 3;
            if (img[9 * WH + idx] == 0.0) {
                out[0] = out[1] = out[2] = nan;
                continue;
            }
            const double scale = 1.0 / sum[9 * WH + idx];
            const double hx = sum[idx] * scale, hy = sum[WH + idx] * scale, hz = sum[2 * WH + idx] * scale;
            double C[9];
            C[0] = sum[3 * WH + idx] * scale - hx * hx;
            C[1] = sum[4 * WH + idx] * scale - hx * hy;
            C[2] = sum[5 * WH + idx] * scale - hx * hz;
            C[4] = sum[6 * WH + idx] * scale - hy * hy;
            C[5] = sum[7 * WH + idx] * scale - hy * hz;
            C[8] = sum[8 * WH + idx] * scale - hz * hz;
            C[3] = C[1];
            C[6] = C[2];
            C[7] = C[5];
            double n[3];
            orc_j3x3_smallest_eigvec(C, n);
            const double d = ((view_point[0] - img[idx]) * n[0] + (view_point[1] - img[WH + idx]) * n[1]) +
                             (view_point[2] - img[2 * WH + idx]) * n[2];
            if (d < 0) {
                n[0] *= -1;
                n[1] *= -1;
                n[2] *= -1;
            }
            out[0] = n[0];
            out[1] = n[1];
            out[2] = n[2];
        }
    free(img);
    free(sum);
    return 0;
}
