/* misc3d_oracle_boundary.c -- CPU restatement of misc3d::features::DetectBoundaryPoints
 * (src/boundary_detection.cpp:14-113; python features.detect_boundary_points, python/py_features.cpp:11-20).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference tests; Open3D / Eigen absent).
 * Restated from the reference: the neighbourhood loop (:92-110: skip a point with fewer than 3 neighbours),
 * the tangent basis (:21-25), the angle criterion (:27-66: skip coincident neighbours, atan2(v.delta, u.delta),
 * sort, largest gap incl. the wrap-around, threshold in degrees).  [RECALL] third-party pieces: Open3D
 * KDTreeFlann::Search with KDTreeSearchParamRadius / Hybrid (brute force here: all points with squared
 * distance <= r^2 for Radius; the max_nn nearest of those with distance < r... see below), Eigen's generic
 * unitOrthogonal for a 4-vector, and -- when the cloud has no normals -- Open3D's EstimateNormals, which is
 * replaced by the covariance of the same neighbourhood + the J3x3 eigen-solver (the boundary decision only
 * depends on the normal DIRECTION, and not on its sign: the angular gaps are invariant under a change of
 * the in-plane basis).  Canonical choices: neighbours ordered by (squared distance, index); output indices
 * ascending (the reference's order depends on thread timing). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void orc_j3x3_smallest_eigvec(const double *Ain, double *n);

typedef struct {
    double d2;
    int64_t idx;
} nb_t;

static int nb_cmp(const void *a, const void *b) {
    const nb_t *x = (const nb_t *)a, *y = (const nb_t *)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static int dbl_cmp(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* Eigen generic unitOrthogonal on (n0, n1, n2, 0) + cross3: v = unitOrthogonal(n), u = n x v ([RECALL]) */
static void tangent_basis(const double *n, double *u, double *v) {
    const double a[4] = {fabs(n[0]), fabs(n[1]), fabs(n[2]), 0.0};
    int maxi = 0;
    for (int i = 1; i < 4; ++i)
        if (a[i] > a[maxi]) maxi = i;
    int sndi = maxi == 0 ? 1 : 0;
    for (int i = 0; i < 4; ++i)
        if (i != maxi && a[i] > a[sndi]) sndi = i;
    const double src[4] = {n[0], n[1], n[2], 0.0};
    const double invnm = 1.0 / sqrt(src[sndi] * src[sndi] + src[maxi] * src[maxi]);
    double p[4] = {0, 0, 0, 0};
    p[maxi] = -src[sndi] * invnm;
    p[sndi] = src[maxi] * invnm;
    v[0] = p[0];
    v[1] = p[1];
    v[2] = p[2];
    u[0] = n[1] * v[2] - n[2] * v[1];
    u[1] = n[2] * v[0] - n[0] * v[2];
    u[2] = n[0] * v[1] - n[1] * v[0];
}

/* search: 0 = KNN (the max_nn nearest, no radius), 1 = Radius (all points with d2 <= r^2), 2 = Hybrid (the max_nn
 * nearest among those with d2 < r^2).
 * normals may be NULL.  out: ascending indices of the boundary points.  Returns their number. */
size_t orc_detect_boundary_points(const double *xyz, const double *normals, size_t n, int search, double radius,
                                  int max_nn, double angle_threshold_deg, int64_t *out) {
    const double r2 = radius * radius;
    nb_t *nb = (nb_t *)malloc(sizeof(nb_t) * (n ? n : 1));
    double *ang = (double *)malloc(sizeof(double) * (n ? n : 1));
    size_t k_out = 0;
    for (size_t i = 0; i < n; ++i) {
        const double *q = xyz + 3 * i;
        size_t m = 0;
        for (size_t j = 0; j < n; ++j) {
            const double dx = q[0] - xyz[3 * j], dy = q[1] - xyz[3 * j + 1], dz = q[2] - xyz[3 * j + 2];
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            if (search == 0 ? 1 : (search == 1 ? d2 <= r2 : d2 < r2)) {
                nb[m].d2 = d2;
                nb[m].idx = (int64_t)j;
                ++m;
            }
        }
        qsort(nb, m, sizeof(nb_t), nb_cmp);
        if (search != 1 && m > (size_t)max_nn) m = (size_t)max_nn;
        if (m < 3) continue;
        double nrm[3];
        if (normals) {
            memcpy(nrm, normals + 3 * i, sizeof(nrm));
        } else {   /* covariance of the neighbourhood (cumulant form), smallest eigenvector */
            double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t t = 0; t < m; ++t) {
                const double *p = xyz + 3 * nb[t].idx;
                s[0] += p[0];
                s[1] += p[1];
                s[2] += p[2];
                s[3] += p[0] * p[0];
                s[4] += p[0] * p[1];
                s[5] += p[0] * p[2];
                s[6] += p[1] * p[1];
                s[7] += p[1] * p[2];
                s[8] += p[2] * p[2];
            }
            const double inv = 1.0 / (double)m;
            for (int t = 0; t < 9; ++t) s[t] *= inv;
            double Cm[9];
            Cm[0] = s[3] - s[0] * s[0];
            Cm[1] = s[4] - s[0] * s[1];
            Cm[2] = s[5] - s[0] * s[2];
            Cm[4] = s[6] - s[1] * s[1];
            Cm[5] = s[7] - s[1] * s[2];
            Cm[8] = s[8] - s[2] * s[2];
            Cm[3] = Cm[1];
            Cm[6] = Cm[2];
            Cm[7] = Cm[5];
            orc_j3x3_smallest_eigvec(Cm, nrm);
        }
        double u[3], v[3];
        tangent_basis(nrm, u, v);
        size_t na = 0;
        for (size_t t = 0; t < m; ++t) {
            const double *p = xyz + 3 * nb[t].idx;
            const double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
            if (d[0] == 0.0 && d[1] == 0.0 && d[2] == 0.0) continue;   /* :36-38 */
            ang[na++] = atan2((v[0] * d[0] + v[1] * d[1]) + v[2] * d[2], (u[0] * d[0] + u[1] * d[1]) + u[2] * d[2]);
        }
        if (na == 0) continue;
        qsort(ang, na, sizeof(double), dbl_cmp);
        double max_dif = 0.0;
        for (size_t t = 0; t + 1 < na; ++t) {
            const double dif = ang[t + 1] - ang[t];
            if (max_dif < dif) max_dif = dif;
        }
        const double wrap = 2 * M_PI - ang[na - 1] + ang[0];
        if (max_dif < wrap) max_dif = wrap;
        if (max_dif > angle_threshold_deg * M_PI / 180.0) out[k_out++] = (int64_t)i;
    }
    free(nb);
    free(ang);
    return k_out;
}
