/*
 * misc3d_oracle.c -- CPU restatement of the Misc3D RANSAC hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the ORACLE for the MI355X implementation in misc3d_amd/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Nothing under misc3d_amd/
 * links, imports or calls it; the product path has no CPU fallback.
 *
 * PARITY UNPINNED: the reference (yuecideng/Misc3D) has no tests and no golden vectors for this
 * path and cannot be compiled in this image (it needs Eigen + Open3D 0.15.1, both
 * absent).  This restatement is therefore pinned only by (a) known-answer tests with
 * analytic results (tests/test_oracle_*.py), (b) libstdc++'s std::mt19937 /
 * std::uniform_int_distribution compiled here (oracle/std_rng_check.cpp), (c) numpy/scipy
 * cross-checks of the linear algebra.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Canonical semantics: the reference is non-deterministic (random_device seed,
 * OpenMP race on the early-exit test).  The oracle restates its OMP_NUM_THREADS=1 behaviour with an
 * explicit std::mt19937 seed: hypotheses are visited in index order, each consumes sampler draws in
 * order, the best-update / adaptive-stop rule of ransac.h:592-613 is applied sequentially.
 *
 * Floating-point operation order ([RECALL] of Eigen >=3.3 with SSE2 packets,
 * no FMA because the reference builds with plain -O3, CMakeLists.txt:16):
 *   (i)  4-element reductions      (e0 + e2) + (e1 + e3)
 *   (ii) 3-element reductions      (e0 + e1) + e2
 *   (iii) cross(a,b) = (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
 *   (iv) Matrix4d::determinant = Eigen 3.3 "bruteforce_det4_helper" (Costabel, 30 multiplies)
 * Build with -ffp-contract=off and without -march so that no FMA is ever formed.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EPS 1.0e-8 /* include/misc3d/common/ransac.h:14 */

enum { ORC_PLANE = 0, ORC_SPHERE = 1, ORC_CYLINDER = 2 };

/* ------------------------------------------------------------------------------------------- */
/* std::mt19937 (the generator of include/misc3d/utils.h:73-77,119)                             */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t mt[624];
    int idx;
} orc_mt19937;

void orc_mt_seed(orc_mt19937 *g, uint64_t seed) {
    /* std::mersenne_twister_engine::seed(value): x0 = value mod 2^32 */
    g->mt[0] = (uint32_t)(seed & 0xffffffffu);
    for (int i = 1; i < 624; ++i) {
        uint32_t prev = g->mt[i - 1];
        g->mt[i] = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
    }
    g->idx = 624;
}

uint32_t orc_mt_next(orc_mt19937 *g) {
    if (g->idx >= 624) {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
            uint32_t v = g->mt[(k + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            g->mt[k] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* RandomSampler<size_t>::operator() -- include/misc3d/utils.h:81-97.
 * idx = rng() % size (biased modulo kept as is); duplicates inside the sample are rejected. */
void orc_sample(orc_mt19937 *g, size_t size, int m, size_t *out) {
    int valid = 0;
    while (valid < m) {
        size_t idx = (size_t)orc_mt_next(g) % size;
        int dup = 0;
        for (int k = 0; k < valid; ++k)
            if (out[k] == idx) dup = 1;
        if (!dup) out[valid++] = idx;
    }
}

/* libstdc++ (GCC >= 11) std::uniform_int_distribution<int>(0, range) on a 32-bit URBG: Lemire's
 * nearly-divisionless method.  Used for the correspondence sampler of the registration RANSAC
 * (Open3D utility::UniformRandIntGenerator, [RECALL]).  Pinned by oracle/std_rng_check.cpp. */
uint32_t orc_uniform_int(orc_mt19937 *g, uint32_t range_incl) {
    if (range_incl == 0xffffffffu) return orc_mt_next(g);
    uint32_t r = range_incl + 1u;
    uint64_t product = (uint64_t)orc_mt_next(g) * (uint64_t)r;
    uint32_t low = (uint32_t)product;
    if (low < r) {
        uint32_t threshold = (uint32_t)(-r) % r;
        while (low < threshold) {
            product = (uint64_t)orc_mt_next(g) * (uint64_t)r;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}

/* ------------------------------------------------------------------------------------------- */
/* Eigen reduction orders                                                                       */
/* ------------------------------------------------------------------------------------------- */
/* ORC_FP_ORDER: which Eigen the association is restated for (the product has the same switch, M3D_FP_ORDER in
 * misc3d_amd/csrc/m3d_fp.hpp, where the three variants are described): 0 = Eigen >= 3.3 (default), 1 = Eigen 3.2
 * (3-element reductions e0 + (e1 + e2)), 2 = Eigen 3.4 (its 4x4 determinant).  `make -C oracle orders` builds
 * _build/order1/ and _build/order2/. */
#ifndef ORC_FP_ORDER
#define ORC_FP_ORDER 0
#endif
int orc_fp_order(void) { return ORC_FP_ORDER; }
static inline double sum3(double e0, double e1, double e2) {
#if ORC_FP_ORDER == 1
    return e0 + (e1 + e2);
#else
    return (e0 + e1) + e2; /* (ii) */
#endif
}
static inline double dot3(const double *a, const double *b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
static inline double norm3(const double *a) { return sqrt(dot3(a, a)); }
static inline double dot4(const double *a, const double *b) {
    return (a[0] * b[0] + a[2] * b[2]) + (a[1] * b[1] + a[3] * b[3]); /* (i) */
}
static inline void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* ------------------------------------------------------------------------------------------- */
/* Plane -- include/misc3d/common/ransac.h:134-221                                             */
/* ------------------------------------------------------------------------------------------- */

/* PlaneEstimator::MinimalFit, ransac.h:138-162.  p = 3 points (9 doubles). */
int orc_plane_minimal_fit(const double *p, double *out) {
    double e0[3], e1[3], abc[3];
    for (int k = 0; k < 3; ++k) {
        e0[k] = p[3 + k] - p[k];
        e1[k] = p[6 + k] - p[k];
    }
    cross3(e0, e1, abc);
    const double norm = norm3(abc);
    if (norm < ORC_EPS) return 0; /* ransac.h:151 */
    const double n2 = norm3(abc); /* ransac.h:154 recomputes the norm */
    abc[0] /= n2;
    abc[1] /= n2;
    abc[2] /= n2;
    out[0] = abc[0];
    out[1] = abc[1];
    out[2] = abc[2];
    out[3] = -dot3(abc, p); /* ransac.h:155 */
    return 1;
}

/* PlaneEstimator::CalcPointToModelDistance, ransac.h:215-220 */
double orc_plane_distance(const double *q, const double *m) {
    const double p4[4] = {q[0], q[1], q[2], 1.0};
    return fabs(dot4(m, p4)) / norm3(m);
}

/* PlaneEstimator::GeneralFit, ransac.h:164-213 (single OpenMP thread => serial sums).
 * pts = n x 3 AoS.  Leaves out[] untouched on failure, like the reference. */
int orc_plane_general_fit(const double *pts, size_t n, double *out) {
    if (n < 3) return 0;
    double mean[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        mean[0] += pts[3 * i];
        mean[1] += pts[3 * i + 1];
        mean[2] += pts[3 * i + 2];
    }
    mean[0] /= (double)n;
    mean[1] /= (double)n;
    mean[2] /= (double)n;
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (size_t i = 0; i < n; ++i) {
        const double r0 = pts[3 * i] - mean[0], r1 = pts[3 * i + 1] - mean[1],
                     r2 = pts[3 * i + 2] - mean[2];
        xx += r0 * r0;
        xy += r0 * r1;
        xz += r0 * r2;
        yy += r1 * r1;
        yz += r1 * r2;
        zz += r2 * r2;
    }
    const double det_x = yy * zz - yz * yz;
    const double det_y = xx * zz - xz * xz;
    const double det_z = xx * yy - xy * xy;
    double abc[3];
    if (det_x > det_y && det_x > det_z) {
        abc[0] = det_x;
        abc[1] = xz * yz - xy * zz;
        abc[2] = xy * yz - xz * yy;
    } else if (det_y > det_z) {
        abc[0] = xz * yz - xy * zz;
        abc[1] = det_y;
        abc[2] = xy * xz - yz * xx;
    } else {
        abc[0] = xy * yz - xz * yy;
        abc[1] = xy * xz - yz * xx;
        abc[2] = det_z;
    }
    const double norm = norm3(abc);
    if (norm < ORC_EPS) return 0;
    abc[0] /= norm;
    abc[1] /= norm;
    abc[2] /= norm;
    out[0] = abc[0];
    out[1] = abc[1];
    out[2] = abc[2];
    out[3] = -dot3(abc, mean);
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* Sphere -- ransac.h:223-344                                                                   */
/* ------------------------------------------------------------------------------------------- */

/* Eigen 3.3 determinant_impl<Derived,4>, row-major m[r][c] ([RECALL]). */
static double det4_helper(double m[4][4], int j, int k, int a, int b) {
    return (m[j][0] * m[k][1] - m[k][0] * m[j][1]) * (m[a][2] * m[b][3] - m[b][2] * m[a][3]);
}
#if ORC_FP_ORDER == 2
/* Eigen 3.4 determinant_impl<Derived,4> ([RECALL]): det2 minors of columns 0-1, det3 cofactors along column 2,
 * expansion along column 3; pmadd(a, b, c) = a * b + c without FMA */
static double det4_d2(double m[4][4], int i0, int i1) { return m[i0][0] * m[i1][1] - m[i1][0] * m[i0][1]; }
static double det4_d3(double m[4][4], int i0, double d0, int i1, double d1, int i2, double d2) {
    return m[i0][2] * d0 + ((-m[i1][2]) * d1 + m[i2][2] * d2);
}
static double det4(double m[4][4]) {
    (void)det4_helper;
    const double d01 = det4_d2(m, 0, 1), d02 = det4_d2(m, 0, 2), d03 = det4_d2(m, 0, 3);
    const double d12 = det4_d2(m, 1, 2), d13 = det4_d2(m, 1, 3), d23 = det4_d2(m, 2, 3);
    const double c0 = det4_d3(m, 1, d23, 2, d13, 3, d12);
    const double c1 = det4_d3(m, 0, d23, 2, d03, 3, d02);
    const double c2 = det4_d3(m, 0, d13, 1, d03, 3, d01);
    const double c3 = det4_d3(m, 0, d12, 1, d02, 2, d01);
    return ((-m[0][3]) * c0 + m[1][3] * c1) + ((-m[2][3]) * c2 + m[3][3] * c3);
}
#else
static double det4(double m[4][4]) {
    return det4_helper(m, 0, 1, 2, 3) - det4_helper(m, 0, 2, 1, 3) + det4_helper(m, 0, 3, 1, 2) +
           det4_helper(m, 1, 2, 0, 3) - det4_helper(m, 1, 3, 0, 2) + det4_helper(m, 2, 3, 0, 1);
}
#endif

/* SphereEstimator::ValidationCheck + MinimalFit, ransac.h:225-234, 239-294.  p = 4 points. */
int orc_sphere_minimal_fit(const double *p, double *out) {
    double plane[4];
    if (!orc_plane_minimal_fit(p, plane)) return 0;       /* ransac.h:228-231 */
    if (orc_plane_distance(p + 9, plane) < ORC_EPS) return 0; /* ransac.h:232-233 */

    double m[4][4];
    double sq[4];
    for (int i = 0; i < 4; ++i) sq[i] = dot3(p + 3 * i, p + 3 * i);
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

    double c[3];
    c[0] = 0.5 * (M12 / M11);
    c[1] = -0.5 * (M13 / M11);
    c[2] = 0.5 * (M14 / M11);
    out[0] = c[0];
    out[1] = c[1];
    out[2] = c[2];
    out[3] = sqrt(dot3(c, c) - (M15 / M11)); /* may be NaN, as in the reference */
    return 1;
}

/* SphereEstimator::CalcPointToModelDistance, ransac.h:332-343 */
double orc_sphere_distance(const double *q, const double *m) {
    const double diff[3] = {q[0] - m[0], q[1] - m[1], q[2] - m[2]};
    const double d = norm3(diff);
    const double r = m[3];
    if (d <= r) return r - d;
    return d - r;
}

/* SphereEstimator::GeneralFit, ransac.h:296-330: least squares of A w = b with A = [2x 2y 2z 1],
 * b = x^2+y^2+z^2.  The reference solves with Eigen bdcSvd(FullU|FullV) (un-restatable and O(n^2)
 * memory); the unique least-squares solution is restated here with Householder QR in fp64.
 * Agreement with any backward-stable solver is ~1e-12 relative for well-conditioned inputs. */
int orc_sphere_general_fit(const double *pts, size_t n, double *out) {
    if (n < 4) return 0;
    double *A = (double *)malloc(sizeof(double) * n * 5);
    if (!A) return 0;
    for (size_t i = 0; i < n; ++i) {
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        A[5 * i] = x * 2;
        A[5 * i + 1] = y * 2;
        A[5 * i + 2] = z * 2;
        A[5 * i + 3] = 1.0;
        A[5 * i + 4] = (x * x + y * y) + z * z; /* rhs carried as 5th column */
    }
    for (int k = 0; k < 4; ++k) {
        double nrm2 = 0;
        for (size_t i = k; i < n; ++i) nrm2 += A[5 * i + k] * A[5 * i + k];
        double nrm = sqrt(nrm2);
        if (nrm == 0) {
            free(A);
            return 0;
        }
        const double alpha = A[5 * k + k] > 0 ? -nrm : nrm;
        const double v0 = A[5 * k + k] - alpha;
        /* v = (v0, A[k+1..,k]); beta = 2/(v^T v) */
        double vtv = v0 * v0;
        for (size_t i = k + 1; i < n; ++i) vtv += A[5 * i + k] * A[5 * i + k];
        if (vtv != 0) {
            for (int j = k + 1; j < 5; ++j) {
                double s = v0 * A[5 * k + j];
                for (size_t i = k + 1; i < n; ++i) s += A[5 * i + k] * A[5 * i + j];
                s = 2.0 * s / vtv;
                A[5 * k + j] -= s * v0;
                for (size_t i = k + 1; i < n; ++i) A[5 * i + j] -= s * A[5 * i + k];
            }
        }
        A[5 * k + k] = alpha;
        for (size_t i = k + 1; i < n; ++i) A[5 * i + k] = 0; /* (v no longer needed) */
    }
    double w[4];
    for (int k = 3; k >= 0; --k) {
        double s = A[5 * k + 4];
        for (int j = k + 1; j < 4; ++j) s -= A[5 * k + j] * w[j];
        w[k] = s / A[5 * k + k];
    }
    free(A);
    out[0] = w[0];
    out[1] = w[1];
    out[2] = w[2];
    out[3] = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3]); /* ransac.h:322-323 */
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* Cylinder -- ransac.h:350-446, utils.h:313-322                                                */
/* ------------------------------------------------------------------------------------------- */

/* CalcPoint2LineDistance, include/misc3d/utils.h:313-322 */
double orc_point2line(const double *q, const double *p1, const double *p2) {
    double a[3], b[3], c[3], x[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = q[k] - p1[k];
        b[k] = q[k] - p2[k];
        c[k] = p2[k] - p1[k];
    }
    cross3(a, b, x);
    return norm3(x) / norm3(c);
}

/* CylinderEstimator::MinimalFit, ransac.h:354-417.  p = 2 points, nrm = their 2 normals. */
int orc_cylinder_minimal_fit(const double *p, const double *nrm, double *out) {
    /* ransac.h:367-374: the parentheses of the reference put the comparisons INSIDE fabs():
     *   fabs( (p0x - p1x <= DBL_EPSILON) && fabs(dy) <= FLT_EPSILON && fabs(dz) <= FLT_EPSILON ) */
    if ((p[0] - p[3] <= DBL_EPSILON) && (fabs(p[1] - p[4]) <= (double)FLT_EPSILON) &&
        (fabs(p[2] - p[5]) <= (double)FLT_EPSILON))
        return 0;
    const double p1[4] = {p[0], p[1], p[2], 0}, p2[4] = {p[3], p[4], p[5], 0};
    const double n1[4] = {nrm[0], nrm[1], nrm[2], 0}, n2[4] = {nrm[3], nrm[4], nrm[5], 0};
    double w[4];
    for (int k = 0; k < 4; ++k) w[k] = (n1[k] + p1[k]) - p2[k];
    const double a = dot4(n1, n1), b = dot4(n1, n2), c = dot4(n2, n2), d = dot4(n1, w),
                 e = dot4(n2, w);
    const double den = a * c - b * b;
    double sc, tc;
    if (den < 1e-8) {
        sc = 0;
        tc = (b > c ? d / b : e / c);
    } else {
        sc = (b * e - c * d) / den;
        tc = (a * e - b * d) / den;
    }
    double line_pt[4], line_dir[4];
    for (int k = 0; k < 4; ++k) line_pt[k] = (p1[k] + n1[k]) + sc * n1[k];
    for (int k = 0; k < 4; ++k) line_dir[k] = (p2[k] + tc * n2[k]) - line_pt[k];
    /* Eigen >=3.3 normalize(): z = squaredNorm(); if (z > 0) *this /= sqrt(z) */
    const double z = dot4(line_dir, line_dir);
    if (z > 0) {
        const double s = sqrt(z);
        for (int k = 0; k < 4; ++k) line_dir[k] /= s;
    }
    out[0] = line_pt[0];
    out[1] = line_pt[1];
    out[2] = line_pt[2];
    out[3] = line_dir[0];
    out[4] = line_dir[1];
    out[5] = line_dir[2];
    /* ransac.h:413-414: the DIRECTION is passed as the second POINT of the line (kept as is) */
    out[6] = orc_point2line(p, line_pt, line_dir);
    return 1;
}

/* CylinderEstimator::CalcPointToModelDistance, ransac.h:435-445 */
double orc_cylinder_distance(const double *q, const double *w) {
    const double center[3] = {w[0], w[1], w[2]};
    const double ref[3] = {w[0] + w[3], w[1] + w[4], w[2] + w[5]};
    const double d = orc_point2line(q, center, ref);
    return fabs(d - w[6]);
}

/* ------------------------------------------------------------------------------------------- */
/* Generic dispatch                                                                             */
/* ------------------------------------------------------------------------------------------- */
int orc_minimal_sample(int kind) { return kind == ORC_PLANE ? 3 : (kind == ORC_SPHERE ? 4 : 2); }
int orc_num_params(int kind) { return kind == ORC_CYLINDER ? 7 : 4; }

double orc_distance(int kind, const double *q, const double *m) {
    if (kind == ORC_PLANE) return orc_plane_distance(q, m);
    if (kind == ORC_SPHERE) return orc_sphere_distance(q, m);
    return orc_cylinder_distance(q, m);
}

/* minimal fit from sample indices into the cloud (mirrors pc_.SelectByIndex(sample) + MinimalFit,
 * ransac.h:576-582) */
int orc_minimal_fit_idx(int kind, const double *xyz, const double *normals, const size_t *idx,
                        double *out) {
    double p[12], nn[6];
    const int m = orc_minimal_sample(kind);
    for (int s = 0; s < m; ++s)
        for (int k = 0; k < 3; ++k) p[3 * s + k] = xyz[3 * idx[s] + k];
    if (kind == ORC_PLANE) return orc_plane_minimal_fit(p, out);
    if (kind == ORC_SPHERE) return orc_sphere_minimal_fit(p, out);
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < 3; ++k) nn[3 * s + k] = normals[3 * idx[s] + k];
    return orc_cylinder_minimal_fit(p, nn, out);
}

/* RANSAC::EvaluateModel, ransac.h:626-654: serial scan, error summed in point order. */
void orc_evaluate_model(int kind, const double *xyz, size_t n, double thr, const double *model,
                        uint64_t *inlier_num, double *error_sum) {
    uint64_t cnt = 0;
    double err = 0;
    for (size_t i = 0; i < n; ++i) {
        const double d = orc_distance(kind, xyz + 3 * i, model);
        if (d < thr) {
            err += d;
            cnt++;
        }
    }
    *inlier_num = cnt;
    *error_sum = err;
}

static void fitness_rmse(uint64_t cnt, double err, size_t n, double *fitness, double *rmse) {
    if (cnt == 0) { /* ransac.h:644-646 */
        *fitness = 0;
        *rmse = 1e+10;
    } else { /* ransac.h:648-650; "rmse" really is error / sqrt(n_inliers) */
        *fitness = (double)cnt / (double)n;
        *rmse = err / sqrt((double)cnt);
    }
}

/* (size_t) of a double as the x86-64 SysV code generated by gcc behaves (cvttsd2si based), made
 * explicit because the C conversion is undefined for NaN / negative / huge values.  Used for the
 * adaptive iteration bound of ransac.h:601-606. */
uint64_t orc_double_to_size_t_x86(double v) {
    const double two63 = 9223372036854775808.0;
    if (v >= two63) {
        const double w = v - two63;
        if (!(w < two63)) return 0x8000000000000000ull ^ 0x8000000000000000ull; /* indefinite ^ sign */
        return ((uint64_t)(int64_t)w) ^ 0x8000000000000000ull;
    }
    if (!(v > -two63)) return 0x8000000000000000ull; /* NaN, -inf, <= -2^63: integer indefinite */
    return (uint64_t)(int64_t)v;                     /* truncation toward zero */
}

typedef struct {
    double fitness;       /* ransac.h:617 best fitness */
    double inlier_rmse;   /* best "rmse" */
    uint64_t count;       /* number of valid hypotheses evaluated ("run {} iterations") */
    uint64_t iterations;  /* loop index after which every further iteration was skipped */
    int64_t best_index;   /* hypothesis index of the best model, -1 if none */
    int general_fit_ok;   /* return value of RefineModel */
} orc_stats;

/* Per-hypothesis trace (all pointers optional, sized max_iter): lets the tests compare the GPU
 * against the oracle hypothesis by hypothesis. */
typedef struct {
    size_t *samples;     /* max_iter x m */
    int *valid;          /* MinimalFit return */
    double *models;      /* max_iter x nparams */
    uint64_t *counts;    /* inlier_num */
    double *errors;      /* serial error sum */
} orc_trace;

/* RANSAC::FitModel -> FitModelParallel -> RefineModel, ransac.h:506-624, sequential semantics.
 * Returns 1 = reference `true`, 0 = reference `false` (GeneralFit failed), <0 = the reference
 * throws (LogError): -1 probability out of (0,1] (ransac.h:483-485), -2 too few points
 * (ransac.h:510-513), -3 cylinder without normals (ransac.h:356-359, py_common.cpp:50-52).
 * params receives best_model after RefineModel (refined in place when GeneralFit succeeds). */
static int fit_impl(int kind, const double *xyz, const double *normals, size_t n, double thr,
                    size_t max_iter, double prob, uint64_t seed, double *params, size_t *inliers,
                    size_t *n_inliers, orc_stats *stats, orc_trace *trace, size_t lookahead);

int orc_fit(int kind, const double *xyz, const double *normals, size_t n, double thr,
            size_t max_iter, double prob, uint64_t seed, double *params, size_t *inliers,
            size_t *n_inliers, orc_stats *stats, orc_trace *trace) {
    return fit_impl(kind, xyz, normals, n, thr, max_iter, prob, seed, params, inliers, n_inliers,
                    stats, trace, 1);
}

/* The SAME sequential loop, for clouds too big to scan hypothesis by hypothesis on one core (the
 * full-size BASELINE configurations): the minimal fit + EvaluateModel of the next `lookahead`
 * hypotheses are computed ahead by an OpenMP team -- a hypothesis' record depends only on its
 * sample, and the samples are the same prefix of the one mt19937 stream, drawn in order --
 * and the loop below consumes them in index order with the reference's own update / stop rules.
 * Records computed past an early stop are discarded (the sampler's extra draws with them: nothing
 * reads the generator afterwards).  Every output, including the serial error sums, is the
 * sequential loop's bit for bit (tests/test_oracle_primitives.py compares the two). */
int orc_fit_parallel(int kind, const double *xyz, const double *normals, size_t n, double thr,
                     size_t max_iter, double prob, uint64_t seed, double *params, size_t *inliers,
                     size_t *n_inliers, orc_stats *stats, orc_trace *trace, size_t lookahead) {
    return fit_impl(kind, xyz, normals, n, thr, max_iter, prob, seed, params, inliers, n_inliers,
                    stats, trace, lookahead < 1 ? 1 : lookahead);
}

static int fit_impl(int kind, const double *xyz, const double *normals, size_t n, double thr,
                    size_t max_iter, double prob, uint64_t seed, double *params, size_t *inliers,
                    size_t *n_inliers, orc_stats *stats, orc_trace *trace, size_t lookahead) {
    const int m = orc_minimal_sample(kind);
    const int np = orc_num_params(kind);
    if (prob <= 0 || prob > 1) return -1;
    if (kind == ORC_CYLINDER && normals == NULL) return -3;
    if (n < (size_t)m) return -2;

    double best_fitness = 0, best_rmse = 0; /* Clear(), ransac.h:519-522 */
    double best_model[7] = {0, 0, 0, 0, 0, 0, 0}; /* reference: uninitialised VectorXd; oracle: 0 */
    int64_t best_index = -1;
    uint64_t count = 0;
    uint64_t current_iteration = UINT64_MAX; /* ransac.h:569 */
    uint64_t last_run = 0;
    orc_mt19937 rng;
    orc_mt_seed(&rng, seed);

    /* ransac.h:572: `for (int i = 0; i < max_iteration_; ++i)` */
    /* look-ahead window [w0, w1) of records computed ahead (lookahead == 1: none, the plain loop) */
    size_t w0 = 0, w1 = 0;
    size_t *w_samples = NULL;
    int *w_valid = NULL;
    double *w_models = NULL, *w_errors = NULL;
    uint64_t *w_counts = NULL;
    if (lookahead > 1) {
        w_samples = (size_t *)malloc(sizeof(size_t) * lookahead * 4);
        w_valid = (int *)malloc(sizeof(int) * lookahead);
        w_models = (double *)malloc(sizeof(double) * lookahead * 7);
        w_errors = (double *)malloc(sizeof(double) * lookahead);
        w_counts = (uint64_t *)malloc(sizeof(uint64_t) * lookahead);
    }
    for (size_t i = 0; i < max_iter; ++i) {
        if (count > current_iteration) break; /* ransac.h:573-575: every later i is skipped too */
        last_run = i + 1;
        size_t sample[4];
        double model[7] = {0, 0, 0, 0, 0, 0, 0};
        int ok;
        if (lookahead > 1) {
            if (i >= w1) {
                w0 = i;
                w1 = i + lookahead < max_iter ? i + lookahead : max_iter;
                for (size_t j = w0; j < w1; ++j) orc_sample(&rng, n, m, w_samples + (j - w0) * 4);
#pragma omp parallel for schedule(dynamic, 1)
                for (long j = 0; j < (long)(w1 - w0); ++j) {
                    double *mj = w_models + 7 * j;
                    for (int q = 0; q < 7; ++q) mj[q] = 0;
                    w_valid[j] = orc_minimal_fit_idx(kind, xyz, normals, w_samples + j * 4, mj);
                    w_counts[j] = 0;
                    w_errors[j] = 0;
                    if (w_valid[j])
                        orc_evaluate_model(kind, xyz, n, thr, mj, &w_counts[j], &w_errors[j]);
                }
            }
            memcpy(sample, w_samples + (i - w0) * 4, sizeof(size_t) * 4);
            memcpy(model, w_models + 7 * (i - w0), sizeof(double) * 7);
            ok = w_valid[i - w0];
        } else {
            orc_sample(&rng, n, m, sample);
            ok = orc_minimal_fit_idx(kind, xyz, normals, sample, model);
        }
        if (trace) {
            if (trace->samples) memcpy(trace->samples + i * m, sample, sizeof(size_t) * m);
            if (trace->valid) trace->valid[i] = ok;
            if (trace->models) memcpy(trace->models + i * np, model, sizeof(double) * np);
            if (trace->counts) trace->counts[i] = 0;
            if (trace->errors) trace->errors[i] = 0;
        }
        if (!ok) continue; /* ransac.h:584-586: no count++ */
        uint64_t cnt;
        double err, fitness, rmse;
        if (lookahead > 1) {
            cnt = w_counts[i - w0];
            err = w_errors[i - w0];
        } else {
            orc_evaluate_model(kind, xyz, n, thr, model, &cnt, &err);
        }
        if (trace) {
            if (trace->counts) trace->counts[i] = cnt;
            if (trace->errors) trace->errors[i] = err;
        }
        fitness_rmse(cnt, err, n, &fitness, &rmse);
        if (fitness > best_fitness || (fitness == best_fitness && rmse < best_rmse)) {
            best_fitness = fitness;
            best_rmse = rmse;
            memcpy(best_model, model, sizeof(double) * np);
            best_index = (int64_t)i;
            if (best_fitness < 1.0) {
                const double k = log(1 - prob) / log(1 - pow(best_fitness, (double)m));
                const double lim = (double)max_iter;
                current_iteration = orc_double_to_size_t_x86(lim < k ? lim : k); /* std::min(k,lim) */
            } else {
                current_iteration = 0;
            }
        }
        count++;
    }
    free(w_samples);
    free(w_valid);
    free(w_models);
    free(w_errors);
    free(w_counts);

    /* RefineModel, ransac.h:534-549 */
    size_t ni = 0;
    for (size_t i = 0; i < n; ++i) {
        const double d = orc_distance(kind, xyz + 3 * i, best_model);
        if (d < thr) inliers[ni++] = i;
    }
    *n_inliers = ni;
    int ok = 1;
    if (kind != ORC_CYLINDER) {
        double *sel = (double *)malloc(sizeof(double) * 3 * (ni ? ni : 1));
        for (size_t k = 0; k < ni; ++k) memcpy(sel + 3 * k, xyz + 3 * inliers[k], 3 * sizeof(double));
        ok = (kind == ORC_PLANE) ? orc_plane_general_fit(sel, ni, best_model)
                                 : orc_sphere_general_fit(sel, ni, best_model);
        free(sel);
    } /* cylinder GeneralFit is a no-op returning true, ransac.h:427-433 */
    memcpy(params, best_model, sizeof(double) * np);
    if (stats) {
        stats->fitness = best_fitness;
        stats->inlier_rmse = best_rmse;
        stats->count = count;
        stats->iterations = last_run;
        stats->best_index = best_index;
        stats->general_fit_ok = ok;
    }
    return ok ? 1 : 0;
}

/* RefineModel alone (ransac.h:534-549) for a given pre-refinement model: inlier list + GeneralFit
 * applied in place.  Returns GeneralFit's return value. */
int orc_refine(int kind, const double *xyz, size_t n, double thr, double *model, size_t *inliers,
               size_t *n_inliers) {
    size_t ni = 0;
    for (size_t i = 0; i < n; ++i) {
        const double d = orc_distance(kind, xyz + 3 * i, model);
        if (d < thr) inliers[ni++] = i;
    }
    *n_inliers = ni;
    if (kind == ORC_CYLINDER) return 1;
    double *sel = (double *)malloc(sizeof(double) * 3 * (ni ? ni : 1));
    for (size_t k = 0; k < ni; ++k) memcpy(sel + 3 * k, xyz + 3 * inliers[k], 3 * sizeof(double));
    const int ok = (kind == ORC_PLANE) ? orc_plane_general_fit(sel, ni, model)
                                       : orc_sphere_general_fit(sel, ni, model);
    free(sel);
    return ok;
}

/* Hypothesis-level helper for kernel parity tests: minimal fit + serial evaluation of H given
 * samples (H x m indices). */
void orc_score_samples(int kind, const double *xyz, const double *normals, size_t n, double thr,
                       const size_t *samples, size_t H, int *valid, double *models,
                       uint64_t *counts, double *errors) {
    const int m = orc_minimal_sample(kind);
    const int np = orc_num_params(kind);
#pragma omp parallel for schedule(dynamic, 4) if ((double)H * (double)n > 2e7)
    for (long h = 0; h < (long)H; ++h) {
        double model[7] = {0, 0, 0, 0, 0, 0, 0};
        const int ok = orc_minimal_fit_idx(kind, xyz, normals, samples + (size_t)h * m, model);
        valid[h] = ok;
        memcpy(models + (size_t)h * np, model, sizeof(double) * np);
        counts[h] = 0;
        errors[h] = 0;
        if (ok) orc_evaluate_model(kind, xyz, n, thr, model, &counts[h], &errors[h]);
    }
}

/* Draw the sample table the sequential driver would draw if no early exit happened. */
void orc_draw_samples(size_t n, int m, size_t H, uint64_t seed, size_t *samples) {
    orc_mt19937 rng;
    orc_mt_seed(&rng, seed);
    for (size_t h = 0; h < H; ++h) orc_sample(&rng, n, m, samples + h * m);
}

/* ------------------------------------------------------------------------------------------- */
/* Reference-shaped OpenMP baseline (cpu_baseline leg of bench.py).                             */
/* Same loop structure as ransac.h:571-613: omp parallel for schedule(static) over hypotheses,   */
/* per hypothesis a small heap allocation for the sample (SelectByIndex, ransac.h:578) and a     */
/* serial AoS scan with per-point sqrt and divide; best-update in a critical section.  To keep   */
/* the result deterministic the samples are pre-drawn sequentially and the best-update rule is   */
/* replayed in index order afterwards (prob = 1 semantics: every hypothesis is evaluated).       */
/* ------------------------------------------------------------------------------------------- */
int orc_fit_omp_baseline(int kind, const double *xyz, const double *normals, size_t n, double thr,
                         size_t H, uint64_t seed, double *best_model_out, uint64_t *best_count,
                         int64_t *best_index) {
    const int m = orc_minimal_sample(kind);
    const int np = orc_num_params(kind);
    size_t *samples = (size_t *)malloc(sizeof(size_t) * H * m);
    int *valid = (int *)malloc(sizeof(int) * H);
    double *models = (double *)malloc(sizeof(double) * H * np);
    uint64_t *counts = (uint64_t *)malloc(sizeof(uint64_t) * H);
    double *errors = (double *)malloc(sizeof(double) * H);
    orc_draw_samples(n, m, H, seed, samples);
#pragma omp parallel for schedule(static)
    for (long h = 0; h < (long)H; ++h) {
        double *sel = (double *)malloc(sizeof(double) * 6 * m); /* mimics SelectByIndex alloc */
        double model[7] = {0, 0, 0, 0, 0, 0, 0};
        const int ok = orc_minimal_fit_idx(kind, xyz, normals, samples + (size_t)h * m, model);
        free(sel);
        valid[h] = ok;
        memcpy(models + (size_t)h * np, model, sizeof(double) * np);
        counts[h] = 0;
        errors[h] = 0;
        if (ok) orc_evaluate_model(kind, xyz, n, thr, model, &counts[h], &errors[h]);
    }
    double bf = 0, br = 0;
    int64_t bi = -1;
    for (size_t h = 0; h < H; ++h) {
        if (!valid[h]) continue;
        double f, r;
        fitness_rmse(counts[h], errors[h], n, &f, &r);
        if (f > bf || (f == bf && r < br)) {
            bf = f;
            br = r;
            bi = (int64_t)h;
        }
    }
    if (bi >= 0) {
        memcpy(best_model_out, models + (size_t)bi * np, sizeof(double) * np);
        *best_count = counts[bi];
    } else {
        memset(best_model_out, 0, sizeof(double) * np);
        *best_count = 0;
    }
    *best_index = bi;
    free(samples);
    free(valid);
    free(models);
    free(counts);
    free(errors);
    return 0;
}

int orc_omp_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads of the next parallel regions (bench.py probes which count does best on the box) */
void orc_set_omp_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* SegmentPlaneIterative -- src/iterative_plane_segmentation.cpp:8-39                           */
/* ------------------------------------------------------------------------------------------- */
/* The reference creates a fresh random_device-seeded sampler each round (ransac.h:570); the
 * seeded restatement uses seed + round for round = 0,1,...  Output: planes (4 per cluster),
 * cluster_offsets (k+1), cluster_indices = indices into the ORIGINAL cloud, cluster by cluster,
 * ascending inside a cluster (SelectByIndex keeps order).  The reference loops forever when a
 * round finds no inlier (:29,:35); the oracle stops there and reports it via return value 2.
 * Returns 0 ok, 1 = N<3 (reference: warning + empty result, :13-17), 2 = stalled. */
static int segment_impl(const double *xyz, size_t n, double thr, int max_iteration,
                        double min_ratio, uint64_t seed, size_t max_clusters, double *planes,
                        size_t *cluster_offsets, size_t *cluster_indices, size_t *n_clusters,
                        size_t lookahead);

int orc_segment_plane_iterative(const double *xyz, size_t n, double thr, int max_iteration,
                                double min_ratio, uint64_t seed, size_t max_clusters,
                                double *planes, size_t *cluster_offsets, size_t *cluster_indices,
                                size_t *n_clusters) {
    return segment_impl(xyz, n, thr, max_iteration, min_ratio, seed, max_clusters, planes,
                        cluster_offsets, cluster_indices, n_clusters, 1);
}

/* the same rounds with orc_fit_parallel's look-ahead inside every fit (10 M-point scenes) */
int orc_segment_plane_iterative_parallel(const double *xyz, size_t n, double thr,
                                         int max_iteration, double min_ratio, uint64_t seed,
                                         size_t max_clusters, double *planes,
                                         size_t *cluster_offsets, size_t *cluster_indices,
                                         size_t *n_clusters, size_t lookahead) {
    return segment_impl(xyz, n, thr, max_iteration, min_ratio, seed, max_clusters, planes,
                        cluster_offsets, cluster_indices, n_clusters, lookahead < 1 ? 1 : lookahead);
}

static int segment_impl(const double *xyz, size_t n, double thr, int max_iteration,
                        double min_ratio, uint64_t seed, size_t max_clusters, double *planes,
                        size_t *cluster_offsets, size_t *cluster_indices, size_t *n_clusters,
                        size_t lookahead) {
    *n_clusters = 0;
    cluster_offsets[0] = 0;
    if (n < 3) return 1;
    double *cur = (double *)malloc(sizeof(double) * 3 * n);
    size_t *orig = (size_t *)malloc(sizeof(size_t) * n);
    size_t *inl = (size_t *)malloc(sizeof(size_t) * n);
    memcpy(cur, xyz, sizeof(double) * 3 * n);
    for (size_t i = 0; i < n; ++i) orig[i] = i;
    size_t cur_n = n, count = 0, k = 0;
    const size_t target = (size_t)((1 - min_ratio) * (double)n); /* :28 */
    double plane[4] = {0, 0, 0, 0}; /* `plane` persists across rounds (:22) */
    int rc = 0;
    while (count < target && k < max_clusters) {
        size_t ni = 0;
        orc_stats st;
        /* probability stays at the RANSAC default 0.9999 (ransac.h:462) */
        int r = fit_impl(ORC_PLANE, cur, NULL, cur_n, thr, (size_t)max_iteration, 0.9999,
                         seed + k, plane, inl, &ni, &st, NULL, lookahead);
        if (r < 0) { /* reference would throw (fewer than 3 points left) */
            rc = 2;
            break;
        }
        if (ni == 0) {
            rc = 2;
            break;
        }
        memcpy(planes + 4 * k, plane, sizeof(double) * 4);
        size_t off = cluster_offsets[k];
        for (size_t j = 0; j < ni; ++j) cluster_indices[off + j] = orig[inl[j]];
        cluster_offsets[k + 1] = off + ni;
        /* remaining = SelectByIndex(inliers, invert=true): order preserved (:33) */
        size_t w = 0, q = 0;
        for (size_t i = 0; i < cur_n; ++i) {
            if (q < ni && inl[q] == i) {
                q++;
                continue;
            }
            cur[3 * w] = cur[3 * i];
            cur[3 * w + 1] = cur[3 * i + 1];
            cur[3 * w + 2] = cur[3 * i + 2];
            orig[w] = orig[i];
            w++;
        }
        cur_n = w;
        count += ni;
        k++;
    }
    *n_clusters = k;
    free(cur);
    free(orig);
    free(inl);
    return rc;
}
