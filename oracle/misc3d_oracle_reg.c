/*
 * misc3d_oracle_reg.c -- CPU restatement of the registration half of the hot path.
 * TEST INFRASTRUCTURE ONLY (see the header of misc3d_oracle.c; same rules, same build flags).
 *
 * PARITY UNPINNED.  The arithmetic of this half lives in third-party code that is NOT under
 * /root/reference and not installed here:
 *   - Open3D v0.15.1 (CI pin .github/workflows/ubuntu.yml:39): RegistrationRANSACBasedOnCorrespondence,
 *     TransformationEstimationPointToPoint, CorrespondenceCheckerBasedOnEdgeLength / ...Distance,
 *     KDTreeFlann::SearchHybrid, GetRegistrationResultAndCorrespondences,
 *     EvaluateInlierCorrespondenceRatio  -- call site src/transform_estimation.cpp:124-164
 *   - Eigen (umeyama, JacobiSVD)          -- call site src/transform_estimation.cpp:59-66
 *   - nanoflann / Annoy nearest neighbour -- call site src/correspondence_matching.cpp:13-44
 * Their published algorithms are restated from memory of the pinned versions ([RECALL], SURVEY.md
 * 8a rows a17-a20).  Reference-side tests for these boundaries: none.
 *
 * Canonical sequential semantics: one thread, std::mt19937(seed) + libstdc++
 * uniform_int_distribution<int>(0, M-1) for the correspondence sampler; sums in index order.
 *
 * The 3x3 SVD inside umeyama cannot be restated bit-for-bit (Eigen JacobiSVD internals); the oracle
 * and the HIP path both implement the SAME fully specified algorithm ("K3x3", below), so that they
 * agree bit-for-bit with each other and to ~1e-14 with Eigen / numpy.linalg.svd (tests check the
 * latter with numpy).
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t mt[624];
    int idx;
} orc_mt19937;
void orc_mt_seed(orc_mt19937 *g, uint64_t seed);
uint32_t orc_mt_next(orc_mt19937 *g);
uint32_t orc_uniform_int(orc_mt19937 *g, uint32_t range_incl);

static inline double dot3(const double *a, const double *b) {
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
/* Eigen::Vector3d reductions of the Open3D checkers (`.norm()`): the association follows ORC_FP_ORDER like the fits'
 * (misc3d_oracle.c "Eigen reduction orders"; product: m3d_fp.hpp sum3 through m3d_reg_fp.hpp reg_checkers).  K3x3 and
 * the nearest-neighbour distances are this repository's own fully specified order on both sides and do not switch. */
#ifndef ORC_FP_ORDER
#define ORC_FP_ORDER 0
#endif
static inline double eig_dot3(const double *a, const double *b) {
#if ORC_FP_ORDER == 1
    return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
#else
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* K3x3: rotation part of umeyama from a 3x3 covariance `sigma` (row-major).                    */
/* One-sided (Hestenes) Jacobi on the columns of A = sigma, V accumulates the rotations:        */
/*   pairs visited in the order (0,1),(0,2),(1,2); at most 30 sweeps; a pair is skipped when     */
/*   gamma == 0 or |gamma| <= 2^-52 * sqrt(alpha*beta).                                          */
/* Singular values = column norms, sorted descending (stable).  u1,u2 = columns / sigma,         */
/* u3 = u1 x u2 (so det U = +1), and R = u1 v1^T + u2 v2^T + sign(det V) u3 v3^T, which equals    */
/* Eigen::umeyama's U * diag(1,1,sign(det U det V)) * V^T (src Eigen/src/Geometry/Umeyama.h       */
/* [RECALL]) including the rank-2 case of three correspondences.                                 */
/* Outputs: R (row-major 3x3), sv[3] singular values with sv[2] SIGNED by the reflection rule.   */
/* ------------------------------------------------------------------------------------------- */
void orc_k3x3_rotation(const double *sigma, double *R, double *sv) {
    double a[3][3], v[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            a[r][c] = sigma[3 * r + c];
            v[r][c] = (r == c) ? 1.0 : 0.0;
        }
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int k = 0; k < 3; ++k) {
            const int p = PQ[k][0], q = PQ[k][1];
            const double alpha = (a[0][p] * a[0][p] + a[1][p] * a[1][p]) + a[2][p] * a[2][p];
            const double beta = (a[0][q] * a[0][q] + a[1][q] * a[1][q]) + a[2][q] * a[2][q];
            const double gamma = (a[0][p] * a[0][q] + a[1][p] * a[1][q]) + a[2][p] * a[2][q];
            if (gamma == 0.0) continue;
            if (fabs(gamma) <= 2.220446049250313e-16 * sqrt(alpha * beta)) continue;
            rotated = 1;
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t);
            const double s = c * t;
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
    for (int c = 0; c < 3; ++c)
        sg[c] = sqrt((a[0][c] * a[0][c] + a[1][c] * a[1][c]) + a[2][c] * a[2][c]);
    int ord[3] = {0, 1, 2};
    /* stable insertion sort, descending */
    for (int i = 1; i < 3; ++i) {
        int j = i;
        while (j > 0 && sg[ord[j]] > sg[ord[j - 1]]) {
            int tmp = ord[j];
            ord[j] = ord[j - 1];
            ord[j - 1] = tmp;
            --j;
        }
    }
    const int i1 = ord[0], i2 = ord[1], i3 = ord[2];
    double u1[3], u2[3], u3[3];
    if (!(sg[i1] > 0.0)) { /* sigma == 0: all points coincide; rotation undefined -> identity */
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[3 * r + c] = (r == c) ? 1.0 : 0.0;
        sv[0] = sv[1] = sv[2] = 0.0;
        return;
    }
    for (int r = 0; r < 3; ++r) u1[r] = a[r][i1] / sg[i1];
    if (sg[i2] > 1e-300 && sg[i2] > 2.220446049250313e-16 * sg[i1]) {
        for (int r = 0; r < 3; ++r) u2[r] = a[r][i2] / sg[i2];
    } else { /* rank 1 (collinear input): deterministic completion */
        int kmin = 0;
        if (fabs(u1[1]) < fabs(u1[kmin])) kmin = 1;
        if (fabs(u1[2]) < fabs(u1[kmin])) kmin = 2;
        double e[3] = {0, 0, 0};
        e[kmin] = 1.0;
        const double pr = u1[kmin];
        double w[3];
        for (int r = 0; r < 3; ++r) w[r] = e[r] - pr * u1[r];
        const double nw = sqrt(dot3(w, w));
        for (int r = 0; r < 3; ++r) u2[r] = w[r] / nw;
    }
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    /* determinant of the SORTED V = [v_i1 v_i2 v_i3] (M[r][c] = v[r][i_c]) */
    const double detv = (v[0][i1] * (v[1][i2] * v[2][i3] - v[1][i3] * v[2][i2]) -
                         v[0][i2] * (v[1][i1] * v[2][i3] - v[1][i3] * v[2][i1])) +
                        v[0][i3] * (v[1][i1] * v[2][i2] - v[1][i2] * v[2][i1]);
    const double sgn = detv < 0.0 ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            R[3 * r + c] = (u1[r] * v[c][i1] + u2[r] * v[c][i2]) + (sgn * u3[r]) * v[c][i3];
    const double a3[3] = {a[0][i3], a[1][i3], a[2][i3]};
    sv[0] = sg[i1];
    sv[1] = sg[i2];
    sv[2] = sgn * dot3(u3, a3); /* = S * sigma_3 of umeyama */
}

/* Eigen::umeyama(src, dst, with_scaling) -- call sites src/transform_estimation.cpp:65 and Open3D
 * TransformationEstimationPointToPoint::ComputeTransformation.  src/dst: n x 3 AoS.  Sums in index
 * order.  T: row-major 4x4. */
void orc_umeyama(const double *src, const double *dst, size_t n, int with_scaling, double *T) {
    const double one_over_n = 1.0 / (double)n;
    double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            ms[k] += src[3 * i + k];
            md[k] += dst[3 * i + k];
        }
    for (int k = 0; k < 3; ++k) {
        ms[k] *= one_over_n;
        md[k] *= one_over_n;
    }
    double sig[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double var[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        double s[3], d[3];
        for (int k = 0; k < 3; ++k) {
            s[k] = src[3 * i + k] - ms[k];
            d[k] = dst[3 * i + k] - md[k];
        }
        for (int r = 0; r < 3; ++r) {
            var[r] += s[r] * s[r];
            for (int c = 0; c < 3; ++c) sig[3 * r + c] += d[r] * s[c];
        }
    }
    for (int k = 0; k < 9; ++k) sig[k] *= one_over_n;
    const double src_var = ((var[0] + var[1]) + var[2]) * one_over_n;
    double R[9], sv[3];
    orc_k3x3_rotation(sig, R, sv);
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

/* Open3D PointCloud::Transform / CorrespondenceCheckerBasedOnDistance: (T * (x,y,z,1)).head<3>(),
 * fixed-size product accumulated column by column ([RECALL]). */
static inline void transform_point(const double *T, const double *p, double *o) {
    for (int r = 0; r < 3; ++r)
        o[r] = ((T[4 * r] * p[0] + T[4 * r + 1] * p[1]) + T[4 * r + 2] * p[2]) + T[4 * r + 3];
}

void orc_transform_points(const double *T, const double *p, size_t n, double *o) {
    for (size_t i = 0; i < n; ++i) transform_point(T, p + 3 * i, o + 3 * i);
}

/* CorrespondenceCheckerBasedOnEdgeLength::Check then ...BasedOnDistance::Check ([RECALL] Open3D
 * 0.15.1 CorrespondenceChecker.cpp), on the 3 sampled correspondences. */
int orc_reg_checkers(const double *src, const double *dst, const int64_t *cs, const int64_t *cd,
                     const double *T, double edge_thr, double dist_thr) {
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) {
            double es[3], et[3];
            for (int k = 0; k < 3; ++k) {
                es[k] = src[3 * cs[i] + k] - src[3 * cs[j] + k];
                et[k] = dst[3 * cd[i] + k] - dst[3 * cd[j] + k];
            }
            const double ds = sqrt(eig_dot3(es, es)), dt = sqrt(eig_dot3(et, et));
            if (ds < dt * edge_thr || dt < ds * edge_thr) return 0;
        }
    for (int i = 0; i < 3; ++i) {
        double pt[3], df[3];
        transform_point(T, src + 3 * cs[i], pt);
        for (int k = 0; k < 3; ++k) df[k] = dst[3 * cd[i] + k] - pt[k];
        if (sqrt(eig_dot3(df, df)) > dist_thr) return 0;
    }
    return 1;
}

/* GetRegistrationResultAndCorrespondences ([RECALL] Open3D Registration.cpp): for every transformed
 * source point, KDTreeFlann::SearchHybrid(p, thr, 1): nearest target point, kept iff dist^2 < thr^2
 * (std::lower_bound on radius*radius).  Brute-force exact nearest neighbour here.  err2 is summed
 * in source order. */
void orc_reg_validate(const double *src, size_t ns, const double *dst, size_t nd, const double *T,
                      double thr, uint64_t *count, double *err2) {
    const double r2 = thr * thr;
    double *best = (double *)malloc(sizeof(double) * (ns ? ns : 1));
    /* small problems stay on one thread: waking a large OpenMP team costs more than the whole scan */
#pragma omp parallel for schedule(static) if ((double)ns * (double)nd > 2e7)
    for (long i = 0; i < (long)ns; ++i) {
        double p[3];
        transform_point(T, src + 3 * i, p);
        double bd = INFINITY;
        for (size_t j = 0; j < nd; ++j) {
            const double dx = p[0] - dst[3 * j], dy = p[1] - dst[3 * j + 1], dz = p[2] - dst[3 * j + 2];
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < bd) bd = d2;
        }
        best[i] = bd;
    }
    uint64_t c = 0;
    double e = 0;
    for (size_t i = 0; i < ns; ++i)
        if (best[i] < r2) {
            e += best[i];
            c++;
        }
    free(best);
    *count = c;
    *err2 = e;
}

/* EvaluateInlierCorrespondenceRatio ([RECALL] Open3D Registration.cpp) */
double orc_reg_corr_inlier_ratio(const double *src, const double *dst, const int64_t *cs,
                                 const int64_t *cd, size_t m, const double *T, double thr) {
    const double r2 = thr * thr;
    uint64_t c = 0;
    for (size_t i = 0; i < m; ++i) {
        double p[3], df[3];
        transform_point(T, src + 3 * cs[i], p);
        for (int k = 0; k < 3; ++k) df[k] = p[k] - dst[3 * cd[i] + k];
        if (dot3(df, df) < r2) c++;
    }
    return (double)c / (double)m;
}

/* static_cast<int>(std::ceil(v)) as x86-64 cvttsd2si (32-bit) behaves */
static int ceil_to_int_x86(double v) {
    const double c = ceil(v);
    if (!(c > -2147483649.0) || !(c < 2147483648.0)) return INT_MIN;
    return (int)c;
}

typedef struct {
    double fitness;
    double inlier_rmse;
    uint64_t validations;  /* total_validation */
    int64_t iterations;    /* number of itr that drew a sample */
    int64_t best_index;
    int64_t est_k;         /* final est_k_global */
} orc_reg_stats;

typedef struct {           /* optional per-iteration trace, sized max_iter */
    int64_t *triples;      /* max_iter x 3 correspondence ids */
    double *T;             /* max_iter x 16 */
    int *passed;           /* checkers passed */
    uint64_t *counts;
    double *err2;
} orc_reg_trace;

/* RANSACSolver::Solve (src/transform_estimation.cpp:124-164) -> Open3D
 * RegistrationRANSACBasedOnCorrespondence(ransac_n = 3, PointToPoint(false), checkers
 * [EdgeLength(edge_thr), Distance(thr)], criteria(max_iter, confidence)) -- [RECALL], 1 thread.
 * The reference's RANSACSolver leaves edge_length_threshold_ uninitialised
 * (transform_estimation.h:126, self-initialisation); the oracle honours the constructor argument.
 * Returns 0 ok; -1 = fewer than 3 points (reference throws, transform_estimation.cpp:130-133). */
int orc_registration_ransac(const double *src, size_t ns, const double *dst, size_t nd,
                            const int64_t *cs, const int64_t *cd, size_t m, double thr,
                            int max_iter, double edge_thr, double confidence, uint64_t seed,
                            double *T_out, orc_reg_stats *stats, orc_reg_trace *trace) {
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(T_out, I4, sizeof(I4));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (stats) stats->best_index = -1;
    if (ns < 3 || nd < 3) return -1;
    if (m < 3 || thr <= 0.0) return 0; /* Open3D returns RegistrationResult() = identity */
    double best_fit = 0, best_rmse = 0;
    int est_k_global = max_iter, est_k_local = max_iter;
    uint64_t total_validation = 0;
    int64_t iters = 0, best_index = -1;
    orc_mt19937 rng;
    orc_mt_seed(&rng, seed);
    for (int itr = 0; itr < max_iter; ++itr) {
        if (!(itr < est_k_global)) continue;
        iters++;
        int64_t tri[3], s3[3], d3[3];
        double ps[9], pd[9], T[16];
        for (int j = 0; j < 3; ++j) {
            tri[j] = (int64_t)orc_uniform_int(&rng, (uint32_t)(m - 1));
            s3[j] = cs[tri[j]];
            d3[j] = cd[tri[j]];
            memcpy(ps + 3 * j, src + 3 * s3[j], 3 * sizeof(double));
            memcpy(pd + 3 * j, dst + 3 * d3[j], 3 * sizeof(double));
        }
        orc_umeyama(ps, pd, 3, 0, T);
        const int pass = orc_reg_checkers(src, dst, s3, d3, T, edge_thr, thr);
        if (trace) {
            if (trace->triples) memcpy(trace->triples + 3 * (size_t)itr, tri, sizeof(tri));
            if (trace->T) memcpy(trace->T + 16 * (size_t)itr, T, sizeof(T));
            if (trace->passed) trace->passed[itr] = pass;
            if (trace->counts) trace->counts[itr] = 0;
            if (trace->err2) trace->err2[itr] = 0;
        }
        if (!pass) continue;
        uint64_t cnt;
        double e2;
        orc_reg_validate(src, ns, dst, nd, T, thr, &cnt, &e2);
        if (trace) {
            if (trace->counts) trace->counts[itr] = cnt;
            if (trace->err2) trace->err2[itr] = e2;
        }
        double fit = 0, rmse = 0;
        if (cnt > 0) {
            fit = (double)cnt / (double)ns;
            rmse = sqrt(e2 / (double)cnt);
        }
        if (fit > best_fit || (fit == best_fit && rmse < best_rmse)) {
            best_fit = fit;
            best_rmse = rmse;
            best_index = itr;
            memcpy(T_out, T, sizeof(T));
            const double ratio = orc_reg_corr_inlier_ratio(src, dst, cs, cd, m, T, thr);
            const double est_d = log(1.0 - confidence) / log(1.0 - pow(ratio, 3.0));
            est_k_local = est_d < (double)est_k_global ? ceil_to_int_x86(est_d) : est_k_local;
        }
        total_validation++;
        if (est_k_local < est_k_global) est_k_global = est_k_local;
    }
    if (stats) {
        stats->fitness = best_fit;
        stats->inlier_rmse = best_rmse;
        stats->validations = total_validation;
        stats->iterations = iters;
        stats->best_index = best_index;
        stats->est_k = est_k_global;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* ANNMatcher::Match -- src/correspondence_matching.cpp:52-84                                    */
/* feat: dim x N column-major (Eigen MatrixXd, correspondence_matching.h:39-41) = N descriptors   */
/* of `dim` contiguous doubles.  Exact nearest neighbour (the FLANN backend's semantics; the      */
/* ANNOY backend approximates the same thing and is not restatable: multi-threaded forest build   */
/* with per-thread seeds, annoylib.h:1144-1146).  L2 accumulated in dimension order (nanoflann    */
/* L2_Simple_Adaptor [RECALL]); ties -> lowest index.                                             */
/* ---------------------------------------------------------------------------------------------
 * Point-to-point ICP.  SURVEY.md 8(f) N1: both of the reference's registration examples chain Open3D's
 * RegistrationICP on the RANSAC pose (examples/cpp/transform_estimation.cpp:82-86 with
 * TransformationEstimationPointToPoint and the default ICPConvergenceCriteria(1e-6, 1e-6, 30)).
 * [RECALL] Open3D 0.15.1 Registration.cpp RegistrationICP:
 *   pcd = source; if init is not identity: pcd.Transform(init)
 *   result = GetRegistrationResultAndCorrespondences(pcd, target, kdtree, max_dist, init)
 *   for i < max_iteration:
 *       update = ComputeTransformation(pcd, target, result.correspondence_set)   (umeyama, no scaling)
 *       transformation = update * transformation;  pcd.Transform(update)
 *       backup = result;  result = GetRegistrationResultAndCorrespondences(...)
 *       if |backup.fitness - result.fitness| < relative_fitness && |backup.rmse - result.rmse| < relative_rmse: break
 * Canonical choices of this restatement (the reference's are thread- / Eigen-order dependent): error2 and
 * the umeyama sums in source order; nearest neighbour = smallest squared distance, lowest target index on
 * exact ties; 4x4 product rows accumulated left to right.
 * Returns the number of ICP iterations executed; corr (size ns, may be NULL): target index or -1. */
static void icp_result(const double *pcd, size_t ns, const double *dst, size_t nd, double max_dist,
                       int64_t *corr, uint64_t *count, double *err2) {
    const double r2 = max_dist * max_dist;
    double *best = (double *)malloc(sizeof(double) * (ns ? ns : 1));
#pragma omp parallel for schedule(static) if ((double)ns * (double)nd > 2e7)
    for (long i = 0; i < (long)ns; ++i) {
        const double *p = pcd + 3 * i;
        double bd = INFINITY;
        int64_t bj = -1;
        for (size_t j = 0; j < nd; ++j) {
            const double dx = p[0] - dst[3 * j], dy = p[1] - dst[3 * j + 1], dz = p[2] - dst[3 * j + 2];
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < bd) {
                bd = d2;
                bj = (int64_t)j;
            }
        }
        best[i] = bd;
        corr[i] = (bd < r2) ? bj : -1;
    }
    uint64_t c = 0;
    double e = 0;
    for (size_t i = 0; i < ns; ++i)
        if (corr[i] >= 0) {
            e += best[i];
            c++;
        }
    free(best);
    *count = c;
    *err2 = e;
}

/* open3d::pipelines::registration::GetInformationMatrixFromPointClouds(source, target, max_dist, T), the
 * acceptance test of ReconstructionPipeline::GlobalRegistration (src/pipeline.cpp:818-824: info(5,5) /
 * min(Ns, Nt) < 0.3 -> reject).  [RECALL] Open3D 0.15.1 Registration.cpp: pcd = source
 * transformed by T; correspondences as in GetRegistrationResultAndCorrespondences; for every correspondence
 * with target point t = (x, y, z): G_r = [0 z -y 1 0 0], [-z 0 x 0 1 0], [y -x 0 0 0 1]; GTG += G_r G_r^T
 * for the three rows in turn.  The reference reduces over OpenMP threads; here: source order, one thread.
 * info: 6 x 6 row-major.  info[35] is the number of correspondences. */
void orc_information_matrix(const double *src, size_t ns, const double *dst, size_t nd, double max_dist,
                            const double *T, double *info) {
    double *pcd = (double *)malloc(sizeof(double) * 3 * (ns ? ns : 1));
    int64_t *corr = (int64_t *)malloc(sizeof(int64_t) * (ns ? ns : 1));
    orc_transform_points(T, src, ns, pcd);
    uint64_t cnt = 0;
    double e2 = 0;
    icp_result(pcd, ns, dst, nd, max_dist, corr, &cnt, &e2);
    for (int k = 0; k < 36; ++k) info[k] = 0.0;
    for (size_t i = 0; i < ns; ++i) {
        if (corr[i] < 0) continue;
        const double x = dst[3 * corr[i]], y = dst[3 * corr[i] + 1], z = dst[3 * corr[i] + 2];
        const double G[3][6] = {{0.0, z, -y, 1.0, 0.0, 0.0}, {-z, 0.0, x, 0.0, 1.0, 0.0}, {y, -x, 0.0, 0.0, 0.0, 1.0}};
        for (int r = 0; r < 3; ++r)
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) info[6 * a + b] += G[r][a] * G[r][b];
    }
    free(pcd);
    free(corr);
}

static void mat4_mul(const double *A, const double *B, double *Cm) {
    double t[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            t[4 * r + c] = ((A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c]) + A[4 * r + 2] * B[8 + c]) + A[4 * r + 3] * B[12 + c];
    memcpy(Cm, t, sizeof(t));
}

int orc_registration_icp(const double *src, size_t ns, const double *dst, size_t nd, double max_dist,
                         const double *T_init, int max_iter, double rel_fitness, double rel_rmse, double *T_out,
                         double *fitness, double *rmse, uint64_t *n_corr, int64_t *corr_out) {
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double T[16];
    memcpy(T, T_init ? T_init : I4, sizeof(T));
    double *pcd = (double *)malloc(sizeof(double) * 3 * (ns ? ns : 1));
    double *gs = (double *)malloc(sizeof(double) * 3 * (ns ? ns : 1));
    double *gd = (double *)malloc(sizeof(double) * 3 * (ns ? ns : 1));
    int64_t *corr = (int64_t *)malloc(sizeof(int64_t) * (ns ? ns : 1));
    memcpy(pcd, src, sizeof(double) * 3 * ns);
    if (memcmp(T, I4, sizeof(T)) != 0) orc_transform_points(T, src, ns, pcd);
    uint64_t cnt = 0;
    double e2 = 0;
    icp_result(pcd, ns, dst, nd, max_dist, corr, &cnt, &e2);
    double fit = ns ? (double)cnt / (double)ns : 0.0;
    double rm = cnt ? sqrt(e2 / (double)cnt) : 0.0;
    int it = 0;
    for (; it < max_iter; ++it) {
        double U[16];
        memcpy(U, I4, sizeof(U));
        if (cnt) {   // ComputeTransformation: identity for an empty correspondence set
            size_t k = 0;
            for (size_t i = 0; i < ns; ++i)
                if (corr[i] >= 0) {
                    memcpy(gs + 3 * k, pcd + 3 * i, 3 * sizeof(double));
                    memcpy(gd + 3 * k, dst + 3 * corr[i], 3 * sizeof(double));
                    ++k;
                }
            orc_umeyama(gs, gd, k, 0, U);
        }
        mat4_mul(U, T, T);
        orc_transform_points(U, pcd, ns, gs);   // pcd.Transform(update): the cloud itself moves (roundings accumulate)
        memcpy(pcd, gs, sizeof(double) * 3 * ns);
        const double fit0 = fit, rm0 = rm;
        icp_result(pcd, ns, dst, nd, max_dist, corr, &cnt, &e2);
        fit = ns ? (double)cnt / (double)ns : 0.0;
        rm = cnt ? sqrt(e2 / (double)cnt) : 0.0;
        if (fabs(fit0 - fit) < rel_fitness && fabs(rm0 - rm) < rel_rmse) {
            ++it;
            break;
        }
    }
    memcpy(T_out, T, sizeof(T));
    *fitness = fit;
    *rmse = rm;
    *n_corr = cnt;
    if (corr_out) memcpy(corr_out, corr, sizeof(int64_t) * ns);
    free(pcd);
    free(gs);
    free(gd);
    free(corr);
    return it;
}

/* ------------------------------------------------------------------------------------------- */
void orc_nearest(const double *q, size_t nq, const double *db, size_t ndb, int dim, int64_t *nn) {
#pragma omp parallel for schedule(static) if ((double)nq * (double)ndb * dim > 2e8)
    for (long i = 0; i < (long)nq; ++i) {
        double bd = INFINITY;
        int64_t bi = -1;
        for (size_t j = 0; j < ndb; ++j) {
            double acc = 0;
            for (int k = 0; k < dim; ++k) {
                const double df = q[(size_t)i * dim + k] - db[j * dim + k];
                acc += df * df;
            }
            if (acc < bd) {
                bd = acc;
                bi = (int64_t)j;
            }
        }
        nn[i] = bi;
    }
}

/* cross-check, correspondence_matching.cpp:64-78.  Returns the number of mutual matches. */
size_t orc_match_mutual_nn(const double *fs, size_t ns, const double *fd, size_t nd, int dim,
                           int64_t *out_src, int64_t *out_dst) {
    int64_t *n01 = (int64_t *)malloc(sizeof(int64_t) * (ns ? ns : 1));
    int64_t *n10 = (int64_t *)malloc(sizeof(int64_t) * (nd ? nd : 1));
    orc_nearest(fs, ns, fd, nd, dim, n01);
    orc_nearest(fd, nd, fs, ns, dim, n10);
    size_t k = 0;
    for (size_t i = 0; i < ns; ++i) {
        if (n01[i] >= 0 && n10[n01[i]] == (int64_t)i) {
            out_src[k] = (int64_t)i;
            out_dst[k] = n01[i];
            k++;
        }
    }
    free(n01);
    free(n10);
    return k;
}

/* Eigen::MatrixBase::isIdentity(prec) on a 4 x 4 row-major matrix (pipeline.cpp:814): diagonal entries
 * internal::isApprox(x, 1, prec) = |x - 1| <= min(|x|, 1) prec, the others internal::isMuchSmallerThan(x, 1, prec)
 * = |x| <= prec.  [RECALL] Eigen/src/Core/CwiseNullaryOp.h, MathFunctions.h. */
int orc_is_identity4(const double *T, double prec) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            const double x = T[4 * r + c];
            if (r == c) {
                const double ax = fabs(x), m = ax < 1.0 ? ax : 1.0;
                if (!(fabs(x - 1.0) <= m * prec)) return 0;
            } else if (!(fabs(x) <= 1.0 * prec)) {
                return 0;
            }
        }
    return 1;
}

/* ReconstructionPipeline::GlobalRegistration with GlobalRegistrationMethod::Ransac (src/pipeline.cpp:790-828), the
 * only in-library caller of the matcher and the RANSAC solver:
 *   max_dis = voxel_size * 1.4                                            :796
 *   matched_list = ANNMatcher(ANNOY).Match(fpfh_s, fpfh_t)               :800-802
 *   pose = RANSACSolver(max_dis).Solve(pcd_s, pcd_t, matched_list)       :806-807  (max_iter 100000, edge 0.9: the
 *          solver's defaults, transform_estimation.h:121-123; confidence 0.999: RANSACConvergenceCriteria's)
 *   pose.isIdentity(1e-8) -> (true, pose, I6)                            :814-816
 *   info = GetInformationMatrixFromPointClouds(pcd_s, pcd_t, max_dis, pose)   :818-820
 *   info(5,5) / min(Ns, Nt) < 0.3 -> (false, pose, I6)                   :821-824  (size_t min, the quotient in double)
 *   else (true, pose, info)                                              :825
 * The caller passes max_iter / edge_thr / confidence explicitly so that tests can bound the run.
 * Returns 1 = success, 0 = rejected, -1 = fewer than 3 points (the solver throws).  T 4 x 4, info 6 x 6 row-major;
 * n_matches: mutual pairs the matcher found. */
int orc_global_registration(const double *src, size_t ns, const double *dst, size_t nd, const double *fs,
                            const double *fd, int dim, double voxel_size, int max_iter, double edge_thr,
                            double confidence, uint64_t seed, double *T, double *info, uint64_t *n_matches) {
    const double max_dis = voxel_size * 1.4;
    int64_t *cs = (int64_t *)malloc(sizeof(int64_t) * (ns ? ns : 1));
    int64_t *cd = (int64_t *)malloc(sizeof(int64_t) * (ns ? ns : 1));
    const size_t m = orc_match_mutual_nn(fs, ns, fd, nd, dim, cs, cd);
    if (n_matches) *n_matches = m;
    for (int k = 0; k < 36; ++k) info[k] = (k % 7 == 0) ? 1.0 : 0.0;
    const int rr = orc_registration_ransac(src, ns, dst, nd, cs, cd, m, max_dis, max_iter, edge_thr, confidence, seed, T,
                                           NULL, NULL);
    free(cs);
    free(cd);
    if (rr < 0) return -1;
    if (orc_is_identity4(T, 1e-8)) return 1;
    double gi[36];
    orc_information_matrix(src, ns, dst, nd, max_dis, T, gi);
    const size_t mn = ns < nd ? ns : nd;
    if (gi[35] / (double)mn < 0.3) return 0;
    memcpy(info, gi, sizeof(gi));
    return 1;
}
