// m3d_eig3.hpp -- "J3x3": the fully specified symmetric 3x3 eigen-solver used for normals
// (EstimateNormalsFromMap, src/normal_estimation.cpp:156-162 calls Eigen::SelfAdjointEigenSolver::compute,
// an iterative QL whose roundings cannot be restated; oracle/misc3d_oracle_normals.c documents the
// substitution).  Cyclic Jacobi, pairs (0,1),(0,2),(1,2), at most 24 sweeps, stop when the three
// off-diagonal entries are exactly zero; eigenvector of the smallest eigenvalue (lowest index on ties),
// normalised.  Host and device run this same code without FMA contraction.
#pragma once
#include <math.h>

#include "m3d_fp.hpp"

namespace m3d {

M3D_HD void j3x3_smallest_eigvec(const double* Ain, double* n) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) A[k] = Ain[k];
    for (int sweep = 0; sweep < 24; ++sweep) {
        if (A[1] == 0.0 && A[2] == 0.0 && A[5] == 0.0) break;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int p = e == 2 ? 1 : 0, q = e == 0 ? 1 : 2;
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) {
                const double akp = A[3 * k + p], akq = A[3 * k + q];
                A[3 * k + p] = c * akp - s * akq;
                A[3 * k + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {
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

}  // namespace m3d
