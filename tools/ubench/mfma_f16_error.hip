// mfma_f16_error.hip -- how far is v_mfma_f32_32x32x16_f16's fp32 result from the exact value of sum a_k b_k + c?
//
// The MFMA screen of the scoring kernels (score_mfma_k) decides a point only when the matrix pipe's value is farther
// from the cut-off than a bound; that bound has to cover the pipe's internal accumulation, which no document specifies.
// This measures it: fp16 operands with random exponents (heavy cancellation, one-huge-term and exactly-cancelling
// cases included), a random fp32 addend, one MFMA and a chain of two (K = 32); the error is reported in units of
// 2^-24 * (sum |a_k b_k| + |c|) -- the worst case over all outputs of all trials.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f16_error.hip -o tools/ubench/mfma_f16_error && tools/ubench/mfma_f16_error
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A: [32 rows][16 k], B: [16 k][32 cols], C: [32][32]; chain = 1: a second (A2, B2) pair accumulates on top
__global__ void k(const _Float16* A, const _Float16* B, const float* C, const _Float16* A2, const _Float16* B2, float* D, int chain) {
    const int lane = threadIdx.x, half = lane >> 5, m = lane & 31;
    const size_t t = blockIdx.x;
    f16x8 a, b, a2, b2;
    for (int i = 0; i < 8; ++i) {
        a[i] = A[t * 512 + m * 16 + half * 8 + i];
        b[i] = B[t * 512 + (half * 8 + i) * 32 + m];
        a2[i] = A2[t * 512 + m * 16 + half * 8 + i];
        b2[i] = B2[t * 512 + (half * 8 + i) * 32 + m];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + m];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (chain) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + m] = acc[r];
}

int main() {
    const int trials = 4096;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<_Float16> A(trials * 512), B(trials * 512), A2(trials * 512), B2(trials * 512);
    std::vector<float> C(trials * 1024), D(trials * 1024);
    auto fill = [&](std::vector<_Float16>& v, int t, int mode) {
        for (int i = 0; i < 512; ++i) {
            double x = U(rng);
            int e = 0;
            if (mode == 0) e = (int)(rng() % 13) - 6;             // random exponents: cancellation of unequal terms
            else if (mode == 1) e = (i % 16 == 0) ? 8 : -8;       // one huge k, the rest tiny
            else if (mode == 2) e = (int)(rng() % 3) - 1;         // similar magnitudes
            else e = -(int)(rng() % 14) - 6;                      // small values, subnormal fp16 included (2^-14 .. 2^-20)
            v[(size_t)t * 512 + i] = (_Float16)std::ldexp(x, e);
        }
    };
    for (int t = 0; t < trials; ++t) {
        const int mode = t % 4;
        fill(A, t, mode);
        fill(B, t, mode);
        fill(A2, t, mode);
        fill(B2, t, mode);
        if (t % 8 >= 4)   // exact cancellation in pairs: k and k + 1 carry opposite products
            for (int m = 0; m < 32; ++m)
                for (int kk = 0; kk < 16; kk += 2) A[(size_t)t * 512 + m * 16 + kk + 1] = -A[(size_t)t * 512 + m * 16 + kk];
        if (t % 8 >= 4)
            for (int kk = 0; kk < 16; kk += 2)
                for (int n = 0; n < 32; ++n) B[(size_t)t * 512 + (kk + 1) * 32 + n] = B[(size_t)t * 512 + kk * 32 + n];
        for (int i = 0; i < 1024; ++i) C[(size_t)t * 1024 + i] = (t % 3 == 0) ? 0.0f : (float)std::ldexp(U(rng), (int)(rng() % 9) - 4);
    }
    _Float16 *dA, *dB, *dA2, *dB2;
    float *dC, *dD;
    hipMalloc(&dA, A.size() * 2);
    hipMalloc(&dB, B.size() * 2);
    hipMalloc(&dA2, A.size() * 2);
    hipMalloc(&dB2, B.size() * 2);
    hipMalloc(&dC, C.size() * 4);
    hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dA2, A2.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB2, B2.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    for (int chain = 0; chain < 2; ++chain) {
        k<<<trials, 64>>>(dA, dB, dC, dA2, dB2, dD, chain);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, worst_rel_result = 0;
        int wt = 0, wm = 0, wn = 0;
        size_t nonfinite = 0;
        double worst_by_mode[4] = {0, 0, 0, 0};
        for (int t = 0; t < trials; ++t)
            for (int m = 0; m < 32; ++m)
                for (int n = 0; n < 32; ++n) {
                    double exact = (double)C[(size_t)t * 1024 + m * 32 + n], mag = std::fabs(exact);
                    for (int kk = 0; kk < 16; ++kk) {
                        const double p = (double)A[(size_t)t * 512 + m * 16 + kk] * (double)B[(size_t)t * 512 + kk * 32 + n];
                        exact += p;
                        mag += std::fabs(p);
                        if (chain) {
                            const double p2 = (double)A2[(size_t)t * 512 + m * 16 + kk] * (double)B2[(size_t)t * 512 + kk * 32 + n];
                            exact += p2;
                            mag += std::fabs(p2);
                        }
                    }
                    const double got = (double)D[(size_t)t * 1024 + m * 32 + n];
                    if (!std::isfinite(got)) {
                        ++nonfinite;
                        continue;
                    }
                    const double err = std::fabs(got - exact);
                    const double unit = std::ldexp(mag, -24) + 1e-300;
                    if (err / unit > worst) { wt = t; wm = m; wn = n; }
                    worst = std::fmax(worst, err / unit);
                    worst_by_mode[t % 4] = std::fmax(worst_by_mode[t % 4], err / unit);
                    if (std::fabs(exact) > 0) worst_rel_result = std::fmax(worst_rel_result, err / (std::ldexp(std::fabs(exact), -24)));
                }
        printf("%s: worst |mfma - exact| = %.3f x 2^-24 x (sum|a b| + |c|)   [by mode: %.3f %.3f %.3f %.3f]   "
               "(relative to |exact|: %.1f ulp-halves)   non-finite outputs: %zu\n",
               chain ? "two chained MFMAs (K = 32)" : "one MFMA (K = 16)         ", worst, worst_by_mode[0], worst_by_mode[1], worst_by_mode[2],
               worst_by_mode[3], worst_rel_result, nonfinite);
        printf("  worst case: trial %d (mode %d, paired cancellation %d) row %d col %d: got %.9g, c = %.9g; terms a*b:", wt, wt % 4, wt % 8 >= 4, wm, wn,
               (double)D[(size_t)wt * 1024 + wm * 32 + wn], (double)C[(size_t)wt * 1024 + wm * 32 + wn]);
        double ex = (double)C[(size_t)wt * 1024 + wm * 32 + wn];
        for (int kk = 0; kk < 16; ++kk) {
            const double p = (double)A[(size_t)wt * 512 + wm * 16 + kk] * (double)B[(size_t)wt * 512 + kk * 32 + wn];
            ex += p;
            printf(" %.6g", p);
        }
        printf("  -> exact (first MFMA) %.9g\n", ex);
    }
    return 0;
}
