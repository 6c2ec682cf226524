// Micro-benchmarks that decide the scoring-kernel design on gfx950:
//   (1) fp64 VALU issue rate (v_mul_f64 / v_add_f64 / v_cmp_lt_f64, no FMA)
//   (2) v_mfma_f64_16x16x4_f64 issue rate
//   (3) both streams in one wave (do the matrix and vector pipes overlap for fp64?)
// Prints ops/clk/SIMD-equivalents derived from wall time at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5, c = 1.0000001, d = 0.9999999;
    double v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    unsigned cnt = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2) {  // 8 independent chains x (mul, add) = 16 VALU fp64 ops
            v0 = v0 * c + d; v1 = v1 * c + d; v2 = v2 * c + d; v3 = v3 * c + d;
            v4 = v4 * c + d; v5 = v5 * c + d; v6 = v6 * c + d; v7 = v7 * c + d;
        }
        if (MODE == 3) {  // 8 compares + ballot popcount
            cnt += __popcll(__ballot(v0 < b)) + __popcll(__ballot(v1 < b)) + __popcll(__ballot(v2 < b)) +
                   __popcll(__ballot(v3 < b)) + __popcll(__ballot(v4 < b)) + __popcll(__ballot(v5 < b)) +
                   __popcll(__ballot(v6 < b)) + __popcll(__ballot(v7 < b));
            b += 1e-12;
        }
        if (MODE == 1 || MODE == 2) {  // 4 independent MFMAs
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + acc0[0] + acc1[1] + acc2[2] + acc3[3] + cnt;
}

template <int MODE>
double run(int blocks, int iters) {
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

int main() {
    const int blocks = 256 * 8, iters = 20000;  // 8 workgroups (32 waves) per CU
    const double waves = blocks * 4.0;
    const double simds = 1024.0;
    double ms;
    ms = run<0>(blocks, iters);
    printf("VALU  mul+add : %.3f ms  -> %.2f cycles@2.4GHz per wave-instruction per SIMD\n", ms,
           ms * 1e-3 * 2.4e9 / (waves * iters * 16.0 / simds));
    ms = run<3>(blocks, iters);
    printf("VALU  cmp+bcnt: %.3f ms  -> %.2f cycles@2.4GHz per compare per SIMD\n", ms,
           ms * 1e-3 * 2.4e9 / (waves * iters * 8.0 / simds));
    ms = run<1>(blocks, iters);
    printf("MFMA f64 16x16x4: %.3f ms -> %.2f cycles@2.4GHz per MFMA per SIMD  (%.1f TFLOP/s)\n", ms,
           ms * 1e-3 * 2.4e9 / (waves * iters * 4.0 / simds), waves * iters * 4.0 * 2048 / (ms * 1e-3) / 1e12);
    double ms2 = run<2>(blocks, iters);
    printf("both in one wave: %.3f ms (sum of parts %.3f ms)\n", ms2, run<0>(blocks, iters) + run<1>(blocks, iters));
    return 0;
}
