// mfma_chain.hip -- does a DEPENDENT chain of v_mfma_f32_32x32x16_f16 (the accumulator of one is SrcC of the next) issue at the
// pipe's rate, and do VALU instructions placed between the chain's MFMAs issue under them?  One wave per SIMD and four.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/mfma_chain.hip -o tools/ubench/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc0 = {0}, acc1 = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // interleaved chains (two accumulators alternate)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
        } else if (MODE == 1) {   // one chain after the other
#pragma unroll
            for (int s = 0; s < 3; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 2) {   // chains, and after every MFMA six VALU instructions that read the OTHER accumulator
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) v[j] = fminf(v[j], acc1[(6 * s + j) & 15]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) v[j] = fminf(v[j], acc0[(6 * s + j) & 15]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {   // the present shape: six MFMAs, then the 36 VALU instructions on their results
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) v[j] = fminf(v[j], acc0[(6 * s + j) & 15]);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) v[j] = fminf(v[j], acc1[(6 * s + j) & 15]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (keep the accumulators small: a fresh start per trip like the scan's)
        acc0 = acc0 * 0.0f + v[0] * 1e-30f;
        acc1 = acc1 * 0.0f + v[1] * 1e-30f;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 2000;   // one workgroup per CU
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, 8 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, cyc, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double c = 0; for (auto x : h) c += x; c /= blocks;
    printf("%-64s W=%d  wave clock ticks per trip %8.1f   wall us per trip per SIMD-wave-slot %.4f\n", name, waves_per_simd, c / iters, ms * 1e3 / iters);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("6 MFMA, two accumulators alternating", w);
        run<1>("6 MFMA, chain of 3 then chain of 3", w);
        run<3>("6 MFMA alternating, then 36 v_min on the results", w);
        run<2>("chain of 3 with 6 v_min (other acc) after each MFMA, twice", w);
    }
    return 0;
}
