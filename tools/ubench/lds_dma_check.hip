// Does global_load_lds_dwordx4 (gfx950) put lane l's 16 bytes at LDS base + 16 l?  (m3d_match_mfma.hip stages its database tiles with it.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_check.hip -o tools/ubench/lds_dma_check && tools/ubench/lds_dma_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void k(const u4* __restrict__ src, u4* __restrict__ out) {
    __shared__ u4 stage[256];
    const int tid = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + blockIdx.x * 256 + tid),
                                     (__attribute__((address_space(3))) void*)(&stage[tid & ~63]), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[blockIdx.x * 256 + tid] = stage[tid];
}
int main() {
    const int n = 256 * 64;
    std::vector<unsigned> h(n * 4), r(n * 4);
    for (int i = 0; i < n * 4; ++i) h[i] = 2654435761u * (unsigned)i + 12345u;
    u4 *d, *o;
    hipMalloc(&d, n * 16);
    hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    k<<<64, 256>>>(d, o);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n * 4; ++i) bad += r[i] != h[i];
    printf("global_load_lds_dwordx4: %d of %d words differ%s\n", bad, n * 4, bad ? "" : " (lane l -> base + 16 l)");
    return bad != 0;
}
