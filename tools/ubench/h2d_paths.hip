// Host -> device upload of a caller's PAGEABLE array, three ways ("register the caller's array with
// hipHostRegister for clouds >= 64 MB and report"): (a) hipMemcpy straight from the pageable pages (the runtime stages),
// (b) hipHostRegister + one async copy + hipHostUnregister, (c) our own staging: worker threads copy slices into
// page-locked buffers, the copy engine ships slice k while slice k + 1 is being staged.
//   hipcc -O2 --offload-arch=gfx950 -o h2d_paths h2d_paths.hip -lpthread && ./h2d_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t sizes[] = {52800000, 105600000, 240000000};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const size_t slice = 8u << 20;
    const int n_stage = 4;
    void* stage[n_stage];
    hipEvent_t ev[n_stage];
    for (int i = 0; i < n_stage; ++i) { CK(hipHostMalloc(&stage[i], slice, hipHostMallocDefault)); CK(hipEventCreate(&ev[i])); }
    for (size_t bytes : sizes) {
        char* h = (char*)malloc(bytes);
        memset(h, 1, bytes);   // touched pages, like a caller's filled array
        void* d;
        CK(hipMalloc(&d, bytes));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now_ms();
            CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
            const double a = now_ms() - t0;
            t0 = now_ms();
            CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
            const double reg = now_ms() - t0;
            CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            const double cp = now_ms() - t0 - reg;
            CK(hipHostUnregister(h));
            const double b = now_ms() - t0;
            double c[3];
            int ci = 0;
            for (int threads : {1, 2, 4}) {
                t0 = now_ms();
                const size_t n_sl = (bytes + slice - 1) / slice;
                for (size_t k = 0; k < n_sl; ++k) {
                    const int sl = (int)(k % n_stage);
                    if (k >= (size_t)n_stage) CK(hipEventSynchronize(ev[sl]));
                    const size_t off = k * slice, len = bytes - off < slice ? bytes - off : slice;
                    if (threads == 1) memcpy(stage[sl], h + off, len);
                    else {
                        std::vector<std::thread> th;
                        const size_t per = (len + threads - 1) / threads;
                        for (int t = 0; t < threads; ++t) {
                            const size_t o = (size_t)t * per;
                            if (o >= len) break;
                            const size_t l = len - o < per ? len - o : per;
                            th.emplace_back([=] { memcpy((char*)stage[sl] + o, h + off + o, l); });
                        }
                        for (auto& x : th) x.join();
                    }
                    CK(hipMemcpyAsync((char*)d + off, stage[sl], len, hipMemcpyHostToDevice, st));
                    CK(hipEventRecord(ev[sl], st));
                }
                CK(hipStreamSynchronize(st));
                c[ci++] = now_ms() - t0;
            }
            printf("%6.1f MB  (a) pageable hipMemcpy %6.2f ms  (b) register %5.2f + copy %5.2f + unregister = %6.2f ms  "
                   "(c) own staging, 1/2/4 threads %6.2f %6.2f %6.2f ms\n", bytes / 1e6, a, reg, cp, b, c[0], c[1], c[2]);
        }
        CK(hipFree(d));
        free(h);
    }
    return 0;
}
