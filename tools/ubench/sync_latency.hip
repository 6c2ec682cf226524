// How long after a kernel has finished does the host know?  hipStreamSynchronize / hipEventSynchronize against a spin on a
// word the kernel's last instruction stores into pinned host memory.  Build: hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void work_k(volatile uint32_t* flag, uint32_t seq, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (flag && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(const_cast<uint32_t*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void tiny_k(uint32_t* p) { if (p) *p = 1; }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    uint32_t* flag; hipHostMalloc((void**)&flag, 64, hipHostMallocDefault); *flag = 0;
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const long long cyc = 100 * 100;   // ~100 us at 100 MHz of clock64
    for (int mode = 0; mode < 4; ++mode) {
        double tot = 0, launch = 0; const int reps = 200;
        for (int r = 0; r < reps + 10; ++r) {
            const double t0 = now_us();
            uint32_t seq = (uint32_t)(mode * 1000 + r + 1);
            if (mode == 3) work_k<<<1, 64, 0, s>>>(flag, seq, cyc); else work_k<<<1, 64, 0, s>>>(nullptr, 0, cyc);
            if (mode == 1) hipEventRecord(ev, s);
            const double t1 = now_us();
            if (mode == 0) hipStreamSynchronize(s);
            else if (mode == 1) hipEventSynchronize(ev);
            else if (mode == 2) { while (hipStreamQuery(s) == hipErrorNotReady) {} }
            else { while (*(volatile uint32_t*)flag != seq) __builtin_ia32_pause(); }
            const double t2 = now_us();
            if (r >= 10) { tot += t2 - t0; launch += t1 - t0; }
        }
        const char* names[4] = {"hipStreamSynchronize", "hipEventRecord + hipEventSynchronize", "hipStreamQuery spin", "pinned word spin"};
        printf("%-40s launch %.2f us, launch + kernel + wait %.2f us\n", names[mode], launch / reps, tot / reps);
    }
    // cost of enqueueing kernels back to back (host side)
    for (int n : {1, 8}) {
        double tot = 0; const int reps = 200;
        for (int r = 0; r < reps; ++r) {
            const double t0 = now_us();
            for (int k = 0; k < n; ++k) tiny_k<<<1, 64, 0, s>>>(nullptr);
            const double t1 = now_us();
            hipStreamSynchronize(s);
            tot += t1 - t0;
        }
        printf("enqueue of %d tiny kernels: %.2f us per kernel\n", n, tot / reps / n);
    }
    return 0;
}
