// Which streams of a process share a hardware queue?  Creates streams in a given order (n = normal, h = high, l = low priority), then for every
// pair launches one small spinning kernel on each and times the pair: streams on ONE queue run their kernels one after the other (2 T),
// streams on different queues side by side (1 T).  (The lanes' stream layout, m3d_device.cpp, is built on what this prints.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_queues.hip -o tools/ubench/stream_queues && tools/ubench/stream_queues nnnnhhnn
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void spin_k(unsigned long long cycles, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned x = 0;
    while (wall_clock64() - t0 < cycles) x += 1;
    if (x == 0xFFFFFFFFu) *sink = x;
}
int main(int argc, char** argv) {
    const char* order = argc > 1 ? argv[1] : "nnnnnnnn";
    const int n = (int)strlen(order);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: lowest %d, highest %d\n", lo, hi);
    std::vector<hipStream_t> st(n);
    for (int i = 0; i < n; ++i) {
        if (order[i] == 'h') hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
        else if (order[i] == 'l') hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, lo);
        else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    unsigned* sink;
    hipMalloc(&sink, 4);
    const unsigned long long cyc = 2000000ull;   // wall_clock64 ticks at 100 MHz: 20 ms
    auto run = [&](int a, int b) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        spin_k<<<8, 64, 0, st[a]>>>(cyc, sink);
        if (b >= 0) spin_k<<<8, 64, 0, st[b]>>>(cyc, sink);
        hipDeviceSynchronize();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    for (int i = 0; i < n; ++i) run(i, -1);   // touch every stream once
    const double one = run(0, -1);
    printf("one kernel: %.2f ms.  pair (i, j): 1 = side by side, 2 = one after the other\n    ", one);
    for (int j = 0; j < n; ++j) printf(" %c%d", order[j], j);
    printf("\n");
    for (int i = 0; i < n; ++i) {
        printf("%c%d  ", order[i], i);
        for (int j = 0; j < n; ++j) {
            if (j <= i) { printf("  ."); continue; }
            const double t = run(i, j);
            printf("  %d", t > 1.5 * one ? 2 : 1);
        }
        printf("\n");
    }
    return 0;
}
