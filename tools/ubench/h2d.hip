// How fast can 240 MB of caller-owned (pageable) memory reach HBM?  hipcc --offload-arch=gfx950 -O2 h2d.hip -o h2d -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 10'000'000, bytes = n * 24;
    std::vector<double> src(3 * n);
    for (size_t i = 0; i < 3 * n; ++i) src[i] = (double)(i % 1000) * 0.001;
    void* d; hipMalloc(&d, bytes);
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); hipMemcpyAsync(d, src.data(), bytes, hipMemcpyHostToDevice, st); double t1 = now(); hipStreamSynchronize(st); double t2 = now();
        printf("pageable hipMemcpyAsync: call %.2f ms, +sync %.2f ms  (%.1f GB/s)\n", t1 - t0, t2 - t0, bytes / (t2 - t0) / 1e6);
    }
    {   // bounding-box pass like m3d_cloud_create
        double t0 = now(); double lo = 1e300, hi = -1e300; size_t nf = 0;
        for (size_t i = 0; i < n; ++i) { const double x = src[3*i], y = src[3*i+1], z = src[3*i+2]; if (std::isfinite(x) && std::isfinite(y) && std::isfinite(z)) { lo = std::min(lo, std::min(x, std::min(y, z))); hi = std::max(hi, std::max(x, std::max(y, z))); ++nf; } }
        printf("host bbox pass: %.2f ms (%g %g %zu)\n", now() - t0, lo, hi, nf);
    }
    for (int nthreads : {1, 2, 4, 8}) {
        const size_t chunk = 8u << 20; const int nbuf = 4;
        char* pin[nbuf]; hipEvent_t ev[nbuf];
        for (int i = 0; i < nbuf; ++i) { hipHostMalloc((void**)&pin[i], chunk, hipHostMallocDefault); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            size_t off = 0; int b = 0; bool used[nbuf] = {false, false, false, false};
            while (off < bytes) {
                const size_t len = std::min(chunk, bytes - off);
                if (used[b]) hipEventSynchronize(ev[b]);
                const char* s = (const char*)src.data() + off;
                if (nthreads == 1) std::memcpy(pin[b], s, len);
                else {
                    std::vector<std::thread> th; const size_t per = (len + nthreads - 1) / nthreads;
                    for (int t = 0; t < nthreads; ++t) { const size_t o = t * per; if (o >= len) break; th.emplace_back([=] { std::memcpy(pin[b] + o, s + o, std::min(per, len - o)); }); }
                    for (auto& t : th) t.join();
                }
                hipMemcpyAsync((char*)d + off, pin[b], len, hipMemcpyHostToDevice, st);
                hipEventRecord(ev[b], st); used[b] = true;
                off += len; b = (b + 1) % nbuf;
            }
            hipStreamSynchronize(st);
            double t1 = now();
            if (rep == 2) printf("staged through %d pinned x %zu MB, %d copy threads: %.2f ms (%.1f GB/s)\n", nbuf, chunk >> 20, nthreads, t1 - t0, bytes / (t1 - t0) / 1e6);
        }
        for (int i = 0; i < nbuf; ++i) { hipHostFree(pin[i]); hipEventDestroy(ev[i]); }
    }
    {
        double t0 = now(); hipHostRegister(src.data(), bytes, hipHostRegisterDefault); double t1 = now();
        hipMemcpyAsync(d, src.data(), bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double t2 = now();
        hipHostUnregister(src.data()); double t3 = now();
        printf("hipHostRegister %.2f ms, copy %.2f ms (%.1f GB/s), unregister %.2f ms\n", t1 - t0, t2 - t1, bytes / (t2 - t1) / 1e6, t3 - t2);
    }
}
