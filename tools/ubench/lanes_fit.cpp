// Lanes without python: T host threads, each with its own resident cloud, each calling m3d_cloud_fit in a loop on ONE device.
//   g++ -O2 -std=c++17 -I include tools/ubench/lanes_fit.cpp -o tools/ubench/lanes_fit -L misc3d_amd/lib -lmisc3d_amd -lpthread -Wl,-rpath,'$ORIGIN/../../misc3d_amd/lib'
//   tools/ubench/lanes_fit [points] [iterations] [probability]
#include <misc3d_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <atomic>
#include <thread>
#include <vector>

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 200000;
    const size_t iters = argc > 2 ? (size_t)atol(argv[2]) : 1000;
    const double prob = argc > 3 ? atof(argv[3]) : 0.9999;
    if (m3d_device_count() < 1) {
        std::printf("no device\n");
        return 1;
    }
    for (int lanes : {1, 4, 8}) {
        m3d_config cfg;
        m3d_get_config(&cfg);
        cfg.lanes = lanes;
        m3d_set_config(&cfg);
        for (int T : {1, 2, 4, 8}) {
            std::vector<std::vector<double>> xyz((size_t)T);
            std::vector<m3d_cloud*> clouds((size_t)T);
            for (int t = 0; t < T; ++t) {
                std::mt19937 g(100 + t);
                std::uniform_real_distribution<double> U(-1.0, 1.0);
                std::normal_distribution<double> G(0.0, 0.002);
                xyz[t].resize(3 * n);
                for (size_t i = 0; i < n; ++i) {
                    xyz[t][3 * i] = U(g);
                    xyz[t][3 * i + 1] = U(g);
                    xyz[t][3 * i + 2] = (i % 5 < 3) ? 1.0 + G(g) : U(g);
                }
            }
            std::vector<std::thread> th;
            std::vector<double> secs((size_t)T, 0.0);
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            std::chrono::steady_clock::time_point t_go;
            const int reps = 400;
            // clouds are created by their own threads (a thread's home lane)
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    clouds[t] = m3d_cloud_create(xyz[t].data(), nullptr, n, 0);
                    std::vector<size_t> inl(n);
                    double par[4];
                    size_t ni = 0;
                    const uint64_t seed = 7;
                    for (int k = 0; k < 20; ++k) m3d_cloud_fit(clouds[t], 0, 0.01, iters, prob, &seed, par, inl.data(), &ni, nullptr);
                    ready.fetch_add(1);
                    while (!go.load()) std::this_thread::yield();   // (all threads start their timed fits together)
                    const auto t0 = std::chrono::steady_clock::now();
                    int bad = 0;
                    size_t ni0 = 0;
                    for (int k = 0; k < reps; ++k) {
                        const int rc = m3d_cloud_fit(clouds[t], 0, 0.01, iters, prob, &seed, par, inl.data(), &ni, nullptr);
                        if (k == 0) ni0 = ni;
                        bad += rc != 1 || ni != ni0 || ni < n / 2;
                    }
                    if (bad) std::printf("   thread %d: %d of %d fits FAILED or differ (last error: %s)\n", t, bad, reps, m3d_last_error());
                    secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                });
            while (ready.load() < T) std::this_thread::yield();
            t_go = std::chrono::steady_clock::now();
            go.store(true);
            for (auto& x : th) x.join();
            const double worst = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_go).count();   // wall clock of all
            for (int t = 0; t < T; ++t) m3d_cloud_destroy(clouds[t]);
            std::printf("lanes %d  threads %d: %8.0f fits/s  (%.3f ms per fit and thread)\n", lanes, T, T * reps / worst, worst / reps * 1e3);
        }
    }
    return 0;
}
