// Host-side check (no GPU, no library) that the GUIDED cut-off searches of m3d_fp.hpp -- the bracket narrowed around a
// computed guess before bisecting -- return exactly what the unguided searches over the whole range return: sphere and
// cylinder cut-offs (sphere_cutoffs / cylinder_cutoffs, guided) against the same compositions without guesses, on random
// models over many orders of magnitude, degenerate radii and thresholds included.  Then the defining property itself on
// sampled values next to the cut-offs: dist(q) < thr  <=>  lo <= s(q) <= hi with the reference's own expression.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>

#define M3D_HD inline
#include "../../misc3d_amd/csrc/m3d_fp.hpp"

using namespace m3d;

static bool same(double a, double b) { return f2u(a) == f2u(b) || (a != a && b != b); }

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 20000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    long wrong = 0, prop_wrong = 0, checked = 0;
    for (int it = 0; it < n; ++it) {
        const double scale = std::pow(10.0, -6.0 + 12.0 * u(rng));
        double r = scale * (0.01 + u(rng));
        double thr = scale * std::pow(10.0, -4.0 + 4.5 * u(rng));
        if (it % 97 == 0) r = 0.0;
        if (it % 101 == 0) thr = r * (1.0 + 1e-15);      // the interval reaches down to d = 0
        if (it % 103 == 0) thr = 0.0;                    // nothing is an inlier
        if (it % 107 == 0) r = std::nan("");
        // ---- sphere
        const double ms[4] = {0.1, 0.2, 0.3, r};
        double lo, hi;
        sphere_cutoffs(ms, thr, &lo, &hi);
        double lo0 = std::nan(""), hi0 = std::nan(""), dA, dB;
        if (radial_interval(r, [=](double d) { return sphere_dist_from_d(d, r) < thr; }, &dA, &dB)) {
            double a, b;
            if (preimage_interval(dA, dB, [](double s) { return std::sqrt(s); }, &a, &b)) {
                lo0 = a;
                hi0 = b;
            }
        }
        if (!same(lo, lo0) || !same(hi, hi0)) ++wrong;
        if (lo == lo) {   // the property next to both ends
            for (int k = -3; k <= 3; ++k)
                for (double base : {lo, hi}) {
                    const uint64_t bb = f2u(base);
                    if ((k < 0 && bb < (uint64_t)(-k)) || bb + 8 >= kInfBits) continue;
                    const double s = u2f(bb + (uint64_t)(int64_t)k);
                    const bool in_cut = s >= lo && s <= hi;
                    const bool in_ref = sphere_dist_from_d(std::sqrt(s), r) < thr;
                    ++checked;
                    if (in_cut != in_ref) ++prop_wrong;
                }
        }
        // ---- cylinder: w = (px,py,pz, nx,ny,nz, r)
        const double L0 = std::pow(10.0, -3.0 + 6.0 * u(rng));
        const double w[7] = {0.0, 0.0, 0.0, L0 * 0.6, L0 * 0.0, L0 * 0.8, r};
        double tlo, thi;
        cylinder_cutoffs(w, thr, &tlo, &thi);
        double ref[3], L;
        cylinder_ref(w, ref, &L);
        double tlo0 = std::nan(""), thi0 = std::nan("");
        if (radial_interval(r, [=](double d) { return std::fabs(d - r) < thr; }, &dA, &dB)) {
            double a, b;
            if (preimage_interval(dA, dB, [=](double t) { return std::sqrt(t) / L; }, &a, &b)) {
                tlo0 = a;
                thi0 = b;
            }
        }
        if (!same(tlo, tlo0) || !same(thi, thi0)) ++wrong;
        if (tlo == tlo) {
            for (int k = -3; k <= 3; ++k)
                for (double base : {tlo, thi}) {
                    const uint64_t bb = f2u(base);
                    if ((k < 0 && bb < (uint64_t)(-k)) || bb + 8 >= kInfBits) continue;
                    const double t = u2f(bb + (uint64_t)(int64_t)k);
                    const bool in_cut = t >= tlo && t <= thi;
                    const bool in_ref = std::fabs(std::sqrt(t) / L - r) < thr;
                    ++checked;
                    if (in_cut != in_ref) ++prop_wrong;
                }
        }
    }
    std::printf("models %d: guided != unguided %ld; property checks %ld, wrong %ld\n", n, wrong, checked, prop_wrong);
    if (wrong || prop_wrong) return 1;
    std::printf("all checks passed\n");
    return 0;
}
