// Host-side check of the candidate cache's certificates (misc3d_amd/csrc/m3d_reg_cache_fp.hpp: the code reg_validate_cached_k
// runs) against an EXACT nearest-neighbour search over ALL target points, in long double: no GPU, no library.
//   winner  => the candidate named is the nearest target point of the whole cloud (its fp64 distance is what the walk returns)
//   nothing => no target point within the search radius
//   else    => lb2 is a lower bound of the nearest point's squared distance
// Scenes: noisy surface patches and clutter, lattices (exact ties), duplicates; coordinates at 0, 1e3 and 3e5; scales 1e-3 .. 1e3;
// reference positions on and off the surface; poses from a tenth of the list's radius to beyond it.
// Built twice more as mutations that MUST fail: -DM3D_CACHE_INFLATE_R (the radius taken 10 % too large) and -DM3D_CACHE_NO_SLACK
// (E(s) = 0, ties and near-ties decided in fp32).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../misc3d_amd/csrc/m3d_reg_cache_fp.hpp"

using namespace m3d;
typedef long double ld;

int main(int argc, char** argv) {
    const int scenes = argc > 1 ? atoi(argv[1]) : 400;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    const int K = 32;
    long long n_q = 0, n_win = 0, n_nothing = 0, n_bound = 0, bad = 0;
    double worst = 0.0;
    for (int sc = 0; sc < scenes; ++sc) {
        const double scale = std::pow(10.0, -3.0 + 6.0 * U(rng));
        const double off = sc % 3 == 0 ? 0.0 : (sc % 3 == 1 ? 1e3 * scale : 3e5 * scale);
        const int kind = sc % 4;   // 0 noisy patch, 1 lattice, 2 duplicates, 3 patch + clutter
        const double sp = 0.0033 * scale;      // point spacing
        std::vector<double> q;
        auto add = [&](double x, double y, double z) {
            q.push_back(x + off);
            q.push_back(y - 0.5 * off);
            q.push_back(z + 0.25 * off);
        };
        const int side = 40;
        for (int i = 0; i < side; ++i)
            for (int j = 0; j < side; ++j) {
                const double x = (i - side / 2) * sp, y = (j - side / 2) * sp;
                if (kind == 1) {
                    for (int l = 0; l < 3; ++l) add(x, y, l * sp);
                } else {
                    const double z = 0.3 * x + 0.6 * sp * N(rng);
                    add(x + 0.3 * sp * N(rng), y + 0.3 * sp * N(rng), z);
                    if (kind == 2 && (i + j) % 3 == 0) add(q[q.size() - 3] - off, q[q.size() - 2] + 0.5 * off, q[q.size() - 1] - 0.25 * off);
                }
            }
        if (kind == 3)
            for (int i = 0; i < 300; ++i) add((U(rng) - 0.5) * side * sp, (U(rng) - 0.5) * side * sp, (U(rng) - 0.5) * 10 * sp);
        const size_t n = q.size() / 3;
        const double r = 9.0 * sp;     // the search radius (C4: 30 mm at 3.3 mm spacing)
        const float r2hi = (float)(r * r) * (1.0f + 0x1p-19f);
        for (int ref = 0; ref < 30; ++ref) {
            // the reference position: near the patch's middle, on or off the surface
            double xa[3] = {(U(rng) - 0.5) * 10 * sp + off, (U(rng) - 0.5) * 10 * sp - 0.5 * off, (ref % 5 == 0 ? 6.0 : 0.5) * sp * N(rng) + 0.25 * off};
            // exact K nearest to xa and the certified radius (what reg_cache_build_k establishes)
            std::vector<std::pair<ld, size_t>> by;
            for (size_t i = 0; i < n; ++i) {
                const ld dx = (ld)q[3 * i] - xa[0], dy = (ld)q[3 * i + 1] - xa[1], dz = (ld)q[3 * i + 2] - xa[2];
                by.push_back({dx * dx + dy * dy + dz * dz, i});
            }
            std::sort(by.begin(), by.end());
            // largest radius whose OPEN ball holds <= K points: just below the (K+1)-th distance, ties included
            ld Rl = std::sqrt((double)by[K].first);
            size_t cnt = 0;
            while (cnt < n && std::sqrt((double)by[cnt].first) < (double)Rl) ++cnt;   // points strictly inside
            if (cnt > (size_t)K) continue;   // (cannot happen: Rl is the (K+1)-th distance)
            double Rd = (double)Rl * (1.0 - 1e-6);
            float R = (float)Rd;
            if ((double)R > Rd) R = std::nextafterf(R, 0.0f);
            float cx[K], cy[K], cz[K];
            size_t cid[K];
            for (int j = 0; j < K; ++j) {
                if ((size_t)j < cnt) {
                    const size_t i = by[j].second;
                    cid[j] = i;
                    cx[j] = (float)(q[3 * i] - xa[0]);
                    cy[j] = (float)(q[3 * i + 1] - xa[1]);
                    cz[j] = (float)(q[3 * i + 2] - xa[2]);
                } else {
                    cid[j] = (size_t)-1;
                    cx[j] = cy[j] = cz[j] = 1e18f;
                }
            }
            for (int qq = 0; qq < 40; ++qq) {
                const double mag = Rd * (qq % 8 == 7 ? 1.3 : 0.1 + 1.0 * U(rng)) * U(rng);
                double d[3] = {N(rng), N(rng), N(rng)};
                const double dn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1e-300;
                const double x[3] = {xa[0] + d[0] / dn * mag, xa[1] + d[1] / dn * mag, xa[2] + d[2] / dn * mag};
                const float ux = (float)(x[0] - xa[0]), uy = (float)(x[1] - xa[1]), uz = (float)(x[2] - xa[2]);
                const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
                const float du = std::sqrt(uu) * (1.0f + 0x1p-20f);
                uint32_t m1 = 0x7F800000u, m2 = 0x7F800000u;
                for (int j = 0; j < K; j += 2) {
                    const uint32_t a = cache_packed_s(cx[j], cy[j], cz[j], ux, uy, uz, (uint32_t)j);
                    const uint32_t b = cache_packed_s(cx[j + 1], cy[j + 1], cz[j + 1], ux, uy, uz, (uint32_t)j + 1u);
                    const uint32_t lo = std::min(a, b), hi = std::max(a, b);
                    m2 = std::min(std::min(std::max(m1, lo), m2), hi);
                    m1 = std::min(m1, lo);
                }
                const CacheVerdict v = cache_certify(m1, m2, R, du, r2hi);
                // the truth
                ld best = 1e300L, second = 1e300L;
                size_t bi = 0;
                for (size_t i = 0; i < n; ++i) {
                    const ld dx = (ld)x[0] - q[3 * i], dy = (ld)x[1] - q[3 * i + 1], dz = (ld)x[2] - q[3 * i + 2];
                    const ld d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < best) {
                        second = best;
                        best = d2;
                        bi = i;
                    } else if (d2 < second) {
                        second = d2;
                    }
                }
                ++n_q;
                if (v.winner) {
                    ++n_win;
                    const size_t w = cid[m1 & kCacheSlotMask];
                    // the winner must be THE nearest point (strictly: an exact tie must not be certified)
                    if (w != bi || !(second > best)) {
                        if (++bad <= 5) fprintf(stderr, "scene %d: winner %zu, true nearest %zu (d2 %.6Le, second %.6Le)\n", sc, w, bi, best, second);
                    }
                } else if (v.nothing) {
                    ++n_nothing;
                    if (best < (ld)r * r) {
                        if (++bad <= 5) fprintf(stderr, "scene %d: 'nothing' but a point at %.6Le < r^2 %.6e\n", sc, best, r * r);
                    }
                } else {
                    ++n_bound;
                    if ((ld)v.lb2 > best) {
                        if (++bad <= 5) fprintf(stderr, "scene %d: bound %.9e above the truth %.9Le\n", sc, (double)v.lb2, best);
                    }
                    if (best > 0) worst = std::max(worst, (double)((ld)v.lb2 / best));
                }
            }
        }
    }
    printf("queries %lld: winner certified %lld, nothing %lld, bounds %lld (tightest bound / truth %.3f); violations %lld\n", n_q, n_win, n_nothing,
           n_bound, worst, bad);
    return bad ? 1 : 0;
}
