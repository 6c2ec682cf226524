// Host-side check of the SCREEN of the registration validation's neighbour search (sorted_walk32 in
// misc3d_amd/csrc/m3d_reg_kernels.hip): 8-byte list entries -- x, y, z as 16-bit fixed point over the list's 3-cell block
// [-h, 2h), and the 16-bit position of the fp64 entry -- evaluated in fp32 in units of 3h / 65535.  The device arithmetic
// is restated with float operations and fmaf (IEEE single precision, the same operations in the same order), the walk
// itself is replayed -- the merged x-sorted list with its pads, the start at the nearest quarter boundary, two entries per
// side and trip (ONE 16-byte load), the cut on the x-distance of a batch's last entry with m1 taken before the batch, the
// count guards, smallest and second smallest distance -- on random grids: cell edges
// over nine orders of magnitude, origins as far out as the host's admission test lets them be, lists of 1 .. 200 points with
// exact duplicates, points at equal distance from the query and points 1e-7 .. 1e-16 (relative) apart in distance.
// Property: whenever the walk calls a query DECIDED, the fp64 distance of its winner is the minimum of the fp64 distances
// over the whole list, to the last bit.  Also checked: |s - d2| <= E(s) for every entry evaluated, and how often the walk
// decides (it must be nearly always on ordinary data).
// Pure host code: no GPU, no library.  Built and run by tests/test_screen_bounds.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace {
std::mt19937_64 rng(4242);
double uni(double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); }
double logu(double a, double b) { return std::exp(uni(std::log(a), std::log(b))); }

constexpr int kWalkB = 2, kWalkPad = 2 * kWalkB;
struct E32 {      // (the fill kernel's intermediate: fp32 offsets from the cell's corner + the fp64 entry's position)
    float x, y, z;
    uint32_t w;
};
struct E16 {      // the 8-byte entry
    uint16_t x, y, z, w;
};
constexpr float kUnitsPerCell = 21845.0f;   // 65535 / 3: the block [-h, 2h) maps to [0, 65535]
float quant_err_bound(float s) { return fmaf(s, 0x1p-18f, fmaf(2.1f, std::sqrt(s), 2.0f)); }   // E(s), units^2 (walk_err_bound)
struct P3 {
    double x, y, z;
};
long long n_queries = 0, n_decided = 0, n_wrong = 0, n_bound_viol = 0, n_entries = 0, n_visited = 0;
long long n_plain = 0, n_plain_decided = 0;   // scenes without planted ties
double worst_ratio = 0.0;
}  // namespace

int main(int argc, char** argv) {
    const int scenes = argc > 1 ? std::atoi(argv[1]) : 4000;
    for (int sc = 0; sc < scenes; ++sc) {
        // ---- a grid: cell edge, origin, the query's cell
        const double h0 = logu(1e-6, 1e3);
        const double inv_h = 1.0 / h0;                 // GridDesc::inv_h
        const double hc = 1.0 / inv_h;                 // what both kernels call h
        const int ix = 1 + (int)(rng() % 200000), iy = 1 + (int)(rng() % 2000), iz = 1 + (int)(rng() % 2000);
        // origin: the admission test wants (|o| + n / inv_h) 2^-50 <= 0.01 * 2^-24 h; stay inside it, sometimes at its edge
        const double far_max = 0.01 * std::ldexp(1.0, -24) * hc / std::ldexp(1.0, -50);
        const double grid_ext = 200002.0 * hc;
        if (grid_ext >= far_max) continue;
        const double omag = (sc % 5 == 0) ? (far_max - grid_ext) * uni(0.5, 0.999) : hc * logu(1.0, 1e4);
        const double o[3] = {uni(-1, 1) * omag, uni(-1, 1) * omag, uni(-1, 1) * omag};
        auto corner = [&](int i, int axis) { return o[axis] + (double)i * hc; };   // the fill kernel's Ox
        // ---- a list: points of the 3x3x3 block, some duplicated, some placed on spheres around the query
        const bool plain = sc % 2 == 1;   // every other scene: ordinary points, no planted duplicates or ties
        const int n = 1 + (int)(rng() % ((sc % 9 == 0) ? 200 : 40));
        std::vector<P3> pts;
        // the query, inside its cell
        const P3 p = {corner(ix, 0) + uni(0.001, 0.999) * hc, corner(iy, 1) + uni(0.001, 0.999) * hc, corner(iz, 2) + uni(0.001, 0.999) * hc};
        for (int k = 0; k < n; ++k) {
            P3 q;
            const int mode = plain ? 9 : (int)(rng() % 10);
            if (mode == 0 && !pts.empty()) {
                q = pts[rng() % pts.size()];   // exact duplicate
            } else if (mode <= 2 && !pts.empty()) {
                // same distance from the query as an earlier point, up to a relative 1e-7 .. 1e-16 (or exactly, by reflection)
                const P3 r = pts[rng() % pts.size()];
                if (mode == 1) {
                    q = {2 * p.x - r.x, 2 * p.y - r.y, 2 * p.z - r.z};
                } else {
                    const double f = 1.0 + logu(1e-16, 1e-7) * (rng() % 2 ? 1 : -1);
                    q = {p.x + (r.y - p.y) * f, p.y + (r.z - p.z) * f, p.z + (r.x - p.x) * f};
                }
            } else {
                const double spread = (mode == 3) ? logu(1e-6, 1.0) : 1.0;   // clusters tight around the query, too
                q = {p.x + uni(-1, 1) * hc * spread, p.y + uni(-1, 1) * hc * spread, p.z + uni(-1, 1) * hc * spread};
            }
            // keep it inside the block [-h, 2h) around the cell's corner
            const double lo[3] = {corner(ix, 0) - hc, corner(iy, 1) - hc, corner(iz, 2) - hc};
            double* c[3] = {&q.x, &q.y, &q.z};
            bool ok = true;
            for (int a = 0; a < 3; ++a) ok = ok && *c[a] >= lo[a] && *c[a] < lo[a] + 3 * hc * 0.999999;
            if (ok) pts.push_back(q);
        }
        if (pts.empty()) continue;
        // ---- the fill kernel: fp32 offsets from the corner, sorted by fp32 x, sentinels, five offsets
        const double Ox = corner(ix, 0), Oy = corner(iy, 1), Oz = corner(iz, 2);
        std::vector<E32> list;
        for (size_t k = 0; k < pts.size(); ++k)
            list.push_back({(float)(pts[k].x - Ox), (float)(pts[k].y - Oy), (float)(pts[k].z - Oz), (uint32_t)k});
        std::stable_sort(list.begin(), list.end(), [](const E32& a, const E32& b) { return a.x < b.x; });
        const int nt = (int)list.size();
        if (nt > 65535) continue;
        uint32_t offs[5];
        for (int k = 0; k < 5; ++k) {
            const float thr = (float)((double)k * hc * 0.25);
            offs[k] = 0;
            for (const E32& e : list) offs[k] += e.x < thr ? 1u : 0u;
        }
        // quantisation: q = rint(v * S + 21845), S = 21845 / h (monotone in v: the order by fp32 x is kept)
        const float S32 = (float)(21845.0 / hc);
        auto quant = [&](float v) {
            const float f = std::rint(fmaf(v, S32, kUnitsPerCell));
            return (uint16_t)std::min(65535.0f, std::max(0.0f, f));
        };
        std::vector<E16> mem(nt + 2 * kWalkPad);
        for (int k = 0; k < kWalkPad; ++k) {
            mem[k] = {0, 0, 0, 0};                                    // farther than the search radius from any query of the cell
            mem[kWalkPad + nt + k] = {65535, 65535, 65535, 0};
        }
        for (int k = 0; k < nt; ++k) mem[kWalkPad + k] = {quant(list[k].x), quant(list[k].y), quant(list[k].z), (uint16_t)list[k].w};
        // ---- the query side: cell_of_frac + sorted_walk32
        const double fx = (p.x - o[0]) * inv_h, fy = (p.y - o[1]) * inv_h, fz = (p.z - o[2]) * inv_h;
        if ((int)fx != ix || (int)fy != iy || (int)fz != iz) continue;   // (a rounding put it into the next cell: another list)
        const double frx = fx - (double)(int)fx, fry = fy - (double)(int)fy, frz = fz - (double)(int)fz;
        const float ux = fmaf((float)(frx * hc), S32, kUnitsPerCell), uy = fmaf((float)(fry * hc), S32, kUnitsPerCell),
                    uz = fmaf((float)(frz * hc), S32, kUnitsPerCell);
        const double S2 = (21845.0 / hc) * (21845.0 / hc);   // units^2 per length^2
        const int ke = std::min(4, std::max(0, (int)fmaf((float)frx, 4.0f, 0.5f)));
        const int start = (int)offs[ke];
        const E16* base = mem.data() + kWalkPad + start;
        float m1 = INFINITY, m2 = INFINITY;
        uint32_t i1 = 0;
        auto d2_exact = [&](uint32_t k) {
            const double ddx = p.x - pts[k].x, ddy = p.y - pts[k].y, ddz = p.z - pts[k].z;
            return (ddx * ddx + ddy * ddy) + ddz * ddz;
        };
        auto visit = [&](const E16& q, bool real) {
            const float dx = (float)q.x - ux, dy = (float)q.y - uy, dz = (float)q.z - uz;
            const float s = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (real) {   // the bound, entry by entry
                const double d2 = d2_exact(q.w) * S2, E = (double)quant_err_bound(s);
                const double ratio = std::fabs((double)s - d2) / E;
                worst_ratio = std::max(worst_ratio, ratio);
                if (!(ratio <= 1.0)) ++n_bound_viol;
                ++n_visited;
            }
            if (s < m1) i1 = q.w;
            // median of (m1, m2, s) with m1 <= m2
            const float med = std::max(std::min(m1, m2), std::min(std::max(m1, m2), s));
            m2 = med;
            m1 = std::min(m1, s);
        };
        int cr = 0, cl = 0;
        bool ar = start < nt, al = start > 0;
        while (ar || al) {
            // everything behind a batch's last entry is at least (|dx| - 0.6)^2 away IN TRUTH; the winner so far is at most
            // m1 + E(m1) away in truth
            // (for the cut a bound without the square root: 2.1 sqrt(s) <= s / 512 + 565, E(s) <= s (2^-9 + 2^-18) + 567)
            const float thr = fmaf(m1, 1.0f + 0x1p-9f + 0x1p-18f, 567.0f);
            if (ar) {
                for (int k = 0; k < kWalkB; ++k) visit(base[cr + k], start + cr + k < nt);
                const float t = ((float)base[cr + kWalkB - 1].x - ux) - 0.6f;
                cr += kWalkB;
                ar = !(t > 0.0f && !(t * t < thr)) && start + cr < nt;
            }
            if (al) {
                for (int k = 0; k < kWalkB; ++k) visit(base[-1 - (cl + k)], start - 1 - (cl + k) >= 0);
                const float t = (ux - (float)base[-1 - (cl + kWalkB - 1)].x) - 0.6f;
                cl += kWalkB;
                al = !(t > 0.0f && !(t * t < thr)) && cl < start;
            }
            if (cr > nt + kWalkPad || cl > nt + kWalkPad) {
                std::printf("walk ran past its pads\n");
                return 1;
            }
        }
        const float bound = quant_err_bound(m1) + quant_err_bound(m2);
        const bool decided = m2 == INFINITY ? m1 < INFINITY : m2 > m1 + bound * 1.0001f;
        ++n_queries;
        n_entries += nt;
        if (plain) {
            ++n_plain;
            n_plain_decided += decided ? 1 : 0;
        }
        if (decided) {
            ++n_decided;
            double best = INFINITY;
            for (size_t k = 0; k < pts.size(); ++k) best = std::min(best, d2_exact((uint32_t)k));
            if (d2_exact(i1) != best) {
                ++n_wrong;
                if (n_wrong < 5) std::printf("WRONG: scene %d h %.3g n %d winner d2 %.17g true min %.17g\n", sc, hc, nt, d2_exact(i1), best);
            }
        }
    }
    std::printf("queries %lld, decided %lld (%.2f %%), wrong %lld; entries visited %lld of %lld; bound violations %lld, worst |s - d2| / E(s) = %.3f\n",
                n_queries, n_decided, n_queries ? 100.0 * (double)n_decided / (double)n_queries : 0.0, n_wrong, n_visited, n_entries,
                n_bound_viol, worst_ratio);
    std::printf("ordinary scenes: %lld of %lld decided by the screen\n", n_plain_decided, n_plain);
    if (n_wrong || n_bound_viol || n_queries < scenes / 2 || n_plain_decided < n_plain * 0.999) return 1;
    std::printf("all checks passed\n");
    return 0;
}
