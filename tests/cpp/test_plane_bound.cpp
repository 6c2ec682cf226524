// Host-side check of the planes' histogram bound (misc3d_amd/csrc/m3d_bound_fp.hpp: plane_pair_ub, the code plane_bound_k runs):
// no GPU, no library.  Random tiles of 512 points -- a noisy plane patch with clutter, an exact plane, two planes, clutter only --
// over scenes scaled by 1e-2 .. 1e2 and moved up to 1e4 scene sizes from the origin; frames good (the patch's normal, slightly off)
// and arbitrary (any frame must give a valid bound); hypotheses through three points of the tile, the patch's own plane tilted and
// shifted, random planes; thresholds from a third of the noise to ten times it.  For every (tile, hypothesis) pair the bound
// must be >= the exact count of the reference's test |((a x + b y) + c z) + d| < T in fp64.
// Built a second time with -DM3D_BOUND_NO_SLACK (no slack, no outward bins) it MUST report violations (tests/test_screen_bounds.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../misc3d_amd/csrc/m3d_bound_fp.hpp"
#include "../../misc3d_amd/csrc/m3d_fp.hpp"

using namespace m3d;

static void cross(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static void unit(double* v) {
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= n; v[1] /= n; v[2] /= n;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? std::atoi(argv[1]) : 4000;
    std::mt19937_64 rng(20260930);
    std::uniform_real_distribution<double> U01(0.0, 1.0);
    std::normal_distribution<double> N01(0.0, 1.0);
    auto rdir = [&](double* v) { v[0] = N01(rng); v[1] = N01(rng); v[2] = N01(rng); unit(v); };
    long long pairs = 0, violations = 0, cyl_pairs = 0, cyl_sum_ub = 0, cyl_sum_exact = 0;
    for (int t = 0; t < trials; ++t) {
        const double sc = std::pow(10.0, -2.0 + 4.0 * U01(rng));
        double off[3];
        rdir(off);
        const double far = sc * std::pow(10.0, 4.0 * U01(rng));
        for (double& o : off) o *= far;
        const int kindt = (int)(U01(rng) * 5.0);   // 0,1: plane + clutter  2: exact plane  3: two planes  4: clutter only
        const double ext = sc * (0.05 + 0.45 * U01(rng));
        const double sigma = kindt == 2 ? 0.0 : sc * std::pow(10.0, -4.0 + 2.0 * U01(rng));
        double n0[3], a0[3], b0[3], n1[3];
        rdir(n0); rdir(n1);
        double tmp[3] = {1, 0, 0};
        if (std::fabs(n0[0]) > 0.9) { tmp[0] = 0; tmp[1] = 1; }
        cross(n0, tmp, a0); unit(a0); cross(n0, a0, b0);
        const double fplane = kindt == 4 ? 0.0 : (kindt == 2 ? 1.0 : 0.3 + 0.7 * U01(rng));
        std::vector<double> P(512 * 3);
        double mabs = 0;
        for (int i = 0; i < 512; ++i) {
            double p[3];
            const double r = U01(rng);
            if (r < fplane) {
                const double x = ext * (2 * U01(rng) - 1), y = ext * (2 * U01(rng) - 1), z = sigma * N01(rng);
                const double* nn = (kindt == 3 && (i & 1)) ? n1 : n0;
                double aa[3], bb[3];
                if (nn == n0) { for (int k = 0; k < 3; ++k) { aa[k] = a0[k]; bb[k] = b0[k]; } }
                else { cross(n1, tmp, aa); unit(aa); cross(n1, aa, bb); }
                for (int k = 0; k < 3; ++k) p[k] = off[k] + x * aa[k] + y * bb[k] + z * nn[k];
            } else {
                for (int k = 0; k < 3; ++k) p[k] = off[k] + ext * (2 * U01(rng) - 1);
            }
            for (int k = 0; k < 3; ++k) { P[3 * i + k] = p[k]; mabs = std::fmax(mabs, std::fabs(p[k])); }
        }
        // ---- a frame: good (the patch's normal, up to 3 degrees off) or arbitrary
        double e[3], u[3], v[3], c[3] = {0, 0, 0};
        const double fsel = U01(rng);
        const bool precise = fsel < 0.35;   // the patch's own normal, the mean as centre, the rms as scale: what tile_frames_k aims at
        if (precise) { for (int k = 0; k < 3; ++k) e[k] = n0[k]; }
        else if (fsel < 0.7) { double d[3]; rdir(d); for (int k = 0; k < 3; ++k) e[k] = n0[k] + 0.05 * U01(rng) * d[k]; unit(e); }
        else rdir(e);
        double ax[3] = {1, 0, 0};
        if (std::fabs(e[0]) > 0.9) { ax[0] = 0; ax[1] = 1; }
        cross(e, ax, u); unit(u); cross(e, u, v);
        for (int i = 0; i < 512; ++i) for (int k = 0; k < 3; ++k) c[k] += P[3 * i + k] / 512.0;
        if (!precise && U01(rng) < 0.3) for (int k = 0; k < 3; ++k) c[k] = P[3 * (int)(U01(rng) * 511) + k];   // (any centre will do)
        double Um = 0, Vm = 0, Wm = 0, Rm = 0, s2 = 0;
        std::vector<double> w(512);
        for (int i = 0; i < 512; ++i) {
            const double dx = P[3 * i] - c[0], dy = P[3 * i + 1] - c[1], dz = P[3 * i + 2] - c[2];
            w[i] = (dx * e[0] + dy * e[1]) + dz * e[2];
            const double uu = (dx * u[0] + dy * u[1]) + dz * u[2], vv = (dx * v[0] + dy * v[1]) + dz * v[2];
            const double rx = ((dx - w[i] * e[0]) - uu * u[0]) - vv * v[0], ry = ((dy - w[i] * e[1]) - uu * u[1]) - vv * v[1],
                         rz = ((dz - w[i] * e[2]) - uu * u[2]) - vv * v[2];
            Um = std::fmax(Um, std::fabs(uu)); Vm = std::fmax(Vm, std::fabs(vv)); Wm = std::fmax(Wm, std::fabs(w[i]));
            Rm = std::fmax(Rm, std::fmax(std::fabs(rx), std::fmax(std::fabs(ry), std::fabs(rz))));
            s2 += w[i] * w[i] / 512.0;
        }
        double s = std::sqrt(s2) * (precise ? 1.0 : 0.2 + U01(rng));   // (any positive scale will do)
        s = std::fmax(s, 1e-7 * ext);
        const double wlo = -5.0 * s, invd = (double)kBoundBins / (10.0 * s);
        std::vector<uint16_t> cum(kCumStride, 0);
        std::vector<int> hist(kBoundBins + 2, 0);
        for (int i = 0; i < 512; ++i) hist[bound_bin(w[i], wlo, invd)]++;
        for (int k = 0; k < kBoundBins + 2; ++k) cum[k + 1] = (uint16_t)(cum[k] + hist[k]);
        double fr[kFrameStride] = {c[0], c[1], c[2], e[0], e[1], e[2], u[0], u[1], u[2], v[0], v[1], v[2],
                                   Um, Vm, Rm + 1e-13 * (mabs + ext), wlo, invd, Wm, 512.0, 1.0};
        float f[kFrameStride];
        for (int k = 0; k < kFrameStride; ++k) f[k] = frame_to_f32(fr[k], (uint32_t)k);
        // ---- hypotheses
        for (int hq = 0; hq < 48; ++hq) {
            double n[3], p0[3];
            const int how = hq % 3;
            if (how == 0) {   // through three points of the tile
                const int i0 = (int)(U01(rng) * 511), i1 = (int)(U01(rng) * 511), i2 = (int)(U01(rng) * 511);
                double d1[3], d2[3];
                for (int k = 0; k < 3; ++k) { p0[k] = P[3 * i0 + k]; d1[k] = P[3 * i1 + k] - p0[k]; d2[k] = P[3 * i2 + k] - p0[k]; }
                cross(d1, d2, n);
                if (!(std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 1e-300)) continue;
                unit(n);
            } else if (how == 1) {   // the patch's plane, tilted by up to ~1 degree and shifted by up to 3 sigma
                double d[3]; rdir(d);
                for (int k = 0; k < 3; ++k) { n[k] = n0[k] + 0.02 * U01(rng) * d[k]; p0[k] = off[k] + 3.0 * sigma * N01(rng) * n0[k]; }
                unit(n);
            } else {
                rdir(n);
                for (int k = 0; k < 3; ++k) p0[k] = off[k] + ext * (2 * U01(rng) - 1);
            }
            const double dpl = -((n[0] * p0[0] + n[1] * p0[1]) + n[2] * p0[2]);
            const double base = sigma > 0 ? sigma : 1e-3 * sc;
            const double T = base * std::pow(10.0, -0.5 + 1.5 * U01(rng));
            const double rec[5] = {n[0], n[1], n[2], dpl, T};
            int exact = 0;
            for (int i = 0; i < 512; ++i) {
                const double s64 = ((rec[0] * P[3 * i] + rec[1] * P[3 * i + 1]) + rec[2] * P[3 * i + 2]) + rec[3];
                exact += std::fabs(s64) < T;
            }
            const PlaneBoundRec pr = plane_bound_record(rec, mabs);
            const uint32_t ub = plane_pair_ub(pr, c, f, cum.data());
            ++pairs;
            if ((long long)ub < exact) {
                if (violations < 5) std::printf("VIOLATION trial %d hyp %d: ub %u < exact %d (kind %d, sc %.3g, far %.3g, sigma %.3g, T %.3g)\n", t, hq, ub, exact, kindt, sc, far, sigma, T);
                ++violations;
            }
        }
        // ---- cylinder hypotheses (cyl_pair_ub): the patch as a piece of the shell (axis in the patch's plane direction a0, one
        // radius below it, radius and axis slightly off), shells that cut the patch anywhere, axes through the tile itself; radii
        // from a third of the tile's extent to a hundred times it; the exact count is the reference's own test
        // fabs(dist(q, axis) - r) < threshold (m3d_fp.hpp cylinder_distance, ransac.h:435-445), the record what the scoring
        // kernels get (centre, centre + direction, the two cut-offs on t)
        for (int hq = 0; hq < 24; ++hq) {
            double w7[7], dir[3];
            const int how = hq % 3;
            const double rad = ext * std::pow(10.0, -0.5 + 2.5 * U01(rng));
            if (how == 0) {
                double dd[3]; rdir(dd);
                for (int k = 0; k < 3; ++k) dir[k] = a0[k] + 0.03 * U01(rng) * dd[k];
                unit(dir);
                const double shift = 3.0 * sigma * N01(rng), along = ext * (2 * U01(rng) - 1);
                for (int k = 0; k < 3; ++k) w7[k] = off[k] - rad * n0[k] + shift * n0[k] + along * dir[k];
                w7[6] = rad * (1.0 + 0.02 * (2 * U01(rng) - 1));
            } else if (how == 1) {
                rdir(dir);
                double away[3]; rdir(away);
                const double dist = rad * (0.5 + U01(rng));
                for (int k = 0; k < 3; ++k) w7[k] = off[k] + ext * (2 * U01(rng) - 1) + dist * away[k];
                w7[6] = rad;
            } else {
                rdir(dir);
                const int i0 = (int)(U01(rng) * 511);
                for (int k = 0; k < 3; ++k) w7[k] = P[3 * i0 + k] + 0.1 * ext * (2 * U01(rng) - 1);
                w7[6] = rad;
            }
            const double dl = std::pow(10.0, -1.0 + 2.0 * U01(rng));   // (the reference's direction is not a unit vector)
            for (int k = 0; k < 3; ++k) w7[3 + k] = dir[k] * dl;
            const double base = sigma > 0 ? sigma : 1e-3 * sc;
            const double T = base * std::pow(10.0, -0.5 + 2.0 * U01(rng));
            double ref[3], Lc, tlo, thi;
            cylinder_ref(w7, ref, &Lc);
            cylinder_cutoffs(w7, T, &tlo, &thi);
            const double rec[8] = {w7[0], w7[1], w7[2], ref[0], ref[1], ref[2], tlo, thi};
            int exact = 0;
            for (int i = 0; i < 512; ++i) exact += cylinder_distance(w7, P[3 * i], P[3 * i + 1], P[3 * i + 2]) < T;
            const CylBoundRec cr = cyl_bound_record(rec, mabs);
            const uint32_t ub = cyl_pair_ub(cr, c, f, cyl_tile_rho(f), cum.data());
            // ... and the sphere of the same radius on the same side of the patch (sphere_bound_record; the reference's test
            // fabs(|q - centre| - r) < threshold, ransac.h:332-343)
            {
                double sp[4] = {w7[0], w7[1], w7[2], w7[6]}, slo, shi;
                sphere_cutoffs(sp, T, &slo, &shi);
                const double srec[5] = {sp[0], sp[1], sp[2], slo, shi};
                int sexact = 0;
                for (int i = 0; i < 512; ++i) sexact += sphere_distance(sp, P[3 * i], P[3 * i + 1], P[3 * i + 2]) < T;
                const CylBoundRec sr = sphere_bound_record(srec, mabs);
                const uint32_t sub = cyl_pair_ub(sr, c, f, cyl_tile_rho(f), cum.data());
                ++pairs;
                ++cyl_pairs;
                cyl_sum_ub += sub;
                cyl_sum_exact += sexact;
                if ((long long)sub < sexact) {
                    if (violations < 5) std::printf("VIOLATION (sphere) trial %d hyp %d: ub %u < exact %d (kind %d, how %d, sc %.3g, far %.3g, sigma %.3g, T %.3g, r %.3g)\n", t, hq, sub, sexact, kindt, how, sc, far, sigma, T, rad);
                    ++violations;
                }
            }
            ++pairs;
            ++cyl_pairs;
            cyl_sum_ub += ub;
            cyl_sum_exact += exact;
            if ((long long)ub < exact) {
                if (violations < 5) std::printf("VIOLATION (cylinder) trial %d hyp %d: ub %u < exact %d (kind %d, how %d, sc %.3g, far %.3g, sigma %.3g, T %.3g, r %.3g)\n", t, hq, ub, exact, kindt, how, sc, far, sigma, T, rad);
                ++violations;
            }
        }
    }
    std::printf("cylinders and spheres: %lld pairs, sum of bounds %lld, sum of exact counts %lld\n", cyl_pairs, cyl_sum_ub, cyl_sum_exact);
    std::printf("%lld (tile, hypothesis) pairs, violations %lld\n", pairs, violations);
    if (violations == 0) std::printf("all checks passed\n");
    return violations == 0 ? 0 : 1;
}
