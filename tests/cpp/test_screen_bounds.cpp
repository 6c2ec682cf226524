// Host-side check of the fp32 SCREEN of the scoring kernels (misc3d_amd/csrc/m3d_fp.hpp: plane_/sphere_/cylinder_screen_record)
// and of the fp32 BOX tests (…_cull32_record): the device arithmetic is re-stated with fmaf / float ops (IEEE single
// precision, the same operations in the same order as score_screen_k / cull_tiles32_k), on random models, tiles and points
// spread over many orders of magnitude, with points placed on purpose around the cut-offs.  Properties:
//   screen:  whenever the screen calls a point decided (|q| >= h), its verdict equals the exact fp64 test's;
//   cull:    a tile the fp32 box test drops contains no point the exact fp64 test accepts.
// Also reports how often the screen decides (it must be nearly always, or the kernel would not be worth having).
// Pure host code: no GPU, no library.  Built and run by tests/test_screen_bounds.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../misc3d_amd/csrc/m3d_fp.hpp"

using namespace m3d;

namespace {
std::mt19937_64 rng(12345);
double uni(double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); }
double logu(double a, double b) { return std::exp(uni(std::log(a), std::log(b))); }

struct Tile {
    std::vector<double> x, y, z;
    double box[6];     // centre, half extents (as tile_boxes_k computes them)
    float box32[6];    // the fp32 box relative to `origin`
};
// box of the points, as tile_boxes_k does it
void make_box(Tile& t, const double* origin) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < t.x.size(); ++i) {
        const double p[3] = {t.x[i], t.y[i], t.z[i]};
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], p[k]);
            hi[k] = std::fmax(hi[k], p[k]);
        }
    }
    for (int k = 0; k < 3; ++k) {
        const double c = 0.5 * lo[k] + 0.5 * hi[k];
        t.box[k] = c;
        t.box[3 + k] = std::fmax(hi[k] - c, c - lo[k]) * (1.0 + 1e-12) + 1e-300;
        const double cr = c - origin[k];
        const float c32 = (float)cr;
        const double hh = (t.box[3 + k] + std::fabs(cr - (double)c32)) * (1.0 + 1e-6) + 1e-30;
        t.box32[k] = c32;
        t.box32[3 + k] = f32_round_up(hh);
    }
}

long long n_points = 0, n_decided = 0, n_wrong = 0, n_tiles = 0, n_culled = 0, n_cull_wrong = 0;
long long k_points[3] = {0, 0, 0}, k_decided[3] = {0, 0, 0}, k_wrong[3] = {0, 0, 0};   // per kind (valid models only)

// ---- the device arithmetic, restated -----------------------------------------------------------
bool screen_plane(const float* r, float xr, float yr, float zr, bool* inside) {
    const float s = fmaf(r[0], xr, fmaf(r[1], yr, fmaf(r[2], zr, r[4])));
    const float q = fmaf(s, s, -r[3]);
    *inside = std::signbit(q);
    return std::fabs(q) >= r[5];   // h = NaN -> false
}
bool screen_sphere(const float* r, float xr, float yr, float zr, bool* inside) {
    const float w = fmaf(zr, zr, fmaf(yr, yr, xr * xr));                            // once per point and tile
    const float u = fmaf(r[6], zr, fmaf(r[5], yr, fmaf(r[4], xr, w + r[0])));      // |x~ - C|^2 - mid, expanded
    const float q = fmaf(u, u, -r[1]);
    *inside = std::signbit(q);
    return std::fabs(q) >= r[7];
}
bool screen_cylinder(const float* r, float xr, float yr, float zr, bool* inside) {
    const float d1 = fmaf(r[0], xr, fmaf(r[1], yr, fmaf(r[2], zr, r[3])));
    const float d2 = fmaf(r[4], xr, fmaf(r[5], yr, fmaf(r[6], zr, r[7])));
    const float t = fmaf(d2, d2, fmaf(d1, d1, -r[8]));
    const float q = fmaf(t, t, -r[9]);
    *inside = std::signbit(q);
    return std::fabs(q) >= r[10];
}
float cull_value(int kind, const float* c, const float* b) {   // negative = the tile is dropped
    const float bx = b[0], by = b[1], bz = b[2], hx = b[3], hy = b[4], hz = b[5];
    if (kind == 0) {
        const float s = fmaf(c[0], bx, fmaf(c[1], by, fmaf(c[2], bz, c[3])));
        const float r = fmaf(c[4], hx, fmaf(c[5], hy, fmaf(c[6], hz, c[7])));
        return r - std::fabs(s);
    }
    auto orbits = [](float a, float bb) {
        uint32_t x, y;
        std::memcpy(&x, &a, 4);
        std::memcpy(&y, &bb, 4);
        x |= y;
        float o;
        std::memcpy(&o, &x, 4);
        return o;
    };
    if (kind == 1) {
        const float dx = std::fabs(c[0] - bx), dy = std::fabs(c[1] - by), dz = std::fabs(c[2] - bz);
        const float nx = std::fmax(0.0f, dx - hx), ny = std::fmax(0.0f, dy - hy), nz = std::fmax(0.0f, dz - hz);
        const float fx = dx + hx, fy = dy + hy, fz = dz + hz;
        const float dmin2 = fmaf(nz, nz, fmaf(ny, ny, nx * nx)), dmax2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
        return orbits(dmax2 - c[3], c[4] - dmin2);
    }
    const float rb = std::sqrt(fmaf(hz, hz, fmaf(hy, hy, hx * hx))) * 1.000001f;
    const float d1 = fmaf(c[0], bx, fmaf(c[1], by, fmaf(c[2], bz, c[3])));
    const float d2 = fmaf(c[4], bx, fmaf(c[5], by, fmaf(c[6], bz, c[7])));
    const float dist = std::sqrt(fmaf(d2, d2, d1 * d1));
    const float rt = c[8] * rb;
    return orbits((c[9] + rt) - dist, (dist + rt) - c[10]);
}

// one (model, tile): every point through the screen and the exact test; the tile through the box test
void check_pair(int kind, const double* score_rec, bool valid, const Tile& t, const double* origin, double radius, double max_abs) {
    float sr[12];
    if (kind == 0) plane_screen_record(score_rec, t.box, max_abs, sr);
    else if (kind == 1) sphere_screen_record(score_rec, t.box, max_abs, sr);
    else cylinder_screen_record(score_rec, t.box, max_abs, sr);
    float cr[12];
    if (kind == 0) plane_cull32_record(score_rec, valid, origin, radius, max_abs, cr);
    else if (kind == 1) sphere_cull32_record(score_rec, valid, origin, radius, max_abs, cr);
    else cylinder_cull32_record(score_rec, valid, origin, radius, max_abs, cr);
    const bool culled = std::signbit(cull_value(kind, cr, t.box32));
    n_tiles++;
    n_culled += culled;
    bool any_inlier = false;
    for (size_t i = 0; i < t.x.size(); ++i) {
        const double x = t.x[i], y = t.y[i], z = t.z[i];
        bool exact;
        if (kind == 0) exact = plane_num(score_rec[0], score_rec[1], score_rec[2], score_rec[3], x, y, z) < score_rec[4];
        else if (kind == 1) {
            const double sv = sphere_s(score_rec[0], score_rec[1], score_rec[2], x, y, z);
            exact = sv >= score_rec[3] && sv <= score_rec[4];
        } else {
            const double tv = line_t(score_rec[0], score_rec[1], score_rec[2], score_rec[3], score_rec[4], score_rec[5], x, y, z);
            exact = tv >= score_rec[6] && tv <= score_rec[7];
        }
        exact = exact && valid;
        any_inlier = any_inlier || exact;
        const float xr = (float)(x - t.box[0]), yr = (float)(y - t.box[1]), zr = (float)(z - t.box[2]);
        bool inside = false, decided;
        if (kind == 0) decided = screen_plane(sr, xr, yr, zr, &inside);
        else if (kind == 1) decided = screen_sphere(sr, xr, yr, zr, &inside);
        else decided = screen_cylinder(sr, xr, yr, zr, &inside);
        n_points++;
        if (valid) k_points[kind]++;
        if (valid && decided) {
            n_decided++;
            k_decided[kind]++;
            if (inside != exact) {
                if (n_wrong < 5) std::fprintf(stderr, "WRONG kind %d: point (%.17g %.17g %.17g) screen %d exact %d\n", kind, x, y, z, inside, exact);
                n_wrong++;
                k_wrong[kind]++;
            }
        }
    }
    if (culled && any_inlier) {
        if (n_cull_wrong < 5) std::fprintf(stderr, "WRONG CULL kind %d\n", kind);
        n_cull_wrong++;
    }
}
}  // namespace

int main(int argc, char** argv) {
    const int trials = argc > 1 ? std::atoi(argv[1]) : 3000;
    for (int trial = 0; trial < trials; ++trial) {
        const int kind = trial % 3;
        // a scene: scale, offset from the origin, threshold relative to the scale
        const double scale = logu(1e-4, 1e4);
        const double off = (trial % 7 == 0) ? scale * logu(1.0, 1e5) : scale * uni(0.0, 3.0);
        const double shift[3] = {uni(-1, 1) * off, uni(-1, 1) * off, uni(-1, 1) * off};
        const double thr = scale * logu(1e-4, 1e-1);
        // the model
        double par[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool valid = false;
        auto rp = [&](double* p) {
            for (int k = 0; k < 3; ++k) p[k] = shift[k] + uni(-1, 1) * scale;
        };
        if (kind == 0) {
            double p[9];
            rp(p);
            rp(p + 3);
            rp(p + 6);
            valid = plane_minimal_fit(p, p + 3, p + 6, par);
            if (valid) {
                for (int k = 0; k < 4; ++k) rec[k] = par[k];
                rec[4] = plane_cutoff(par, thr);
                const double mag = ((std::fabs(par[0]) + std::fabs(par[1])) + std::fabs(par[2])) * (std::fabs(shift[0]) + std::fabs(shift[1]) + std::fabs(shift[2]) + 2 * scale) + std::fabs(par[3]);
                rec[5] = rec[4] + 1e-12 * (mag + rec[4]);
            }
        } else if (kind == 1) {
            double c[3];
            rp(c);
            const double r = scale * logu(0.05, (trial % 11 == 0) ? 200.0 : 2.0);
            par[0] = c[0];
            par[1] = c[1];
            par[2] = c[2];
            par[3] = r;
            valid = true;
            for (int k = 0; k < 3; ++k) rec[k] = par[k];
            sphere_cutoffs(par, thr, &rec[3], &rec[4]);
            valid = rec[3] <= rec[4];
        } else {
            double c[3], dir[3];
            rp(c);
            do {
                for (int k = 0; k < 3; ++k) dir[k] = uni(-1, 1);
            } while (dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2] < 0.01);
            const double dl = logu(0.1, 10.0);
            for (int k = 0; k < 3; ++k) {
                par[k] = c[k] + ((trial % 13 == 0) ? 50.0 * scale * dir[k] : 0.0);   // axis point far along the axis
                par[3 + k] = dir[k] * dl;
            }
            par[6] = scale * logu(0.05, 2.0);
            double ref[3], L;
            cylinder_ref(par, ref, &L);
            for (int k = 0; k < 3; ++k) {
                rec[k] = par[k];
                rec[3 + k] = ref[k];
            }
            cylinder_cutoffs(par, thr, &rec[6], &rec[7]);
            valid = rec[6] <= rec[7];
        }
        if (!valid) continue;
        // cloud frame: origin and radius as m3d_cloud_create computes them, from the scene's extent
        const double origin[3] = {shift[0], shift[1], shift[2]};
        const double radius = 1.5 * scale * (1.0 + 1e-12);
        const double max_abs = std::fabs(shift[0]) + std::fabs(shift[1]) + std::fabs(shift[2]) + 1.5 * scale;
        // tiles: small boxes inside the scene, some far from the model, some cut by its surface, with points on purpose
        // around the cut-off (relative offsets from 1e-16 to 1e-3 of the threshold, both sides)
        for (int tt = 0; tt < 6; ++tt) {
            Tile t;
            const double ext = scale * logu(1e-3, 0.3);
            double c0[3];
            for (int k = 0; k < 3; ++k) c0[k] = shift[k] + uni(-1, 1) * scale;
            for (int i = 0; i < 96; ++i) {
                double p[3];
                for (int k = 0; k < 3; ++k) p[k] = c0[k] + uni(-1, 1) * ext;
                if (i >= 32) {
                    // move the point to the model's surface +- thr (1 + eps): along the plane normal / radially
                    const double eps = (i & 1 ? 1.0 : -1.0) * logu(1e-16, 1e-3);
                    const double side = (i & 2) ? 1.0 : -1.0;
                    if (kind == 0) {
                        const double nn = std::sqrt(par[0] * par[0] + par[1] * par[1] + par[2] * par[2]);
                        const double s = (par[0] * p[0] + par[1] * p[1] + par[2] * p[2] + par[3]) / nn;
                        const double want = side * thr * (1.0 + eps);
                        for (int k = 0; k < 3; ++k) p[k] += (want - s) * par[k] / nn;
                    } else if (kind == 1) {
                        double d[3] = {p[0] - par[0], p[1] - par[1], p[2] - par[2]};
                        const double dn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                        const double want = par[3] + side * thr * (1.0 + eps);
                        if (dn > 0 && want > 0)
                            for (int k = 0; k < 3; ++k) p[k] = par[k] + d[k] / dn * want;
                    } else {
                        const double L2 = par[3] * par[3] + par[4] * par[4] + par[5] * par[5];
                        double d[3] = {p[0] - par[0], p[1] - par[1], p[2] - par[2]};
                        const double a = (d[0] * par[3] + d[1] * par[4] + d[2] * par[5]) / L2;
                        double foot[3], rad[3];
                        for (int k = 0; k < 3; ++k) {
                            foot[k] = par[k] + a * par[3 + k];
                            rad[k] = p[k] - foot[k];
                        }
                        const double rn = std::sqrt(rad[0] * rad[0] + rad[1] * rad[1] + rad[2] * rad[2]);
                        const double want = par[6] + side * thr * (1.0 + eps);
                        if (rn > 0 && want > 0)
                            for (int k = 0; k < 3; ++k) p[k] = foot[k] + rad[k] / rn * want;
                    }
                    // keep the tile a tile: points that the move threw far away are dropped
                    bool far = false;
                    for (int k = 0; k < 3; ++k) far = far || std::fabs(p[k] - shift[k]) > 1.4 * scale;
                    if (far) continue;
                }
                t.x.push_back(p[0]);
                t.y.push_back(p[1]);
                t.z.push_back(p[2]);
            }
            make_box(t, origin);
            check_pair(kind, rec, valid, t, origin, radius, max_abs);
        }
    }
    std::printf("points %lld, decided by the screen %lld (%.2f %%), wrong %lld; tiles %lld, dropped by the box test %lld, wrongly %lld\n",
                n_points, n_decided, 100.0 * (double)n_decided / (double)(n_points ? n_points : 1), n_wrong, n_tiles, n_culled, n_cull_wrong);
    for (int k = 0; k < 3; ++k)
        std::printf("kind %d: decided %.2f %% of %lld points, wrong %lld\n", k, 100.0 * (double)k_decided[k] / (double)(k_points[k] ? k_points[k] : 1), k_points[k], k_wrong[k]);
    if (n_wrong || n_cull_wrong || n_points < 1000) return 1;
    std::printf("all checks passed\n");
    return 0;
}
