// m3d_cull_kernels.hip -- spatially culled inlier counting (the production scoring path).
//
// EvaluateModel (include/misc3d/common/ransac.h:626-641) visits every point for every hypothesis,
// but a plane / sphere-shell / cylinder-shell slab of half-width `threshold` only intersects a small
// part of space.  The resident cloud therefore carries a Z-order sorted copy cut into TILES of 512
// consecutive points (one wave: 8 rows of 64) with an axis-aligned bounding box each.  Per chunk of
// hypotheses:
//   cull_k        one wave per tile, one hypothesis per lane: conservative box-vs-slab test written
//                 against the EXACT cut-offs of the scoring record (m3d_fp.hpp) with a margin three
//                 orders of magnitude above the fp64 rounding of the per-point arithmetic; surviving
//                 hypothesis ids are appended to the tile's list (ballot + prefix, ascending).
//   score_list_k  one wave per (tile, list segment): the tile's 512 points stay in VGPRs, the listed
//                 hypothesis records stream through SGPRs, the per-pair arithmetic and the compare are
//                 exactly those of score_k (bit-identical decisions), counts go to counts[h] with
//                 integer atomics (order-free, exact).
// A culled (tile, hypothesis) pair provably contains no inlier, so the counts equal the dense ones;
// tests compare both against the oracle.  RefineModel / tie-break passes keep using the
// original-order arrays, so inlier index lists and serial sums are unaffected by the sort.
#include "m3d_cull_kernels.hpp"

#include "m3d_fp.hpp"

#pragma clang fp contract(off)

namespace m3d {

// ------------------------------------------------------------------------------------------------
// tile boxes: one wave per tile; NaN padding is ignored (fmin/fmax drop NaN); an empty tile gets a
// negative half-extent, which cull_k treats as "never intersects".
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_boxes_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                     const double* __restrict__ sz, uint32_t n_tiles,
                                                     double* __restrict__ boxes) {
    const int lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = 0; j < kTilePoints / 64; ++j) {
        const size_t i = (size_t)tile * kTilePoints + j * 64 + lane;
        const double p[3] = {sx[i], sy[i], sz[i]};
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin(lo[k], p[k]);
            hi[k] = fmax(hi[k], p[k]);
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmax(hi[k], __shfl_xor(hi[k], off, 64));
        }
    if (lane == 0) {
        double* b = boxes + (size_t)tile * 6;
        const bool empty = !(lo[0] <= hi[0]);
        for (int k = 0; k < 3; ++k) {
            const double c = 0.5 * lo[k] + 0.5 * hi[k];
            b[k] = empty ? 0.0 : c;
            // half extent measured from the ROUNDED centre and inflated, so the box contains its points
            b[3 + k] = empty ? -1.0 : fmax(hi[k] - c, c - lo[k]) * (1.0 + 1e-12) + 1e-300;
        }
    }
}

void launch_tile_boxes(const SortedView& s, double* boxes, hipStream_t st) {
    if (s.n_tiles) tile_boxes_k<<<(s.n_tiles + 3) / 4, 256, 0, st>>>(s.x, s.y, s.z, s.n_tiles, boxes);
}

// ------------------------------------------------------------------------------------------------
// conservative box tests.  `true` = the box cannot contain an inlier of this hypothesis.
// ------------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ bool box_culled(const double* __restrict__ rec, const double* __restrict__ box) {
    const double cx = box[0], cy = box[1], cz = box[2], hx = box[3], hy = box[4], hz = box[5];
    if (hx < 0.0) return true;  // empty tile
    if (KIND == 0) {
        // inlier <=> |fl(a x + b y + c z + d)| < T.  Over the box, a x + b y + c z + d ranges over
        // [s - r, s + r]; the rounded per-point value differs from the exact one by < 8 u * mag.
        const double a = rec[0], b = rec[1], c = rec[2], d = rec[3], T = rec[4];
        if (!(T > 0.0)) return true;  // `num < T` can never hold
        const double s = ((a * cx + b * cy) + c * cz) + d;
        const double r = (fabs(a) * hx + fabs(b) * hy) + fabs(c) * hz;
        const double mag = ((fabs(a * cx) + fabs(b * cy)) + (fabs(c * cz) + fabs(d))) + r;
        return fabs(s) - r > T + 1e-12 * (mag + T);
    } else if (KIND == 1) {
        // inlier <=> lo <= |q - c|^2 <= hi
        const double lo = rec[3], hi = rec[4];
        if (!(lo <= hi)) return true;  // NaN cut-offs = "no inlier" record
        const double dx = fabs(rec[0] - cx), dy = fabs(rec[1] - cy), dz = fabs(rec[2] - cz);
        const double nx = fmax(0.0, dx - hx), ny = fmax(0.0, dy - hy), nz = fmax(0.0, dz - hz);
        const double fx = dx + hx, fy = dy + hy, fz = dz + hz;
        const double dmin2 = (nx * nx + ny * ny) + nz * nz;
        const double dmax2 = (fx * fx + fy * fy) + fz * fz;
        return dmax2 * (1.0 + 1e-12) < lo || dmin2 * (1.0 - 1e-12) > hi;
    } else {
        // inlier <=> t_lo <= |(q - c) x (q - ref)|^2 <= t_hi, and |(q - c) x (q - ref)| = dist(q, axis) * |ref - c|
        const double t_lo = rec[6], t_hi = rec[7];
        if (!(t_lo <= t_hi)) return true;
        const double ax = cx - rec[0], ay = cy - rec[1], az = cz - rec[2];
        const double bx = cx - rec[3], by = cy - rec[4], bz = cz - rec[5];
        const double ux = rec[3] - rec[0], uy = rec[4] - rec[1], uz = rec[5] - rec[2];
        const double L2 = (ux * ux + uy * uy) + uz * uz;
        const double kx = ay * bz - az * by, ky = az * bx - ax * bz, kz = ax * by - ay * bx;
        const double tc = (kx * kx + ky * ky) + kz * kz;
        const double dist_c = sqrt(tc / L2);
        const double R = sqrt((hx * hx + hy * hy) + hz * hz);
        const double dmax = dist_c + R, dmin = fmax(0.0, dist_c - R);
        const double tmax = dmax * dmax * L2, tmin = dmin * dmin * L2;
        const double D = (sqrt((ax * ax + ay * ay) + az * az) + sqrt(L2)) + R;  // >= |q - c|, |q - ref|
        const double marg = 1e-12 * ((D * D) * (D * D)) + 1e-12 * tmax;
        return tmax + marg < t_lo || tmin - marg > t_hi;  // NaN anywhere -> false -> kept
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void cull_k(const double* __restrict__ boxes, uint32_t n_tiles,
                                               const double* __restrict__ score,
                                               const uint8_t* __restrict__ valid, uint32_t h_count,
                                               uint32_t h_cap, uint32_t* __restrict__ lists,
                                               uint32_t* __restrict__ list_count) {
    const int lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    double box[6];
    for (int k = 0; k < 6; ++k) box[k] = boxes[(size_t)tile * 6 + k];
    uint32_t* __restrict__ out = lists + (size_t)tile * h_cap;
    uint32_t cnt = 0;
    for (uint32_t h0 = 0; h0 < h_count; h0 += 64) {
        const uint32_t h = h0 + lane;
        bool keep = false;
        if (h < h_count && valid[h]) {
            double rec[kModelStride];
            for (int k = 0; k < kModelStride; ++k) rec[k] = score[(size_t)h * kModelStride + k];
            keep = !box_culled<KIND>(rec, box);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) out[cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = h;
        cnt += (uint32_t)__popcll(m);
    }
    if (lane == 0) list_count[tile] = cnt;
}

void launch_cull(int kind, const SortedView& s, const double* score, const uint8_t* valid, uint32_t h_count,
                 uint32_t h_cap, uint32_t* lists, uint32_t* list_count, hipStream_t st) {
    if (!s.n_tiles) return;
    const dim3 g((s.n_tiles + 3) / 4), b(256);
    if (kind == 0)
        cull_k<0><<<g, b, 0, st>>>(s.boxes, s.n_tiles, score, valid, h_count, h_cap, lists, list_count);
    else if (kind == 1)
        cull_k<1><<<g, b, 0, st>>>(s.boxes, s.n_tiles, score, valid, h_count, h_cap, lists, list_count);
    else
        cull_k<2><<<g, b, 0, st>>>(s.boxes, s.n_tiles, score, valid, h_count, h_cap, lists, list_count);
}

// ------------------------------------------------------------------------------------------------
// counting over the surviving (tile, hypothesis) pairs
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(64) void score_list_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                    const double* __restrict__ sz,
                                                    const double* __restrict__ score,
                                                    const uint32_t* __restrict__ lists,
                                                    const uint32_t* __restrict__ list_count, uint32_t h_cap,
                                                    uint32_t* __restrict__ counts) {
    const uint32_t tile = blockIdx.x;
    const uint32_t n_list = list_count[tile];
    const uint32_t e0 = blockIdx.y * kListSegment;
    if (e0 >= n_list) return;
    const uint32_t e1 = min(n_list, e0 + kListSegment);
    const int lane = threadIdx.x;
    const size_t base = (size_t)tile * kTilePoints + lane;
    constexpr int P = kTilePoints / 64;
    double x[P], y[P], z[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        x[j] = sx[base + 64 * j];
        y[j] = sy[base + 64 * j];
        z[j] = sz[base + 64 * j];
    }
    const uint32_t* __restrict__ lst = lists + (size_t)tile * h_cap;
    constexpr int kUsed = KIND == 2 ? 8 : 5;
    // two-deep software pipeline: list entry e+2 and record e+1 are in flight while e is evaluated
    uint32_t h_cur = lst[e0];
    uint32_t h_nxt = lst[min(e0 + 1, e1 - 1)];
    double rec[kModelStride];
    {
        const double* __restrict__ m = score + (size_t)h_cur * kModelStride;
#pragma unroll
        for (int k = 0; k < kUsed; ++k) rec[k] = m[k];
    }
    uint32_t park_cnt = 0, park_h = 0;
    for (uint32_t e = e0; e < e1; ++e) {
        const uint32_t h_nn = lst[min(e + 2, e1 - 1)];
        const double* __restrict__ mn = score + (size_t)h_nxt * kModelStride;
        double nxt[kModelStride];
#pragma unroll
        for (int k = 0; k < kUsed; ++k) nxt[k] = mn[k];
        uint32_t cnt = 0;
        if (KIND == 0) {
            const double a = rec[0], b = rec[1], c = rec[2], d = rec[3], T = rec[4];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const double num = plane_num(a, b, c, d, x[j], y[j], z[j]);
                cnt += (uint32_t)__popcll(__ballot(num < T));
            }
        } else if (KIND == 1) {
            const double cx = rec[0], cy = rec[1], cz = rec[2], lo = rec[3], hi = rec[4];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const double sv = sphere_s(cx, cy, cz, x[j], y[j], z[j]);
                cnt += (uint32_t)__popcll(__ballot(sv >= lo) & __ballot(sv <= hi));
            }
        } else {
            const double cx = rec[0], cy = rec[1], cz = rec[2], rx = rec[3], ry = rec[4], rz = rec[5];
            const double lo = rec[6], hi = rec[7];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const double tv = line_t(cx, cy, cz, rx, ry, rz, x[j], y[j], z[j]);
                cnt += (uint32_t)__popcll(__ballot(tv >= lo) & __ballot(tv <= hi));
            }
        }
        const uint32_t slot = (e - e0) & 63u;
        park_cnt = ((uint32_t)lane == slot) ? cnt : park_cnt;
        park_h = ((uint32_t)lane == slot) ? h_cur : park_h;
        if (slot == 63u || e + 1 == e1) {  // wave-uniform: flush the parked counts
            if ((uint32_t)lane <= slot && park_cnt) atomicAdd(&counts[park_h], park_cnt);
            park_cnt = 0;
        }
        h_cur = h_nxt;
        h_nxt = h_nn;
#pragma unroll
        for (int k = 0; k < kUsed; ++k) rec[k] = nxt[k];
    }
}

void launch_score_list(int kind, const SortedView& s, const double* score, const uint32_t* lists,
                       const uint32_t* list_count, uint32_t h_cap, uint32_t h_count, uint32_t* counts,
                       hipStream_t st) {
    if (!s.n_tiles || !h_count) return;
    const dim3 g(s.n_tiles, (h_count + kListSegment - 1) / kListSegment), b(64);
    if (kind == 0)
        score_list_k<0><<<g, b, 0, st>>>(s.x, s.y, s.z, score, lists, list_count, h_cap, counts);
    else if (kind == 1)
        score_list_k<1><<<g, b, 0, st>>>(s.x, s.y, s.z, score, lists, list_count, h_cap, counts);
    else
        score_list_k<2><<<g, b, 0, st>>>(s.x, s.y, s.z, score, lists, list_count, h_cap, counts);
}

}  // namespace m3d
