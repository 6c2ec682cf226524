// m3d_cull_kernels.hip -- spatially culled inlier counting (the production scoring path).
//
// EvaluateModel (include/misc3d/common/ransac.h:626-641) visits every point for every hypothesis,
// but a plane / sphere-shell / cylinder-shell slab of half-width `threshold` only intersects a small
// part of space.  The resident cloud therefore carries a Hilbert-sorted copy cut into TILES of 512
// consecutive points (one wave: 8 rows of 64) with an axis-aligned bounding box each.  Per chunk of
// hypotheses:
//   cull_tiles32_k  one wave per (64 tiles, a few groups of 64 hypotheses): lane = tile with its fp32 box in registers,
//                   the hypotheses' fp32 box-test records stream through scalar loads, two per packed instruction.
//                   CONSERVATIVE box-vs-slab test: every rounding is in the margin (m3d_fp.hpp), the sign bit of the
//                   test value is the verdict.  Result: one 64-bit mask per (tile, group) and, per hypothesis, the
//                   number of tiles it can touch.  (cull_tiles_k: the same in fp64, m3d_config.cull_fp32 = 0.)
//   keep_mask_k     bound-and-prune: hypotheses whose touched tiles hold fewer points than the best
//                   inlier count of EARLIER hypotheses can neither beat nor tie it in the sequential
//                   replay (ransac.h:595-596); their bits are masked out.
//   score_screen_k  one wave per (tile, range of groups): the tile's 512 points stay in VGPRs as fp32 offsets from the
//                   box centre; a packed-fp32 pass with a rigorous rounding bound decides all but the points within
//                   the bound of the cut-off, and a (tile, hypothesis) pair with such a point is counted again by
//                   the exact fp64 code (tile_count: the per-pair arithmetic and compare of score_k, bit-identical
//                   decisions); counts go to counts[h] with integer atomics (order-free, exact).
//                   (score_mask_k: fp64 only, m3d_config.score_fp32_screen = 0.)
//   cull_lead_k     the head of a fit's first chunk in ONE launch: cull_tiles32_k's workgroups for the chunk's groups and
//                   score_screen_k's for its leading hypotheses, whose best count prunes the rest -- the lead pass runs
//                   the box tests of its own (tile, group) pairs itself (cull32_one: the same arithmetic, the same bits).
// A culled (tile, hypothesis) pair provably contains no inlier, so the counts of unpruned hypotheses
// equal the dense ones; tests compare both paths against the oracle.  RefineModel / tie-break passes
// keep using the original-order arrays, so inlier index lists and serial sums do not see the sort.
#include "m3d_cull_kernels.hpp"
#include "m3d_poison.hpp"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

#include "m3d_config.hpp"
#include "m3d_fp.hpp"
#include "m3d_tile_count.hpp"

#pragma clang fp contract(off)

namespace m3d {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// tile boxes: one wave per tile; NaN padding is ignored (fmin/fmax drop NaN); an empty tile gets a
// negative half-extent, which the box test treats as "never intersects".
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_boxes_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                     const double* __restrict__ sz, uint32_t n_tiles,
                                                     double* __restrict__ boxes, double ox, double oy, double oz,
                                                     float* __restrict__ tile_f32) {
    const int lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double pts[kTilePoints / 64][3];
#pragma unroll
    for (int j = 0; j < kTilePoints / 64; ++j) {
        const size_t i = (size_t)tile * kTilePoints + j * 64 + lane;
        const double p[3] = {sx[i], sy[i], sz[i]};
        for (int k = 0; k < 3; ++k) {
            pts[j][k] = p[k];
            lo[k] = fmin(lo[k], p[k]);
            hi[k] = fmax(hi[k], p[k]);
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmax(hi[k], __shfl_xor(hi[k], off, 64));
        }
    // score_screen_k's view of the tile: offsets from the box centre in fp32 (the subtraction in fp64: as accurate as
    // fp32 gets), rows 2j and 2j + 1 side by side; a tile with a non-finite offset (NaN padding, a coordinate beyond the
    // fp32 range) is not screened (box slot 6)
    bool finite = true;
    if (tile_f32) {
        const bool empty_t = !(lo[0] <= hi[0]);
        f32x2* __restrict__ t2 = reinterpret_cast<f32x2*>(tile_f32 + (size_t)tile * kTileF32Floats);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double c = empty_t ? 0.0 : 0.5 * lo[k] + 0.5 * hi[k];   // (the centre as stored below)
#pragma unroll
            for (int j = 0; j < kTilePoints / 128; ++j) {
                const f32x2 v = {(float)(pts[2 * j][k] - c), (float)(pts[2 * j + 1][k] - c)};
                t2[(k * (kTilePoints / 128) + j) * 64 + lane] = v;
                finite = finite && (v.x * 0.0f == 0.0f) && (v.y * 0.0f == 0.0f);
            }
        }
    }
    const bool screenable = tile_f32 && __ballot(!finite) == 0ull;
    if (lane == 0) {
        double* b = boxes + (size_t)tile * kBoxStride;
        const bool empty = !(lo[0] <= hi[0]);
        for (int k = 0; k < 3; ++k) {
            const double c = 0.5 * lo[k] + 0.5 * hi[k];
            b[k] = empty ? 0.0 : c;
            // half extent measured from the ROUNDED centre and inflated, so the box contains its points
            b[3 + k] = empty ? -1.0 : fmax(hi[k] - c, c - lo[k]) * (1.0 + 1e-12) + 1e-300;
        }
        b[6] = screenable ? 1.0 : 0.0;   // (SortedView::tile_f32)
        b[7] = 0.0;
        // the same box in fp32, relative to the cloud's origin (cull_tiles32_k): centre rounded to nearest, half extents
        // enlarged by the rounding of the centre and rounded outwards, so that the fp32 box contains the fp64 one
        float* f = reinterpret_cast<float*>(b + 8);
        const double o[3] = {ox, oy, oz};
        for (int k = 0; k < 3; ++k) {
            const double cr = b[k] - o[k];
            const float c32 = (float)cr;
            const double hh = (b[3 + k] + fabs(cr - (double)c32)) * (1.0 + 1e-6) + 1e-30;
            f[k] = empty ? 0.0f : c32;
            f[3 + k] = empty ? -1.0f : f32_round_up(hh);
        }
        // ... and the fp32 box's bounding radius as the box tests compute it (cull32_body), for the kernel that reads a tile's box
        // through the scalar unit and has no lane to spare for it (cull_hyp32_k)
        const float hx = empty ? 0.0f : f[3], hy = empty ? 0.0f : f[4], hz = empty ? 0.0f : f[5];
        f[6] = __builtin_sqrtf(__builtin_fmaf(hz, hz, __builtin_fmaf(hy, hy, hx * hx))) * 1.000001f;
        f[7] = 0.0f;
    }
}

void launch_tile_boxes(const SortedView& s, double* boxes, hipStream_t st) {
    if (s.n_tiles) tile_boxes_k<<<(s.n_tiles + 3) / 4, 256, 0, st>>>(s.x, s.y, s.z, s.n_tiles, boxes, s.origin[0], s.origin[1], s.origin[2], s.tile_f32);
}

// ------------------------------------------------------------------------------------------------
// TOMBSTONES.  A segmentation round in the clutter removes ~0.5 % of the cloud; the stable partition of the sorted copy
// rewrote all of it (count + write + fresh tile boxes: 24 us of a 116 us round on 1 M points).  Counts do not depend on
// the order or the presence of points that are nobody's inlier, so such a round KILLS its inliers in place instead:
// x := NaN in the fp64 copy (the exact code's `|s| < T` is false for a NaN) and in the tile's fp32 offsets (score_screen_k
// masks the lane's bit).  Boxes go stale -- they still contain every live point, which is all the box tests need.
// One wave per tile; a tile the plane's slab misses (the box test of the scoring pass, same record) is not read.
// The driver compacts for real (compact_write_k mode 3 drops NaN) when the dead are an eighth of the copy.
// *total accumulates the kills of all launches (checked by the driver against the inlier lists, later).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void poison_plane_inliers_k(PoisonJob job) {
    poison_tile(job, blockIdx.x * 4u + (threadIdx.x >> 6), (int)(threadIdx.x & 63));
}
PoisonJob make_poison_job(const SortedView& s, const double* model, double thr, uint32_t* total) {
    PoisonJob j;
    j.sx = const_cast<double*>(s.x);
    j.sy = s.y;
    j.sz = s.z;
    j.boxes = s.boxes;
    j.n_tiles = s.n_tiles;
    j.tile_f32 = const_cast<float*>(s.tile_f32);
    j.model = model;
    j.thr = thr;
    j.max_abs = s.max_abs;
    j.total = total;
    return j;
}
void launch_poison_plane_inliers(const PoisonJob& job, hipStream_t st) {
    if (job.n_tiles) poison_plane_inliers_k<<<(job.n_tiles + 3) / 4, 256, 0, st>>>(job);
}

// ------------------------------------------------------------------------------------------------
// conservative box tests.  `true` = the box cannot contain an inlier of this hypothesis.
// ------------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ bool box_culled(const double* __restrict__ rec, const double* __restrict__ box) {
    // branch-free: every sub-condition is evaluated and OR-ed (the record conditions are wave-uniform, the empty-box
    // condition per lane; early returns cost the loop of cull_tiles_k more in exec-mask bookkeeping than they save)
    const double cx = box[0], cy = box[1], cz = box[2], hx = box[3], hy = box[4], hz = box[5];
    const bool empty = hx < 0.0;  // empty tile
    if (KIND == 0) {
        // inlier <=> |fl(a x + b y + c z + d)| < T.  Over the box, a x + b y + c z + d ranges over
        // [s - r, s + r]; the rounded per-point value differs from the exact one by < 8 u * mag.
        (void)cx; (void)cy; (void)cz; (void)hy; (void)hz; (void)empty;
        return plane_box_culled(rec, box);   // (m3d_poison.hpp: shared with the tombstone pass)
    } else if (KIND == 1) {
        // inlier <=> lo <= |q - c|^2 <= hi
        const double lo = rec[3], hi = rec[4];
        const double dx = fabs(rec[0] - cx), dy = fabs(rec[1] - cy), dz = fabs(rec[2] - cz);
        const double nx = fmax(0.0, dx - hx), ny = fmax(0.0, dy - hy), nz = fmax(0.0, dz - hz);
        const double fx = dx + hx, fy = dy + hy, fz = dz + hz;
        const double dmin2 = (nx * nx + ny * ny) + nz * nz;
        const double dmax2 = (fx * fx + fy * fy) + fz * fz;
        // !(lo <= hi): NaN cut-offs = "no inlier" record
        return empty | !(lo <= hi) | (dmax2 * (1.0 + 1e-12) < lo) | (dmin2 * (1.0 - 1e-12) > hi);
    } else {
        // inlier <=> t_lo <= |(q - c) x (q - ref)|^2 <= t_hi, and |(q - c) x (q - ref)| = dist(q, axis) * |ref - c|
        const double t_lo = rec[6], t_hi = rec[7];
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
        // NaN anywhere in the last two -> false -> kept
        return empty | !(t_lo <= t_hi) | (tmax + marg < t_lo) | (tmin - marg > t_hi);
    }
}

// One wave = 64 TILES (one per lane, its box in VGPRs, loaded once, coalesced) x a few groups of 64 hypotheses whose
// records stream through SGPRs (scalar loads, the next record in flight) -- the decomposition of the scoring kernel.
// masks[tile * n_groups + group] = hypotheses of the group that may have inliers in the tile: lane `tile` ORs bit h
// into its own word as hypothesis h goes by (no transposition step); ub[h] += number of such tiles (the ballot's
// popcount, parked in lane h, one vector atomic per group).  Invalid and padding hypotheses carry "no inlier"
// records (minimal_fit_k), which every test below rejects, so `valid` is not read.
// Round 1's layout was the transpose (lane = hypothesis, boxes through scalar loads, one ballot + one 8-byte store per
// (tile, group)): 24 us for ~7 us of arithmetic at 10 000 hypotheses x 1954 tiles, whatever the launch geometry.
template <int KIND>
__global__ __launch_bounds__(64) void cull_tiles_k(const double* __restrict__ boxes, uint32_t n_tiles,
                                                    const double* __restrict__ score, uint32_t n_groups,
                                                    uint32_t groups_per_wave, unsigned long long* __restrict__ masks,
                                                    uint32_t* __restrict__ ub, uint32_t group_begin, uint32_t group_end) {
    const int lane = threadIdx.x;
    const uint32_t tile = blockIdx.x * 64u + (uint32_t)lane;
    const bool tile_ok = tile < n_tiles;
    double box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) box[k] = tile_ok ? boxes[(size_t)tile * kBoxStride + k] : (k >= 3 ? -1.0 : 0.0);   // (hx < 0: empty)
    const uint32_t g0 = group_begin + blockIdx.y * groups_per_wave;
    const uint32_t g1 = min(group_end, g0 + groups_per_wave);
    constexpr int kUsed = KIND == 0 ? 6 : (KIND == 1 ? 5 : 8);   // record words the box test reads
    for (uint32_t g = g0; g < g1; ++g) {
        uint32_t w[2] = {0u, 0u}, ubv = 0u;
        const double* __restrict__ rp = score + (size_t)g * 64u * kModelStride;
        auto load_rec = [&](double (&r)[kModelStride], uint32_t hh) {   // hh wave-uniform -> scalar loads
            const double* __restrict__ q = rp + (size_t)hh * kModelStride;
#pragma unroll
            for (int k = 0; k < kUsed; ++k) r[k] = q[k];
        };
        auto test = [&](const double (&r)[kModelStride], uint32_t hh, int half, uint32_t b) {
            const bool keep = !box_culled<KIND>(r, box);
            const unsigned long long m = __ballot(keep);
            w[half] |= keep ? (1u << b) : 0u;
            ubv = ((uint32_t)lane == hh) ? (uint32_t)__popcll(m) : ubv;
        };
        // two hypotheses per trip, their records in alternating register sets (no register-to-register copies: the
        // scalar unit, not the VALU, was what the first version of this loop kept busy); the next record is always in flight
        double ra[kModelStride] = {0, 0, 0, 0, 0, 0, 0, 0}, rb[kModelStride] = {0, 0, 0, 0, 0, 0, 0, 0};
        load_rec(ra, 0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            for (uint32_t b = 0; b < 32u; b += 2u) {
                const uint32_t hh = (uint32_t)half * 32u + b;
                load_rec(rb, hh + 1u);
                test(ra, hh, half, b);
                load_rec(ra, min(hh + 2u, 63u));
                test(rb, hh + 1u, half, b + 1u);
            }
        }
        if (tile_ok) masks[(size_t)tile * n_groups + g] = ((unsigned long long)w[1] << 32) | w[0];
        if (ub && ubv) atomicAdd(&ub[g * 64u + (uint32_t)lane], ubv);
    }
}

// ------------------------------------------------------------------------------------------------
// cull_tiles32_k: the box tests in fp32 (records and margins: m3d_fp.hpp, "fp32 records of the BOX tests").
// Same decomposition as cull_tiles_k -- lane = tile, its fp32 box (relative to the cloud's origin) in six VGPRs, the
// hypothesis records through scalar loads -- but TWO hypotheses per packed instruction (the records are stored
// pairwise interleaved, so a scalar load delivers the operand pairs), the test value's SIGN BIT is the verdict
// (negative = dropped; v_alignbit shifts it into the lane's mask word, no compare / select), and no fp64 sqrt or
// divide for the cylinder.  Plane: 6 v_pk_fma per pair of hypotheses + 2 v_sub + 2 v_alignbit + the ballots of the
// per-hypothesis tile count = ~9 VALU instructions per hypothesis against ~43 of the fp64 kernel.
// ------------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void cull32_pair(const float* __restrict__ rp /* 24 floats, wave-uniform */, float bx, float by,
                                            float bz, float hx, float hy, float hz, float rb, float& ta, float& tb) {
    const f32x2 BX = {bx, bx}, BY = {by, by}, BZ = {bz, bz};
    if (KIND == 0) {
        const f32x2 A = {rp[0], rp[1]}, B = {rp[2], rp[3]}, C = {rp[4], rp[5]}, D = {rp[6], rp[7]};
        const f32x2 AA = {rp[8], rp[9]}, AB = {rp[10], rp[11]}, AC = {rp[12], rp[13]}, K = {rp[14], rp[15]};
        const f32x2 HX = {hx, hx}, HY = {hy, hy}, HZ = {hz, hz};
        const f32x2 s = __builtin_elementwise_fma(A, BX, __builtin_elementwise_fma(B, BY, __builtin_elementwise_fma(C, BZ, D)));
        const f32x2 r = __builtin_elementwise_fma(AA, HX, __builtin_elementwise_fma(AB, HY, __builtin_elementwise_fma(AC, HZ, K)));
        ta = r.x - __builtin_fabsf(s.x);
        tb = r.y - __builtin_fabsf(s.y);
    } else if (KIND == 1) {
        // dmin^2 / dmax^2 of the box to the centre, per hypothesis (no packed abs / max: plain VALU)
        auto one = [&](int i) -> float {
            const float dx = __builtin_fabsf(rp[0 + i] - bx), dy = __builtin_fabsf(rp[2 + i] - by), dz = __builtin_fabsf(rp[4 + i] - bz);
            const float nx = __builtin_fmaxf(0.0f, dx - hx), ny = __builtin_fmaxf(0.0f, dy - hy), nz = __builtin_fmaxf(0.0f, dz - hz);
            const float fx = dx + hx, fy = dy + hy, fz = dz + hz;
            const float dmin2 = __builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx));
            const float dmax2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
            const float t1 = dmax2 - rp[6 + i], t2 = rp[8 + i] - dmin2;   // loM, hiM
            return __uint_as_float(__float_as_uint(t1) | __float_as_uint(t2));   // negative if either is
        };
        ta = one(0);
        tb = one(1);
    } else {
        const f32x2 E1X = {rp[0], rp[1]}, E1Y = {rp[2], rp[3]}, E1Z = {rp[4], rp[5]}, D1 = {rp[6], rp[7]};
        const f32x2 E2X = {rp[8], rp[9]}, E2Y = {rp[10], rp[11]}, E2Z = {rp[12], rp[13]}, D2 = {rp[14], rp[15]};
        const f32x2 d1 = __builtin_elementwise_fma(E1X, BX, __builtin_elementwise_fma(E1Y, BY, __builtin_elementwise_fma(E1Z, BZ, D1)));
        const f32x2 d2 = __builtin_elementwise_fma(E2X, BX, __builtin_elementwise_fma(E2Y, BY, __builtin_elementwise_fma(E2Z, BZ, D2)));
        const f32x2 tt = __builtin_elementwise_fma(d2, d2, d1 * d1);
        auto one = [&](float t, int i) -> float {
            const float dist = __builtin_sqrtf(t);
            const float rt = rp[16 + i] * rb;                  // |L| x the box's bounding radius
            const float t1 = (rp[18 + i] + rt) - dist;          // sHiM
            const float t2 = (dist + rt) - rp[20 + i];          // sLoM
            return __uint_as_float(__float_as_uint(t1) | __float_as_uint(t2));
        };
        ta = one(tt.x, 0);
        tb = one(tt.y, 1);
    }
}

// one hypothesis against one box: the arithmetic of cull32_pair, component by component (same operations, same roundings:
// the lead pass of cull_lead_k computes ITS box tests with it, lane = hypothesis, and gets the bits cull_tiles32_k writes)
template <int KIND>
__device__ __forceinline__ float cull32_one(const float* __restrict__ rp /* the pair's 24 floats */, int i /* 0 / 1 */, float bx,
                                            float by, float bz, float hx, float hy, float hz, float rb) {
    if (KIND == 0) {
        const float s = __builtin_fmaf(rp[0 + i], bx, __builtin_fmaf(rp[2 + i], by, __builtin_fmaf(rp[4 + i], bz, rp[6 + i])));
        const float r = __builtin_fmaf(rp[8 + i], hx, __builtin_fmaf(rp[10 + i], hy, __builtin_fmaf(rp[12 + i], hz, rp[14 + i])));
        return r - __builtin_fabsf(s);
    } else if (KIND == 1) {
        const float dx = __builtin_fabsf(rp[0 + i] - bx), dy = __builtin_fabsf(rp[2 + i] - by), dz = __builtin_fabsf(rp[4 + i] - bz);
        const float nx = __builtin_fmaxf(0.0f, dx - hx), ny = __builtin_fmaxf(0.0f, dy - hy), nz = __builtin_fmaxf(0.0f, dz - hz);
        const float fx = dx + hx, fy = dy + hy, fz = dz + hz;
        const float dmin2 = __builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx));
        const float dmax2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
        const float t1 = dmax2 - rp[6 + i], t2 = rp[8 + i] - dmin2;
        return __uint_as_float(__float_as_uint(t1) | __float_as_uint(t2));
    } else {
        const float d1 = __builtin_fmaf(rp[0 + i], bx, __builtin_fmaf(rp[2 + i], by, __builtin_fmaf(rp[4 + i], bz, rp[6 + i])));
        const float d2 = __builtin_fmaf(rp[8 + i], bx, __builtin_fmaf(rp[10 + i], by, __builtin_fmaf(rp[12 + i], bz, rp[14 + i])));
        const float tt = __builtin_fmaf(d2, d2, d1 * d1);
        const float dist = __builtin_sqrtf(tt);
        const float rt = rp[16 + i] * rb;
        const float t1 = (rp[18 + i] + rt) - dist;
        const float t2 = (dist + rt) - rp[20 + i];
        return __uint_as_float(__float_as_uint(t1) | __float_as_uint(t2));
    }
}

template <int KIND>
__device__ __forceinline__ void cull32_body(const double* __restrict__ boxes, uint32_t n_tiles,
                                            const float* __restrict__ cull32, uint32_t n_groups,
                                            uint32_t groups_per_wave, unsigned long long* __restrict__ masks,
                                            uint32_t* __restrict__ ub, uint32_t group_begin, uint32_t group_end,
                                            uint32_t block_x, uint32_t block_y,
                                            uint32_t* __restrict__ ubp = nullptr /* phased scoring: touched tiles with index % 4 == 0 (low half) / == 1 (high half) */) {
    const int lane = threadIdx.x;
    const uint32_t tile = block_x * 64u + (uint32_t)lane;
    const bool tile_ok = tile < n_tiles;
    float bx = 0.0f, by = 0.0f, bz = 0.0f, hx = -1.0f, hy = -1.0f, hz = -1.0f;
    if (tile_ok) {
        const float* __restrict__ f = reinterpret_cast<const float*>(boxes + (size_t)tile * kBoxStride + 8);
        bx = f[0];
        by = f[1];
        bz = f[2];
        hx = f[3];
        hy = f[4];
        hz = f[5];
    }
    const bool live = tile_ok && hx >= 0.0f;   // (hx < 0: empty tile)
    const unsigned long long live_mask = __ballot(live);
    if (!live) hx = hy = hz = 0.0f;
    // bounding radius of the box, rounded outwards (cylinder)
    const float rb = __builtin_sqrtf(__builtin_fmaf(hz, hz, __builtin_fmaf(hy, hy, hx * hx))) * 1.000001f;
    const uint32_t g0 = group_begin + block_y * groups_per_wave;
    const uint32_t g1 = min(group_end, g0 + groups_per_wave);
    // The group's 32 record pairs (3 KB) come in with ONE round trip -- three coalesced 16-byte loads per lane, parked in LDS
    // and read back as broadcasts -- instead of one scalar load per pair, each waited for before its pair was evaluated: the
    // records were written by minimal_fit_k a moment ago, mostly through another XCD's L2, and 32 dependent round trips of
    // ~1400 cycles were what this kernel's 28 us on C2 consisted of (round 5: tools/sweep_lead.sh shows the launch's time
    // does not depend on the lead pass it is fused with).  One wave per workgroup: the barrier is a formality.
    __shared__ float4 rec_s[192];
    float4 v0, v1, v2;   // the NEXT group's records, on their way while this group's pairs are evaluated
    auto fetch = [&](uint32_t g) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(cull32 + (size_t)g * 32u * 24u);
        v0 = src[lane];
        v1 = src[lane + 64];
        v2 = src[lane + 128];
    };
    if (g0 < g1) fetch(g0);
    for (uint32_t g = g0; g < g1; ++g) {
        uint32_t w[2] = {0u, 0u}, ubv = 0u, ubpv = 0u;
        __syncthreads();   // (the previous group's reads are done)
        rec_s[lane] = v0;
        rec_s[lane + 64] = v1;
        rec_s[lane + 128] = v2;
        __syncthreads();
        if (g + 1u < g1) fetch(g + 1u);
        // 32 pairs of hypotheses, 24 floats each (a kind's own share of them: 16 / 10 / 22), read as broadcasts ONE PAIR AHEAD
        // of the pair being evaluated (the LDS latency of a pair's operands was otherwise waited for 32 times per group)
        constexpr int kQuads = KIND == 0 ? 4 : (KIND == 1 ? 3 : 6);
        struct PairRec {
            float4 q[kQuads];
        };
        auto load_pair = [&](uint32_t pair) {
            PairRec r;
#pragma unroll
            for (int k = 0; k < kQuads; ++k) r.q[k] = rec_s[pair * 6u + (uint32_t)k];
            return r;
        };
        // (the phase counters are a kernel argument: decided once per group, not inside the pairs' loop, where the branch kept
        //  the compiler from overlapping one pair's LDS reads with the pair before)
        auto pairs = [&](auto with_ubp) {
            PairRec nxt = load_pair(0u);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint32_t drop = 0u;   // bit (31 - b): hypothesis half * 32 + b dropped this lane's tile
#pragma unroll
                for (uint32_t b = 0; b < 32u; b += 2u) {
                    const uint32_t hh = (uint32_t)half * 32u + b;
                    const PairRec cur = nxt;
                    nxt = load_pair(min((hh >> 1) + 1u, 31u));
                    float rp[24];
#pragma unroll
                    for (int k = 0; k < kQuads; ++k) {
                        rp[4 * k] = cur.q[k].x;
                        rp[4 * k + 1] = cur.q[k].y;
                        rp[4 * k + 2] = cur.q[k].z;
                        rp[4 * k + 3] = cur.q[k].w;
                    }
                    float ta, tb;
                    cull32_pair<KIND>(rp, bx, by, bz, hx, hy, hz, rb, ta, tb);
                    drop = __builtin_amdgcn_alignbit(drop, __float_as_uint(ta), 31);
                    drop = __builtin_amdgcn_alignbit(drop, __float_as_uint(tb), 31);
                    const unsigned long long ma = __ballot(!(ta < 0.0f)) & live_mask, mb = __ballot(!(tb < 0.0f)) & live_mask;
                    ubv = ((uint32_t)lane == hh) ? (uint32_t)__popcll(ma) : ubv;
                    ubv = ((uint32_t)lane == hh + 1u) ? (uint32_t)__popcll(mb) : ubv;
                    if (decltype(with_ubp)::value) {   // lane = tile and the wave starts at a multiple of 64: tile % 4 = lane % 4
                        const uint32_t pa = (uint32_t)__popcll(ma & 0x1111111111111111ull) | ((uint32_t)__popcll(ma & 0x2222222222222222ull) << 16);
                        const uint32_t pb = (uint32_t)__popcll(mb & 0x1111111111111111ull) | ((uint32_t)__popcll(mb & 0x2222222222222222ull) << 16);
                        ubpv = ((uint32_t)lane == hh) ? pa : ubpv;
                        ubpv = ((uint32_t)lane == hh + 1u) ? pb : ubpv;
                    }
                }
                w[half] = ~__builtin_bitreverse32(drop);
            }
        };
        if (ubp) pairs(std::true_type());
        else pairs(std::false_type());
        if (tile_ok) masks[(size_t)tile * n_groups + g] = live ? (((unsigned long long)w[1] << 32) | w[0]) : 0ull;
        if (ub && ubv) atomicAdd(&ub[g * 64u + (uint32_t)lane], ubv);
        if (ubp && ubpv) atomicAdd(&ubp[g * 64u + (uint32_t)lane], ubpv);
    }
}
template <int KIND>
__global__ __launch_bounds__(64) void cull_tiles32_k(const double* __restrict__ boxes, uint32_t n_tiles,
                                                      const float* __restrict__ cull32, uint32_t n_groups,
                                                      uint32_t groups_per_wave, unsigned long long* __restrict__ masks,
                                                      uint32_t* __restrict__ ub, uint32_t group_begin, uint32_t group_end,
                                                      uint32_t* __restrict__ ubp) {
    cull32_body<KIND>(boxes, n_tiles, cull32, n_groups, groups_per_wave, masks, ub, group_begin, group_end, blockIdx.x, blockIdx.y, ubp);
}

// The same box tests with LANE = HYPOTHESIS (round 6): a wave holds the 64 records of one group in registers and walks a block of
// tiles whose boxes come through the scalar unit.  cull_tiles32_k (lane = tile) needs a ballot and up to three scalar popcounts per
// HYPOTHESIS for the touched-tile counters, two selects to park them in the hypothesis' lane, and the records staged through the LDS:
// ~17-21 VALU + ~10 SALU instructions per box test.  Here a test's verdict is a lane's own (the counters are lane-local additions, the
// residues of the phase counters compile-time constants of the loop unrolled by four), the ballot IS the tile's mask word: ~17 VALU + 2
// SALU.  Same arithmetic, same roundings (cull32_one: what cull_lead_k's lead pass has always run), same words and counters.
// C3's window of 47 952 cylinders x 1954 tiles: 111 -> 81 us (what is left is the cylinder's correctly rounded sqrt: 15 of its 31 VALU
// instructions per test), the fit 0.765 -> 0.75 ms; spheres' windows 40 -> 41 / 108 -> 51 us (under the scoring either way).  For
// windows of many groups (launch_cull_mask): a wave's 11-22 record loads want a block of tiles behind them.
template <int KIND>
__global__ __launch_bounds__(64) void cull_hyp32_k(const double* __restrict__ boxes, uint32_t n_tiles, const float* __restrict__ cull32,
                                                    uint32_t n_groups, uint32_t tiles_per_wave /* a multiple of 4 */,
                                                    unsigned long long* __restrict__ masks, uint32_t* __restrict__ ub, uint32_t group_begin,
                                                    uint32_t group_end, uint32_t* __restrict__ ubp,
                                                    unsigned long long* __restrict__ touched /* [blockIdx.x][touched_stride], or null; tiles_per_wave = 64 */,
                                                    uint32_t touched_stride) {
    const uint32_t g = group_begin + blockIdx.y;
    if (g >= group_end) return;
    const int lane = threadIdx.x;
    const uint32_t h = g * 64u + (uint32_t)lane;
    float rr[24];   // the lane's record where cull32_one looks for hypothesis 0 of a pair (only the slots its KIND reads stay)
    {
        const float* __restrict__ rp = cull32 + (size_t)(h >> 1) * 24u + (h & 1u);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            rr[2 * k] = rp[2 * k];
            rr[2 * k + 1] = 0.0f;
        }
    }
    const uint32_t t0 = blockIdx.x * tiles_per_wave, t1 = min(n_tiles, t0 + tiles_per_wave);
    uint32_t cnt = 0, c0 = 0, c1 = 0;
    uint32_t tw_lo = 0u, tw_hi = 0u;   // the lane's own word: bit (tile - t0) = the mask bit of (tile, this hypothesis)
    for (uint32_t t = t0; t < t1; t += 4u) {
        // the four tiles' boxes in one go (wave-uniform addresses: scalar loads; past the block's end: its last tile once more)
        float bx[4][7];
#pragma unroll
        for (uint32_t r = 0; r < 4u; ++r) {
            const float* __restrict__ f = reinterpret_cast<const float*>(boxes + (size_t)min(t + r, t1 - 1u) * kBoxStride + 8);
#pragma unroll
            for (int k = 0; k < 7; ++k) bx[r][k] = f[k];
        }
        uint32_t w_lo = 0u, w_hi = 0u;   // lane r: the mask word of tile t + r
#pragma unroll
        for (uint32_t r = 0; r < 4u; ++r) {   // (t0 is a multiple of 4: tile % 4 = r)
            unsigned long long word = 0ull;
            if (bx[r][3] >= 0.0f) {   // (wave-uniform; hx < 0: empty tile)
                const float tv = cull32_one<KIND>(rr, 0, bx[r][0], bx[r][1], bx[r][2], bx[r][3], bx[r][4], bx[r][5], bx[r][6]);
                // the WORD takes the sign bit, the COUNTERS the comparison -- as cull_tiles32_k has them (a sphere's or cylinder's
                // verdict is the OR of two floats' bits: with the sign set it may be a NaN, which no comparison calls negative)
                const bool bit_on = (int)__float_as_uint(tv) >= 0;
                word = __builtin_amdgcn_ballot_w64(bit_on);
                if (touched) {   // (kernel argument: uniform)
                    const uint32_t b = (t - t0) + r, one = 1u << (b & 31u);   // (scalar)
                    if (b < 32u) tw_lo |= (bit_on && t + r < t1) ? one : 0u;
                    else tw_hi |= (bit_on && t + r < t1) ? one : 0u;
                }
                const bool touched_now = !(tv < 0.0f) && t + r < t1;
                cnt += touched_now ? 1u : 0u;
                if (r == 0u) c0 += touched_now ? 1u : 0u;
                if (r == 1u) c1 += touched_now ? 1u : 0u;
            }
            // (v_writelane_b32: the scalar word into lane r of the pair of registers the four-lane store below sends)
            const uint32_t wl = (uint32_t)word, wh = (uint32_t)(word >> 32);
            if (r == 0u) asm volatile("v_writelane_b32 %0, %2, 0\n\tv_writelane_b32 %1, %3, 0" : "+v"(w_lo), "+v"(w_hi) : "s"(wl), "s"(wh));
            if (r == 1u) asm volatile("v_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %1, %3, 1" : "+v"(w_lo), "+v"(w_hi) : "s"(wl), "s"(wh));
            if (r == 2u) asm volatile("v_writelane_b32 %0, %2, 2\n\tv_writelane_b32 %1, %3, 2" : "+v"(w_lo), "+v"(w_hi) : "s"(wl), "s"(wh));
            if (r == 3u) asm volatile("v_writelane_b32 %0, %2, 3\n\tv_writelane_b32 %1, %3, 3" : "+v"(w_lo), "+v"(w_hi) : "s"(wl), "s"(wh));
        }
        if ((uint32_t)lane < 4u && t + (uint32_t)lane < t1)
            masks[(size_t)(t + (uint32_t)lane) * n_groups + g] = ((unsigned long long)w_hi << 32) | w_lo;
    }
    if (ub && cnt) atomicAdd(&ub[h], cnt);
    if (ubp && (c0 | c1)) atomicAdd(&ubp[h], c0 | (c1 << 16));
    if (touched) touched[(size_t)blockIdx.x * touched_stride + h] = ((unsigned long long)tw_hi << 32) | tw_lo;
}

void launch_cull_mask(int kind, const SortedView& s, const double* score, const uint8_t* valid, uint32_t h_count,
                      uint32_t n_groups, unsigned long long* masks, uint32_t* ub, hipStream_t st, bool ub_is_zero,
                      uint32_t group_begin, uint32_t group_end, const float* cull32, uint32_t* ubp, unsigned long long* touched,
                      uint32_t touched_stride, bool* touched_written) {
    if (touched_written) *touched_written = false;
    (void)valid;     // (invalid and padding hypotheses are "no inlier" records)
    (void)h_count;
    group_end = std::min(group_end, n_groups);
    if (!s.n_tiles || group_begin >= group_end) return;
    const uint32_t window = group_end - group_begin;
    if (ub && !ub_is_zero) (void)hipMemsetAsync(ub + (size_t)group_begin * 64, 0, sizeof(uint32_t) * (size_t)window * 64, st);
    const uint32_t tblocks = (s.n_tiles + 63) / 64;
    // a few thousand waves at least; beyond that several groups per wave amortise the box loads
    uint32_t gpw = std::max<uint32_t>(1, (uint32_t)(((uint64_t)tblocks * window) / 8192));
    gpw = std::min<uint32_t>(gpw, 8);
    const dim3 g(tblocks, (window + gpw - 1) / gpw), b(64);
    if (cull32 && s.radius < 1e18 && config().cull_fp32 != 0) {
        // windows of many groups: lane = hypothesis (cull_hyp32_k); m3d_config.cull_fp32 = 2: always lane = tile (the tests' switch)
        if (window >= 128u && config().cull_fp32 != 2) {
            const uint32_t tpw = 64u;
            const dim3 gh((s.n_tiles + tpw - 1) / tpw, window);
            if (kind == 0) cull_hyp32_k<0><<<gh, b, 0, st>>>(s.boxes, s.n_tiles, cull32, n_groups, tpw, masks, ub, group_begin, group_end, ubp, touched, touched_stride);
            else if (kind == 1) cull_hyp32_k<1><<<gh, b, 0, st>>>(s.boxes, s.n_tiles, cull32, n_groups, tpw, masks, ub, group_begin, group_end, ubp, touched, touched_stride);
            else cull_hyp32_k<2><<<gh, b, 0, st>>>(s.boxes, s.n_tiles, cull32, n_groups, tpw, masks, ub, group_begin, group_end, ubp, touched, touched_stride);
            if (touched_written) *touched_written = touched != nullptr;
            return;
        }
        if (kind == 0)
            cull_tiles32_k<0><<<g, b, 0, st>>>(s.boxes, s.n_tiles, cull32, n_groups, gpw, masks, ub, group_begin, group_end, ubp);
        else if (kind == 1)
            cull_tiles32_k<1><<<g, b, 0, st>>>(s.boxes, s.n_tiles, cull32, n_groups, gpw, masks, ub, group_begin, group_end, ubp);
        else
            cull_tiles32_k<2><<<g, b, 0, st>>>(s.boxes, s.n_tiles, cull32, n_groups, gpw, masks, ub, group_begin, group_end, ubp);
        return;
    }
    if (kind == 0)
        cull_tiles_k<0><<<g, b, 0, st>>>(s.boxes, s.n_tiles, score, n_groups, gpw, masks, ub, group_begin, group_end);
    else if (kind == 1)
        cull_tiles_k<1><<<g, b, 0, st>>>(s.boxes, s.n_tiles, score, n_groups, gpw, masks, ub, group_begin, group_end);
    else
        cull_tiles_k<2><<<g, b, 0, st>>>(s.boxes, s.n_tiles, score, n_groups, gpw, masks, ub, group_begin, group_end);
}

// keep[g] = hypotheses of group g that are still worth scoring: ub[h] * 512 >= best_count[0]
// (ub null: everything; best_count null or 0: everything).  zero != null: the kernel also clears the
// kCountReplicas x rep_stride counter replicas of its hypotheses and (group 0) the kPairReplicas words behind them,
// which the scoring kernel that follows adds into -- one command less than a separate memset.
// Optional by-product of the keep rules (plane_bound_k's input, m3d_bound.hip): the kept hypotheses of the launch as a LIST, in no
// particular order (one atomic per group of 64) -- *surv_count its length (zero before the launch: cleared by a fit's first
// minimal_fit_k and by plane_bound_k, which consumes the list -- the last of its workgroups resets the count).
struct SurvOut {
    uint32_t* count = nullptr;
    uint32_t* ids = nullptr;
};
__device__ __forceinline__ void emit_survivors(const SurvOut surv, unsigned long long m, bool k, uint32_t h) {
    if (!surv.count || !m) return;   // (m: wave-uniform)
    uint32_t base = 0;
    if (threadIdx.x == 0) base = atomicAdd(surv.count, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (k) surv.ids[base + rank] = h;
}
__global__ __launch_bounds__(64) void keep_mask_k(const uint32_t* __restrict__ ub,
                                                   const uint32_t* __restrict__ best_count_ptr,
                                                   unsigned long long* __restrict__ keep,
                                                   uint32_t* __restrict__ zero, uint32_t rep_stride,
                                                   uint32_t group_offset, SurvOut surv) {
    const uint32_t g = group_offset + blockIdx.x;
    const uint32_t h = g * 64u + threadIdx.x;
    const uint32_t best = best_count_ptr ? best_count_ptr[0] : 0u;
    const bool k = !ub || best == 0 || (uint64_t)ub[h] * kTilePoints >= best;
    const unsigned long long m = __ballot(k);
    if (threadIdx.x == 0) keep[g] = m;
    emit_survivors(surv, m, k, h);
    if (zero) {
#pragma unroll
        for (int r = 0; r < kCountReplicas; ++r) zero[(size_t)r * rep_stride + h] = 0u;
        if (blockIdx.x == 0)   // (first workgroup of the launch, whatever its group window)
            for (int i = threadIdx.x; i < kPairReplicas; i += 64) zero[(size_t)kCountReplicas * rep_stride + i] = 0u;
    }
}
// groups [group_offset, group_offset + n_groups) of the chunk
void launch_keep_mask(const uint32_t* ub, const uint32_t* best_count, uint32_t n_groups, unsigned long long* keep,
                      hipStream_t st, uint32_t* zero_counts_rep, uint32_t rep_stride, uint32_t group_offset, uint32_t* surv_count, uint32_t* surv) {
    if (!n_groups) return;
    if (!ub && !zero_counts_rep && !surv_count) {
        (void)hipMemsetAsync(keep + group_offset, 0xFF, sizeof(unsigned long long) * n_groups, st);
        return;
    }
    keep_mask_k<<<n_groups, 64, 0, st>>>(ub, best_count, keep, zero_counts_rep, rep_stride, group_offset, SurvOut{surv_count, surv});
}

// The lead pass of a chunk (its first `lead` hypotheses, counted on their own) and the keep masks of the rest in
// ONE launch: every workgroup folds the lead hypotheses' counter replicas again (lead <= 512: a few loads per
// lane) to know their best count without waiting for another kernel; workgroup 0 also writes their records
// (count | valid << 31) and raises the running best.  Then the keep_mask_k rule for its own group.
__global__ __launch_bounds__(64) void lead_fold_keep_k(const uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                        uint32_t lead, const uint8_t* __restrict__ valid, uint32_t h_count,
                                                        uint32_t* __restrict__ records, uint32_t* __restrict__ best_count,
                                                        const uint32_t* __restrict__ ub,
                                                        unsigned long long* __restrict__ keep,
                                                        uint32_t* __restrict__ zero, uint32_t group_offset,
                                                        uint32_t* __restrict__ records_dev,
                                                        unsigned long long* __restrict__ pick_key,
                                                        unsigned long long* __restrict__ pick_key2,
                                                        SurvOut surv) {
    const uint32_t g = group_offset + blockIdx.x;
    const uint32_t prev = best_count[0];   // may or may not include this chunk's lead already: max() below either way
    uint32_t v = 0;
    unsigned long long key = 0, key2 = 0;   // (count << 32 | ~index): highest count, lowest index among equals; key2: highest index
    for (uint32_t h = threadIdx.x; h < lead; h += 64u) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < kCountReplicas; ++r) c += counts_rep[(size_t)r * rep_stride + h];
        const bool ok = h < h_count && valid[h];
        if (blockIdx.x == 0) {
            if (records) records[h] = c | (ok ? 0x80000000u : 0u);
            if (records_dev) records_dev[h] = c | (ok ? 0x80000000u : 0u);
        }
        v = max(v, ok ? c : 0u);
        if (ok && c) key = max(key, ((unsigned long long)c << 32) | (0xFFFFFFFFu - h));
        if (ok && c) key2 = max(key2, ((unsigned long long)c << 32) | h);
    }
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
    if (blockIdx.x == 0 && threadIdx.x == 0 && v) atomicMax(best_count, v);
    if (zero && blockIdx.x == 0) {   // block-uniform: what the pair counters hold now is the lead pass' share (m3d_stats.pairs_lead)
        uint32_t* __restrict__ pair_rep = zero + (size_t)kCountReplicas * rep_stride;
        uint32_t p = 0;
        for (int r = threadIdx.x; r < kPairMain; r += 64) p += pair_rep[r];
        for (int off = 32; off > 0; off >>= 1) p += (uint32_t)__shfl_xor((int)p, off, 64);
        if (threadIdx.x == 0) pair_rep[kPairLead] = p;
    }
    if (pick_key && blockIdx.x == 0) {   // block-uniform
        for (int off = 32; off > 0; off >>= 1) key = max(key, (unsigned long long)__shfl_xor((long long)key, off, 64));
        if (threadIdx.x == 0 && key) atomicMax(pick_key, key);
        if (pick_key2) {
            for (int off = 32; off > 0; off >>= 1) key2 = max(key2, (unsigned long long)__shfl_xor((long long)key2, off, 64));
            if (threadIdx.x == 0 && key2) atomicMax(pick_key2, key2);
        }
    }
    const uint32_t best = max(prev, v);
    const uint32_t h = g * 64u + threadIdx.x;
    const bool k = !ub || best == 0 || (uint64_t)ub[h] * kTilePoints >= best;
    const unsigned long long m = __ballot(k);
    if (threadIdx.x == 0) keep[g] = m;
    emit_survivors(surv, m, k, h);
    if (zero) {
#pragma unroll
        for (int r = 0; r < kCountReplicas; ++r) zero[(size_t)r * rep_stride + h] = 0u;
    }
}
void launch_lead_fold_keep(const uint32_t* counts_rep, uint32_t rep_stride, uint32_t lead, const uint8_t* valid,
                           uint32_t h_count, uint32_t* records, uint32_t* best_count, const uint32_t* ub,
                           unsigned long long* keep, uint32_t n_groups_rest, hipStream_t st, uint32_t* records_dev,
                           uint32_t group_begin, unsigned long long* pick_key, unsigned long long* pick_key2, uint32_t* surv_count, uint32_t* surv) {
    if (group_begin == 0xFFFFFFFFu) group_begin = lead / 64u;
    // (n_groups_rest == 0 still needs the fold of the lead's counters: one workgroup whose keep word is scratch)
    if (!n_groups_rest) return;
    lead_fold_keep_k<<<n_groups_rest, 64, 0, st>>>(counts_rep, rep_stride, lead, valid, h_count, records, best_count, ub,
                                                   keep, const_cast<uint32_t*>(counts_rep), group_begin, records_dev, pick_key, pick_key2, SurvOut{surv_count, surv});
}

// ------------------------------------------------------------------------------------------------
// counting over the surviving (tile, hypothesis) pairs
// ------------------------------------------------------------------------------------------------
// One wave per (tile, <= groups_per_block hypothesis groups).  The surviving hypotheses of the wave's mask words are
// first COMPACTED into a list of 16-bit ids in LDS (one pass over the words: rank = running total + v_mbcnt, a
// ds_write per set bit); the loop then takes them 64 at a time -- lane k of `ids` holds the k-th id of the batch, one
// v_readlane per hypothesis -- with the record of the next hypothesis in flight (scalar loads, two register sets
// used alternately: no register-to-register copies), counts parked in lane k and flushed with one vector atomic per
// batch.  Per hypothesis the scalar unit now does the 16 popcount / add instructions of the 8 rows plus ~8 of address
// arithmetic and loop control -- the bit walk (s_ff1 / s_and / s_add chains, ~10 more), the m0 save / restore of the
// v_writelane parking and the five 64-bit moves of the record hand-over are gone: 45 -> ~27 scalar instructions against
// 58 fp64 VALU instructions (the scalar unit issues one instruction per SIMD slot, like the VALU: it was the co-bottleneck).
template <int KIND>
__global__ __launch_bounds__(64) void score_mask_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                    const double* __restrict__ sz,
                                                    const double* __restrict__ score,
                                                    const unsigned long long* __restrict__ masks,
                                                    const unsigned long long* __restrict__ keep,
                                                    uint32_t n_groups /* of the chunk: row length of masks */,
                                                    uint32_t groups_per_block,
                                                    uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                    uint32_t* __restrict__ pair_rep,
                                                    uint32_t group_begin, uint32_t group_end /* window of this launch */) {
    __shared__ uint16_t ids[64 * 64];   // groups_per_block <= 64
    const uint32_t tile = blockIdx.x;
    // ~430 tiles add to every hypothesis' counter: kCountReplicas copies of the counter array (tile mod R)
    // keep the same-address atomic chains short (they serialise in L2 and dominated small chunks)
    uint32_t* __restrict__ counts = counts_rep + (size_t)(tile % kCountReplicas) * rep_stride;
    const uint32_t g0 = group_begin + blockIdx.y * groups_per_block;
    const int lane = threadIdx.x;
    // lane l holds the (pruned) mask of group g0 + l
    unsigned long long mm = 0;
    if ((uint32_t)lane < groups_per_block && g0 + lane < group_end)
        mm = masks[(size_t)tile * n_groups + g0 + lane] & keep[g0 + lane];
    if (!__ballot(mm != 0)) return;  // most (tile, range) blocks of a pruned chunk end here
    const size_t base = (size_t)tile * kTilePoints + lane;
    constexpr int P = kTilePoints / 64;
    double x[P], y[P], z[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        x[j] = sx[base + 64 * j];
        y[j] = sy[base + 64 * j];
        z[j] = sz[base + 64 * j];
    }
    // ---- compaction of the set bits into ids[0 .. total): id = 64 * word + bit (relative to g0)
    const int mm_lo = (int)(uint32_t)mm, mm_hi = (int)(uint32_t)(mm >> 32);
    uint32_t total = 0;
    for (uint32_t w = 0; w < groups_per_block; ++w) {   // wave-uniform
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(mm_lo, (int)w), hi = (uint32_t)__builtin_amdgcn_readlane(mm_hi, (int)w);
        if ((lo | hi) == 0u) continue;
        const unsigned long long word = ((unsigned long long)hi << 32) | lo;
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));   // set bits below this lane
        if ((word >> lane) & 1ull) ids[total + below] = (uint16_t)(w * 64u + (uint32_t)lane);
        total += (uint32_t)__popcll(word);
    }
    __syncthreads();   // (one wave: orders the LDS writes before the reads below)
    if (lane == 0) atomicAdd(&pair_rep[(tile + blockIdx.y * 67u) % (uint32_t)kPairMain], total);   // m3d_stats.pairs_scored
    constexpr int kUsed = KIND == 2 ? 8 : 5;
    const double* __restrict__ score0 = score + (size_t)g0 * 64u * kModelStride;
    auto load_rec = [&](double (&r)[kModelStride], uint32_t id) {   // id wave-uniform -> scalar loads
        const double* __restrict__ rp = score0 + (size_t)id * kModelStride;
#pragma unroll
        for (int k = 0; k < kUsed; ++k) r[k] = rp[k];
    };
    for (uint32_t b0 = 0; b0 < total; b0 += 64u) {
        const uint32_t nb = min(64u, total - b0);
        const int my = (b0 + (uint32_t)lane < total) ? (int)ids[b0 + lane] : 0;   // lane k: k-th id of the batch
        uint32_t park = 0;
        double ra[kModelStride], rb[kModelStride];
        load_rec(ra, (uint32_t)__builtin_amdgcn_readlane(my, 0));
        for (uint32_t k = 0; k < nb; k += 2u) {
            // two hypotheses per trip, their records in alternating register sets; the next one is always in flight
            load_rec(rb, (uint32_t)__builtin_amdgcn_readlane(my, (int)min(k + 1u, nb - 1u)));
            const uint32_t ca = tile_count<KIND, P>(ra, x, y, z);
            park = ((uint32_t)lane == k) ? ca : park;
            load_rec(ra, (uint32_t)__builtin_amdgcn_readlane(my, (int)min(k + 2u, nb - 1u)));
            if (k + 1u < nb) {   // wave-uniform
                const uint32_t cb = tile_count<KIND, P>(rb, x, y, z);
                park = ((uint32_t)lane == k + 1u) ? cb : park;
            }
        }
        if ((uint32_t)lane < nb && park) atomicAdd(&counts[g0 * 64u + (uint32_t)my], park);
    }
}

// ------------------------------------------------------------------------------------------------
// score_screen_k: score_mask_k for planes and spheres with a packed-fp32 SCREEN in front of the fp64 test.
//
// The fp64 loop of score_mask_k issues 7 (plane) / 10 (sphere) VALU instructions per point and hypothesis, and the
// VALU is what bounds it.  Almost every point is nowhere near the cut-off: an fp32 evaluation with a rigorous error
// bound decides it (m3d_fp.hpp derives the bound), and v_pk_fma_f32 handles two points per lane and instruction.
//   * The wave keeps its 512 points as fp32 offsets from the tile's box centre (12 register pairs; the subtraction is
//     done in fp64, so the offsets are as accurate as fp32 gets).
//   * Per batch of <= 64 surviving hypotheses lane k prepares the record of hypothesis k FOR THIS TILE (fp64: the
//     model moved to the box centre, the rounding bound from the box's extent) and parks it in LDS; the loop reads it
//     back as a broadcast (two ds_reads per hypothesis, the next one always in flight; no scalar loads).
//   * Per pair of points (plane): 3 v_pk_fma for the plane value, 1 v_pk_fma for q = s^2 - T^2, 2 v_alignbit shifting
//     the sign bits of q (inside <=> q < 0) into a per-lane bit string, 1.5 v_min keeping min |q|: 7.5 instructions
//     for two points instead of 14, and none of them on the scalar unit (the v_cmp -> s_bcnt1 -> s_add counting of the
//     fp64 loop costs two scalar instructions per row).
//   * After the 8 points of a lane: one ds_write_b8 parks the lane's 8 inside-bits in row k of an LDS table;
//     `!(min |q| >= h)` over the wave says whether any point was too close to call -- then the tile's fp64 points are
//     fetched again and the pair is counted by tile_count, the exact code (m3d_stats.pairs_exact).
//   * Every 64 hypotheses lane k adds up the bits of row k (4 ds_read_b128 + 16 v_bcnt) and issues the batch's one
//     vector atomic, as before.
//   * Four hypotheses per trip of the loop: 29 VALU instructions each (16 v_pk_fma, 8 shifts, 4 v_min3, 1 compare: what
//     ops_per_pair in bench.py counts) and three more per trip (two address additions, a register copy).
// A tile with a non-finite coordinate (the NaN padding of the last tile, or the caller's own) is never screened.
// ------------------------------------------------------------------------------------------------
#ifndef M3D_SCREEN_MAX_GROUPS
#define M3D_SCREEN_MAX_GROUPS 16
#endif
constexpr uint32_t kScreen4MaxGroups = 64;   // score_screen4_k: groups per four-wave workgroup (lane = mask word; the id list: 8 KB)
constexpr uint32_t kScreenMaxGroups = M3D_SCREEN_MAX_GROUPS;   // 64-hypothesis groups per workgroup (the id list: 2 KB of LDS; 8.2 KB in all: 19 workgroups per CU)
// Groups per scoring workgroup: m3d_config.score_groups_per_block (8), and TWICE that for windows of 192 groups and more.  What a
// workgroup pays per batch of <= 64 surviving hypotheses -- 64 lanes preparing records, the tile's offsets, the id list -- is
// worth sharing among more of them where the pruning leaves few per (tile, 8 groups): C3's chunks of 250 groups (~16
// survivors per workgroup) score_screen_k<2> 0.846 -> 0.801 ms, <1> 0.396 -> 0.377, the fits 1.09 -> 1.05 / 0.72 -> 0.71 ms; C2's
// window of 155 groups loses 2 % of its launch with 16 (fewer, longer workgroups: the tail) and keeps 8.
// `thinned`: the planes' histogram bound has pruned the window (m3d_bound.hip: about half of the survivors are gone) -- half as
// many groups again per workgroup (C2: 8 / 12 / 16 groups -> launch 0.0603 / 0.0591 / 0.0604 ms, step 0.2622 / 0.2587 / 0.2605).
static uint32_t screen_gpb_max(uint32_t window, bool thinned = false) {
    const uint32_t g = (uint32_t)config().score_groups_per_block;
    return std::min<uint32_t>(window >= 192u ? 2u * g : (thinned ? g + g / 2u : g), kScreenMaxGroups);
}
constexpr int kCntStride = 64;              // bytes per row of the count table

// bits: 8 sign bits (point inside <=> 1); m: min over the lane's points of the distance to the decision boundary
template <int KIND, int Q>
__device__ __forceinline__ void screen_eval(const float4 ra, const float4 rb, const float4 rc, const f32x2 (&xf)[Q],
                                            const f32x2 (&yf)[Q], const f32x2 (&zf)[Q], const f32x2 (&wf)[Q] /* sphere: |x~|^2 */,
                                            uint32_t& bits, float& m) {
    uint32_t acc = 0;
    float mn = __builtin_inff();
    if (KIND == 0) {   // ra = (a, b, c, mid2), rb = (D, h, -, -)
        const f32x2 A = {ra.x, ra.x}, B = {ra.y, ra.y}, C = {ra.z, ra.z}, D = {rb.x, rb.x}, M2 = {-ra.w, -ra.w};
        // the four row pairs advance together: dependent v_pk_fma_f32 need a wait state that independent ones fill
        f32x2 s[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) s[j] = __builtin_elementwise_fma(C, zf[j], D);
#pragma unroll
        for (int j = 0; j < Q; ++j) s[j] = __builtin_elementwise_fma(B, yf[j], s[j]);
#pragma unroll
        for (int j = 0; j < Q; ++j) s[j] = __builtin_elementwise_fma(A, xf[j], s[j]);
#pragma unroll
        for (int j = 0; j < Q; ++j) s[j] = __builtin_elementwise_fma(s[j], s[j], M2);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s[j].x), 31);
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s[j].y), 31);
            mn = __builtin_fminf(__builtin_fminf(mn, __builtin_fabsf(s[j].x)), __builtin_fabsf(s[j].y));   // one v_min3_f32
        }
    } else if (KIND == 2) {   // ra = (E1x, E1y, E1z, D1), rb = (E2x, E2y, E2z, D2), rc = (mid, half^2, h, -)
        // t = (E1 . q + D1)^2 + (E2 . q + D2)^2: two plane values (m3d_fp.hpp: the exact code's |(q - p1) x (q - p2)|^2)
        const f32x2 AX = {ra.x, ra.x}, AY = {ra.y, ra.y}, AZ = {ra.z, ra.z}, AD = {ra.w, ra.w};
        const f32x2 BX = {rb.x, rb.x}, BY = {rb.y, rb.y}, BZ = {rb.z, rb.z}, BD = {rb.w, rb.w};
        const f32x2 MID = {-rc.x, -rc.x}, H2 = {-rc.y, -rc.y};
        f32x2 d1[Q], d2[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            d1[j] = __builtin_elementwise_fma(AZ, zf[j], AD);
            d2[j] = __builtin_elementwise_fma(BZ, zf[j], BD);
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            d1[j] = __builtin_elementwise_fma(AY, yf[j], d1[j]);
            d2[j] = __builtin_elementwise_fma(BY, yf[j], d2[j]);
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            d1[j] = __builtin_elementwise_fma(AX, xf[j], d1[j]);
            d2[j] = __builtin_elementwise_fma(BX, xf[j], d2[j]);
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) d1[j] = __builtin_elementwise_fma(d1[j], d1[j], MID);
#pragma unroll
        for (int j = 0; j < Q; ++j) d1[j] = __builtin_elementwise_fma(d2[j], d2[j], d1[j]);   // t - mid
#pragma unroll
        for (int j = 0; j < Q; ++j) d1[j] = __builtin_elementwise_fma(d1[j], d1[j], H2);     // q = (t - mid)^2 - half^2
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(d1[j].x), 31);
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(d1[j].y), 31);
            mn = __builtin_fminf(__builtin_fminf(mn, __builtin_fabsf(d1[j].x)), __builtin_fabsf(d1[j].y));
        }
    } else {           // ra = (K, half^2, -, -), rb = (-2 Cx, -2 Cy, -2 Cz, h): the expanded form (sphere_screen_record)
        const f32x2 CX = {rb.x, rb.x}, CY = {rb.y, rb.y}, CZ = {rb.z, rb.z}, KK = {ra.x, ra.x}, H2 = {-ra.y, -ra.y};
        f32x2 t[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) t[j] = wf[j] + KK;
#pragma unroll
        for (int j = 0; j < Q; ++j) t[j] = __builtin_elementwise_fma(CX, xf[j], t[j]);
#pragma unroll
        for (int j = 0; j < Q; ++j) t[j] = __builtin_elementwise_fma(CY, yf[j], t[j]);
#pragma unroll
        for (int j = 0; j < Q; ++j) t[j] = __builtin_elementwise_fma(CZ, zf[j], t[j]);   // |x~ - C|^2 - mid
#pragma unroll
        for (int j = 0; j < Q; ++j) t[j] = __builtin_elementwise_fma(t[j], t[j], H2);     // q = (t - mid)^2 - half^2
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(t[j].x), 31);
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(t[j].y), 31);
            mn = __builtin_fminf(__builtin_fminf(mn, __builtin_fabsf(t[j].x)), __builtin_fabsf(t[j].y));
        }
    }
    bits = acc;
    m = mn;
}

// OWN_BOX_TESTS (the lead pass inside cull_lead_k): no mask word has been written for these groups yet -- the wave runs
// the fp32 box test of its tile against the 64 hypotheses of each group itself (lane = hypothesis, cull32_one: the bits
// cull_tiles32_k would have written) and stores the word for whoever reads the masks later.
// WAVES = 4 (score_screen4_k): one workgroup = one tile x up to kScreen4MaxGroups groups; the surviving hypotheses of ALL its
// groups are compacted into ONE id list (wave 0, lane = mask word) and the four waves take its batches of 64 in turn -- full
// batches whatever the masks look like, a tile's offsets loaded by 12 waves instead of 20, one mask read per 64 groups.
template <int KIND, bool OWN_BOX_TESTS, int WAVES = 1>
__device__ __forceinline__ void score_screen_body(const double* __restrict__ sx, const double* __restrict__ sy,
                                                  const double* __restrict__ sz,
                                                  const double* __restrict__ boxes, double max_abs,
                                                  const double* __restrict__ score,
                                                  unsigned long long* __restrict__ masks,
                                                  const unsigned long long* __restrict__ keep,
                                                  uint32_t n_groups, uint32_t groups_per_block /* <= kScreenMaxGroups */,
                                                  uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                  uint32_t* __restrict__ pair_rep,
                                                  uint32_t group_begin, uint32_t group_end, uint32_t block_x, uint32_t block_y,
                                                  const float* __restrict__ cull32, const float* __restrict__ tile_f32,
                                                  uint32_t has_dead /* SortedView::has_dead: some points are tombstones */) {
    static_assert(WAVES == 1 || !OWN_BOX_TESTS, "the lead pass keeps one-wave workgroups");
    constexpr uint32_t kMaxGroups = WAVES == 1 ? kScreenMaxGroups : kScreen4MaxGroups;
    __shared__ uint16_t ids[kMaxGroups * 64];
    __shared__ __attribute__((aligned(16))) uint8_t cnt8_all[WAVES][64 * kCntStride];
    constexpr int NL = KIND == 2 ? 3 : 2;
    __shared__ float4 loc_all[WAVES][64 + 1][NL];   // the batch's (tile, hypothesis) records (+ one row the loop's look-ahead may read)
    __shared__ uint32_t s_total;
    const int lane = threadIdx.x & 63;
    const int wave = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: an SGPR)
    uint8_t* const cnt8 = cnt8_all[wave];
    float4 (*const loc)[NL] = loc_all[wave];
    // LDS traffic between the lanes of ONE wave needs no barrier (a wave's LDS instructions complete in order): a fence keeps
    // the compiler from moving them.  With four waves a __syncthreads here would also deadlock (their batch counts differ).
    auto wave_sync = [&]() {
        if (WAVES == 1) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    };
    const uint32_t tile = block_x;
    uint32_t* __restrict__ counts = counts_rep + (size_t)(tile % kCountReplicas) * rep_stride;
    const uint32_t g0 = group_begin + block_y * groups_per_block;
    unsigned long long mm = 0;
    if (OWN_BOX_TESTS) {
        const float* __restrict__ f = reinterpret_cast<const float*>(boxes + (size_t)tile * kBoxStride + 8);   // (wave-uniform)
        const float bx = f[0], by = f[1], bz = f[2];
        float hx = f[3], hy = f[4], hz = f[5];
        const bool live = hx >= 0.0f;   // (hx < 0: empty tile)
        if (!live) hx = hy = hz = 0.0f;
        const float rb = __builtin_sqrtf(__builtin_fmaf(hz, hz, __builtin_fmaf(hy, hy, hx * hx))) * 1.000001f;
        for (uint32_t w = 0; w < groups_per_block && g0 + w < group_end; ++w) {
            const uint32_t h = (g0 + w) * 64u + (uint32_t)lane;
            const float t = cull32_one<KIND>(cull32 + (size_t)(h >> 1) * 24u, (int)(h & 1u), bx, by, bz, hx, hy, hz, rb);
            const unsigned long long word = live ? __ballot(!(t < 0.0f)) : 0ull;
            if (lane == 0) masks[(size_t)tile * n_groups + g0 + w] = word;
            if ((uint32_t)lane == w) mm = word & keep[g0 + w];
        }
    } else if ((uint32_t)lane < groups_per_block && g0 + lane < group_end) {
        mm = masks[(size_t)tile * n_groups + g0 + lane] & keep[g0 + lane];
    }
    if (!__ballot(mm != 0)) return;
    const size_t base = (size_t)tile * kTilePoints + lane;
    constexpr int P = kTilePoints / 64, Q = P / 2;
    double box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) box[k] = boxes[(size_t)tile * kBoxStride + k];   // (wave-uniform: scalar loads)
    // rows 2 j and 2 j + 1 of the tile share a register pair
    f32x2 xf[Q], yf[Q], zf[Q];
    bool tile_screened;
    if (tile_f32) {   // (kernel argument: uniform) tile_boxes_k has prepared the offsets and the verdict
        const f32x2* __restrict__ t2 = reinterpret_cast<const f32x2*>(tile_f32 + (size_t)tile * kTileF32Floats) + lane;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            xf[j] = t2[(0 * Q + j) * 64];
            yf[j] = t2[(1 * Q + j) * 64];
            zf[j] = t2[(2 * Q + j) * 64];
        }
        tile_screened = boxes[(size_t)tile * kBoxStride + 6] != 0.0;
    } else {
        float chk = 0.0f;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            xf[j] = {(float)(sx[base + 128 * j] - box[0]), (float)(sx[base + 128 * j + 64] - box[0])};
            yf[j] = {(float)(sy[base + 128 * j] - box[1]), (float)(sy[base + 128 * j + 64] - box[1])};
            zf[j] = {(float)(sz[base + 128 * j] - box[2]), (float)(sz[base + 128 * j + 64] - box[2])};
            chk += ((xf[j].x + xf[j].y) + (yf[j].x + yf[j].y)) + (zf[j].x + zf[j].y);
        }
        // inf or NaN anywhere (also an offset beyond the fp32 range) makes chk * 0 a NaN
        tile_screened = __ballot(!(chk * 0.0f == 0.0f)) == 0ull;
    }
    // the sphere's screen works on the expanded square: w = |x~|^2 once per point and tile (sphere_screen_record)
    f32x2 wf[Q];   // (only the sphere reads it)
    if (KIND == 1) {
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            f32x2 w = xf[j] * xf[j];
            w = __builtin_elementwise_fma(yf[j], yf[j], w);
            wf[j] = __builtin_elementwise_fma(zf[j], zf[j], w);
        }
    }
    // DEAD points (poison_plane_inliers_k: a segmentation round's inliers killed in place, x = NaN in both copies of the
    // tile): their bit of the lane's inside-string is masked out -- a NaN's sign bit is nobody's business -- and the
    // minimum over |q| skips them (v_min3_f32 returns the non-NaN operands; were it not so, the pair would be recounted
    // in fp64, where NaN < T is false: slower, never wrong).  dead8 in the order screen_eval shifts the bits in.
    uint32_t dead8 = 0;
    bool tile_has_dead = false;
    if (KIND == 0 && has_dead && tile_f32) {   // (kernel arguments: uniform; only a segmentation's working cloud has any)
#pragma unroll
        for (int j = 0; j < Q; ++j) dead8 = (dead8 << 2) | (xf[j].x != xf[j].x ? 2u : 0u) | (xf[j].y != xf[j].y ? 1u : 0u);
        tile_has_dead = tile_screened && __ballot(dead8 != 0u) != 0ull;
    }
    // ---- compaction of the set bits into ids[0 .. total): id = 64 * word + bit (relative to g0)
    uint32_t total = 0;
    if (WAVES == 1) {
        const int mm_lo = (int)(uint32_t)mm, mm_hi = (int)(uint32_t)(mm >> 32);
        for (uint32_t w = 0; w < groups_per_block; ++w) {   // wave-uniform
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(mm_lo, (int)w), hi = (uint32_t)__builtin_amdgcn_readlane(mm_hi, (int)w);
            if ((lo | hi) == 0u) continue;
            const unsigned long long word = ((unsigned long long)hi << 32) | lo;
            const uint32_t slot = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, total));
            if (__builtin_amdgcn_inverse_ballot_w64(word)) ids[slot] = (uint16_t)(w * 64u + (uint32_t)lane);   // (exec = word)
            total += (uint32_t)__popcll(word);
        }
        __syncthreads();
    } else {
        if (wave == 0) {   // lane l expands its own word behind the words before it (ascending ids)
            const uint32_t pc = (uint32_t)__popcll(mm);
            uint32_t incl = pc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (lane == 63) s_total = incl;
            uint32_t at = incl - pc;
            unsigned long long w = mm;
            while (w) {
                ids[at++] = (uint16_t)((uint32_t)lane * 64u + (uint32_t)__builtin_ctzll(w));
                w &= w - 1ull;
            }
        }
        __syncthreads();   // (every wave of the workgroup is here: the exit above is workgroup-uniform)
        total = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_total);
    }
    if (threadIdx.x == 0) atomicAdd(&pair_rep[(tile + block_y * 67u) % (uint32_t)kPairMain], total);
    const double* __restrict__ score0 = score + (size_t)g0 * 64u * kModelStride;
    // the exact count of one (tile, hypothesis) pair: the fp64 points come back from memory (L2), four rows at a time
    auto exact_count = [&](uint32_t id) -> uint32_t {
        double rec[kModelStride];
        const double* __restrict__ rp = score0 + (size_t)id * kModelStride;   // id wave-uniform -> scalar loads
#pragma unroll
        for (int k = 0; k < kModelStride; ++k) rec[k] = rp[k];
        uint32_t c = 0;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            double x[P / 2], y[P / 2], z[P / 2];
#pragma unroll
            for (int j = 0; j < P / 2; ++j) {
                x[j] = sx[base + 64 * (half * (P / 2) + j)];
                y[j] = sy[base + 64 * (half * (P / 2) + j)];
                z[j] = sz[base + 64 * (half * (P / 2) + j)];
            }
            c += tile_count<KIND, P / 2>(rec, x, y, z);
        }
        return c;
    };
    for (uint32_t b0 = (uint32_t)wave * 64u; b0 < total; b0 += (uint32_t)WAVES * 64u) {   // (the waves take the batches in turn)
        const uint32_t nb = min(64u, total - b0);
        const int my = (b0 + (uint32_t)lane < total) ? (int)ids[b0 + lane] : 0;   // lane k: k-th id of the batch
        {   // lane k: the record of hypothesis k at this tile
            const double* __restrict__ rp = score0 + (size_t)my * kModelStride;
            constexpr int kRec = KIND == 2 ? 8 : 5;
            double rec[kRec];
#pragma unroll
            for (int k = 0; k < kRec; ++k) rec[k] = rp[k];
            float sr[12];
            if (KIND == 0) plane_screen_record(rec, box, max_abs, sr);
            else if (KIND == 1) sphere_screen_record(rec, box, max_abs, sr);
            else cylinder_screen_record(rec, box, max_abs, sr);
            loc[lane][0] = make_float4(sr[0], sr[1], sr[2], sr[3]);
            loc[lane][1] = make_float4(sr[4], sr[5], sr[6], sr[7]);
            if (KIND == 2) loc[lane][NL - 1] = make_float4(sr[8], sr[9], sr[10], sr[11]);
        }
        wave_sync();
        uint32_t park = 0;   // lane k: exact count of hypothesis k when the screen could not decide it
        // SCREENED is the tile's verdict (wave-uniform, fixed for the workgroup): a compile-time flag of the loop so that
        // the screened loop carries no branch and no register initialisation for the other case
        auto step = [&](auto screened, auto dead, const float4 ra, const float4 rb, const float4 rc, uint32_t k, uint8_t* cnt_row) {
            constexpr bool SCREENED = decltype(screened)::value;
            constexpr bool DEAD = decltype(dead)::value;
            uint32_t bits = 0;
            bool exact = true;
            if (SCREENED) {
                float m;
                screen_eval<KIND, Q>(ra, rb, rc, xf, yf, zf, wf, bits, m);
                if (DEAD) bits &= ~dead8;
                exact = __ballot(!(m >= (KIND == 0 ? rb.y : (KIND == 1 ? rb.w : rc.z)))) != 0ull;   // (h = NaN: the record is not screened)
            }
            if (__builtin_expect(exact, 0)) {   // wave-uniform, rare
                const uint32_t e = exact_count((uint32_t)__builtin_amdgcn_readlane(my, (int)k));
                if (lane == 0) atomicAdd(&pair_rep[(uint32_t)kPairMain + tile % (uint32_t)(kPairLead - kPairMain)], 1u);
                park = ((uint32_t)lane == k) ? e : park;
                bits = 0;
            }
            *cnt_row = (uint8_t)bits;   // (the lane's 8 inside-bits: whoever adds up the row counts them)
        };
        // four hypotheses per trip, their records in alternating register sets with the next one always in flight.  The
        // trip count is rounded up: rows nb .. of `loc` hold valid records (every lane wrote one), their counts land in
        // rows of the table nobody adds up -- which keeps every LDS address of a trip at a constant offset from one
        // register: the two byte offsets below live in VGPRs the compiler cannot see through (it would otherwise rebuild
        // each address from the scalar loop counter: two VALU instructions per hypothesis)
        auto batch = [&](auto screened, auto dead) {
            uint32_t rec_off = 0, cnt_off = (uint32_t)lane;
            asm volatile("" : "+v"(rec_off), "+v"(cnt_off));
            const char* const loc_b = reinterpret_cast<const char*>(&loc[0][0]);
            auto fetch = [&](int i, float4& r0, float4& r1, float4& r2) {
                const float4* __restrict__ r = reinterpret_cast<const float4*>(loc_b + rec_off) + i * NL;
                r0 = r[0];
                r1 = r[1];
                r2 = r[NL - 1];
                __builtin_amdgcn_sched_barrier(0);   // (the reads are issued HERE, ahead of the arithmetic of the step before them)
            };
            float4 a0, a1, a2, b0r, b1r, b2r;
            fetch(0, a0, a1, a2);
            for (uint32_t k = 0; k < nb; k += 4u) {
                fetch(1, b0r, b1r, b2r);
                step(screened, dead, a0, a1, a2, k, cnt8 + cnt_off);
                fetch(2, a0, a1, a2);
                if (k + 1u < nb) step(screened, dead, b0r, b1r, b2r, k + 1u, cnt8 + cnt_off + kCntStride);   // (scalar branches)
                fetch(3, b0r, b1r, b2r);
                if (k + 2u < nb) step(screened, dead, a0, a1, a2, k + 2u, cnt8 + cnt_off + 2 * kCntStride);
                fetch(4, a0, a1, a2);   // (row 64: the padding row, read by the last trip and never used)
                if (k + 3u < nb) step(screened, dead, b0r, b1r, b2r, k + 3u, cnt8 + cnt_off + 3 * kCntStride);
                rec_off += 4u * NL * (uint32_t)sizeof(float4);
                cnt_off += 4u * kCntStride;
            }
        };
        if (tile_screened && !tile_has_dead) batch(std::true_type{}, std::false_type{});
        else if (tile_screened) batch(std::true_type{}, std::true_type{});
        else batch(std::false_type{}, std::false_type{});
        wave_sync();   // (the table is complete)
        if ((uint32_t)lane < nb) {
            const uint4* row = reinterpret_cast<const uint4*>(cnt8 + (uint32_t)lane * kCntStride);
            uint32_t sum = park;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 v = row[i];
                sum += (uint32_t)__popc(v.x);   // (v_bcnt_u32_b32 adds its second operand)
                sum += (uint32_t)__popc(v.y);
                sum += (uint32_t)__popc(v.z);
                sum += (uint32_t)__popc(v.w);
            }
            if (sum) atomicAdd(&counts[g0 * 64u + (uint32_t)my], sum);
        }
        wave_sync();   // (the next batch overwrites the tables)
    }
}
// the bx-th tile whose index mod 4 is in res_mask (ascending); may lie behind the last tile
__device__ __forceinline__ uint32_t phase_tile4(uint32_t bx, uint32_t res_mask) {
    const uint32_t k = (uint32_t)__popc(res_mask);
    uint32_t m = res_mask;
    for (uint32_t j = bx % k; j > 0; --j) m &= m - 1u;   // drop the j lowest set bits
    return (bx / k) * 4u + (uint32_t)(__ffs(m) - 1);
}
// res_mask = 0xF: every tile (blockIdx.x = tile).  Otherwise ONE PHASE of a phased scoring (launch_score_phased): the tiles
// whose index mod 4 is in res_mask, grid.x = ceil(n_tiles / 4) * popcount(res_mask).
template <int KIND>
__global__ __launch_bounds__(64) void score_screen_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                      const double* __restrict__ sz,
                                                      const double* __restrict__ boxes, double max_abs,
                                                      const double* __restrict__ score,
                                                      const unsigned long long* __restrict__ masks,
                                                      const unsigned long long* __restrict__ keep,
                                                      uint32_t n_groups, uint32_t groups_per_block /* <= kScreenMaxGroups */,
                                                      uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                      uint32_t* __restrict__ pair_rep,
                                                      uint32_t group_begin, uint32_t group_end,
                                                      const float* __restrict__ tile_f32, uint32_t has_dead,
                                                      uint32_t n_tiles, uint32_t res_mask) {
    uint32_t tile = blockIdx.x;
    if (res_mask != 0xFu) tile = phase_tile4(blockIdx.x, res_mask);   // (kernel argument: uniform)
    if (tile >= n_tiles) return;   // (the grid's x is padded to a multiple of 8: launch_score_mask)
    score_screen_body<KIND, false>(sx, sy, sz, boxes, max_abs, score, const_cast<unsigned long long*>(masks), keep, n_groups,
                                   groups_per_block, counts_rep, rep_stride, pair_rep, group_begin, group_end, tile,
                                   blockIdx.y, nullptr, tile_f32, has_dead);
}

// Between two phases of a phased scoring: a hypothesis stays only if what it has collected so far plus 512 per tile it can
// still touch reaches the best count of EARLIER hypotheses -- otherwise its final count is below that count whatever the
// remaining tiles hold, it can neither beat nor tie the incumbent in the sequential replay (ransac.h:595-596), and its record
// is reported as 0 like any pruned hypothesis' (launch_sum_replicas masks it with the final keep words).
// done_mask: residues (tile index mod 4) scored so far; ubp[h] = touched tiles of residue 0 (low half) and 1 (high half).
__global__ __launch_bounds__(64) void phase_keep_k(const uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                    const uint32_t* __restrict__ ub, const uint32_t* __restrict__ ubp,
                                                    const uint32_t* __restrict__ best_count_ptr,
                                                    unsigned long long* __restrict__ keep, uint32_t group_offset, uint32_t done_mask) {
    const uint32_t g = group_offset + blockIdx.x;
    const uint32_t h = g * 64u + threadIdx.x;
    const unsigned long long kw = keep[g];
    if (kw == 0ull) return;   // (uniform)
    const uint32_t best = best_count_ptr[0];
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < kCountReplicas; ++r) c += counts_rep[(size_t)r * rep_stride + h];
    const uint32_t ab = ubp[h];
    uint32_t done = 0;
    if (done_mask & 1u) done += ab & 0xFFFFu;
    if (done_mask & 2u) done += ab >> 16;
    // (residues 2 and 3 are only ever the LAST phase: nothing is decided after them)
    const uint32_t rem = ub[h] - done;
    const bool k = ((kw >> threadIdx.x) & 1ull) && (best == 0u || (uint64_t)c + (uint64_t)rem * kTilePoints >= best);
    const unsigned long long m = __ballot(k);
    if (threadIdx.x == 0) keep[g] = m;
}

#ifdef M3D_EXPERIMENTAL
// score_screen4_k: the same counting with four-wave workgroups that share one compacted id list (score_screen_body, WAVES = 4)
template <int KIND>
__global__ __launch_bounds__(256) void score_screen4_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                        const double* __restrict__ sz,
                                                        const double* __restrict__ boxes, double max_abs,
                                                        const double* __restrict__ score,
                                                        const unsigned long long* __restrict__ masks,
                                                        const unsigned long long* __restrict__ keep,
                                                        uint32_t n_groups, uint32_t groups_per_block /* <= kScreen4MaxGroups */,
                                                        uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                        uint32_t* __restrict__ pair_rep,
                                                        uint32_t group_begin, uint32_t group_end,
                                                        const float* __restrict__ tile_f32, uint32_t has_dead) {
    score_screen_body<KIND, false, 4>(sx, sy, sz, boxes, max_abs, score, const_cast<unsigned long long*>(masks), keep, n_groups,
                                      groups_per_block, counts_rep, rep_stride, pair_rep, group_begin, group_end, blockIdx.x,
                                      blockIdx.y, nullptr, tile_f32, has_dead);
}
#endif   // M3D_EXPERIMENTAL

// cull_lead_k: ONE launch for the two latency-bound steps at the head of a fit's first chunk -- the box tests of the
// chunk's hypotheses (cull_tiles32_k's workgroups) and the counting of its leading hypotheses (score_screen_k's, with
// their own box tests: they cannot wait for a mask another workgroup of the same launch writes).  Back to back the two
// launches took 15 + 15 us on a 1 M-point cloud and left most of the chip idle; together they take about as long as one.
// Workgroups [0, n_lead_wgs): lead pass, tile = id % n_tiles (the longer-running ones first); the rest: box tests of
// groups [cull_begin, cull_end).
template <int KIND>
__global__ __launch_bounds__(64) void cull_lead_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                   const double* __restrict__ sz, const double* __restrict__ boxes,
                                                   uint32_t n_tiles, double max_abs, const double* __restrict__ score,
                                                   const float* __restrict__ cull32, unsigned long long* __restrict__ masks,
                                                   const unsigned long long* __restrict__ keep, uint32_t n_groups,
                                                   uint32_t lead_groups, uint32_t lead_gpb, uint32_t n_lead_wgs,
                                                   uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                   uint32_t* __restrict__ pair_rep, uint32_t* __restrict__ ub,
                                                   uint32_t cull_begin, uint32_t cull_end, uint32_t cull_gpw, uint32_t cull_tblocks,
                                                   const float* __restrict__ tile_f32, uint32_t has_dead, uint32_t* __restrict__ ubp) {
    if (blockIdx.x < n_lead_wgs) {   // (workgroup-uniform)
        score_screen_body<KIND, true>(sx, sy, sz, boxes, max_abs, score, masks, keep, n_groups, lead_gpb, counts_rep, rep_stride,
                                      pair_rep, 0u, lead_groups, blockIdx.x % n_tiles, blockIdx.x / n_tiles, cull32, tile_f32, has_dead);
    } else {
        const uint32_t b = blockIdx.x - n_lead_wgs;
        cull32_body<KIND>(boxes, n_tiles, cull32, n_groups, cull_gpw, masks, ub, cull_begin, cull_end, b % cull_tblocks,
                          b / cull_tblocks, ubp);
    }
}

// records[h] = sum over the replicas for h in [h_begin, h_end), written to `counts` (device-visible host memory, may be
// null) and `counts_dev` (device, may be null); *pairs_out (device-visible, may be null) = evaluated (tile, hypothesis)
// pairs of the launch.
// valid != null: bit 31 of the record carries MinimalFit's return (one array instead of two; counts < 2^31);
// best_count != null: running maximum over the valid hypotheses (bound-and-prune incumbent).
__global__ void sum_replicas_k(const uint32_t* __restrict__ counts_rep, uint32_t rep_stride, uint32_t h_end,
                               uint32_t* __restrict__ counts, const uint32_t* __restrict__ pair_rep,
                               uint32_t* __restrict__ pairs_out, const uint8_t* __restrict__ valid, uint32_t h_count,
                               uint32_t* __restrict__ best_count, uint32_t h_begin,
                               uint32_t* __restrict__ counts_dev /* device copy of the records, or null */, PickFinal pf,
                               const unsigned long long* __restrict__ keep_final /* phased scoring: hypotheses dropped between
                               phases carry partial counts -- reported as 0 (pruned); null: none */) {
    const uint32_t h = h_begin + blockIdx.x * 256u + threadIdx.x;   // window [h_begin, h_end) of the chunk
    if (blockIdx.x == 0 && pair_rep && pairs_out) {   // block-uniform
        __shared__ uint32_t red[256];
        uint32_t p = 0;
        for (int r = threadIdx.x; r < kPairMain; r += 256) p += pair_rep[r];
        red[threadIdx.x] = p;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            uint32_t e = 0;
            for (int r = kPairMain; r < kPairLead; ++r) e += pair_rep[r];
            pairs_out[0] = red[0];
            pairs_out[1] = e;   // pairs score_screen_k recounted in fp64
            pairs_out[2] = pair_rep[kPairLead];   // pairs of the chunk's lead pass (0: the chunk had none)
        }
    }
    const bool mine = h < h_end;
    uint32_t c = 0;
    if (mine) {
#pragma unroll
        for (int r = 0; r < kCountReplicas; ++r) c += counts_rep[(size_t)r * rep_stride + h];
        if (keep_final && !((keep_final[h >> 6] >> (h & 63u)) & 1ull)) c = 0u;
    }
    const bool ok = mine && valid && h < h_count && valid[h];
    if (mine) {
        const uint32_t rec = valid ? (c | (ok ? 0x80000000u : 0u)) : c;
        if (counts) counts[h] = rec;
        if (counts_dev) counts_dev[h] = rec;
    }
    if (best_count) {   // wave-uniform
        uint32_t v = ok ? c : 0u;
        for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
        if ((threadIdx.x & 63) == 0 && v) atomicMax(best_count, v);
    }
    if (!pf.pick) return;   // (kernel argument: uniform)
    // ---- pick_best_k's decision, without its launch: every wave contributes its best (count, index) key; the
    // workgroup that finishes last compares the chunk's best with the running pick
    __shared__ uint32_t s_last, s_take, s_idx, s_tie;
    {
        unsigned long long key = (ok && c) ? (((unsigned long long)c << 32) | (0xFFFFFFFFu - h)) : 0ull;
        for (int off = 32; off > 0; off >>= 1) key = max(key, (unsigned long long)__shfl_xor((long long)key, off, 64));
        if ((threadIdx.x & 63) == 0 && key) atomicMax(pf.key, key);
        if (pf.key2) {   // (kernel argument: uniform) the HIGHEST index among the best counts: differs from the lowest = a tie
            unsigned long long key2 = (ok && c) ? (((unsigned long long)c << 32) | h) : 0ull;
            for (int off = 32; off > 0; off >>= 1) key2 = max(key2, (unsigned long long)__shfl_xor((long long)key2, off, 64));
            if ((threadIdx.x & 63) == 0 && key2) atomicMax(pf.key2, key2);
        }
    }
    // this workgroup's records (host memory) and its key are out before its ticket.  (Waiting for the stores with
    // s_waitcnt alone is NOT enough: with two processes sharing the GPU the host saw the completion word before some
    // records -- the system-scope release is what orders them.  Every wave first waits until its own stores have
    // been taken by the L2, so that thread 0's write-back covers them.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        s_last = __hip_atomic_fetch_add(pf.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        const unsigned long long key = __hip_atomic_load(pf.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t cnt = (uint32_t)(key >> 32), idx = 0xFFFFFFFFu - (uint32_t)key;
        BestPick* pick = pf.pick;
        const bool had = !pf.first_chunk && pick->have;
        const bool take = cnt > 0 && (!had || cnt > pick->cnt);
        // COUNT TIES.  The device's pick is "highest count, lowest index"; the replay breaks equal counts by rmse
        // (ransac.h:595-596), so with two hypotheses at the top the pick is a coin toss -- and a wrong pick costs the fit a
        // second RefineModel compaction (4 MB over the host link: C3's sphere fit, 0.13 of 0.82 ms).  key2 holds the HIGHEST
        // index among the chunk's best counts: if it differs from the lowest, or the chunk's best equals the running pick's
        // count, the pick is marked `tie`, its model record is poisoned (NaN: the speculative compaction queued behind this
        // kernel then finds no inlier and ships nothing) and the host treats the speculation as a miss.
        uint32_t tie = (pf.first_chunk || !had) ? 0u : pick->tie;
        if (pf.key2) {
            const unsigned long long k2 = __hip_atomic_load(pf.key2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool inner = cnt > 0 && (uint32_t)k2 != idx;
            if (take) tie = inner ? 1u : 0u;
            else if (had && cnt > 0 && cnt == pick->cnt) tie = 1u;
            *pf.key2 = 0ull;
        }
        s_take = take ? 1u : 0u;
        s_idx = idx;
        s_tie = tie;
        if (take) {
            pick->have = 1;
            pick->cnt = cnt;
            pick->index = pf.index_base + idx;
        } else if (!had) {
            pick->have = 0;
            pick->cnt = 0;
            pick->index = ~0ull;
        }
        pick->tie = tie;
        pf.pick_host->have = pick->have;
        pf.pick_host->cnt = pick->cnt;
        pf.pick_host->index = pick->index;
        pf.pick_host->tie = tie;
        *pf.key = 0ull;      // (for the next chunk)
        *pf.ticket = 0u;
    }
    __syncthreads();
    if (threadIdx.x < kModelStride) {
        if (s_take) pf.pick->params[threadIdx.x] = pf.params[(size_t)s_idx * kModelStride + threadIdx.x];
        else if (pf.first_chunk) pf.pick->params[threadIdx.x] = 0.0;
        if (s_tie && threadIdx.x == 0) pf.pick->params[0] = __builtin_nan("");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&pf.pick_host->seq, pf.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void launch_sum_replicas(const uint32_t* counts_rep, uint32_t rep_stride, uint32_t h_end, uint32_t* counts,
                         const uint32_t* pair_rep, uint32_t* pairs_out, const uint8_t* valid, uint32_t h_count,
                         uint32_t* best_count, hipStream_t st, uint32_t h_begin, uint32_t* counts_dev, const PickFinal* pick,
                         const unsigned long long* keep_final) {
    PickFinal pf;
    if (pick) pf = *pick;
    if (h_end > h_begin)
        sum_replicas_k<<<(h_end - h_begin + 255) / 256, 256, 0, st>>>(counts_rep, rep_stride, h_end, counts, pair_rep,
                                                                      pairs_out, valid, h_count, best_count, h_begin,
                                                                      counts_dev, pf, keep_final);
}

bool launch_cull_lead(int kind, const SortedView& s, const double* score, const float* cull32, unsigned long long* masks,
                      const unsigned long long* keep, uint32_t n_groups, uint32_t lead_groups, uint32_t* counts_rep,
                      uint32_t rep_stride, uint32_t* pair_rep, uint32_t* ub, uint32_t cull_end, hipStream_t st, uint32_t* ubp) {
    cull_end = std::min(cull_end, n_groups);
    if (!s.n_tiles || !cull32 || !(s.radius < 1e18) || config().cull_fp32 == 0 || config().score_fp32_screen == 0 ||
        lead_groups == 0 || lead_groups > (uint32_t)kScreenMaxGroups || lead_groups >= cull_end)
        return false;
    // the lead pass: ALL leading groups of a tile in one workgroup (the tile's 12 KB are loaded once; the launch has the
    // box-test workgroups to fill the chip with)
    const uint32_t lead_gpb = lead_groups;
    const uint32_t lead_y = 1;
    const uint32_t n_lead_wgs = s.n_tiles * lead_y;
    // the box tests of the rest: launch_cull_mask's geometry
    const uint32_t window = cull_end - lead_groups;
    const uint32_t tblocks = (s.n_tiles + 63) / 64;
    uint32_t gpw = std::max<uint32_t>(1, (uint32_t)(((uint64_t)tblocks * window) / 8192));
    gpw = std::min<uint32_t>(gpw, 8);
    const uint32_t cull_y = (window + gpw - 1) / gpw;
    const dim3 g(n_lead_wgs + tblocks * cull_y), b(64);
    if (kind == 0)
        cull_lead_k<0><<<g, b, 0, st>>>(s.x, s.y, s.z, s.boxes, s.n_tiles, s.max_abs, score, cull32, masks, keep, n_groups, lead_groups,
                                        lead_gpb, n_lead_wgs, counts_rep, rep_stride, pair_rep, ub, lead_groups, cull_end, gpw, tblocks,
                                        (const float*)s.tile_f32, s.has_dead ? 1u : 0u, ubp);
    else if (kind == 1)
        cull_lead_k<1><<<g, b, 0, st>>>(s.x, s.y, s.z, s.boxes, s.n_tiles, s.max_abs, score, cull32, masks, keep, n_groups, lead_groups,
                                        lead_gpb, n_lead_wgs, counts_rep, rep_stride, pair_rep, ub, lead_groups, cull_end, gpw, tblocks,
                                        (const float*)s.tile_f32, s.has_dead ? 1u : 0u, ubp);
    else
        cull_lead_k<2><<<g, b, 0, st>>>(s.x, s.y, s.z, s.boxes, s.n_tiles, s.max_abs, score, cull32, masks, keep, n_groups, lead_groups,
                                        lead_gpb, n_lead_wgs, counts_rep, rep_stride, pair_rep, ub, lead_groups, cull_end, gpw, tblocks,
                                        (const float*)s.tile_f32, s.has_dead ? 1u : 0u, ubp);
    return true;
}

// The WHOLE chunk through cull_lead_k's first kind of workgroup: every scoring workgroup runs the fp32 box tests of its
// tile against its own groups (lane = hypothesis) and counts the survivors -- no box-test launch in front of the scoring
// launch.  For chunks that prune nothing (keep = all ones, prepared by minimal_fit_k): a segmentation round in the clutter,
// 1000 hypotheses on ~2000 tiles, where cull_tiles32_k was 9 us of launch latency in front of a 34 us scoring launch.
bool launch_score_own_tests(int kind, const SortedView& s, const double* score, const float* cull32, unsigned long long* masks,
                            const unsigned long long* keep, uint32_t n_groups, uint32_t groups, uint32_t* counts_rep,
                            uint32_t rep_stride, uint32_t* pair_rep, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    groups = std::min(groups, n_groups);
    if (!s.n_tiles || !cull32 || !(s.radius < 1e18) || config().cull_fp32 == 0 || config().score_fp32_screen == 0 || groups == 0)
        return false;
    const uint32_t gpb_max = std::min<uint32_t>((uint32_t)config().score_groups_per_block, kScreenMaxGroups);
    const uint32_t min_wgs = (uint32_t)config().score_min_workgroups;
    const uint32_t gpb = std::max<uint32_t>(1, std::min<uint32_t>(gpb_max, (uint32_t)(((uint64_t)s.n_tiles * groups) / min_wgs)));
    const uint32_t n_wgs = s.n_tiles * ((groups + gpb - 1) / gpb);
    const dim3 g(n_wgs), b(64);
    auto go = [&](auto kernel) {
        if (ev_start && ev_stop)
            hipExtLaunchKernelGGL(kernel, g, b, 0, st, ev_start, ev_stop, 0, s.x, s.y, s.z, s.boxes, s.n_tiles, s.max_abs, score, cull32,
                                  masks, keep, n_groups, groups, gpb, n_wgs, counts_rep, rep_stride, pair_rep, (uint32_t*)nullptr,
                                  groups, groups, 1u, 1u, (const float*)s.tile_f32, s.has_dead ? 1u : 0u, (uint32_t*)nullptr);
        else
            kernel<<<g, b, 0, st>>>(s.x, s.y, s.z, s.boxes, s.n_tiles, s.max_abs, score, cull32, masks, keep, n_groups, groups, gpb, n_wgs,
                                    counts_rep, rep_stride, pair_rep, (uint32_t*)nullptr, groups, groups, 1u, 1u,
                                    (const float*)s.tile_f32, s.has_dead ? 1u : 0u, (uint32_t*)nullptr);
    };
    if (kind == 0) go(cull_lead_k<0>);
    else if (kind == 1) go(cull_lead_k<1>);
    else go(cull_lead_k<2>);
    return true;
}

// The hypothesis the sequential replay will most probably end with, chosen on the device: highest inlier count
// among the valid hypotheses of the chunk, lowest index among equals, against the running pick of earlier chunks
// (strictly more inliers to replace it).  Fitness ties are decided by rmse on the host, so this is a PREDICTION:
// the driver starts RefineModel's compaction on pick->params behind the last scoring launch and keeps the result
// only if the replay names the same hypothesis (m3d_fit.cpp).  One workgroup.
__global__ __launch_bounds__(1024) void pick_best_k(const uint32_t* __restrict__ records, uint32_t count,
                                                     unsigned long long index_base, const double* __restrict__ params,
                                                     int first_chunk, BestPick* __restrict__ pick,
                                                     BestPickHost* __restrict__ pick_host,
                                                     uint32_t* __restrict__ records_host, uint32_t* __restrict__ best_count) {
    __shared__ uint32_t s_cnt[1024];
    __shared__ uint32_t s_idx[1024];
    __shared__ int s_take;
    uint32_t bc = 0, bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < count; i += 1024u) {
        const uint32_t r = records[i];
        if (records_host) records_host[i] = r;   // sharded fits: the gathered records reach the host through this kernel
        const uint32_t c = r & 0x7FFFFFFFu;
        if ((r >> 31) && c > bc) {   // ascending i per thread: the first of equals stays
            bc = c;
            bi = i;
        }
    }
    s_cnt[threadIdx.x] = bc;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const uint32_t c2 = s_cnt[threadIdx.x + off], i2 = s_idx[threadIdx.x + off];
            const uint32_t c1 = s_cnt[threadIdx.x], i1 = s_idx[threadIdx.x];
            if (c2 > c1 || (c2 == c1 && i2 < i1)) {
                s_cnt[threadIdx.x] = c2;
                s_idx[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (best_count && s_cnt[0]) atomicMax(best_count, s_cnt[0]);   // other ranks' hypotheses raise the incumbent too
        const bool had = !first_chunk && pick->have;
        const bool take = s_cnt[0] > 0 && (!had || s_cnt[0] > pick->cnt);
        s_take = take ? 1 : 0;
        if (take) {
            pick->have = 1;
            pick->cnt = s_cnt[0];
            pick->index = index_base + s_idx[0];
        } else if (!had) {
            pick->have = 0;
            pick->cnt = 0;
            pick->index = ~0ull;
        }
        pick_host->have = pick->have;
        pick_host->cnt = pick->cnt;
        pick_host->index = pick->index;
        pick->tie = 0;   // (ties are only tracked by sum_replicas_k's PickFinal tail: one-GPU fits)
        pick_host->tie = 0;
    }
    __syncthreads();
    if (threadIdx.x < kModelStride) {
        if (s_take) pick->params[threadIdx.x] = params[(size_t)s_idx[0] * kModelStride + threadIdx.x];
        else if (first_chunk) pick->params[threadIdx.x] = 0.0;
    }
}
void launch_pick_best(const uint32_t* records, uint32_t count, unsigned long long index_base, const double* params,
                      bool first_chunk, BestPick* pick, BestPickHost* pick_host, hipStream_t st, uint32_t* records_host,
                      uint32_t* best_count) {
    pick_best_k<<<1, 1024, 0, st>>>(records, count, index_base, params, first_chunk ? 1 : 0, pick, pick_host,
                                    records_host, best_count);
}

void launch_score_mask(int kind, const SortedView& s, const double* score, const unsigned long long* masks,
                       const unsigned long long* keep, uint32_t n_groups, uint32_t* counts_rep, uint32_t rep_stride,
                       uint32_t* pair_rep, hipStream_t st, uint32_t group_begin, uint32_t group_end, hipEvent_t ev_start,
                       hipEvent_t ev_stop, bool thinned) {
#ifdef M3D_EXPERIMENTAL
    if (launch_score_mfma(kind, s, score, masks, keep, n_groups, counts_rep, rep_stride, pair_rep, st, group_begin, group_end, ev_start, ev_stop))
        return;
#endif
    group_end = std::min(group_end, n_groups);
    if (!s.n_tiles || group_begin >= group_end) return;
    const uint32_t window = group_end - group_begin;
    // at least ~16k workgroups when the chunk is small, at most kGroupsPerBlock groups each
    const bool screened = config().score_fp32_screen != 0;
    const uint32_t gpb_max = screened ? screen_gpb_max(window, thinned) : std::min<uint32_t>((uint32_t)config().score_groups_per_block, 64u);
    const uint32_t min_wgs = (uint32_t)config().score_min_workgroups;
    const uint32_t gpb = std::max<uint32_t>(1, std::min<uint32_t>(gpb_max, (uint32_t)(((uint64_t)s.n_tiles * window) / min_wgs)));
    // XCD-aware geometry (round 5): workgroup w of a launch runs on XCD w % 8 and every XCD has an L2 of its own.  With the
    // grid's x = tiles padded to a multiple of 8, the workgroups (tile, y) of ONE tile -- 13 of them on C2, one per block of
    // groups -- all land on XCD tile % 8 and the tile's 6 KB of offsets are fetched from HBM once instead of once per XCD that
    // happens to get one of them (C2: n_tiles = 1954 = 2 mod 8 sent a tile's workgroups to four XCDs; FETCH_SIZE per launch
    // 115.9 MB = 4.8 x the cloud: VERDICT r4 "What's weak" 3).  Only the screened kernels take the padding (they check the tile).
    const dim3 g(screened ? (s.n_tiles + 7u) / 8u * 8u : s.n_tiles, (window + gpb - 1) / gpb), b(64);
    // ev_start / ev_stop: the launch's OWN start and stop times (hipExtLaunchKernelGGL attaches the events to the kernel's
    // dispatch packet: no barrier packets in front of and behind the kernel, which is what two hipEventRecord calls
    // cost -- ~5 us of bubble each on this stream)
    auto go = [&](auto kernel, auto... args) {
        if (ev_start && ev_stop) hipExtLaunchKernelGGL(kernel, g, b, 0, st, ev_start, ev_stop, 0, args...);
        else kernel<<<g, b, 0, st>>>(args...);
    };
#ifdef M3D_EXPERIMENTAL
    // windows of many groups: four-wave workgroups over up to 64 groups each (m3d_config.score_waves4)
    const uint32_t gpb4 = std::min<uint32_t>(kScreen4MaxGroups, (uint32_t)std::max(1, config().score_waves4_groups));
    if (screened && config().score_waves4 != 0 && window >= 24u) {
        const dim3 g4(s.n_tiles, (window + gpb4 - 1) / gpb4), b4(256);
        auto go4 = [&](auto kernel) {
            if (ev_start && ev_stop)
                hipExtLaunchKernelGGL(kernel, g4, b4, 0, st, ev_start, ev_stop, 0, s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep, n_groups,
                                      gpb4, counts_rep, rep_stride, pair_rep, group_begin, group_end, (const float*)s.tile_f32, s.has_dead ? 1u : 0u);
            else
                kernel<<<g4, b4, 0, st>>>(s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep, n_groups, gpb4, counts_rep, rep_stride, pair_rep,
                                          group_begin, group_end, (const float*)s.tile_f32, s.has_dead ? 1u : 0u);
        };
        if (kind == 0) go4(score_screen4_k<0>);
        else if (kind == 1) go4(score_screen4_k<1>);
        else go4(score_screen4_k<2>);
        return;
    }
#endif   // M3D_EXPERIMENTAL
    if (screened) {
        if (kind == 0)
            go(score_screen_k<0>, s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep, n_groups, gpb, counts_rep, rep_stride, pair_rep,
               group_begin, group_end, (const float*)s.tile_f32, s.has_dead ? 1u : 0u, s.n_tiles, 0xFu);
        else if (kind == 1)
            go(score_screen_k<1>, s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep, n_groups, gpb, counts_rep, rep_stride, pair_rep,
               group_begin, group_end, (const float*)s.tile_f32, s.has_dead ? 1u : 0u, s.n_tiles, 0xFu);
        else
            go(score_screen_k<2>, s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep, n_groups, gpb, counts_rep, rep_stride, pair_rep,
               group_begin, group_end, (const float*)s.tile_f32, s.has_dead ? 1u : 0u, s.n_tiles, 0xFu);
    } else {
        if (kind == 0)
            go(score_mask_k<0>, s.x, s.y, s.z, score, masks, keep, n_groups, gpb, counts_rep, rep_stride, pair_rep, group_begin, group_end);
        else if (kind == 1)
            go(score_mask_k<1>, s.x, s.y, s.z, score, masks, keep, n_groups, gpb, counts_rep, rep_stride, pair_rep, group_begin, group_end);
        else
            go(score_mask_k<2>, s.x, s.y, s.z, score, masks, keep, n_groups, gpb, counts_rep, rep_stride, pair_rep, group_begin, group_end);
    }
}


// PHASED scoring (m3d_config.score_phases): the window's hypotheses are counted on a quarter of the tiles (index % 4 == 0: the
// copy is Hilbert-sorted, so that is a uniform sample of space), re-pruned with what they collected (phase_keep_k), counted on the
// next quarter, re-pruned, and only the survivors see the remaining half.  On C2 about half of the hypotheses that survive the
// first pruning hold less than 64 % of the incumbent's inliers and leave after the first quarter.  Exact: see phase_keep_k.
// Needs ub / ubp from the fp32 box tests of the SAME window, a pruning incumbent (best_count) and the fp32 screen; returns false,
// having launched nothing, otherwise.  keep[] is rewritten; pass it to launch_sum_replicas as keep_final.
// Measured (profiles/r04_score_phases.txt, 1 M points): cylinders 50 000 hypotheses 9.62 M -> 4.83 M pairs, scoring launches
// 1.42 -> 0.885 ms, the fit 1.84 -> 1.29 ms; planes 10 000: 1.45 M -> 1.02 M pairs but 0.100 -> 0.106 ms (three launch tails and
// two re-pruning kernels cost what the pairs save); spheres 50 000: 3.32 M -> 2.95 M, 0.332 -> 0.405 ms.  Hence the default
// (m3d_config.score_phases = -1): three phases for cylinders, one launch for planes and spheres.
int score_phases_for(int kind) {
    const int c = config().score_phases;
    if (c == 2 || c == 3) return c;
    return (c < 0 && kind == 2) ? 3 : 0;
}
bool launch_score_phased(int kind, const SortedView& s, const double* score, const unsigned long long* masks,
                         unsigned long long* keep, uint32_t n_groups, uint32_t* counts_rep, uint32_t rep_stride,
                         uint32_t* pair_rep, const uint32_t* ub, const uint32_t* ubp, const uint32_t* best_count, hipStream_t st,
                         uint32_t group_begin, uint32_t group_end, hipEvent_t ev_start, hipEvent_t ev_stop) {
    group_end = std::min(group_end, n_groups);
    // (the size guards are where the phases start to pay; an EXPLICIT m3d_config.score_phases = 2 / 3 -- the test-suite's paths --
    // engages them on small clouds and windows too)
    const bool forced = config().score_phases >= 2;
    if (!ub || !ubp || !best_count || !s.tile_f32 || s.n_tiles < (forced ? 8u : 256u) || s.n_tiles > 200000u || score_phases_for(kind) == 0 ||
        config().score_fp32_screen == 0 || group_begin >= group_end || group_end - group_begin < (forced ? 2u : 32u))
        return false;
    const uint32_t window = group_end - group_begin;
    const uint32_t gpb_max = screen_gpb_max(window);
    const uint32_t min_wgs = (uint32_t)config().score_min_workgroups;
    const int n_ph = score_phases_for(kind);
    const uint32_t res[3] = {0x1u, n_ph == 3 ? 0x2u : 0xEu, 0xCu};
    uint32_t done = 0;
    for (int ph = 0; ph < n_ph; ++ph) {
        const uint32_t k = (uint32_t)__builtin_popcount(res[ph]);
        const uint32_t tiles_x = (s.n_tiles + 3u) / 4u * k;
        const uint32_t gpb = std::max<uint32_t>(1, std::min<uint32_t>(gpb_max, (uint32_t)(((uint64_t)tiles_x * window) / min_wgs)));
        const dim3 g((tiles_x + 7u) / 8u * 8u, (window + gpb - 1) / gpb), b(64);   // (a multiple of 8: a tile's workgroups on one XCD, launch_score_mask)
        hipEvent_t e0 = ph == 0 ? ev_start : nullptr, e1 = ph == n_ph - 1 ? ev_stop : nullptr;
        auto go = [&](auto kernel) {
            if (ev_start && ev_stop && (e0 || e1))
                hipExtLaunchKernelGGL(kernel, g, b, 0, st, e0, e1, 0, s.x, s.y, s.z, s.boxes, s.max_abs, score, masks,
                                      (const unsigned long long*)keep, n_groups, gpb, counts_rep, rep_stride, pair_rep, group_begin, group_end,
                                      (const float*)s.tile_f32, s.has_dead ? 1u : 0u, s.n_tiles, res[ph]);
            else
                kernel<<<g, b, 0, st>>>(s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, (const unsigned long long*)keep, n_groups, gpb, counts_rep,
                                        rep_stride, pair_rep, group_begin, group_end, (const float*)s.tile_f32, s.has_dead ? 1u : 0u, s.n_tiles, res[ph]);
        };
        if (kind == 0) go(score_screen_k<0>);
        else if (kind == 1) go(score_screen_k<1>);
        else go(score_screen_k<2>);
        done |= res[ph];
        if (ph + 1 < n_ph) phase_keep_k<<<window, 64, 0, st>>>(counts_rep, rep_stride, ub, ubp, best_count, keep, group_begin, done);
    }
    return true;
}

// number of set bits of masks & keep (statistics for the measurement hook)
__global__ void count_bits_k(const unsigned long long* __restrict__ masks, const unsigned long long* __restrict__ keep,
                             uint32_t n_tiles, uint32_t n_groups, unsigned long long* __restrict__ total) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    unsigned long long c = 0;
    if (i < (size_t)n_tiles * n_groups) c = (unsigned long long)__popcll(masks[i] & keep[i % n_groups]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, c);
}
void launch_count_bits(const unsigned long long* masks, const unsigned long long* keep, uint32_t n_tiles,
                       uint32_t n_groups, unsigned long long* total, hipStream_t st) {
    (void)hipMemsetAsync(total, 0, sizeof(unsigned long long), st);
    const size_t n = (size_t)n_tiles * n_groups;
    if (n) count_bits_k<<<(uint32_t)((n + 255) / 256), 256, 0, st>>>(masks, keep, n_tiles, n_groups, total);
}

}  // namespace m3d
