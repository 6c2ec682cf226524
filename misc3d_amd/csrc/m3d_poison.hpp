// m3d_poison.hpp -- the tombstone pass of a segmentation round (device code shared by poison_plane_inliers_k,
// m3d_cull_kernels.hip, and the launch of minimal_fit_k<0> it may ride in, m3d_kernels.hip).
//
// A clutter round removes ~0.5 % of the cloud; the stable partition of the Hilbert-sorted copy rewrote all of it (count +
// write + fresh tile boxes: 24 us of a round on 1 M points).  Counts do not depend on the presence of points that are
// nobody's inlier, so such a round KILLS its inliers in place: x := NaN in the fp64 copy (the exact code's `|s| < T` is
// false for a NaN) and in the tile's fp32 offsets (score_screen_k masks the lane's bit).  Boxes go stale -- they still
// contain every live point, which is all the box tests need.  One wave per tile; a tile the plane's slab misses (the box
// test of the scoring pass, same record) is not read.  The driver compacts for real (compact_write_k mode 3 drops NaN) when
// the dead reach an eighth of the copy.  *total accumulates the kills (checked against the inlier lists, later).
#pragma once
#include "m3d_cull_kernels.hpp"
#include "m3d_fp.hpp"

namespace m3d {

// `true` = the box cannot contain an inlier of the plane record rec = (a, b, c, d, T, cut): inlier <=> |fl(a x + b y + c z + d)|
// < T; over the box the value ranges over [s - r, s + r], the rounded per-point value differs from it by < 8 u * mag;
// cut = T + margin (minimal_fit_k); inf or NaN keeps the tile; !(T > 0): `num < T` can never hold; hx < 0: empty tile
__device__ __forceinline__ bool plane_box_culled(const double* __restrict__ rec, const double* __restrict__ box) {
    const double cx = box[0], cy = box[1], cz = box[2], hx = box[3], hy = box[4], hz = box[5];
    const double a = rec[0], b = rec[1], c = rec[2], d = rec[3], T = rec[4], cut = rec[5];
    const double s = ((a * cx + b * cy) + c * cz) + d;
    const double r = (fabs(a) * hx + fabs(b) * hy) + fabs(c) * hz;
    return (hx < 0.0) | !(T > 0.0) | (fabs(s) - r > cut);
}

__device__ __forceinline__ void poison_tile(const PoisonJob& job, uint32_t tile, int lane) {
    if (tile >= job.n_tiles) return;   // (wave-uniform)
    double m[4];
    for (int k = 0; k < 4; ++k) m[k] = job.model[k];
    // the scoring record of this model (minimal_fit_k): the exact cut-off and the box test's margin
    double rec[6];
    rec[0] = m[0];
    rec[1] = m[1];
    rec[2] = m[2];
    rec[3] = m[3];
    rec[4] = plane_cutoff(m, job.thr);
    rec[5] = rec[4] + 1e-12 * ((((fabs(m[0]) + fabs(m[1])) + fabs(m[2])) * job.max_abs + fabs(m[3])) + rec[4]);
    if (plane_box_culled(rec, job.boxes + (size_t)tile * kBoxStride)) return;   // (wave-uniform)
    uint32_t kills = 0;
    const double nan = u2f(0x7FF8000000000000ull);
    float* __restrict__ tf = job.tile_f32 ? job.tile_f32 + (size_t)tile * kTileF32Floats : nullptr;
#pragma unroll
    for (int r = 0; r < kTilePoints / 64; ++r) {
        const size_t i = (size_t)tile * kTilePoints + (size_t)r * 64 + lane;
        const bool inl = plane_distance(m, job.sx[i], job.sy[i], job.sz[i]) < job.thr;   // RefineModel's own predicate (NaN: false)
        if (inl) {
            job.sx[i] = nan;
            if (tf) tf[(((r >> 1) * 64) + lane) * 2 + (r & 1)] = f32_nan();   // x offsets: rows 2 j, 2 j + 1 side by side
        }
        kills += (uint32_t)__popcll(__ballot(inl));
    }
    // (kills are counted by waves that killed: ~250 of ~2000 in a clutter round; nobody waits for the sum)
    if (lane == 0 && kills) atomicAdd(job.total, kills);
}

}  // namespace m3d
