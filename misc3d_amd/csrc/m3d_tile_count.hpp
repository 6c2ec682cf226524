// m3d_tile_count.hpp -- the exact fp64 count of one (tile, hypothesis) pair, shared by the scoring kernels
// (m3d_cull_kernels.hip: score_mask_k and the recount path of score_screen_k; m3d_score_mfma.hip: the recount path of
// score_mfma_k).  The per-point arithmetic is m3d_fp.hpp's -- the reference's own expressions and association.
#pragma once
#include <hip/hip_runtime.h>

#include "m3d_fp.hpp"
#include "m3d_kernels.hpp"

#pragma clang fp contract(off)

namespace m3d {

// Inner loop of score_mask_k: inlier count of ONE hypothesis record over the wave's 512 points (wave-uniform result)
template <int KIND, int P>
__device__ __forceinline__ uint32_t tile_count(const double (&rec)[kModelStride], const double (&x)[P], const double (&y)[P],
                                               const double (&z)[P]) {
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
    return cnt;
}

}  // namespace m3d
