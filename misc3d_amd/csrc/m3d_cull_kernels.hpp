// m3d_cull_kernels.hpp -- launch interface of the spatially culled scoring path (m3d_cull_kernels.hip).
#pragma once
#include "m3d_kernels.hpp"

namespace m3d {

constexpr int kTilePoints = 512;    // points per tile = one wave x 8 rows of 64 (kept in VGPRs)
constexpr int kListSegment = 256;   // listed hypotheses per score_list_k workgroup

// Z-order sorted copy of a resident cloud: SoA padded with NaN to a multiple of kTilePoints, plus
// one bounding box per tile (centre xyz, half extents xyz; half < 0 marks an empty tile).
struct SortedView {
    const double* x;
    const double* y;
    const double* z;
    const double* boxes;  // n_tiles x 6
    uint32_t n_tiles;
};

void launch_tile_boxes(const SortedView& s, double* boxes, hipStream_t st);

// lists: n_tiles x h_cap uint32 (surviving hypothesis ids per tile, ascending); list_count: n_tiles.
void launch_cull(int kind, const SortedView& s, const double* score, const uint8_t* valid, uint32_t h_count,
                 uint32_t h_cap, uint32_t* lists, uint32_t* list_count, hipStream_t st);

// counts[h] += inliers of hypothesis h inside the listed tiles; counts must be zero on entry.
void launch_score_list(int kind, const SortedView& s, const double* score, const uint32_t* lists,
                       const uint32_t* list_count, uint32_t h_cap, uint32_t h_count, uint32_t* counts,
                       hipStream_t st);

}  // namespace m3d
