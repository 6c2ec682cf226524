// misc3d/common/normal_estimation.h -- host mirror of the reference's
// include/misc3d/common/normal_estimation.h (EstimateNormalsFromMap, src/normal_estimation.cpp:180-207)
// over the C ABI (m3d_normals_from_map).  Header-only; no Eigen / Open3D needed.
#pragma once
#include <array>
#include <tuple>
#include <vector>

#include <misc3d/geometry.h>
#include <misc3d/logging.h>
#include <misc3d_amd.h>

namespace misc3d {
namespace common {

/**
 * @brief Estimate normals from an organised point map (pc.points_ laid out row by row, shape = (w, h)).
 * Same signature and behaviour as the reference: pc.normals_ is overwritten; a size mismatch is an
 * error (LogError throws, normal_estimation.cpp:187-191).  Normals of invalid pixels (NaN depth) are
 * NaN (the reference leaves them uninitialised).
 */
inline void EstimateNormalsFromMap(PointCloud& pc, const std::tuple<int, int> shape, int k,
                                   const std::array<double, 3>& view_point = {0, 0, 0}, int device = 0) {
    const size_t num = pc.points_.size();
    const int w = std::get<0>(shape), h = std::get<1>(shape);
    if (w < 0 || h < 0 || k < 0 || num != (size_t)w * (size_t)h) {
        LogError("The point cloud size is not equal to given point map size.");
        return;
    }
    std::vector<Vector3d> normals(num);
    const int rc = m3d_normals_from_map(num ? pc.points_[0].data() : nullptr, (uint32_t)w, (uint32_t)h, (uint32_t)k,
                                        view_point.data(), device, num ? normals[0].data() : nullptr, nullptr);
    if (rc < 0) LogError(m3d_last_error());
    pc.normals_ = std::move(normals);
}

}  // namespace common
}  // namespace misc3d
