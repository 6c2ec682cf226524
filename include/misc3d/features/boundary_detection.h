// misc3d/features/boundary_detection.h -- host mirror of the reference's
// include/misc3d/features/boundary_detection.h (DetectBoundaryPoints, src/boundary_detection.cpp:68-113) over the
// C ABI (m3d_detect_boundary_points).  Header-only; no Eigen / Open3D needed.
#pragma once
#include <vector>

#include <misc3d/geometry.h>
#include <misc3d/logging.h>
#include <misc3d_amd.h>

namespace misc3d {
namespace features {

// Stand-ins for open3d::geometry::KDTreeSearchParamKNN / Radius / Hybrid (same member names).
struct KDTreeSearchParamKNN {
    int knn_;
    explicit KDTreeSearchParamKNN(int knn = 30) : knn_(knn) {}
};
struct KDTreeSearchParamRadius {
    double radius_;
    explicit KDTreeSearchParamRadius(double radius) : radius_(radius) {}
};
struct KDTreeSearchParamHybrid {
    double radius_;
    int max_nn_;
    KDTreeSearchParamHybrid(double radius, int max_nn) : radius_(radius), max_nn_(max_nn) {}
};

namespace detail {
inline std::vector<size_t> Detect(const CloudView& pc, int search, double radius, int max_nn, double angle_threshold,
                                  int device) {
    std::vector<size_t> idx;
    if (pc.n == 0) {
        LogError("No PointCloud data.");  // :72-76
        return idx;
    }
    idx.resize(pc.n);
    size_t k = 0;
    const int rc = m3d_detect_boundary_points(pc.xyz, pc.normals, pc.n, search, radius, max_nn, angle_threshold, device,
                                              idx.data(), &k);
    if (rc < 0) LogError(m3d_last_error());
    idx.resize(k);
    return idx;
}
}  // namespace detail

/**
 * @brief Detect boundary points of a point cloud (indices, ascending).  If the cloud has no normals they are
 * estimated from the same neighbourhood.  angle_threshold in degrees.
 */
inline std::vector<size_t> DetectBoundaryPoints(const CloudView& pc, const KDTreeSearchParamHybrid& param,
                                                double angle_threshold = 90.0, int device = 0) {
    return detail::Detect(pc, 2, param.radius_, param.max_nn_, angle_threshold, device);
}
inline std::vector<size_t> DetectBoundaryPoints(const CloudView& pc, const KDTreeSearchParamKNN& param,
                                                double angle_threshold = 90.0, int device = 0) {
    return detail::Detect(pc, 0, 0.0, param.knn_, angle_threshold, device);
}
inline std::vector<size_t> DetectBoundaryPoints(const CloudView& pc, const KDTreeSearchParamRadius& param,
                                                double angle_threshold = 90.0, int device = 0) {
    return detail::Detect(pc, 1, param.radius_, 0, angle_threshold, device);
}

}  // namespace features
}  // namespace misc3d
