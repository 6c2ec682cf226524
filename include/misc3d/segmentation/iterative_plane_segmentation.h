// misc3d/segmentation/iterative_plane_segmentation.h -- mirror of
// include/misc3d/segmentation/iterative_plane_segmentation.h:25-28 /
// src/iterative_plane_segmentation.cpp:8-39 over m3d_segment_plane_iterative.
#pragma once
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

#include "../../misc3d_amd.h"
#include "../geometry.h"
#include "../logging.h"

namespace misc3d {
namespace segmentation {

struct PlaneCluster {
    Vector4d plane;
    std::vector<size_t> indices;  // into the input cloud, ascending (extension: the reference only
                                  // returns the points)
    PointCloud cloud;             // cluster points (what the reference returns)
};

// devices: more than one ordinal = every round's hypothesis loop sharded over those GPUs from this one process
// (m3d_segment_plane_iterative_multi: a thread and a replica of the cloud per device).
// comm: one process per GPU instead (m3d_segment_plane_iterative_sharded; every rank makes the same call).
inline std::vector<PlaneCluster> SegmentPlaneIterativeIndexed(const CloudView& pcd, double threshold,
                                                               int max_iteration = 100, double min_ratio = 0.05,
                                                               const uint64_t* seed = nullptr, int device = 0,
                                                               const std::vector<int>& devices = {},
                                                               m3d_comm* comm = nullptr) {
    std::vector<PlaneCluster> result;
    if (pcd.n < 3) {  // :13-17
        LogWarning("Point cloud size has less than 3.");
        return result;
    }
    const size_t max_clusters = 4096;
    std::vector<double> planes(4 * max_clusters);
    std::vector<size_t> offsets(max_clusters + 1), indices(pcd.n);
    std::unique_ptr<double[]> gathered;
    size_t k = 0;
    int status;
    if (!devices.empty())
        status = m3d_segment_plane_iterative_multi(pcd.xyz, pcd.n, threshold, max_iteration, min_ratio, seed,
                                                   devices.data(), (int)devices.size(), max_clusters, planes.data(),
                                                   offsets.data(), indices.data(), &k);
    else if (comm)
        status = m3d_segment_plane_iterative_sharded(pcd.xyz, pcd.n, threshold, max_iteration, min_ratio, seed, device,
                                                     comm, max_clusters, planes.data(), offsets.data(),
                                                     indices.data(), &k);
    else {
        // one device: the clusters' points are gathered on the device (m3d_segment_plane_iterative_clouds)
        gathered.reset(new double[3 * pcd.n]);
        status = m3d_segment_plane_iterative_clouds(pcd.xyz, pcd.n, threshold, max_iteration, min_ratio, seed, device,
                                                    max_clusters, planes.data(), offsets.data(), indices.data(),
                                                    gathered.get(), &k);
    }
    const int rc = CheckStatus(status);
    if (rc == 2) LogWarning("segment_plane_iterative: a round found no inlier; stopping early");
    result.resize(k);
    static_assert(sizeof(Vector3d) == 3 * sizeof(double), "Vector3d is three packed doubles");
    for (size_t c = 0; c < k; ++c) {
        for (int j = 0; j < 4; ++j) result[c].plane[j] = planes[4 * c + j];
        const size_t lo = offsets[c], cnt = offsets[c + 1] - offsets[c];
        result[c].indices.assign(indices.begin() + lo, indices.begin() + lo + cnt);
        auto& P = result[c].cloud.points_;
        if (gathered) {
            P.resize(cnt);
            if (cnt) std::memcpy(static_cast<void*>(P.data()), gathered.get() + 3 * lo, sizeof(double) * 3 * cnt);
        } else {
            P.reserve(cnt);
            for (size_t i : result[c].indices) P.push_back({pcd.xyz[3 * i], pcd.xyz[3 * i + 1], pcd.xyz[3 * i + 2]});
        }
    }
    return result;
}

// same signature and return shape as the reference
inline std::vector<std::pair<Vector4d, PointCloud>> SegmentPlaneIterative(const CloudView& pcd, double threshold,
                                                                          int max_iteration = 100,
                                                                          double min_ratio = 0.05) {
    std::vector<std::pair<Vector4d, PointCloud>> out;
    for (auto& c : SegmentPlaneIterativeIndexed(pcd, threshold, max_iteration, min_ratio))
        out.emplace_back(c.plane, std::move(c.cloud));
    return out;
}
// the reference's call on several GPUs of this node (extension: the reference has no device notion)
inline std::vector<std::pair<Vector4d, PointCloud>> SegmentPlaneIterative(const CloudView& pcd, double threshold,
                                                                          int max_iteration, double min_ratio,
                                                                          const std::vector<int>& devices) {
    std::vector<std::pair<Vector4d, PointCloud>> out;
    for (auto& c : SegmentPlaneIterativeIndexed(pcd, threshold, max_iteration, min_ratio, nullptr, 0, devices))
        out.emplace_back(c.plane, std::move(c.cloud));
    return out;
}

}  // namespace segmentation
}  // namespace misc3d
