// misc3d/geometry.h -- the point-cloud types the host API accepts.
//
// The reference passes open3d::geometry::PointCloud (points_/normals_ = std::vector<Eigen::Vector3d>,
// contiguous N x 3 float64).  Open3D and Eigen are not dependencies of this implementation: the API
// takes a non-owning CloudView, constructible from misc3d::PointCloud below (same member names as
// Open3D's class) or, with MISC3D_WITH_OPEN3D defined, directly from an Open3D cloud without a copy.
#pragma once
#include "../misc3d_amd.h"

#include <array>
#include <cstddef>
#include <vector>

#ifdef MISC3D_WITH_OPEN3D
#include <open3d/geometry/PointCloud.h>
#endif

namespace misc3d {

using Vector3d = std::array<double, 3>;
using Vector4d = std::array<double, 4>;
using Matrix4d = std::array<double, 16>;  // row-major

struct PointCloud {
    std::vector<Vector3d> points_;
    std::vector<Vector3d> normals_;
    PointCloud() = default;
    explicit PointCloud(std::vector<Vector3d> points) : points_(std::move(points)) {}
    bool HasPoints() const { return !points_.empty(); }
    bool HasNormals() const { return !points_.empty() && normals_.size() == points_.size(); }
    // open3d::geometry::PointCloud::SelectByIndex: order preserved, normals follow
    PointCloud SelectByIndex(const std::vector<size_t>& indices, bool invert = false) const {
        PointCloud out;
        const bool nrm = HasNormals();
        if (!invert) {
            for (size_t i : indices) {
                out.points_.push_back(points_[i]);
                if (nrm) out.normals_.push_back(normals_[i]);
            }
        } else {
            std::vector<char> mask(points_.size(), 0);
            for (size_t i : indices) mask[i] = 1;
            for (size_t i = 0; i < points_.size(); ++i)
                if (!mask[i]) {
                    out.points_.push_back(points_[i]);
                    if (nrm) out.normals_.push_back(normals_[i]);
                }
        }
        return out;
    }
};

struct CloudView {
    const double* xyz = nullptr;      // n x 3
    const double* normals = nullptr;  // n x 3 or null
    size_t n = 0;
    CloudView() = default;
    CloudView(const double* p, const double* nrm, size_t count) : xyz(p), normals(nrm), n(count) {}
    CloudView(const PointCloud& pc)  // NOLINT: implicit by design
        : xyz(pc.points_.empty() ? nullptr : pc.points_[0].data()),
          normals(pc.HasNormals() ? pc.normals_[0].data() : nullptr),
          n(pc.points_.size()) {}
#ifdef MISC3D_WITH_OPEN3D
    CloudView(const open3d::geometry::PointCloud& pc)  // NOLINT: Eigen::Vector3d is 24-byte POD
        : xyz(pc.points_.empty() ? nullptr : pc.points_[0].data()),
          normals(pc.HasNormals() ? pc.normals_[0].data() : nullptr),
          n(pc.points_.size()) {}
#endif
};

// ---- page-locked scratch of the host mirrors ---------------------------------------------------------------------------------
// Large results (index lists, gathered cluster points) land in page-locked blocks the CALLING THREAD keeps between calls: the
// kernels store into them directly and the next call of the same size pins nothing.  Retention is bounded: a block
// is given back when a request needs less than a quarter of it, a thread keeps at most kHostScratchBudget bytes, and
// ReleaseHostScratch() frees the calling thread's blocks at once (a long-lived host calls it after its last big cloud).
namespace detail {
constexpr size_t kHostScratchBudget = (size_t)1 << 30;   // 1 GiB of page-locked memory per thread, at most
struct PinnedScratch {
    void* p = nullptr;
    size_t cap = 0;
    void release() {
        if (p) m3d_host_free(p);
        p = nullptr;
        cap = 0;
    }
    // nullptr: over the budget or the allocation failed -- the caller falls back to pageable memory
    void* get(size_t bytes) {
        if (bytes > kHostScratchBudget) {
            release();
            return nullptr;
        }
        if (bytes > cap || (cap > ((size_t)64 << 20) && bytes < cap / 4)) {
            release();
            p = m3d_host_alloc(bytes);
            cap = p ? bytes : 0;
        }
        return p;
    }
    // (no destructor: a thread-local of the main thread dies at process exit, possibly after the HIP runtime)
};
inline PinnedScratch& host_scratch(int which) {   // 0: index lists, 1: gathered points
    static thread_local PinnedScratch s[2];
    return s[which & 1];
}
}  // namespace detail
inline void ReleaseHostScratch() {
    detail::host_scratch(0).release();
    detail::host_scratch(1).release();
}

}  // namespace misc3d
