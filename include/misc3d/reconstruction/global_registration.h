// misc3d/reconstruction/global_registration.h -- the loop-closure half of
// misc3d::reconstruction::ReconstructionPipeline over the C ABI: GlobalRegistration (src/pipeline.cpp:790-828, the Ransac
// method) and the loop over fragment pairs that calls it, BuildPoseGraphForScene's one std::thread per pair
// (src/pipeline.cpp:428-439).  The rest of the pipeline class (RGBD odometry, TSDF integration, pose-graph optimisation:
// Open3D calls) is outside the accelerated path.
#pragma once
#include <array>
#include <tuple>
#include <vector>

#include "../../misc3d_amd.h"
#include "../geometry.h"
#include "../logging.h"
#include "../registration/correspondence_matching.h"

namespace misc3d {
namespace reconstruction {

using Matrix6d = std::array<double, 36>;  // row-major

struct GlobalRegistrationOption {
    double voxel_size = 0.01;             // PipelineConfig::voxel_size_; max_dis = 1.4 voxel_size (pipeline.cpp:796)
    int max_iter = 100000;                // RANSACSolver's defaults (transform_estimation.h:121-123)
    double edge_length_threshold = 0.9;
    double confidence = 0.999;            // Open3D RANSACConvergenceCriteria's default
};

// std::tuple<bool, Eigen::Matrix4d, Eigen::Matrix6d> ReconstructionPipeline::GlobalRegistration(int s, int t) for one pair of
// preprocessed fragments and their FPFH features.  seed == nullptr: std::random_device (the reference's behaviour).
inline std::tuple<bool, Matrix4d, Matrix6d> GlobalRegistration(const CloudView& pcd_s, const CloudView& pcd_t,
                                                              const registration::FeatureView& fpfh_s,
                                                              const registration::FeatureView& fpfh_t,
                                                              const GlobalRegistrationOption& opt = {},
                                                              const uint64_t* seed = nullptr, int device = 0,
                                                              m3d_global_reg_stats* stats = nullptr) {
    if (fpfh_s.dim != fpfh_t.dim || fpfh_s.n != pcd_s.n || fpfh_t.n != pcd_t.n)
        LogError("one descriptor per point and equal descriptor widths are required");
    Matrix4d pose;
    Matrix6d info;
    const int rc = CheckStatus(m3d_global_registration(pcd_s.xyz, pcd_s.n, pcd_t.xyz, pcd_t.n, fpfh_s.data, fpfh_t.data,
                                                       fpfh_s.dim, opt.voxel_size, opt.max_iter, opt.edge_length_threshold,
                                                       opt.confidence, seed, device, pose.data(), info.data(), stats));
    return std::make_tuple(rc == M3D_OK, pose, info);
}

// MatchingResult of the reference (pipeline.h): the pair's indices, success, pose, information.
struct MatchingResult {
    int s_ = 0, t_ = 0;
    bool success_ = false;
    Matrix4d transformation_{};
    Matrix6d information_{};
    m3d_global_reg_stats stats_{};
};

// The loop-closure part of BuildPoseGraphForScene (pipeline.cpp:415-440): every pair (s, t) of `pairs` registered with
// GlobalRegistration -- the reference spawns one std::thread per pair; here the pairs are dealt to `devices` and every
// device keeps `inflight` of them in flight on lanes of its own, its fragments resident (m3d_register_fragment_pairs).  seeds: one per pair, or
// empty for std::random_device.  A failed pair (success_ false) carries identity pose / information, as
// RegisterFragmentPair leaves it (pipeline.cpp:770-775).
inline std::vector<MatchingResult> RegisterFragmentPairs(const std::vector<CloudView>& fragments,
                                                         const std::vector<registration::FeatureView>& features,
                                                         const std::vector<std::pair<int, int>>& pairs,
                                                         const GlobalRegistrationOption& opt = {},
                                                         const std::vector<uint64_t>& seeds = {},
                                                         const std::vector<int>& devices = {0}, int inflight = 0) {
    if (fragments.size() != features.size()) LogError("one feature set per fragment is required");
    if (!seeds.empty() && seeds.size() != pairs.size()) LogError("one seed per pair (or none) is required");
    std::vector<m3d_fragment_view> fv(fragments.size());
    const int dim = features.empty() ? 0 : features[0].dim;
    for (size_t i = 0; i < fragments.size(); ++i) {
        if (features[i].dim != dim || features[i].n != fragments[i].n)
            LogError("one descriptor per point and equal descriptor widths are required");
        fv[i].xyz = fragments[i].xyz;
        fv[i].feat = features[i].data;
        fv[i].n = fragments[i].n;
    }
    std::vector<m3d_pair_result> fp(pairs.size());
    for (size_t k = 0; k < pairs.size(); ++k) {
        const int s = pairs[k].first, t = pairs[k].second;
        if (s < 0 || t < 0 || (size_t)s >= fragments.size() || (size_t)t >= fragments.size())
            LogError("fragment index out of range");
        fp[k] = m3d_pair_result{};
        fp[k].s = s;
        fp[k].t = t;
        if (!seeds.empty()) {
            fp[k].seed = seeds[k];
            fp[k].has_seed = 1;
        }
    }
    // every fragment is uploaded once per device and stays resident for the call (m3d_register_fragment_pairs)
    CheckStatus(m3d_register_fragment_pairs(fv.data(), fv.size(), dim, fp.data(), fp.size(), opt.voxel_size, opt.max_iter,
                                            opt.edge_length_threshold, opt.confidence, devices.data(), (int)devices.size(),
                                            inflight));
    std::vector<MatchingResult> out(pairs.size());
    static const Matrix4d I4 = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (size_t k = 0; k < pairs.size(); ++k) {
        MatchingResult& r = out[k];
        r.s_ = pairs[k].first;
        r.t_ = pairs[k].second;
        r.success_ = fp[k].rc == M3D_OK;
        r.stats_ = fp[k].stats;
        if (r.success_) {
            for (int i = 0; i < 16; ++i) r.transformation_[i] = fp[k].T[i];
            for (int i = 0; i < 36; ++i) r.information_[i] = fp[k].info[i];
        } else {   // pipeline.cpp:770-775
            r.transformation_ = I4;
            for (int i = 0; i < 36; ++i) r.information_[i] = (i % 7 == 0) ? 1.0 : 0.0;
        }
    }
    return out;
}

}  // namespace reconstruction
}  // namespace misc3d
