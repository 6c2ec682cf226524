// misc3d/registration/transform_estimation.h -- mirror of
// include/misc3d/registration/transform_estimation.h:68-146 over the C ABI.
// TeaserSolver (TEASER++ max-clique + GNC, CPU graph code) is outside the accelerated path.
#pragma once
#include <utility>
#include <vector>

#include "../../misc3d_amd.h"
#include "../geometry.h"
#include "../logging.h"

namespace misc3d {
namespace registration {

class TransformationSolver {
public:
    enum class SolverType { LeastSquare = 0, TEASER = 1, RANSAC = 2 };
    virtual ~TransformationSolver() {}
    SolverType GetSolverType() const { return solver_type_; }

protected:
    explicit TransformationSolver(SolverType type) : solver_type_(type) {}

private:
    SolverType solver_type_;
};

// src/transform_estimation.cpp:49-66 (Eigen::umeyama)
class LeastSquareSolver : public TransformationSolver {
public:
    explicit LeastSquareSolver(bool scaling, int device = 0)
        : TransformationSolver(SolverType::LeastSquare), scaling_(scaling), device_(device) {}
    Matrix4d Solve(const CloudView& src, const CloudView& dst) const {
        if (src.n < 3 || dst.n < 3) LogError("The number of points pair is less than 3.");   // :17-19
        if (src.n != dst.n) LogError("The number of points pair is not equal.");             // :20-22
        Matrix4d T;
        CheckStatus(m3d_kabsch(src.xyz, dst.xyz, src.n, scaling_ ? 1 : 0, device_, T.data()));
        return T;
    }

private:
    bool scaling_;
    int device_;
};

// src/transform_estimation.cpp:124-164.  The reference leaves edge_length_threshold_ uninitialised
// (transform_estimation.h:126 initialises the member with itself); here the argument is honoured.
class RANSACSolver : public TransformationSolver {
public:
    explicit RANSACSolver(double threshold, int max_iter = 100000, double edge_length_threshold = 0.9)
        : TransformationSolver(SolverType::RANSAC),
          threshold_(threshold),
          max_iter_(max_iter),
          edge_length_threshold_(edge_length_threshold) {}
    void SetSeed(uint64_t seed) {
        seed_ = seed;
        has_seed_ = true;
    }
    void SetConfidence(double c) { confidence_ = c; }  // Open3D RANSACConvergenceCriteria default 0.999
    void SetDevice(int device) { device_ = device; }
    // false: the call only returns the pose (what the reference's RANSACSolver::Solve does, transform_estimation.cpp:163);
    // the winner's deterministic inlier_rmse -- one more nearest-neighbour pass over the source -- is then not formed and
    // GetStats() is all zeros
    void SetWantStats(bool want) { want_stats_ = want; }
    Matrix4d Solve(const CloudView& src, const CloudView& dst,
                   const std::pair<std::vector<size_t>, std::vector<size_t>>& corres) const {
        if (corres.first.size() != corres.second.size()) LogError("correspondence lists differ in length");
        Matrix4d T;
        if (!want_stats_) stats_ = m3d_reg_stats{};   // (not an earlier call's values)
        CheckStatus(m3d_registration_ransac(src.xyz, src.n, dst.xyz, dst.n, corres.first.data(),
                                            corres.second.data(), corres.first.size(), threshold_, max_iter_,
                                            edge_length_threshold_, confidence_, has_seed_ ? &seed_ : nullptr,
                                            device_, T.data(), want_stats_ ? &stats_ : nullptr));
        return T;
    }
    const m3d_reg_stats& GetStats() const { return stats_; }

private:
    bool want_stats_ = true;
    double threshold_;
    int max_iter_;
    double edge_length_threshold_;
    double confidence_ = 0.999;
    uint64_t seed_ = 0;
    bool has_seed_ = false;
    int device_ = 0;
    mutable m3d_reg_stats stats_{};
};

}  // namespace registration
}  // namespace misc3d
