// misc3d/registration/correspondence_matching.h -- mirror of
// include/misc3d/registration/correspondence_matching.h:20-91 over m3d_match_mutual_nn.
// Descriptors: dim x N column-major (Eigen::MatrixXd / open3d Feature::data_), i.e. N contiguous
// descriptors of `dim` doubles.
#pragma once
#include <utility>
#include <vector>

#include "../../misc3d_amd.h"
#include "../logging.h"

namespace misc3d {
namespace registration {

enum class MatchMethod { FLANN = 0, ANNOY = 1 };

struct FeatureView {
    const double* data = nullptr;  // dim x n column-major
    int dim = 0;
    size_t n = 0;
};

class ANNMatcher {
public:
    ANNMatcher() : match_method_(MatchMethod::FLANN), n_tress_(0) {}
    explicit ANNMatcher(const MatchMethod& method) : match_method_(method), n_tress_(4) {}
    ANNMatcher(const MatchMethod& method, int n_trees) : match_method_(method), n_tress_(n_trees) {}
    MatchMethod GetMatcherType() const { return match_method_; }
    void SetDevice(int device) { device_ = device; }

    // src/correspondence_matching.cpp:52-84.  Both methods run the exact mutual nearest-neighbour
    // search on the GPU (FLANN is exact; ANNOY approximates the same thing).
    std::pair<std::vector<size_t>, std::vector<size_t>> Match(const FeatureView& src, const FeatureView& dst) const {
        if (src.dim != dst.dim) LogError("descriptor dimensions differ");
        std::pair<std::vector<size_t>, std::vector<size_t>> res;
        res.first.resize(src.n);
        res.second.resize(src.n);
        size_t k = 0;
        CheckStatus(m3d_match_mutual_nn(src.data, src.n, dst.data, dst.n, src.dim, (int)match_method_, n_tress_,
                                        device_, res.first.data(), res.second.data(), &k));
        res.first.resize(k);
        res.second.resize(k);
        return res;
    }

private:
    MatchMethod match_method_;
    int n_tress_;
    int device_ = 0;
};

}  // namespace registration
}  // namespace misc3d
