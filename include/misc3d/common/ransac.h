// misc3d/common/ransac.h -- host-side mirror of the reference's RANSAC classes
// (include/misc3d/common/ransac.h:24-80 models, :455-669 RANSAC + aliases) over the C ABI.
// Same class and method names, same argument meaning, same error behaviour; the sample / fit /
// score loop runs on the MI355X behind m3d_cloud_fit.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../misc3d_amd.h"
#include "../geometry.h"
#include "../logging.h"

namespace misc3d {
namespace common {

// ransac.h:24-80: parameters_ is the model vector ([a,b,c,d] / [x,y,z,r] / [x,y,z,nx,ny,nz,r])
class Model {
public:
    std::vector<double> parameters_;
    Model() = default;
    explicit Model(size_t n) : parameters_(n, 0.0) {}
};
class Plane : public Model {
public:
    Plane() : Model(4) {}
};
class Sphere : public Model {
public:
    Sphere() : Model(4) {}
};
class Cylinder : public Model {
public:
    Cylinder() : Model(7) {}
};

template <int KIND, class ModelT>
class RANSAC {
public:
    RANSAC() = default;
    RANSAC(const RANSAC&) = delete;
    RANSAC& operator=(const RANSAC&) = delete;
    ~RANSAC() { Release(); }

    // ransac.h:469-475: the reference deep-copies the cloud; here the copy lives in HBM (SoA)
    void SetPointCloud(const CloudView& pc) {
        Release();
        has_normals_ = pc.normals != nullptr;
        size_ = pc.n;
        if (size_ < kMinimalSample) return;  // FitModel raises, like ransac.h:509-513
        cloud_ = m3d_cloud_create(pc.xyz, pc.normals, pc.n, device_);
        if (!cloud_) LogError(m3d_last_error());
    }
    // ransac.h:482-487
    void SetProbability(double probability) {
        if (probability <= 0 || probability > 1) LogError("Probability must be > 0 or <= 1.0");
        probability_ = probability;
    }
    // ransac.h:495
    void SetMaxIteration(size_t num) { max_iteration_ = num; }
    // extensions (the reference seeds from std::random_device and has no device notion)
    void SetSeed(uint64_t seed) {
        seed_ = seed;
        has_seed_ = true;
    }
    void ClearSeed() { has_seed_ = false; }
    void SetDevice(int device) { device_ = device; }
    // One process per GPU: every rank holds its own RANSAC object with the same cloud and settings
    // and a communicator of the group (m3d_comm_create_rccl / _host); FitModel then shards the hypothesis loop
    // and returns the same model and inliers on every rank.  nullptr = single GPU.
    void SetComm(m3d_comm* comm) { comm_ = comm; }

    // ransac.h:506-516 -> FitModelParallel + RefineModel
    bool FitModel(double threshold, ModelT& model, std::vector<size_t>& inlier_indices) {
        if (size_ < kMinimalSample || !cloud_) LogError("Can not fit model due to lack of points");
        if (KIND == M3D_CYLINDER && !has_normals_) LogError("Cylinder estimation requires normals.");  // ransac.h:356-359
        inlier_indices.resize(size_);
        size_t ni = 0;
        model.parameters_.assign(KIND == M3D_CYLINDER ? 7 : 4, 0.0);
        const int rc = CheckStatus(m3d_cloud_fit_sharded(cloud_, comm_, KIND, threshold, max_iteration_, probability_,
                                                         has_seed_ ? &seed_ : nullptr, model.parameters_.data(),
                                                         inlier_indices.data(), &ni, &stats_));
        inlier_indices.resize(ni);
        char buf[160];  // ransac.h:616-619
        std::snprintf(buf, sizeof(buf), "Find best model with %g%% inliers and run %llu iterations",
                      stats_.fitness * 100, (unsigned long long)stats_.count);
        LogInfo(buf);
        return rc == M3D_OK;
    }
    const m3d_stats& GetStats() const { return stats_; }

private:
    static constexpr size_t kMinimalSample = KIND == M3D_PLANE ? 3 : (KIND == M3D_SPHERE ? 4 : 2);
    void Release() {
        if (cloud_) m3d_cloud_destroy(cloud_);
        cloud_ = nullptr;
    }
    m3d_cloud* cloud_ = nullptr;
    size_t size_ = 0;
    bool has_normals_ = false;
    double probability_ = 0.9999;   // ransac.h:462
    size_t max_iteration_ = 1000;   // ransac.h:461
    uint64_t seed_ = 0;
    bool has_seed_ = false;
    int device_ = 0;
    m3d_comm* comm_ = nullptr;
    m3d_stats stats_{};
};

using RANSACPlane = RANSAC<M3D_PLANE, Plane>;
using RANSACShpere = RANSAC<M3D_SPHERE, Sphere>;  // (sic) the reference's spelling, ransac.h:667
using RANSACSphere = RANSACShpere;
using RANSACCylinder = RANSAC<M3D_CYLINDER, Cylinder>;

}  // namespace common
}  // namespace misc3d
