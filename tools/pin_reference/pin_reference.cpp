// pin_reference.cpp -- dumps what the REAL reference computes for a fixed sample table, so that the oracle (and through
// it the HIP path) can be pinned to it.  NOT built in this repository's image (it needs Eigen + Open3D 0.15.1 and the
// yuecideng/Misc3D headers); tools/pin_reference/run.sh builds it on a machine that has them.
//
// It uses the reference's own estimator classes (include/misc3d/common/ransac.h:134-446) exactly as
// RANSAC::FitModelParallel does (ransac.h:576-590): sample = pc.SelectByIndex(sample_indices),
// estimator.MinimalFit(*sample, model), then the EvaluateModel scan (ransac.h:626-641) over all points with
// CalcPointToModelDistance -- only the sampler is replaced by the table read from the input file, because the
// reference seeds it from std::random_device (utils.h:74-77).  For the hypothesis with the most inliers it also runs
// GeneralFit on the inliers (RefineModel, ransac.h:534-549).
//
// Input  (<dir>/k<kind>.in, little-endian): u64 n, u64 has_normals, u64 H, u64 m, f64 threshold,
//        n x 3 f64 points, [n x 3 f64 normals], H x m u64 sample indices.
// Output (<dir>/k<kind>.ref): u64 H, u64 npar, H x u8 valid, H x npar f64 models (zeros when invalid), H x u64 inlier
//        counts, H x f64 serial error sums, i64 best index, u64 general_fit_ok, npar f64 refined parameters.
//
// Second leg ("driver"): the reference's OWN RANSAC<...>::FitModel -- sampler, sequential best-update, adaptive stop
// (ransac.h:561-613), RefineModel and GeneralFit -- on a fixed mt19937 seed (pin_seed.h, force-included), one thread:
//   <dir>/d<kind>_<case>.in : u64 n, u64 has_normals, u64 max_iteration, u64 seed, f64 threshold, f64 probability,
//                             points, [normals]
//   <dir>/d<kind>_<case>.ref: i64 ret, u64 npar, npar f64 parameters, i64 "run {} iterations" count parsed from the
//                             reference's own log line (ransac.h:616-619; -1 if not seen), u64 n_inliers, indices
// Third leg: misc3d::segmentation::SegmentPlaneIterative itself (src/iterative_plane_segmentation.cpp, compiled with
// pin_seed.h force-included) -- <dir>/seg_<case>.in: u64 n, u64 max_iteration, u64 seed, f64 threshold, f64 min_ratio,
// points; .ref: u64 k, k x (4 f64 plane, u64 size), then the clusters' points in order.
// Optional fourth leg (-DPIN_WITH_REGISTRATION): Open3D's RegistrationRANSACBasedOnCorrespondence as RANSACSolver::Solve
// calls it (src/transform_estimation.cpp:142-161) with an explicit seed -- <dir>/reg.in / reg.ref (see run_registration).
#include "pin_seed.h"

#include <misc3d/common/ransac.h>
#include <misc3d/logging.h>
#include <misc3d/segmentation/iterative_plane_segmentation.h>
#include <open3d/geometry/PointCloud.h>
#ifdef PIN_WITH_REGISTRATION
#include <open3d/pipelines/registration/Registration.h>
#include <open3d/pipelines/registration/CorrespondenceChecker.h>
#endif

extern "C" {
uint64_t m3d_pin_seed = 0;
uint64_t m3d_pin_calls = 0;
}

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace {

template <class T>
bool read_vec(FILE* f, std::vector<T>& v, size_t n) {
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

template <class Estimator, class ModelT>
int run(int kind, const std::string& dir) {
    const std::string in = dir + "/k" + std::to_string(kind) + ".in", out = dir + "/k" + std::to_string(kind) + ".ref";
    FILE* f = std::fopen(in.c_str(), "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", in.c_str());
        return 1;
    }
    uint64_t hdr[4];
    double thr;
    if (std::fread(hdr, 8, 4, f) != 4 || std::fread(&thr, 8, 1, f) != 1) return 1;
    const size_t n = hdr[0], has_normals = hdr[1], H = hdr[2], m = hdr[3];
    std::vector<double> pts, nrm;
    std::vector<uint64_t> samples;
    if (!read_vec(f, pts, 3 * n) || (has_normals && !read_vec(f, nrm, 3 * n)) || !read_vec(f, samples, H * m)) return 1;
    std::fclose(f);

    open3d::geometry::PointCloud pc;
    pc.points_.resize(n);
    for (size_t i = 0; i < n; ++i) pc.points_[i] = Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (has_normals) {
        pc.normals_.resize(n);
        for (size_t i = 0; i < n; ++i) pc.normals_[i] = Eigen::Vector3d(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    }

    Estimator estimator;
    const size_t npar = kind == 2 ? 7 : 4;
    std::vector<uint8_t> valid(H, 0);
    std::vector<double> models(H * npar, 0.0), errors(H, 0.0);
    std::vector<uint64_t> counts(H, 0);
    int64_t best = -1;
    uint64_t best_count = 0;
    ModelT best_model;
    for (size_t h = 0; h < H; ++h) {
        std::vector<size_t> idx(samples.begin() + h * m, samples.begin() + (h + 1) * m);
        const auto sample = pc.SelectByIndex(idx);          // ransac.h:578
        ModelT model;
        const bool ok = estimator.MinimalFit(*sample, model);   // ransac.h:582
        valid[h] = ok ? 1 : 0;
        if (!ok) continue;                                   // ransac.h:584-586
        for (size_t k = 0; k < npar; ++k) models[h * npar + k] = model.parameters_(k);
        uint64_t cnt = 0;                                    // EvaluateModel, ransac.h:626-641
        double err = 0;
        for (size_t i = 0; i < n; ++i) {
            const double d = estimator.CalcPointToModelDistance(pc.points_[i], model);
            if (d < thr) {
                err += d;
                cnt++;
            }
        }
        counts[h] = cnt;
        errors[h] = err;
        if (best < 0 || cnt > best_count) {
            best = (int64_t)h;
            best_count = cnt;
            best_model = model;
        }
    }
    // RefineModel (ransac.h:534-549) of the hypothesis with the most inliers (first of equals)
    uint64_t gf_ok = 0;
    std::vector<double> refined(npar, 0.0);
    if (best >= 0) {
        std::vector<size_t> inl;
        for (size_t i = 0; i < n; ++i)
            if (estimator.CalcPointToModelDistance(pc.points_[i], best_model) < thr) inl.push_back(i);
        const auto inliers_pc = pc.SelectByIndex(inl);
        ModelT m2 = best_model;
        gf_ok = estimator.GeneralFit(*inliers_pc, m2) ? 1 : 0;
        for (size_t k = 0; k < npar; ++k) refined[k] = m2.parameters_(k);
    }
    FILE* g = std::fopen(out.c_str(), "wb");
    if (!g) return 1;
    const uint64_t oh[2] = {H, npar};
    std::fwrite(oh, 8, 2, g);
    std::fwrite(valid.data(), 1, H, g);
    std::fwrite(models.data(), 8, H * npar, g);
    std::fwrite(counts.data(), 8, H, g);
    std::fwrite(errors.data(), 8, H, g);
    std::fwrite(&best, 8, 1, g);
    std::fwrite(&gf_ok, 8, 1, g);
    std::fwrite(refined.data(), 8, npar, g);
    std::fclose(g);
    std::printf("kind %d: %zu hypotheses, %zu points, best %lld with %llu inliers -> %s\n", kind, H, n, (long long)best,
                (unsigned long long)best_count, out.c_str());
    return 0;
}

// ---- the reference's own driver ------------------------------------------------------------------------------------
long long g_logged_count = -1;   // "... and run {} iterations" of the last FitModel (ransac.h:616-619)

void capture_log(const std::string& line) {
    const char* key = "and run ";
    const size_t p = line.find(key);
    if (p != std::string::npos) g_logged_count = std::atoll(line.c_str() + p + std::strlen(key));
}

bool fill_cloud(FILE* f, size_t n, bool has_normals, open3d::geometry::PointCloud& pc) {
    std::vector<double> pts, nrm;
    if (!read_vec(f, pts, 3 * n) || (has_normals && !read_vec(f, nrm, 3 * n))) return false;
    pc.points_.resize(n);
    for (size_t i = 0; i < n; ++i) pc.points_[i] = Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (has_normals) {
        pc.normals_.resize(n);
        for (size_t i = 0; i < n; ++i) pc.normals_[i] = Eigen::Vector3d(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    }
    return true;
}

template <class Ransac, class ModelT>
int run_driver(int kind, const std::string& dir, const std::string& tag) {
    const std::string base = dir + "/d" + std::to_string(kind) + "_" + tag;
    FILE* f = std::fopen((base + ".in").c_str(), "rb");
    if (!f) return 0;   // case not exported
    uint64_t hdr[4];
    double par[2];
    if (std::fread(hdr, 8, 4, f) != 4 || std::fread(par, 8, 2, f) != 2) return 1;
    open3d::geometry::PointCloud pc;
    if (!fill_cloud(f, hdr[0], hdr[1] != 0, pc)) return 1;
    std::fclose(f);
    m3d_pin_seed = hdr[3];
    m3d_pin_calls = 0;
    g_logged_count = -1;
    Ransac fit;                                    // python/py_common.cpp:11-27 (FitPlane and its siblings)
    fit.SetMaxIteration(hdr[2]);
    fit.SetProbability(par[1]);
    fit.SetPointCloud(pc);
    ModelT model;
    std::vector<size_t> inliers;
    const bool ret = fit.FitModel(par[0], model, inliers);
    const uint64_t npar = kind == 2 ? 7 : 4, ni = inliers.size();
    FILE* g = std::fopen((base + ".ref").c_str(), "wb");
    if (!g) return 1;
    const int64_t r64 = ret ? 1 : 0, cnt = g_logged_count;
    std::fwrite(&r64, 8, 1, g);
    std::fwrite(&npar, 8, 1, g);
    for (uint64_t k = 0; k < npar; ++k) {
        const double v = model.parameters_(k);
        std::fwrite(&v, 8, 1, g);
    }
    std::fwrite(&cnt, 8, 1, g);
    std::fwrite(&ni, 8, 1, g);
    for (size_t v : inliers) {
        const uint64_t u = v;
        std::fwrite(&u, 8, 1, g);
    }
    std::fclose(g);
    std::printf("driver kind %d %s: ret %d, %llu inliers, %lld iterations logged\n", kind, tag.c_str(), (int)ret,
                (unsigned long long)ni, (long long)cnt);
    return 0;
}

int run_segmentation(const std::string& dir, const std::string& tag) {
    const std::string base = dir + "/seg_" + tag;
    FILE* f = std::fopen((base + ".in").c_str(), "rb");
    if (!f) return 0;
    uint64_t hdr[3];
    double par[2];
    if (std::fread(hdr, 8, 3, f) != 3 || std::fread(par, 8, 2, f) != 2) return 1;
    open3d::geometry::PointCloud pc;
    if (!fill_cloud(f, hdr[0], false, pc)) return 1;
    std::fclose(f);
    m3d_pin_seed = hdr[2];
    m3d_pin_calls = 0;
    const auto res = misc3d::segmentation::SegmentPlaneIterative(pc, par[0], (int)hdr[1], par[1]);
    FILE* g = std::fopen((base + ".ref").c_str(), "wb");
    if (!g) return 1;
    const uint64_t k = res.size();
    std::fwrite(&k, 8, 1, g);
    for (const auto& pr : res) {
        for (int c = 0; c < 4; ++c) {
            const double v = pr.first(c);
            std::fwrite(&v, 8, 1, g);
        }
        const uint64_t sz = pr.second.points_.size();
        std::fwrite(&sz, 8, 1, g);
    }
    for (const auto& pr : res)
        for (const auto& p : pr.second.points_) std::fwrite(p.data(), 8, 3, g);
    std::fclose(g);
    std::printf("segmentation %s: %llu clusters\n", tag.c_str(), (unsigned long long)k);
    return 0;
}

#ifdef PIN_WITH_REGISTRATION
// reg.in: u64 ns, u64 nd, u64 m, u64 max_iter, u64 seed, f64 threshold, f64 edge, f64 confidence, src, dst points,
// m x 2 u64 correspondences.  reg.ref: 16 f64 transformation (row-major), f64 fitness, f64 inlier_rmse, u64 correspondences.
// Open3D 0.15.x takes the seed as the call's last argument; later versions use utility::random::Seed (-DPIN_O3D_GLOBAL_SEED).
int run_registration(const std::string& dir) {
    namespace reg = open3d::pipelines::registration;
    FILE* f = std::fopen((dir + "/reg.in").c_str(), "rb");
    if (!f) return 0;
    uint64_t hdr[5];
    double par[3];
    if (std::fread(hdr, 8, 5, f) != 5 || std::fread(par, 8, 3, f) != 3) return 1;
    open3d::geometry::PointCloud src, dst;
    if (!fill_cloud(f, hdr[0], false, src) || !fill_cloud(f, hdr[1], false, dst)) return 1;
    std::vector<uint64_t> cs;
    if (!read_vec(f, cs, 2 * hdr[2])) return 1;
    std::fclose(f);
    reg::CorrespondenceSet corres(hdr[2]);
    for (size_t i = 0; i < hdr[2]; ++i) corres[i] = Eigen::Vector2i((int)cs[2 * i], (int)cs[2 * i + 1]);
    std::vector<std::reference_wrapper<const reg::CorrespondenceChecker>> checkers;   // transform_estimation.cpp:142-152
    auto edge = reg::CorrespondenceCheckerBasedOnEdgeLength(par[1]);
    auto dist = reg::CorrespondenceCheckerBasedOnDistance(par[0]);
    checkers.push_back(edge);
    checkers.push_back(dist);
#ifdef PIN_O3D_GLOBAL_SEED
    open3d::utility::random::Seed((int)hdr[4]);
    const auto res = reg::RegistrationRANSACBasedOnCorrespondence(src, dst, corres, par[0], reg::TransformationEstimationPointToPoint(false), 3,
                                                                  checkers, reg::RANSACConvergenceCriteria((int)hdr[3], par[2]));
#else
    const auto res = reg::RegistrationRANSACBasedOnCorrespondence(src, dst, corres, par[0], reg::TransformationEstimationPointToPoint(false), 3,
                                                                  checkers, reg::RANSACConvergenceCriteria((int)hdr[3], par[2]),
                                                                  (unsigned int)hdr[4]);
#endif
    FILE* g = std::fopen((dir + "/reg.ref").c_str(), "wb");
    if (!g) return 1;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            const double v = res.transformation_(r, c);
            std::fwrite(&v, 8, 1, g);
        }
    const uint64_t nc = res.correspondence_set_.size();
    std::fwrite(&res.fitness_, 8, 1, g);
    std::fwrite(&res.inlier_rmse_, 8, 1, g);
    std::fwrite(&nc, 8, 1, g);
    std::fclose(g);
    std::printf("registration: fitness %.6f rmse %.6g, %llu correspondences\n", res.fitness_, res.inlier_rmse_, (unsigned long long)nc);
    return 0;
}
#endif

}  // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    int rc = 0;
    rc |= run<misc3d::common::PlaneEstimator, misc3d::common::Plane>(0, dir);
    rc |= run<misc3d::common::SphereEstimator, misc3d::common::Sphere>(1, dir);
    rc |= run<misc3d::common::CylinderEstimator, misc3d::common::Cylinder>(2, dir);
    // the reference's own driver: one thread (the loop is then sequential), Info-level log captured for the count
    misc3d::Logger::GetInstance().SetPrintFunction(capture_log);
    for (const char* tag : {"adaptive", "exhaustive"}) {
        rc |= run_driver<misc3d::common::RANSACPlane, misc3d::common::Plane>(0, dir, tag);
        rc |= run_driver<misc3d::common::RANSACShpere, misc3d::common::Sphere>(1, dir, tag);
        rc |= run_driver<misc3d::common::RANSACCylinder, misc3d::common::Cylinder>(2, dir, tag);
    }
    for (const char* tag : {"example", "room"}) rc |= run_segmentation(dir, tag);
#ifdef PIN_WITH_REGISTRATION
    rc |= run_registration(dir);
#endif
    return rc;
}
