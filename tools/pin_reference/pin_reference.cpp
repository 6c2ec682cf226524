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
#include <misc3d/common/ransac.h>
#include <open3d/geometry/PointCloud.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
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

}  // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    int rc = 0;
    rc |= run<misc3d::common::PlaneEstimator, misc3d::common::Plane>(0, dir);
    rc |= run<misc3d::common::SphereEstimator, misc3d::common::Sphere>(1, dir);
    rc |= run<misc3d::common::CylinderEstimator, misc3d::common::Cylinder>(2, dir);
    return rc;
}
