// C++ user of the host mirror (include/misc3d/**): the code shape of the reference's
// examples/cpp/ransac_and_boundary.cpp:36-41, segment_plane_iterative.cpp:15-19 and
// transform_estimation.cpp:69-87, with analytic expectations.  Exit code 0 = all checks passed.
#include <cmath>
#include <cstring>
#include <cstdio>
#include <random>
#include <thread>

#include <algorithm>

#include <misc3d/common/normal_estimation.h>
#include <misc3d/features/boundary_detection.h>
#include <misc3d/common/ransac.h>
#include <misc3d/reconstruction/global_registration.h>
#include <misc3d/registration/correspondence_matching.h>
#include <misc3d/registration/transform_estimation.h>
#include <misc3d/segmentation/iterative_plane_segmentation.h>

#define CHECK(cond)                                                   \
    do {                                                              \
        if (!(cond)) {                                                \
            std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main() {
    misc3d::SetVerbosityLevel(misc3d::VerbosityLevel::Warning);
    std::mt19937 gen(1);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::normal_distribution<double> G(0.0, 0.002);

    // ---- RANSACPlane on 60 % plane z = 1 + outliers
    misc3d::PointCloud pc;
    for (int i = 0; i < 20000; ++i) {
        if (i % 5 < 3)
            pc.points_.push_back({U(gen), U(gen), 1.0 + G(gen)});
        else
            pc.points_.push_back({U(gen), U(gen), U(gen)});
    }
    misc3d::common::RANSACPlane fit;
    fit.SetMaxIteration(1000);
    fit.SetProbability(0.9999);
    fit.SetSeed(7);
    fit.SetPointCloud(pc);
    misc3d::common::Plane plane;
    std::vector<size_t> inliers;
    const bool ret = fit.FitModel(0.01, plane, inliers);
    CHECK(ret);
    CHECK(plane.parameters_.size() == 4);
    CHECK(std::fabs(std::fabs(plane.parameters_[2]) - 1.0) < 1e-3);
    CHECK(std::fabs(std::fabs(plane.parameters_[3]) - 1.0) < 1e-3);
    CHECK(inliers.size() > 11500 && inliers.size() < 12800);
    for (size_t k = 1; k < inliers.size(); ++k) CHECK(inliers[k] > inliers[k - 1]);
    // determinism for a fixed seed
    misc3d::common::Plane plane2;
    std::vector<size_t> inliers2;
    CHECK(fit.FitModel(0.01, plane2, inliers2));
    CHECK(inliers2 == inliers && plane2.parameters_ == plane.parameters_);

    // ---- error convention: exceptions with the reference's messages
    bool threw = false;
    try {
        fit.SetProbability(1.5);
    } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("Probability must be") != std::string::npos;
    }
    CHECK(threw);
    threw = false;
    try {
        misc3d::common::RANSACCylinder cyl;
        cyl.SetPointCloud(pc);  // no normals
        misc3d::common::Cylinder c;
        std::vector<size_t> idx;
        cyl.FitModel(0.01, c, idx);
    } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("requires normals") != std::string::npos;
    }
    CHECK(threw);

    // ---- sphere
    misc3d::PointCloud sp;
    for (int i = 0; i < 8000; ++i) {
        double x = G(gen) * 500, y = G(gen) * 500, z = G(gen) * 500;
        const double n = std::sqrt(x * x + y * y + z * z);
        if (i % 2 == 0)
            sp.points_.push_back({0.3 + 0.5 * x / n, -0.2 + 0.5 * y / n, 1.0 + 0.5 * z / n});
        else
            sp.points_.push_back({U(gen) * 2, U(gen) * 2, U(gen) * 2});
    }
    misc3d::common::RANSACShpere sfit;
    sfit.SetMaxIteration(500);
    sfit.SetSeed(3);
    sfit.SetPointCloud(sp);
    misc3d::common::Sphere sphere;
    CHECK(sfit.FitModel(0.01, sphere, inliers));
    CHECK(std::fabs(sphere.parameters_[0] - 0.3) < 1e-3 && std::fabs(sphere.parameters_[3] - 0.5) < 1e-3);

    // ---- SegmentPlaneIterative: two planes
    misc3d::PointCloud two;
    for (int i = 0; i < 6000; ++i) two.points_.push_back({U(gen), U(gen), G(gen)});
    for (int i = 0; i < 4000; ++i) two.points_.push_back({2.0 + G(gen), U(gen), U(gen)});
    for (int i = 0; i < 500; ++i) two.points_.push_back({3 * U(gen), 3 * U(gen), 3 * U(gen)});
    auto seg = misc3d::segmentation::SegmentPlaneIterative(two, 0.01, 100, 0.1);
    CHECK(seg.size() >= 2);
    CHECK(std::fabs(std::fabs(seg[0].first[2]) - 1.0) < 1e-2);
    CHECK(std::fabs(std::fabs(seg[1].first[0]) - 1.0) < 1e-2);
    CHECK(seg[0].second.points_.size() > 5500 && seg[1].second.points_.size() > 3500);
    // the devices[] form (one process, a replica per GPU; here the one GPU of the test box) and the communicator
    // form (a world-1 host communicator: the exchange is the caller's function) give the one-call result
    {
        const uint64_t sd = 5;
        auto one = misc3d::segmentation::SegmentPlaneIterativeIndexed(two, 0.01, 100, 0.1, &sd, 0);
        auto multi = misc3d::segmentation::SegmentPlaneIterativeIndexed(two, 0.01, 100, 0.1, &sd, 0, {0});
        struct Echo {
            static int gather(void* user, const void* send, void* recv, size_t bytes) {
                ++*static_cast<int*>(user);
                std::memcpy(recv, send, bytes);
                return 0;
            }
        };
        int exchanges = 0;
        m3d_comm* comm = m3d_comm_create_host(1, 0, &Echo::gather, &exchanges);
        CHECK(comm != nullptr && m3d_comm_world(comm) == 1 && m3d_comm_rank(comm) == 0);
        auto sharded = misc3d::segmentation::SegmentPlaneIterativeIndexed(two, 0.01, 100, 0.1, &sd, 0, {}, comm);
        CHECK(exchanges > 0 && (uint64_t)exchanges == m3d_comm_collectives(comm));
        CHECK(one.size() == multi.size() && one.size() == sharded.size() && one.size() >= 2);
        for (size_t k = 0; k < one.size(); ++k) {
            CHECK(one[k].indices == multi[k].indices && one[k].indices == sharded[k].indices);
            // the cluster cloud is SelectByIndex of the list (iterative_plane_segmentation.cpp:32): gathered on the device in
            // the one-GPU form (m3d_segment_plane_iterative_clouds), on the host in the other two -- bit for bit the same
            CHECK(one[k].cloud.points_.size() == one[k].indices.size());
            CHECK(one[k].cloud.points_ == multi[k].cloud.points_ && one[k].cloud.points_ == sharded[k].cloud.points_);
            for (size_t i = 0; i < one[k].indices.size(); i += 97) CHECK(one[k].cloud.points_[i] == two.points_[one[k].indices[i]]);
            for (int j = 0; j < 4; ++j) CHECK(one[k].plane[j] == multi[k].plane[j] && one[k].plane[j] == sharded[k].plane[j]);
        }
        misc3d::common::RANSACPlane f1, f2;
        misc3d::common::Plane p1, p2;
        std::vector<size_t> i1, i2;
        f1.SetSeed(9);
        f2.SetSeed(9);
        f2.SetComm(comm);
        f1.SetPointCloud(two);
        f2.SetPointCloud(two);
        CHECK(f1.FitModel(0.01, p1, i1) == f2.FitModel(0.01, p2, i2));
        CHECK(i1 == i2 && p1.parameters_ == p2.parameters_);
        m3d_comm_destroy(comm);
        bool threw = false;
        try {
            misc3d::segmentation::SegmentPlaneIterativeIndexed(two, 0.01, 100, 0.1, &sd, 0, {0, 0});
        } catch (const std::runtime_error&) {
            threw = true;   // devices must be distinct
        }
        CHECK(threw);
    }

    // ---- LeastSquareSolver + RANSACSolver + ANNMatcher
    const double c = std::cos(0.5), s = std::sin(0.5);
    misc3d::PointCloud src, dst;
    std::vector<double> fsrc, fdst;
    for (int i = 0; i < 3000; ++i) {
        const misc3d::Vector3d p = {U(gen), U(gen), U(gen)};
        src.points_.push_back(p);
        dst.points_.push_back({c * p[0] - s * p[1] + 0.3, s * p[0] + c * p[1] - 0.1, p[2] + 0.2});
        for (int k = 0; k < 8; ++k) {
            const double f = U(gen);
            fsrc.push_back(f);
            fdst.push_back(f + 1e-3 * U(gen));
        }
    }
    misc3d::registration::LeastSquareSolver ls(false);
    const misc3d::Matrix4d T = ls.Solve(src, dst);
    CHECK(std::fabs(T[0] - c) < 1e-12 && std::fabs(T[1] + s) < 1e-12 && std::fabs(T[3] - 0.3) < 1e-12);
    CHECK(std::fabs(T[11] - 0.2) < 1e-12 && T[15] == 1.0);
    misc3d::registration::ANNMatcher matcher(misc3d::registration::MatchMethod::ANNOY, 4);
    misc3d::registration::FeatureView fa{fsrc.data(), 8, 3000}, fb{fdst.data(), 8, 3000};
    auto corres = matcher.Match(fa, fb);
    CHECK(corres.first.size() > 2900);
    for (size_t k = 0; k < corres.first.size(); ++k) CHECK(corres.first[k] == corres.second[k]);
    misc3d::registration::RANSACSolver rs(0.03, 2000, 0.9);
    rs.SetSeed(17);
    const misc3d::Matrix4d Tr = rs.Solve(src, dst, corres);
    CHECK(std::fabs(Tr[0] - c) < 1e-9 && std::fabs(Tr[3] - 0.3) < 1e-9);
    CHECK(rs.GetStats().fitness == 1.0);
    // ---- GlobalRegistration + RegisterFragmentPairs (include/misc3d/reconstruction/global_registration.h): the shape of
    // src/pipeline.cpp:428-439 -- three fragments, every pair; one call per std::thread, and the batch entry point
    {
        std::vector<misc3d::PointCloud> frag(3);
        std::vector<std::vector<double>> feat(3);
        const double ang[3] = {0.0, 0.5, -0.3};
        for (int i = 0; i < 2500; ++i) {
            // a scene without symmetries: two orthogonal wall patches and a blob
            misc3d::Vector3d p;
            if (i % 3 == 0) p = {U(gen), U(gen) * 0.6, 0.1 * U(gen) * U(gen)};
            else if (i % 3 == 1) p = {0.9 + 0.05 * U(gen), U(gen) * 0.7, 0.5 + 0.5 * U(gen)};
            else p = {0.3 * U(gen) - 0.4, 0.2 * U(gen) + 0.3, 0.3 * U(gen) + 0.4};
            double f[8];
            for (int k = 0; k < 8; ++k) f[k] = U(gen);
            for (int j = 0; j < 3; ++j) {
                const double cj = std::cos(ang[j]), sj = std::sin(ang[j]);
                frag[j].points_.push_back({cj * p[0] - sj * p[1] + 0.1 * j, sj * p[0] + cj * p[1] - 0.05 * j, p[2] + 0.02 * j});
                for (int k = 0; k < 8; ++k) feat[j].push_back(f[k] + 1e-3 * U(gen));
            }
        }
        std::vector<misc3d::CloudView> views(frag.begin(), frag.end());
        std::vector<misc3d::registration::FeatureView> fviews;
        for (int j = 0; j < 3; ++j) fviews.push_back({feat[j].data(), 8, frag[j].points_.size()});
        const std::vector<std::pair<int, int>> pairs = {{0, 1}, {0, 2}, {1, 2}};
        const std::vector<uint64_t> seeds = {5, 6, 7};
        misc3d::reconstruction::GlobalRegistrationOption opt;
        opt.voxel_size = 0.03 / 1.4;
        opt.max_iter = 2000;
        const auto batch = misc3d::reconstruction::RegisterFragmentPairs(views, fviews, pairs, opt, seeds, {0}, 3);
        CHECK(batch.size() == 3);
        // the same three pairs, one std::thread each (the reference's own loop)
        std::vector<std::tuple<bool, misc3d::Matrix4d, misc3d::reconstruction::Matrix6d>> single(3);
        std::vector<std::thread> th;
        for (int k = 0; k < 3; ++k)
            th.emplace_back([&, k] {
                single[k] = misc3d::reconstruction::GlobalRegistration(views[pairs[k].first], views[pairs[k].second],
                                                                       fviews[pairs[k].first], fviews[pairs[k].second], opt,
                                                                       &seeds[k]);
            });
        for (auto& t : th) t.join();
        for (int k = 0; k < 3; ++k) {
            CHECK(batch[k].success_ && std::get<0>(single[k]));
            CHECK(batch[k].transformation_ == std::get<1>(single[k]) && batch[k].information_ == std::get<2>(single[k]));
            const double da = ang[pairs[k].second] - ang[pairs[k].first];
            CHECK(std::fabs(batch[k].transformation_[0] - std::cos(da)) < 1e-6 && std::fabs(batch[k].transformation_[4] - std::sin(da)) < 1e-6);
            CHECK(batch[k].information_[35] == 2500.0 && batch[k].stats_.n_matches > 2400);
        }
    }

    // ---- EstimateNormalsFromMap (include/misc3d/common/normal_estimation.h): tilted plane seen from the origin
    {
        const int w = 64, h = 48;
        misc3d::PointCloud map;
        for (int r = 0; r < h; ++r)
            for (int c = 0; c < w; ++c) {
                const double x = (c - w / 2) * 0.01, y = (r - h / 2) * 0.01;
                map.points_.push_back({x, y, 1.0 + 0.2 * x - 0.1 * y});
            }
        map.points_[100][2] = std::nan("");   // one invalid pixel
        misc3d::common::EstimateNormalsFromMap(map, {w, h}, 3);
        CHECK(map.normals_.size() == map.points_.size());
        const double inv = 1.0 / std::sqrt(0.2 * 0.2 + 0.1 * 0.1 + 1.0);
        const auto& nn = map.normals_[h / 2 * w + w / 2];
        CHECK(std::fabs(nn[0] - 0.2 * inv) < 1e-6 && std::fabs(nn[1] + 0.1 * inv) < 1e-6 && std::fabs(nn[2] + inv) < 1e-6);
        CHECK(std::isnan(map.normals_[100][0]));
        bool threw_size = false;
        try {
            misc3d::common::EstimateNormalsFromMap(map, {w + 1, h}, 3);
        } catch (const std::runtime_error& e) {
            threw_size = std::string(e.what()).find("not equal to given point map size") != std::string::npos;
        }
        CHECK(threw_size);
    }

    // ---- DetectBoundaryPoints (include/misc3d/features/boundary_detection.h): a square patch, Hybrid(0.08, 30)
    {
        misc3d::PointCloud patch;
        for (int i = 0; i < 3000; ++i) patch.points_.push_back({0.5 + 0.5 * U(gen), 0.5 + 0.5 * U(gen), 0.001 * U(gen)});
        const auto b = misc3d::features::DetectBoundaryPoints(patch, misc3d::features::KDTreeSearchParamHybrid(0.08, 30));
        CHECK(b.size() > 100 && b.size() < 1200);
        size_t near_edge = 0;
        for (size_t i : b) {
            const auto& q = patch.points_[i];
            const double e = std::min(std::min(q[0], 1.0 - q[0]), std::min(q[1], 1.0 - q[1]));
            near_edge += e < 0.08;
        }
        CHECK(near_edge * 10 > b.size() * 9);   // >= 90 % of the flagged points lie within one radius of the edge
        for (size_t k = 1; k < b.size(); ++k) CHECK(b[k] > b[k - 1]);
    }

    std::printf("host api: all checks passed\n");
    return 0;
}
