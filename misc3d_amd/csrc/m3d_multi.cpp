// m3d_multi.cpp -- one process, several devices: a thread, a replica of the cloud and a LOCAL communicator per device
// (m3d_segment_plane_iterative_multi, m3d_fit_multi; SURVEY.md 8(e): hypotheses sharded, points replicated).
#include "m3d_driver_internal.hpp"

#pragma clang fp contract(off)

using namespace m3d;

extern "C" {

// ---- one process, several devices: a thread, a replica and a LOCAL communicator per device ----------------------
namespace {
int run_on_devices(const int* devices, int n_dev, const std::function<int(int, m3d_comm*)>& per_rank) {
    if (!devices || n_dev < 1) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    for (int a = 0; a < n_dev; ++a)
        for (int b = a + 1; b < n_dev; ++b)
            if (devices[a] == devices[b]) return fail(M3D_ERR_INVALID_ARG, "devices must be distinct");
    if (n_dev == 1) return per_rank(0, nullptr);
    std::vector<m3d_comm*> comms((size_t)n_dev, nullptr);
    int rc = m3d_comm_create_local(n_dev, comms.data());
    if (rc != M3D_OK) return rc;
    std::vector<int> rcs((size_t)n_dev, M3D_OK);
    std::vector<std::string> errs((size_t)n_dev);
    auto rank_body = [&](int r) {
        try {
            rcs[(size_t)r] = per_rank(r, comms[(size_t)r]);
        } catch (...) {   // (extern "C" above us: no exception may pass -- bad_alloc in a rank's buffers)
            rcs[(size_t)r] = fail(M3D_ERR_INTERNAL, "an exception in a rank's thread (out of memory?)");
        }
        errs[(size_t)r] = m3d_last_error();
        if (rcs[(size_t)r] < 0) comms[(size_t)r]->local->abort(r);   // nobody waits for a rank that has given up
    };
    std::vector<std::thread> th;
    int started = 1;
    try {
        for (int r = 1; r < n_dev; ++r, ++started) th.emplace_back(rank_body, r);
    } catch (...) {   // (no more threads: the ranks that exist must not wait for the ones that do not)
        comms[0]->local->abort(started);
        rcs[(size_t)started] = M3D_ERR_INTERNAL;
        errs[(size_t)started] = "could not start a thread for the device";
    }
    rank_body(0);
    for (auto& t : th) t.join();
    // the error of the rank that gave up FIRST is the call's: the others only report that somebody did
    const int first = comms[0]->local->first_failed;
    for (m3d_comm* q : comms) m3d_comm_destroy(q);
    if (first >= 0 && first < n_dev && rcs[(size_t)first] < 0) {
        set_error("device " + std::to_string(devices[first]) + ": " + errs[(size_t)first]);
        return rcs[(size_t)first];
    }
    for (int r = 0; r < n_dev; ++r)
        if (rcs[(size_t)r] < 0) {
            set_error("device " + std::to_string(devices[r]) + ": " + errs[(size_t)r]);
            return rcs[(size_t)r];
        }
    return rcs[0];
}
}  // namespace

int m3d_segment_plane_iterative_multi(const double* xyz, size_t n, double threshold, int max_iteration,
                                      double min_ratio, const uint64_t* seed, const int* devices, int n_dev,
                                      size_t max_clusters, double* planes, size_t* cluster_offsets,
                                      size_t* cluster_indices, size_t* n_clusters) {
    if (!planes || !cluster_offsets || !cluster_indices || !n_clusters || (!xyz && n))
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    const uint64_t seed0 = resolve_seed(seed);   // one seed for all ranks
    const size_t cap = std::min(max_clusters, n);
    return run_on_devices(devices, n_dev, [&](int r, m3d_comm* comm) -> int {
        if (r == 0)
            return segment_impl(xyz, n, threshold, max_iteration, min_ratio, &seed0, devices[0], comm, max_clusters, planes,
                                cluster_offsets, cluster_indices, n_clusters);
        // helper ranks compute the same result into scratch (every rank removes the same inliers from its replica)
        std::vector<double> pl(4 * std::max<size_t>(cap, 1));
        std::vector<size_t> off(cap + 2), idx(std::max<size_t>(n, 1));
        size_t k = 0;
        return segment_impl(xyz, n, threshold, max_iteration, min_ratio, &seed0, devices[r], comm, max_clusters, pl.data(),
                            off.data(), idx.data(), &k);
    });
}

int m3d_fit_multi(int kind, const double* xyz, const double* normals, size_t n, double threshold, size_t max_iteration,
                  double probability, const uint64_t* seed, const int* devices, int n_dev, double* params,
                  size_t* inliers, size_t* n_inliers, m3d_stats* stats) {
    if (!params || (!xyz && n) || kind < 0 || kind > 2) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    const int vr = validate_fit_args(kind, n, normals != nullptr, probability);
    if (vr != M3D_OK) return vr;
    const uint64_t seed0 = resolve_seed(seed);
    return run_on_devices(devices, n_dev, [&](int r, m3d_comm* comm) -> int {
        m3d_cloud* c = m3d_cloud_create(xyz, normals, n, devices[r]);
        if (!c) return M3D_ERR_DEVICE;
        int rc;
        if (r == 0) {
            rc = m3d_cloud_fit_sharded(c, comm, kind, threshold, max_iteration, probability, &seed0, params, inliers,
                                       n_inliers, stats);
        } else {
            double par[kModelStride];
            size_t ni = 0;
            rc = m3d_cloud_fit_sharded(c, comm, kind, threshold, max_iteration, probability, &seed0, par, nullptr, &ni,
                                       nullptr);
        }
        m3d_cloud_destroy(c);
        return rc;
    });
}


}  // extern "C"
