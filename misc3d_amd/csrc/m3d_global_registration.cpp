// m3d_global_registration.cpp -- ReconstructionPipeline::GlobalRegistration (/root/reference/src/pipeline.cpp:790-828, the
// Ransac method) and its caller's shape, BuildPoseGraphForScene's one std::thread per fragment pair (pipeline.cpp:428-439), as
// entry points of the C ABI.
//
//   reference (one std::thread per pair, each ...)                 here
//   ------------------------------------------------               ---------------------------------------------------------
//   ANNMatcher(ANNOY).Match(fpfh_s, fpfh_t)        :800-802        match_mutual_nn_on      (exact mutual NN, m3d_registration.cpp)
//   RANSACSolver(1.4 voxel).Solve(...)             :806-807        registration_ransac_on  on the pair's two RESIDENT clouds
//   pose.isIdentity(1e-8) -> (true, pose, I)       :814-816        is_identity4 (Eigen's rule, on the host)
//   GetInformationMatrixFromPointClouds(...)       :818-820        information_matrix_on   (the same resident clouds: one upload)
//   info(5,5) / min(Ns, Nt) < 0.3 -> reject        :821-824        host
//
// One pair holds ONE lane of a device (m3d_driver.hpp: DeviceCtx) from its first upload to its result.  The batch form deals
// the pairs round-robin to the devices and runs `inflight` of them side by side per device, each on a lane of its own: pair k + 1's
// uploads (105 MB of descriptors at C4 size) and host work (cross-check, RANSAC replay) run under pair k's kernels.  No
// collective: pairs are independent ("replicas").
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "m3d_config.hpp"
#include "m3d_driver.hpp"

using namespace m3d;

namespace {

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Eigen::MatrixBase::isIdentity(prec) (pipeline.cpp:814): diagonal internal::isApprox(x, 1, prec) = |x - 1| <= min(|x|, 1) prec,
// off-diagonal internal::isMuchSmallerThan(x, 1, prec) = |x| <= prec.  (oracle: orc_is_identity4)
bool is_identity4(const double* T, double prec) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            const double x = T[4 * r + c];
            if (r == c) {
                if (!(std::fabs(x - 1.0) <= std::min(std::fabs(x), 1.0) * prec)) return false;
            } else if (!(std::fabs(x) <= prec)) {
                return false;
            }
        }
    return true;
}

void identity6(double* info) {
    for (int k = 0; k < 36; ++k) info[k] = (k % 7 == 0) ? 1.0 : 0.0;
}

int check_pair_args(const double* src, size_t n_src, const double* dst, size_t n_dst, const double* feat_src,
                    const double* feat_dst, int dim, double voxel_size, const double* T, const double* info) {
    if (!T || !info || dim <= 0 || dim > 1024 || (!src && n_src) || (!dst && n_dst) || (!feat_src && n_src) ||
        (!feat_dst && n_dst) || !(voxel_size == voxel_size))
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (n_src < 3 || n_dst < 3)   // RANSACSolver::Solve, transform_estimation.cpp:130-133 (LogError throws)
        return fail(M3D_ERR_TOO_FEW_POINTS, "The number of points pair is less than 3.");
    if (n_src >= ((size_t)1 << 31) || n_dst >= ((size_t)1 << 31)) return fail(M3D_ERR_INVALID_ARG, "too many points");
    return M3D_OK;
}

// A fragment resident on a device for the length of a batch call (m3d_register_fragment_pairs): its points as an m3d_cloud
// (SoA + bounding box: what the solver and the information matrix read) and its descriptors as uploaded, with their largest
// |value| (what the matcher's screen scales by).  Read-only once `ready`: any lane of the device may use it.
struct ResidentFragment {
    std::mutex mu;            // held while the fragment is being uploaded (the first pair that needs it does that)
    bool ready = false, failed = false;
    std::string error;
    m3d_cloud* cloud = nullptr;
    DevBuf feat;
    double max_abs = -1.0;
};

// the pair on a lane the caller holds; arguments checked.  rs / rt: the pair's fragments if they are resident (else null:
// uploaded here from src / dst / feat_*).
int global_registration_on(DeviceCtx* ctx, const double* src, size_t n_src, const double* dst, size_t n_dst,
                           const double* feat_src, const double* feat_dst, int dim, double voxel_size, int max_iter,
                           double edge_length_threshold, double confidence, const uint64_t* seed, double* T, double* info,
                           m3d_global_reg_stats* stats, const ResidentFragment* rs = nullptr,
                           const ResidentFragment* rt = nullptr) {
    const double t0 = now_ms();
    const double max_dis = voxel_size * 1.4;   // pipeline.cpp:796
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(T, I4, sizeof(I4));
    identity6(info);
    m3d_global_reg_stats st;
    std::memset(&st, 0, sizeof(st));
    st.ransac.best_index = -1;
    st.device = ctx->logical;   // (the ordinal the caller dealt the pair to: under m3d_config.device_aliases several share a physical device)
    st.lane = ctx->lane;
    auto leave = [&](int rc) {
        st.ms_total = now_ms() - t0;
        if (stats) *stats = st;
        return rc;
    };
    // ---- ANNMatcher::Match (:800-802)
    std::vector<size_t> cs(n_src), cd(n_src);
    size_t m = 0;
    const MatchSide ms = rs ? MatchSide{nullptr, rs->feat.as<double>(), rs->max_abs} : MatchSide{feat_src, nullptr, -1.0};
    const MatchSide mt = rt ? MatchSide{nullptr, rt->feat.as<double>(), rt->max_abs} : MatchSide{feat_dst, nullptr, -1.0};
    int rc = match_mutual_nn_on(ctx, ms, n_src, mt, n_dst, dim, cs.data(), cd.data(), &m);
    if (rc != M3D_OK) return leave(rc);
    st.n_matches = m;
    const double t1 = now_ms();
    st.ms_match = t1 - t0;
    // ---- RANSACSolver(max_dis).Solve (:806-807).  With fewer than 3 matches or max_dis <= 0 Open3D returns the default
    // RegistrationResult -- the identity -- without touching the clouds: the shortcut below (:814-816) ends the call.
    const bool trivial = m < 3 || !(max_dis > 0.0);
    m3d_cloud *csrc = rs ? rs->cloud : nullptr, *cdst = rt ? rt->cloud : nullptr;
    if (!trivial) {
        if (!csrc) csrc = m3d_cloud_create_on(ctx, src, nullptr, n_src, 0);
        if (csrc && !cdst) cdst = m3d_cloud_create_on(ctx, dst, nullptr, n_dst, 0);
        if (!csrc || !cdst) {
            if (csrc && !rs) m3d_cloud_destroy_on(csrc);
            return leave(M3D_ERR_DEVICE);
        }
    }
    m3d_reg* session = nullptr;   // (the solver's session stays for the information matrix: one target grid for both)
    rc = registration_ransac_on(trivial ? nullptr : ctx, csrc, cdst, src, n_src, dst, n_dst, cs.data(), cd.data(), m, max_dis,
                                max_iter, edge_length_threshold, confidence, seed, T, &st.ransac, &session);
    const double t2 = now_ms();
    st.ms_ransac = t2 - t1;
    if (rc == M3D_OK) {
        if (is_identity4(T, 1e-8)) {   // :814-816
            st.identity_shortcut = 1;
        } else {
            double gi[36];
            uint64_t nc = 0;
            rc = information_matrix_on(ctx, csrc, cdst, dst, n_dst, max_dis, T, gi, &nc, session);   // :818-820
            st.n_info_correspondences = nc;
            if (rc == M3D_OK) {
                if (gi[35] / (double)std::min(n_src, n_dst) < 0.3)   // :821-824
                    rc = M3D_FALSE;
                else
                    std::memcpy(info, gi, sizeof(gi));
            }
            st.ms_info = now_ms() - t2;
        }
    }
    if (session) reg_session_release(session);
    if (csrc && !rs) m3d_cloud_destroy_on(csrc);
    if (cdst && !rt) m3d_cloud_destroy_on(cdst);
    return leave(rc);
}

// The batch entry points' workers: worker(d, w) drains device d's share of the pairs.  extern "C" code must not let an
// exception out: a worker that throws (bad_alloc in a pair's buffers) is caught and its pair keeps
// M3D_ERR_INTERNAL; a thread that cannot be created is done without -- whoever runs drains the device's queue, and the
// calling thread sweeps every device once more at the end.  The caller's current HIP device is put back.
template <class W>
void run_pair_workers(int n_dev, int per_dev, size_t n_pairs, W&& worker) {
    int caller_dev = -1;
    if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
    auto guarded = [&](int d, int w) {
        try {
            worker(d, w);
        } catch (...) {
            set_error("an exception in a pair's worker (out of memory?)");
        }
    };
    std::vector<std::thread> th;
    const size_t busiest = (n_pairs + (size_t)n_dev - 1) / (size_t)n_dev;
    try {
        for (int d = 0; d < n_dev; ++d)
            for (int w = 0; w < per_dev && (size_t)w < busiest; ++w)
                if (d || w) th.emplace_back(guarded, d, w);
    } catch (...) {   // (std::system_error: no more threads)
    }
    guarded(0, 0);
    for (auto& t : th) t.join();
    for (int d = 0; d < n_dev; ++d) guarded(d, 0);   // (nothing left unless a thread was missing)
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
}

}  // namespace

extern "C" {

int m3d_global_registration(const double* src, size_t n_src, const double* dst, size_t n_dst, const double* feat_src,
                            const double* feat_dst, int dim, double voxel_size, int max_iter, double edge_length_threshold,
                            double confidence, const uint64_t* seed, int device, double* T, double* info,
                            m3d_global_reg_stats* stats) {
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const int chk = check_pair_args(src, n_src, dst, n_dst, feat_src, feat_dst, dim, voxel_size, T, info);
    if (chk != M3D_OK) return chk;
    LaneLock lane(device);
    if (!lane.ctx) return M3D_ERR_DEVICE;
    return global_registration_on(lane.ctx, src, n_src, dst, n_dst, feat_src, feat_dst, dim, voxel_size, max_iter,
                                  edge_length_threshold, confidence, seed, T, info, stats);
}

int m3d_global_registration_batch(m3d_fragment_pair* pairs, size_t n_pairs, int dim, double voxel_size, int max_iter,
                                  double edge_length_threshold, double confidence, const int* devices, int n_dev,
                                  int inflight) {
    if ((!pairs && n_pairs) || !devices || n_dev < 1) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    for (int a = 0; a < n_dev; ++a)
        for (int b = a + 1; b < n_dev; ++b)
            if (devices[a] == devices[b]) return fail(M3D_ERR_INVALID_ARG, "devices must be distinct");
    for (size_t k = 0; k < n_pairs; ++k) {
        pairs[k].rc = M3D_ERR_INTERNAL;
        std::memset(&pairs[k].stats, 0, sizeof(pairs[k].stats));
    }
    if (n_pairs == 0) return M3D_OK;
    for (int d = 0; d < n_dev; ++d)
        if (!get_ctx(devices[d])) return M3D_ERR_DEVICE;   // (a bad ordinal fails the call, not its pairs one by one)
    const int lanes = lane_count();
    const int per_dev = std::min(inflight > 0 ? inflight : lanes, lanes);
    // pair k belongs to device k % n_dev; a device's workers take its pairs in order, whoever is free first
    std::vector<std::atomic<size_t>> next((size_t)n_dev);
    for (auto& a : next) a.store(0);
    std::vector<std::string> errs(n_pairs);
    auto worker = [&](int d, int w) {
        for (;;) {
            const size_t j = next[(size_t)d].fetch_add(1);
            const size_t k = j * (size_t)n_dev + (size_t)d;
            if (k >= n_pairs) break;
            m3d_fragment_pair& p = pairs[k];
            const int chk = check_pair_args(p.src, p.n_src, p.dst, p.n_dst, p.feat_src, p.feat_dst, dim, voxel_size, p.T, p.info);
            if (chk != M3D_OK) {
                p.rc = chk;
            } else {
                LaneLock lane(devices[d], w);
                p.rc = lane.ctx ? global_registration_on(lane.ctx, p.src, p.n_src, p.dst, p.n_dst, p.feat_src, p.feat_dst, dim,
                                                         voxel_size, max_iter, edge_length_threshold, confidence,
                                                         p.has_seed ? &p.seed : nullptr, p.T, p.info, &p.stats)
                                : M3D_ERR_DEVICE;
            }
            if (p.rc < 0) errs[k] = m3d_last_error();
        }
    };
    run_pair_workers(n_dev, per_dev, n_pairs, worker);
    for (size_t k = 0; k < n_pairs; ++k)
        if (pairs[k].rc < 0) {   // the first failed pair's message is the call's; every pair keeps its own code
            set_error("pair " + std::to_string(k) + ": " + errs[k]);
            return pairs[k].rc;
        }
    return M3D_OK;
}

// BuildPoseGraphForScene as the reference holds its data: n fragments, pairs by index.  Every fragment a device needs is
// uploaded to it ONCE -- by the first pair that asks for it, on that pair's lane -- and stays resident for the call (288 GB of
// HBM: 58 MB per 200 000-point fragment with 33-D descriptors); a pair then costs no host-to-device traffic beyond its
// correspondences.  Results are those of m3d_global_registration on the same arrays and seeds.
int m3d_register_fragment_pairs(const m3d_fragment_view* frags, size_t n_frags, int dim, m3d_pair_result* pairs,
                                size_t n_pairs, double voxel_size, int max_iter, double edge_length_threshold,
                                double confidence, const int* devices, int n_dev, int inflight) {
    if ((!frags && n_frags) || (!pairs && n_pairs) || !devices || n_dev < 1) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    for (int a = 0; a < n_dev; ++a)
        for (int b = a + 1; b < n_dev; ++b)
            if (devices[a] == devices[b]) return fail(M3D_ERR_INVALID_ARG, "devices must be distinct");
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (size_t k = 0; k < n_pairs; ++k) {
        pairs[k].rc = M3D_ERR_INTERNAL;
        std::memcpy(pairs[k].T, I4, sizeof(I4));
        identity6(pairs[k].info);
        std::memset(&pairs[k].stats, 0, sizeof(pairs[k].stats));
    }
    // (every pair initialised before the first is judged: the header promises each its own rc -- ADVICE r5)
    for (size_t k = 0; k < n_pairs; ++k)
        if (pairs[k].s < 0 || pairs[k].t < 0 || (size_t)pairs[k].s >= n_frags || (size_t)pairs[k].t >= n_frags) {
            pairs[k].rc = M3D_ERR_INVALID_ARG;
            return fail(M3D_ERR_INVALID_ARG, "pair " + std::to_string(k) + ": fragment index out of range");
        }
    if (n_pairs == 0) return M3D_OK;
    for (int d = 0; d < n_dev; ++d)
        if (!get_ctx(devices[d])) return M3D_ERR_DEVICE;
    const int lanes = lane_count();
    const int per_dev = std::min(inflight > 0 ? inflight : lanes, lanes);
    std::vector<std::unique_ptr<ResidentFragment[]>> resident((size_t)n_dev);
    for (auto& r : resident) r.reset(new ResidentFragment[n_frags]);
    // the fragment on this device, uploaded now if nobody has yet (the caller holds lane `ctx`)
    auto fetch = [&](int d, DeviceCtx* ctx, size_t i, const ResidentFragment** out) -> int {
        ResidentFragment& r = resident[(size_t)d][i];
        std::lock_guard<std::mutex> lock(r.mu);
        if (!r.ready && !r.failed) {
            const m3d_fragment_view& f = frags[i];
            int rc = M3D_OK;
            r.cloud = m3d_cloud_create_on(ctx, f.xyz, nullptr, f.n, 0);
            if (!r.cloud) rc = M3D_ERR_DEVICE;
            const size_t bytes = sizeof(double) * (size_t)dim * f.n;
            if (rc == M3D_OK && !r.feat.reserve(std::max<size_t>(bytes, 8))) rc = M3D_ERR_DEVICE;
            if (rc == M3D_OK &&
                (hipMemcpyAsync(r.feat.p, f.feat, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                 hipStreamSynchronize(ctx->stream) != hipSuccess))
                rc = fail(M3D_ERR_DEVICE, "fragment upload failed");
            if (rc == M3D_OK && dim == 33) rc = device_max_abs(ctx, r.feat.as<double>(), f.n * 33, &r.max_abs);
            if (rc == M3D_OK) {
                r.ready = true;
            } else {
                // Not latched: whatever was taken goes back, this pair uploads its own copies like
                // m3d_global_registration_batch does (*out stays null), and a later pair may try again -- residency is an
                // optimisation, a device short of memory is not a failed loop closure.
                if (r.cloud) m3d_cloud_destroy_on(r.cloud);
                r.cloud = nullptr;
                r.feat.release();
            }
        }
        *out = r.ready ? &r : nullptr;
        return M3D_OK;
    };
    std::vector<std::atomic<size_t>> next((size_t)n_dev);
    for (auto& a : next) a.store(0);
    std::vector<std::string> errs(n_pairs);
    auto worker = [&](int d, int w) {
        for (;;) {
            const size_t j = next[(size_t)d].fetch_add(1);
            const size_t k = j * (size_t)n_dev + (size_t)d;
            if (k >= n_pairs) break;
            m3d_pair_result& p = pairs[k];
            const m3d_fragment_view &fs = frags[p.s], &ft = frags[p.t];
            int rc = check_pair_args(fs.xyz, fs.n, ft.xyz, ft.n, fs.feat, ft.feat, dim, voxel_size, p.T, p.info);
            if (rc == M3D_OK) {
                LaneLock lane(devices[d], w);
                const ResidentFragment *rs = nullptr, *rt = nullptr;
                if (!lane.ctx) rc = M3D_ERR_DEVICE;
                if (rc == M3D_OK) rc = fetch(d, lane.ctx, (size_t)p.s, &rs);
                if (rc == M3D_OK) rc = fetch(d, lane.ctx, (size_t)p.t, &rt);
                if (rc == M3D_OK)
                    rc = global_registration_on(lane.ctx, fs.xyz, fs.n, ft.xyz, ft.n, fs.feat, ft.feat, dim, voxel_size, max_iter,
                                                edge_length_threshold, confidence, p.has_seed ? &p.seed : nullptr, p.T, p.info,
                                                &p.stats, rs, rt);
            }
            p.rc = rc;
            if (rc < 0) errs[k] = m3d_last_error();
        }
    };
    int caller_dev = -1;
    if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
    run_pair_workers(n_dev, per_dev, n_pairs, worker);
    for (int d = 0; d < n_dev; ++d) {   // every lane of the call has drained (each pair ends with its stream idle)
        {
            const int phys = physical_device(devices[d]);
            if (phys >= 0) (void)hipSetDevice(phys);
        }
        for (size_t i = 0; i < n_frags; ++i) {
            ResidentFragment& r = resident[(size_t)d][i];
            if (r.cloud) {
                CtxLock lock(r.cloud->ctx);
                m3d_cloud_destroy_on(r.cloud);
            }
            r.feat.release();
        }
    }
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    for (size_t k = 0; k < n_pairs; ++k)
        if (pairs[k].rc < 0) {
            set_error("pair " + std::to_string(k) + ": " + errs[k]);
            return pairs[k].rc;
        }
    return M3D_OK;
}

}  // extern "C"
