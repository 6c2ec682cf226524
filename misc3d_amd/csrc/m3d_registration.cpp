// m3d_registration.cpp -- host drivers of the registration entry points of the C ABI.
//
//   m3d_kabsch               registration::LeastSquareSolver::Solve   src/transform_estimation.cpp:49-66
//   m3d_registration_ransac  registration::RANSACSolver::Solve        src/transform_estimation.cpp:124-164
//                            (= Open3D 0.15.1 RegistrationRANSACBasedOnCorrespondence)
//   (m3d_match_mutual_nn -- registration::ANNMatcher::Match, src/correspondence_matching.cpp:52-84 -- lives in m3d_match.cpp)
//
// Sequential (one-thread) semantics of the Open3D loop with an explicit seed: iteration `itr` draws
// three correspondences from std::uniform_int_distribution<int>(0, M-1) on std::mt19937 iff
// itr < est_k_global; 3-point umeyama; EdgeLength + Distance checkers; survivors are validated
// against the whole source cloud (fitness = #points with a target neighbour closer than threshold);
// best = (fitness, then rmse); every improvement re-estimates est_k from the inlier ratio of the
// correspondence set.  The data-parallel parts run on the GPU in chunks of iterations; the host
// replays the sequential rule over each chunk in index order.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "../../include/misc3d_amd_bench.h"
#include "m3d_comm.hpp"
#include "m3d_config.hpp"
#include "m3d_driver.hpp"
#include "m3d_reg_fp.hpp"

namespace m3d {
int stream_wait_spin(DeviceCtx* ctx);   // (m3d_fit.cpp: the end of the stream's work, polled in page-locked memory)
}
#include "m3d_reg_kernels.hpp"

#pragma clang fp contract(off)

using namespace m3d;

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(M3D_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define RESERVE(buf, bytes)                               \
    do {                                                  \
        if (!(buf).reserve(bytes)) return M3D_ERR_DEVICE; \
    } while (0)

namespace {

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// static_cast<int>(std::ceil(v)) as the x86-64 conversion behaves (cvttsd2si -> INT_MIN when out of range)
int ceil_to_int_x86(double v) {
    const double c = std::ceil(v);
    if (!(c > -2147483649.0) || !(c < 2147483648.0)) return INT_MIN;
    return (int)c;
}

struct Scratch {  // per-call device buffers (registration calls are rare and large: no caching)
    DevBuf corr_src, corr_dst, triples, T12, pass, list, Ts, partial, counts, cell_of_point, cell_start, fill,
        tile_sums, total, qx, qy, qz, best, vals, block_counts, sums, one_T, ratio, partial_sum, sum2,
        s_cell_of_point, s_cell_start, s_fill, s_tile_sums, sx, sy, sz, keep, nl_start, nl_pts, nl_hdr, nl32, nl_rec, nl32_start, nl32_fallbacks, cell_orig, tile_sph, q4,
        cc_x, cc_y, cc_z, cc_64, cc_xa, cc_ya, cc_za, cc_R, cc_stats, cc_redo, cc_ring, cc_ring_tmp, cc_pairs;   // the validation's candidate cache (m3d_reg_cache.hip)
    void release() {
        for (DevBuf* b : {&corr_src, &corr_dst, &triples, &T12, &pass, &list, &Ts, &partial, &counts,
                          &cell_of_point, &cell_start, &fill, &tile_sums, &total, &qx, &qy, &qz, &best, &vals,
                          &block_counts, &sums, &one_T, &ratio, &partial_sum, &sum2, &s_cell_of_point,
                          &s_cell_start, &s_fill, &s_tile_sums, &sx, &sy, &sz, &keep, &nl_start, &nl_pts, &nl_hdr, &nl32, &nl_rec, &nl32_start, &nl32_fallbacks, &cell_orig,
                          &tile_sph, &q4, &cc_x, &cc_y, &cc_z, &cc_64, &cc_xa, &cc_ya, &cc_za, &cc_R, &cc_stats, &cc_redo, &cc_ring, &cc_ring_tmp, &cc_pairs})
            b->release();
    }
};

struct RegCtx {
    DeviceCtx* ctx;
    CloudView src, dst;
    GridDesc g;
    Scratch* s;
    uint32_t m;
    double thr;
};

// Uniform grid over the target cloud for radius-bounded exact nearest-neighbour queries (KDTreeFlann stand-in):
// cell edge h = 1.001 radius / K, K = 4, 2, 1 while the dense cell table fits (<= 2^27 cells incl. 2K+1 pad
// cells per side; if even K = 1
// does not fit the cell is doubled: a coarser grid with K = 1 still covers the radius); points counting-
// sorted by cell (qx/qy/qz + cell_start), optional neighbour lists (3x3x3 block of every cell, contiguous;
// bounded to 2 M target points; M3D_REG_NL=0 switches them off).  with_orig: also keep the original index of
// every sorted point (S.cell_orig) and store it in the w component of the neighbour-list entries.
int add_neighbour_lists(DeviceCtx* ctx, Scratch& S, GridDesc* gp, size_t n_dst, const uint32_t* orig);
// bbox6 != null: the bounding box (lo, hi) of the points that have three finite coordinates -- the resident cloud knows it
// from its creation (m3d_cloud::bb); the host pass over the caller's array it replaces took 0.1 ms per 200 000 points
int build_target_grid(DeviceCtx* ctx, Scratch& S, const CloudView& dst_view, const double* dst, size_t n_dst,
                      double radius, bool with_orig, GridDesc* g_out, int K0 = 4, bool with_nl = true,
                      const double* bbox6 = nullptr) {
    GridDesc g;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (bbox6) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = bbox6[k];
            hi[k] = bbox6[3 + k];
        }
    } else {
        for (size_t i = 0; i < n_dst; ++i)   // bounding box on the host: one pass over n_dst points
            for (int k = 0; k < 3; ++k) {
                const double v = dst[3 * i + k];
                if (std::isfinite(v)) {
                    lo[k] = std::min(lo[k], v);
                    hi[k] = std::max(hi[k], v);
                }
            }
    }
    for (int k = 0; k < 3; ++k)
        if (!(lo[k] <= hi[k])) lo[k] = hi[k] = 0.0;
    int K = K0;
    double h = radius * 1.001 / K;
    uint64_t dims[3];
    for (;;) {
        bool fits = true;
        uint64_t cells = 1;
        for (int k = 0; k < 3; ++k) {
            const double ext = (hi[k] - lo[k]) / h;
            if (!(ext < 1e9)) {
                fits = false;
                break;
            }
            dims[k] = (uint64_t)ext + 1 + 2 * (uint64_t)(2 * K + 1);
            cells *= dims[k];
            if (cells > ((uint64_t)1 << 27)) fits = false;
        }
        if (fits) break;
        if (K > 1)
            K /= 2;
        h *= 2.0;
    }
    g.K = K;
    g.morton_bits = 0;
    // 2K+1 pad cells on every side: a query up to one radius (< K cells) outside the bounding box still has its
    // whole (2K+1)^3 search block inside the table, and anything further out cannot have a neighbour
    g.ox = lo[0] - (2 * K + 1) * h;
    g.oy = lo[1] - (2 * K + 1) * h;
    g.oz = lo[2] - (2 * K + 1) * h;
    g.inv_h = 1.0 / h;
    g.r2 = radius * radius;  // radius * radius, KDTreeFlann::SearchHybrid
    g.h2_in = (0.999 * h) * (0.999 * h);
    g.nx = (uint32_t)dims[0];
    g.ny = (uint32_t)dims[1];
    g.nz = (uint32_t)dims[2];
    const uint32_t ncell = g.nx * g.ny * g.nz;
    RESERVE(S.cell_of_point, sizeof(uint32_t) * n_dst);
    RESERVE(S.cell_start, sizeof(uint32_t) * ((size_t)ncell + 1));
    RESERVE(S.fill, sizeof(uint32_t) * std::max<size_t>(dst_view.n, 1));   // rank of every point in its cell
    RESERVE(S.tile_sums, sizeof(uint32_t) * ((size_t)(ncell + 2047) / 2048 + 1));
    RESERVE(S.total, 16);
    RESERVE(S.qx, sizeof(double) * n_dst);
    RESERVE(S.qy, sizeof(double) * n_dst);
    RESERVE(S.qz, sizeof(double) * n_dst);
    if (with_orig) RESERVE(S.cell_orig, sizeof(uint32_t) * n_dst);
    uint32_t* orig = with_orig ? S.cell_orig.as<uint32_t>() : nullptr;
    RESERVE(S.q4, sizeof(double4) * std::max<size_t>(n_dst, 1));
    g.q4 = S.q4.as<double4>();
    launch_grid_build(dst_view, g, S.cell_of_point.as<uint32_t>(), S.cell_start.as<uint32_t>(),
                      S.fill.as<uint32_t>(), S.tile_sums.as<uint32_t>(), S.total.as<uint32_t>(),
                      S.qx.as<double>(), S.qy.as<double>(), S.qz.as<double>(), ctx->stream, orig, S.q4.as<double4>());
    if (with_nl) {
        const int rn = add_neighbour_lists(ctx, S, &g, n_dst, orig);
        if (rn != M3D_OK) return rn;
    }
    *g_out = g;
    return M3D_OK;
}

// Neighbour lists on top of a grid built by build_target_grid (its cell_start / q arrays in S): g gains nl_start /
// nl_pts.  Separate because they only pay off once enough queries follow (173 MB for 200 k target points).
int add_neighbour_lists(DeviceCtx* ctx, Scratch& S, GridDesc* gp, size_t n_dst, const uint32_t* orig) {
    GridDesc& g = *gp;
    const uint32_t ncell = g.nx * g.ny * g.nz;
    if (config().reg_neighbour_lists && n_dst <= ((size_t)2 << 20)) {
        RESERVE(S.nl_start, sizeof(uint32_t) * ((size_t)ncell + 1));
        launch_nl_count(g, S.cell_start.as<uint32_t>(), S.nl_start.as<uint32_t>(), S.tile_sums.as<uint32_t>(),
                        S.total.as<uint32_t>() + 1, ctx->stream);
        uint32_t entries = 0;
        HIPCHK(hipMemcpyAsync(&entries, S.nl_start.as<uint32_t>() + ncell, sizeof(uint32_t), hipMemcpyDeviceToHost,
                              ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        RESERVE(S.nl_pts, sizeof(double4) * std::max<size_t>(entries, 1));
        // without original indices riding in w (validation grid): x-sorted three-column lists with early termination
        const bool sorted = orig == nullptr && config().reg_sorted_lists != 0;
        if (sorted) RESERVE(S.nl_hdr, sizeof(uint32_t) * (size_t)ncell);
        // fp32 copies of the entries (GridDesc::nl32) when the offsets from a cell corner are good to fp32's last bit: the
        // corner g.o + i h and the subtraction are rounded in fp64, which must stay far below 2^-24 h
        double far = 0.0;
        for (double v : {g.ox, g.oy, g.oz}) far = std::max(far, std::fabs(v));
        far += (double)std::max(g.nx, std::max(g.ny, g.nz)) / g.inv_h;
        bool screen = sorted && config().reg_fp32_screen != 0 && far * 0x1p-50 <= 0x1p-24 / g.inv_h * 0.01 &&
                      1.0 / g.inv_h > 1e-15 && 1.0 / g.inv_h < 1e15 && entries < (1u << 28);
        uint32_t* overflow = S.total.as<uint32_t>() + 3;
        if (screen) {
            RESERVE(S.nl32_start, sizeof(uint32_t) * ((size_t)ncell + 1));
            launch_nl32_offsets(S.nl_start.as<uint32_t>(), ncell, S.nl32_start.as<uint32_t>(), S.tile_sums.as<uint32_t>(),
                                overflow, ctx->stream);   // (the word receives the length first, then serves as the flag)
            uint32_t entries32 = 0;
            HIPCHK(hipMemcpyAsync(&entries32, overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            RESERVE(S.nl32, sizeof(uint2) * std::max<size_t>(entries32, 1));
            RESERVE(S.nl_rec, sizeof(uint4) * (size_t)ncell);
            HIPCHK(hipMemsetAsync(overflow, 0, sizeof(uint32_t), ctx->stream));
        }
        launch_nl_fill(g, S.cell_start.as<uint32_t>(), S.nl_start.as<uint32_t>(), S.qx.as<double>(),
                       S.qy.as<double>(), S.qz.as<double>(), S.nl_pts.as<double4>(), ctx->stream, orig, sorted,
                       sorted ? S.nl_hdr.as<uint32_t>() : nullptr, screen ? S.nl32.as<uint2>() : nullptr,
                       screen ? S.nl_rec.as<uint4>() : nullptr, overflow, screen ? S.nl32_start.as<uint32_t>() : nullptr);
        if (screen) {   // a list beyond the 16-bit offsets of nl_rec: no screen for this grid
            uint32_t over = 0;
            HIPCHK(hipMemcpyAsync(&over, overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            screen = over == 0;
        }
        g.nl_hdr = sorted ? S.nl_hdr.as<uint32_t>() : nullptr;
        g.nl32 = screen ? S.nl32.as<uint2>() : nullptr;
        g.nl_rec = screen ? S.nl_rec.as<uint4>() : nullptr;
        if (screen) {
            RESERVE(S.nl32_fallbacks, sizeof(unsigned long long) * 64);   // [0] the counter; [8 ..): M3D_REG_TRIP_STATS builds
            HIPCHK(hipMemsetAsync(S.nl32_fallbacks.p, 0, sizeof(unsigned long long) * 64, ctx->stream));
            g.nl32_fallbacks = S.nl32_fallbacks.as<unsigned long long>();
        }
        g.nl_start = S.nl_start.as<uint32_t>();
        g.nl_pts = S.nl_pts.as<double4>();
        g.nl_sorted = sorted ? 1 : 0;
    }
    return M3D_OK;
}

// serial-order sum of the nearest squared distances below r^2 (GetRegistrationResult... error2)
int exact_err2(RegCtx& rc, const double* T_dev, uint64_t* count, double* err2) {
    DeviceCtx* ctx = rc.ctx;
    Scratch& s = *rc.s;
    const uint32_t n = rc.src.n;
    const uint32_t nb = (n + 2047) / 2048;
    RESERVE(s.best, sizeof(double) * std::max<uint32_t>(n, 1));
    RESERVE(s.vals, sizeof(double) * std::max<uint32_t>(n, 1));
    RESERVE(s.block_counts, sizeof(uint32_t) * ((size_t)nb + 1));
    RESERVE(s.total, 16);
    RESERVE(s.sums, sizeof(double) * 4);
    RESERVE(ctx->h_small, 256);
    launch_reg_min_d2(rc.src, T_dev, rc.g, s.cell_start.as<uint32_t>(), s.qx.as<double>(), s.qy.as<double>(),
                      s.qz.as<double>(), s.best.as<double>(), ctx->stream);
    launch_compact_vals(s.best.as<double>(), n, rc.g.r2, s.block_counts.as<uint32_t>(), s.total.as<uint32_t>(),
                        s.vals.as<double>(), ctx->stream);
    launch_serial_sum(s.vals.as<double>(), s.total.as<uint32_t>(), s.sums.as<double>(), ctx->stream);
    uint8_t* h = ctx->h_small.as<uint8_t>();
    HIPCHK(hipMemcpyAsync(h, s.total.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(h + 8, s.sums.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipGetLastError());
    if (const int wr = stream_wait_spin(ctx); wr != M3D_OK) return wr;   // (results in page-locked memory: a polled word instead of the runtime's wait, which wakes 10-20 us late)
    uint32_t c;
    std::memcpy(&c, h, 4);
    std::memcpy(err2, h + 8, 8);
    *count = c;
    return M3D_OK;
}

// Same quantity with a FIXED-shape tree sum over the points in their original order: deterministic from run
// to run (unlike the sums of the validation kernel, whose source copy is sorted with atomics) and ~20x cheaper
// than the serial chain.  Used for the reported inlier_rmse when no fitness tie needed the serial value.
int tree_err2(RegCtx& rc, const double* T_dev, uint64_t* count, double* err2) {
    DeviceCtx* ctx = rc.ctx;
    Scratch& s = *rc.s;
    const uint32_t n = rc.src.n;
    RESERVE(s.best, sizeof(double) * std::max<uint32_t>(n, 1));
    RESERVE(s.sums, sizeof(double) * 32);
    RESERVE(s.partial_sum, sizeof(double) * 256 * 16);
    RESERVE(ctx->h_small, 256);
    launch_reg_min_d2(rc.src, T_dev, rc.g, s.cell_start.as<uint32_t>(), s.qx.as<double>(), s.qy.as<double>(),
                      s.qz.as<double>(), s.best.as<double>(), ctx->stream);
    launch_icp_err(s.best.as<double>(), n, rc.g.r2, s.partial_sum.as<double>(), s.sums.as<double>() + 24, ctx->stream);
    uint8_t* h = ctx->h_small.as<uint8_t>();
    HIPCHK(hipMemcpyAsync(h, s.sums.as<double>() + 24, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipGetLastError());
    if (const int wr = stream_wait_spin(ctx); wr != M3D_OK) return wr;   // (results in page-locked memory: a polled word instead of the runtime's wait, which wakes 10-20 us late)
    double es[2];
    std::memcpy(es, h, 16);
    *err2 = es[0];
    *count = (uint64_t)es[1];
    return M3D_OK;
}

int corr_inlier_ratio(RegCtx& rc, const double* T_dev, double* ratio) {
    DeviceCtx* ctx = rc.ctx;
    Scratch& s = *rc.s;
    RESERVE(s.ratio, 16);
    RESERVE(ctx->h_small, 256);
    launch_corr_ratio(rc.src, rc.dst, s.corr_src.as<uint32_t>(), s.corr_dst.as<uint32_t>(), rc.m, T_dev,
                      rc.g.r2, s.ratio.as<uint32_t>(), ctx->stream);
    HIPCHK(hipMemcpyAsync(ctx->h_small.p, s.ratio.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipGetLastError());
    if (const int wr = stream_wait_spin(ctx); wr != M3D_OK) return wr;   // (results in page-locked memory: a polled word instead of the runtime's wait, which wakes 10-20 us late)
    uint32_t c;
    std::memcpy(&c, ctx->h_small.p, 4);
    *ratio = (double)c / (double)rc.m;
    return M3D_OK;
}

}  // namespace

extern "C" {

int m3d_kabsch(const double* src, const double* dst, size_t n, int scaling, int device, double* T) {
    if (!T || ((!src || !dst) && n)) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (n < 3)  // CheckValid, transform_estimation.cpp:27-31
        return fail(M3D_ERR_TOO_FEW_POINTS, "The number of points pair is less than 3.");
    if (n >= ((size_t)1 << 31)) return fail(M3D_ERR_INVALID_ARG, "too many points");
    LaneLock lane(device);
    DeviceCtx* ctx = lane.ctx;
    if (!ctx) return M3D_ERR_DEVICE;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf ds, dd, partial, sums;
    int rc = M3D_OK;
    auto done = [&](int r) {
        ds.release(); dd.release(); partial.release(); sums.release();
        return r;
    };
    if (!ds.reserve(sizeof(double) * 3 * n) || !dd.reserve(sizeof(double) * 3 * n) ||
        !partial.reserve(sizeof(double) * 256 * 16) || !sums.reserve(sizeof(double) * 32) ||
        !ctx->h_small.reserve(256))
        return done(M3D_ERR_DEVICE);
    bool ok = hipMemcpyAsync(ds.p, src, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
              hipMemcpyAsync(dd.p, dst, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    if (ok) {
        launch_kabsch_sums(ds.as<double>(), dd.as<double>(), (uint32_t)n, partial.as<double>(), sums.as<double>(),
                           ctx->stream);
        ok = hipMemcpyAsync(ctx->h_small.p, sums.p, sizeof(double) * 18, hipMemcpyDeviceToHost, ctx->stream) ==
                 hipSuccess &&
             hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!ok) return done(fail(M3D_ERR_DEVICE, "m3d_kabsch: HIP error"));
    double h[18];
    std::memcpy(h, ctx->h_small.p, sizeof(h));
    // Eigen::umeyama: means, sigma = (1/n) sum d s^T, src_var = (1/n) sum |s|^2
    const double one_over_n = 1.0 / (double)n;
    double ms[3], md[3], sig[9];
    for (int k = 0; k < 3; ++k) {
        ms[k] = h[k] * one_over_n;
        md[k] = h[3 + k] * one_over_n;
    }
    for (int k = 0; k < 9; ++k) sig[k] = h[6 + k] * one_over_n;
    const double src_var = ((h[15] + h[16]) + h[17]) * one_over_n;
    umeyama_assemble(ms, md, sig, src_var, scaling != 0, T);
    return done(rc);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Registration session: the Open3D RANSAC loop cut into the steps a multi-GPU driver needs
// (hypotheses sharded, clouds and grid replicated):
//   begin_chunk   draw the triples of the next chunk of iterations (the generator lives in the session, so
//                 identically seeded ranks draw identical triples), 3-point Kabsch + checkers for all of
//                 them, survivor list + their transformations on the device
//   validate      count + sum of squared nearest distances for survivors [s_begin, s_end) -- the shard
//   replay        the sequential best-update / est_k rule over the whole chunk, given every survivor's
//                 (count, sum): identical inputs on every rank -> identical state on every rank
// m3d_registration_ransac is the single-GPU loop over the same three steps.
// ------------------------------------------------------------------------------------------------
struct m3d_reg {
    DeviceCtx* ctx = nullptr;
    m3d_cloud *csrc = nullptr, *cdst = nullptr;
    bool own_clouds = true;   // false: the caller's resident clouds (global_registration_on), left alone by the destructor
    bool keep_orig = false;   // the target grid keeps its points' original indices: information_matrix_on then searches THIS grid (global_registration_on)
    Scratch S;
    RegCtx R;
    GridDesc g;
    CloudView src_sorted;
    size_t n_src = 0, n_dst = 0, m = 0;
    double threshold = 0, edge_length_threshold = 0, confidence = 0;
    int max_iter = 0;
    bool trivial = false;   // Open3D returns the default RegistrationResult without looping
    std::mt19937 rng;
    std::uniform_int_distribution<int> pick;
    double best_fit = 0, best_rmse = 0;
    uint32_t best_cnt = 0;   // inlier count behind best_fit
    bool reg_prune = true;
    bool best_rmse_known = true;
    int64_t best_index = -1;
    int est_k_global = 0, est_k_local = 0;
    uint64_t total_validation = 0, ties = 0, exact_evals = 0;
    int64_t iters = 0;
    double* best_T_dev = nullptr;
    double best_T_host[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    uint32_t n_tiles = 0;
    std::vector<uint32_t> tri, survivors, h_counts;
    std::vector<double> h_sum2;  // order-free sums of the nearest squared distances per survivor
    double best_sum2 = 0.0;      // the same for the current best
    std::vector<uint8_t> pass;
    // Triples per chunk: starts small and doubles.  A block of reg_validate_k works through its (<= 64) survivors one
    // after the other at ~45 us each (dependent neighbour-list gathers), and with the reference's default confidence
    // (0.999) the loop usually ends after two or three dozen iterations: a first chunk of 256 triples validated ~65
    // survivors (3 ms) of which the replay used 5.
    size_t chunk = 32;
    size_t validated_total = 0, n_dst_points = 0;
    bool nl_built = false;
    // the candidate cache (m3d_reg_cache.hip): built for the incumbent of some earlier chunk, rebuilt when the incumbent has moved on
    RegCache cache;
    bool cache_valid = false;
    int64_t cache_ref_index = -1;
    uint32_t cache_ref_cnt = 0, cache_builds = 0;
    double cache_ref_sum2 = 0.0;
    int ensure_cache(uint32_t s_pad);
    int itr = 0;
    int n_exec = 0;          // iterations of the chunk in flight
    bool finished = false;
    double t_begin = 0;

    int setup(const double* src, const double* dst, const size_t* corr_src, const size_t* corr_dst,
              const uint64_t* seed);
    int begin_chunk(size_t* n_survivors);   // M3D_OK with *n_survivors set, or M3D_FALSE when the loop is over
    int validate(size_t s_begin, size_t s_end, uint32_t* counts_out, double* sums_out);
    int replay(const uint32_t* counts_in, const double* sums_in);
    int finish(double* T_out, m3d_reg_stats* stats);
};


int m3d_reg::setup(const double* src, const double* dst, const size_t* corr_src, const size_t* corr_dst,
                   const uint64_t* seed) {

    HIPCHK(hipSetDevice(ctx->device));
    R.ctx = ctx;
    R.src = this->csrc->view();
    // the count kernel reads whole tiles of kRegTile points: the cloud padding (kScoreTile) covers it
    static_assert(kScoreTile % kRegTile == 0, "padding of resident clouds must cover the reg tiles");
    R.dst = this->cdst->view();
    R.s = &S;
    R.m = (uint32_t)m;
    R.thr = threshold;

    {
        // cells per radius of the validation grid: m3d_config.reg_cells_per_radius (4) is the figure for a 200 000-point target; a
        // smaller one gets coarser cells in proportion to the cube root (about the same points per cell: a 50 000-point target
        // builds its grid and is searched in 0.64 ms per default-confidence call with 3 cells per radius, 0.77 with 4; 200 000
        // points: 1.33 with 4, 1.51 with 3, 2.09 with 2)
        const int k_cfg = config().reg_cells_per_radius;
        const int k0 = std::max(1, std::min(k_cfg, (int)std::lround(k_cfg * std::cbrt((double)n_dst / 200000.0))));
        const int rc_grid = build_target_grid(ctx, S, R.dst, dst, n_dst, threshold, keep_orig, &g, k0, /*with_nl=*/false,
                                              cdst->bb_known ? cdst->bb : nullptr);
        if (rc_grid != M3D_OK) return rc_grid;
        R.g = g;
        n_dst_points = n_dst;
    }

    // ---- spatially sorted copy of the SOURCE cloud for the validation kernel: counts and
    // order-free sums do not depend on the point order, and lanes of a wave that hold
    // neighbouring source points probe the same target cells (L1/L2 hits instead of scattered
    // gathers).  Same counting sort over the source bounding box.  Points with
    // non-finite coordinates drop out (they can never have a neighbour).
    src_sorted = R.src;
    {
        double slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (csrc->bb_known) {   // (the box of the points with three finite coordinates: the others are no queries)
            for (int k = 0; k < 3; ++k) {
                slo[k] = csrc->bb[k];
                shi[k] = csrc->bb[3 + k];
            }
        } else {
            for (size_t i = 0; i < n_src; ++i)
                for (int k = 0; k < 3; ++k) {
                    const double v = src[3 * i + k];
                    if (std::isfinite(v)) {
                        slo[k] = std::min(slo[k], v);
                        shi[k] = std::max(shi[k], v);
                    }
                }
        }
        double ext = 0.0;
        for (int k = 0; k < 3; ++k) {
            if (!(slo[k] <= shi[k])) slo[k] = shi[k] = 0.0;
            ext = std::max(ext, shi[k] - slo[k]);
        }
        if (ext > 0.0 && std::isfinite(ext)) {
            GridDesc gs;
            // Hilbert order over 128^3 cells: the 64 points of a wave form a compact patch, so the lanes probe the same few
            // target cells
            // (round 5: 64^3 / 128^3 / 256^3 / 512^3 cells: the forced C4 run 142.3 / 138.5 / 136.8 / 139.0 ms -- and with the source
            //  pre-aligned to the target's axes, so that a wave's points share target cells, 135.2: the number of DISTINCT cells
            //  among a wave's lanes is not what the validation kernel's time goes with; profiles/r05_reg_validate_findings.txt)
            const uint32_t fbits = 7u;
            const double hs = ext / 127.0;
            gs.K = 0;
            gs.morton_bits = fbits | 0x100u;
            gs.ox = slo[0];
            gs.oy = slo[1];
            gs.oz = slo[2];
            gs.inv_h = 1.0 / hs;
            gs.r2 = gs.h2_in = 0.0;
            const uint32_t side = 1u << fbits;
            gs.nx = gs.ny = gs.nz = side;
            const uint32_t ncs = gs.nx * gs.ny * gs.nz;
            RESERVE(S.s_cell_of_point, sizeof(uint32_t) * n_src);
            RESERVE(S.s_cell_start, sizeof(uint32_t) * ((size_t)ncs + 1));
            RESERVE(S.s_fill, sizeof(uint32_t) * std::max<size_t>(R.src.n, 1));   // rank of every point in its cell
            RESERVE(S.s_tile_sums, sizeof(uint32_t) * ((size_t)(ncs + 2047) / 2048 + 1));
            launch_grid_count_scan(R.src, gs, S.s_cell_of_point.as<uint32_t>(), S.s_cell_start.as<uint32_t>(),
                                   S.s_fill.as<uint32_t>(), S.s_tile_sums.as<uint32_t>(), S.total.as<uint32_t>() + 2,
                                   ctx->stream);
            const uint32_t np = R.src.n_pad;
            RESERVE(S.sx, sizeof(double) * np);
            RESERVE(S.sy, sizeof(double) * np);
            RESERVE(S.sz, sizeof(double) * np);
            launch_fill_nan(S.sx.as<double>(), np, ctx->stream);
            launch_fill_nan(S.sy.as<double>(), np, ctx->stream);
            launch_fill_nan(S.sz.as<double>(), np, ctx->stream);
            launch_grid_scatter(R.src, S.s_cell_of_point.as<uint32_t>(), S.s_cell_start.as<uint32_t>(),
                                S.s_fill.as<uint32_t>(), S.sx.as<double>(), S.sy.as<double>(), S.sz.as<double>(), ctx->stream);
            src_sorted.x = S.sx.as<double>();
            src_sorted.y = S.sy.as<double>();
            src_sorted.z = S.sz.as<double>();
            src_sorted.n_pad = np;   // (NaN slots are no queries: counts and sums only see the real points)
            src_sorted.n = np;
        }
    }

    // ---- correspondences (CorrespondenceSet of Vector2i, transform_estimation.cpp:135-140)
    std::vector<uint32_t> cs(m), cd(m);
    for (size_t i = 0; i < m; ++i) {
        cs[i] = (uint32_t)corr_src[i];
        cd[i] = (uint32_t)corr_dst[i];
    }
    RESERVE(S.corr_src, sizeof(uint32_t) * m);
    RESERVE(S.corr_dst, sizeof(uint32_t) * m);
    HIPCHK(hipMemcpyAsync(S.corr_src.p, cs.data(), sizeof(uint32_t) * m, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(S.corr_dst.p, cd.data(), sizeof(uint32_t) * m, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));

    // ---- RANSAC state
    std::random_device rd;
    this->rng = std::mt19937((std::mt19937::result_type)((seed ? *seed : (uint64_t)rd()) & 0xffffffffull));
    this->pick = std::uniform_int_distribution<int>(0, (int)m - 1);  // utility::UniformRandIntGenerator(0, M-1)
    this->reg_prune = config().reg_prune != 0;
    this->est_k_global = this->est_k_local = max_iter;
    RESERVE(S.one_T, sizeof(double) * kRegTStride * 2);
    this->best_T_dev = S.one_T.as<double>();
    this->n_tiles = src_sorted.n_pad / kRegTile;
    return M3D_OK;
}

// returns M3D_OK with *n_survivors set, or M3D_FALSE when the loop is over
int m3d_reg::begin_chunk(size_t* n_survivors) {
    *n_survivors = 0;
    if (this->trivial || this->finished || !(itr < max_iter && itr < est_k_global)) {
        this->finished = true;
        return M3D_FALSE;
    }
    HIPCHK(hipSetDevice(ctx->device));

    // iterations [itr, itr + n_exec) all satisfy itr < est_k_global as of now; est_k can only
    // shrink while replaying, in which case the tail of the chunk is discarded (its draws
    // would not have happened: the generator is rewound by re-drawing from a saved state)
    n_exec = (int)std::min<size_t>(chunk, (size_t)(std::min(max_iter, est_k_global) - itr));
    tri.resize((size_t)n_exec * 3);
    for (int k = 0; k < n_exec * 3; ++k) tri[k] = (uint32_t)pick(rng);
    RESERVE(S.triples, sizeof(uint32_t) * 3 * (size_t)n_exec);
    RESERVE(S.T12, sizeof(double) * kRegTStride * (size_t)n_exec);
    RESERVE(S.pass, (size_t)n_exec);
    HIPCHK(hipMemcpyAsync(S.triples.p, tri.data(), sizeof(uint32_t) * 3 * (size_t)n_exec,
                          hipMemcpyHostToDevice, ctx->stream));
    launch_kabsch3_check(R.src, R.dst, S.corr_src.as<uint32_t>(), S.corr_dst.as<uint32_t>(),
                         S.triples.as<uint32_t>(), (uint32_t)n_exec, edge_length_threshold, threshold,
                         S.T12.as<double>(), S.pass.as<uint8_t>(), ctx->stream);
    pass.resize(n_exec);
    // (through page-locked memory and a polled word: a copy into the vector's pageable pages is staged, and the runtime's wait wakes
    //  the caller 10-20 us after the stream has drained -- twice per chunk of a call that is 1.3 ms long)
    RESERVE(ctx->h_reg, (size_t)std::max(n_exec, 1));
    HIPCHK(hipMemcpyAsync(ctx->h_reg.p, S.pass.p, (size_t)n_exec, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipGetLastError());
    if (const int wr = stream_wait_spin(ctx); wr != M3D_OK) return wr;
    std::memcpy(pass.data(), ctx->h_reg.p, (size_t)n_exec);
    survivors.clear();
    for (int k = 0; k < n_exec; ++k)
        if (pass[k]) survivors.push_back((uint32_t)k);
    const uint32_t ns = (uint32_t)survivors.size();
    h_counts.assign(ns, 0);
    // neighbour lists once the call has turned out to validate more than a handful of hypotheses (with the
    // reference's default confidence it usually has not: ~8 validations, against 0.4 ms to build the lists)
    validated_total += ns;
    if (!nl_built && validated_total > 48) {
        nl_built = true;
        const int rn = add_neighbour_lists(ctx, S, &g, n_dst_points, keep_orig ? S.cell_orig.as<uint32_t>() : nullptr);
        if (rn != M3D_OK) return rn;
        R.g = g;
    }

    if (ns) {
        const uint32_t s_pad = round_up(ns, 64);
        RESERVE(S.list, sizeof(uint32_t) * ns);
        RESERVE(S.Ts, sizeof(double) * kRegTStride * ((size_t)s_pad + 1));
        RESERVE(S.partial, sizeof(uint32_t) * (size_t)n_tiles * kRegValidateRows * s_pad);
        RESERVE(S.partial_sum, sizeof(double) * (size_t)n_tiles * kRegValidateRows * s_pad);
        RESERVE(S.sum2, sizeof(double) * s_pad);
        RESERVE(S.counts, sizeof(uint32_t) * s_pad);
        HIPCHK(hipMemcpyAsync(S.list.p, survivors.data(), sizeof(uint32_t) * ns, hipMemcpyHostToDevice,
                              ctx->stream));
        launch_gather_T(S.T12.as<double>(), S.list.as<uint32_t>(), ns, s_pad + 1, S.Ts.as<double>(),
                        ctx->stream);
    }
    *n_survivors = ns;
    return M3D_OK;
}

// The candidate cache for the incumbent's pose (best_T_dev), (re)built when there is none yet or the incumbent has improved
// by more than a little since the last build (the hypotheses worth a full evaluation are near-copies of the best pose; an
// early incumbent 30 % worse in rmse certifies almost as many pairs as the final one: profiles/r06_reg_cache.txt).
int m3d_reg::ensure_cache(uint32_t s_pad) {
    const size_t np = src_sorted.n_pad, nt = np / kRegTile;
    const bool moved = cache_valid && best_index != cache_ref_index &&
                       ((double)best_cnt > 1.02 * (double)cache_ref_cnt || best_sum2 < 0.7 * cache_ref_sum2);
    RESERVE(S.cc_redo, nt * (size_t)s_pad);
    cache.redo = S.cc_redo.as<uint8_t>();
    RESERVE(S.cc_pairs, sizeof(uint32_t) * (nt * (size_t)s_pad + 4));
    cache.n_pairs = S.cc_pairs.as<uint32_t>();
    cache.pairs = cache.n_pairs + 4;
    if (cache_valid && !moved) return M3D_OK;
    constexpr size_t slots = (size_t)kRegCacheK * kRegCacheTiers;
    RESERVE(S.cc_x, sizeof(float2) * nt * (slots / 2) * 256);
    RESERVE(S.cc_y, sizeof(float2) * nt * (slots / 2) * 256);
    RESERVE(S.cc_z, sizeof(float2) * nt * (slots / 2) * 256);
    RESERVE(S.cc_64, sizeof(double4) * nt * slots * 256);
    RESERVE(S.cc_xa, sizeof(double) * np);
    RESERVE(S.cc_ya, sizeof(double) * np);
    RESERVE(S.cc_za, sizeof(double) * np);
    RESERVE(S.cc_R, sizeof(float) * np * kRegCacheTiers);
    if (!S.cc_stats.p) {
        RESERVE(S.cc_stats, sizeof(unsigned long long) * 4);
        HIPCHK(hipMemsetAsync(S.cc_stats.p, 0, sizeof(unsigned long long) * 4, ctx->stream));
    }
    cache.cx = S.cc_x.as<float2>();
    cache.cy = S.cc_y.as<float2>();
    cache.cz = S.cc_z.as<float2>();
    cache.c64 = S.cc_64.as<double4>();
    cache.xa = S.cc_xa.as<double>();
    cache.ya = S.cc_ya.as<double>();
    cache.za = S.cc_za.as<double>();
    cache.R = S.cc_R.as<float>();
    cache.stats = S.cc_stats.as<unsigned long long>();
    if (!cache.ring) {   // the target grid's cell rings, once per session (K + 1 rounds: a ring beyond K certifies "nothing in reach")
        const size_t ncell = (size_t)g.nx * g.ny * g.nz;
        RESERVE(S.cc_ring, ncell);
        RESERVE(S.cc_ring_tmp, 2 * ncell);
        launch_reg_rings(g, S.cell_start.as<uint32_t>(), S.cc_ring.as<uint8_t>(), S.cc_ring_tmp.as<uint8_t>(), g.K + 1, ctx->stream);
        S.cc_ring_tmp.release();
        cache.ring = S.cc_ring.as<uint8_t>();
        g.ring = cache.ring;     // the walk's outer search looks them up as well (nearest_d2)
        R.g = g;
    }
    launch_reg_cache_build(src_sorted, best_T_dev, g, S.cell_start.as<uint32_t>(), S.qx.as<double>(), S.qy.as<double>(),
                           S.qz.as<double>(), cache, ctx->stream);
    cache_valid = true;
    cache_ref_index = best_index;
    cache_ref_cnt = best_cnt;
    cache_ref_sum2 = best_sum2;
    cache_builds++;
    return M3D_OK;
}

// (count, sum of squared nearest distances) of survivors [s_begin, s_end) of the chunk in flight;
// s_begin must be a multiple of 64 (the kernel works on groups of 64 hypotheses)
int m3d_reg::validate(size_t s_begin, size_t s_end, uint32_t* counts_out, double* sums_out) {
    const uint32_t ns_all = (uint32_t)survivors.size();
    if (s_begin == s_end) return M3D_OK;
    if (s_begin > s_end || s_end > ns_all || (s_begin % 64) != 0)
        return fail(M3D_ERR_INVALID_ARG, "m3d_reg_validate: bad survivor range");
    const uint32_t ns = (uint32_t)(s_end - s_begin);
    HIPCHK(hipSetDevice(ctx->device));
    {
        const uint32_t s_pad = round_up(ns, 64);
        const double* Ts = S.Ts.as<double>() + s_begin * kRegTStride;   // records behind the shard are real or NaN padding
        RESERVE(S.partial, sizeof(uint32_t) * (size_t)n_tiles * kRegValidateRows * s_pad);
        RESERVE(S.partial_sum, sizeof(double) * (size_t)n_tiles * kRegValidateRows * s_pad);
        RESERVE(S.sum2, sizeof(double) * s_pad);
        RESERVE(S.counts, sizeof(uint32_t) * s_pad);

        RESERVE(S.keep, s_pad);
        // the candidate cache: once the call has turned out long (the neighbour lists are built), has an incumbent, and the shard
        // is large enough to fill whole 64-hypothesis groups per workgroup (m3d_config.reg_cache: 0 off, 1 auto, 2 whenever an
        // incumbent exists -- the tests' switch)
        const int cc = config().reg_cache;
        // (a cell edge within [1e-12, 1e12]: the cache's fp32 squares neither underflow below its rounding bound nor reach the
        //  empty slots' 1e18 -- m3d_reg_cache.hip)
        const double h_cell = 1.0 / g.inv_h;
        const bool use_cache = best_index >= 0 && src_sorted.n_pad <= ((size_t)2 << 20) && h_cell > 1e-12 && h_cell < 1e12 &&
                               (cc == 2 || (cc == 1 && nl_built && ns >= 192));
        if (use_cache) {
            const int rcache = ensure_cache(s_pad);
            if (rcache != M3D_OK) return rcache;
        }
        // bound-and-prune against the best of EARLIER chunks (m3d_config.reg_prune = 0 switches it off)
        const uint32_t rows = launch_reg_validate(src_sorted, Ts, s_pad, g, S.cell_start.as<uint32_t>(),
                            S.qx.as<double>(), S.qy.as<double>(), S.qz.as<double>(),
                            S.partial.as<uint32_t>(), S.partial_sum.as<double>(), S.sum2.as<double>(),
                            reg_prune ? best_cnt : 0u, (uint32_t)n_src, S.keep.as<uint8_t>(), ctx->stream, best_sum2, ns,
                            use_cache ? &cache : nullptr);
        HIPCHK(hipMemsetAsync(S.counts.p, 0, sizeof(uint32_t) * s_pad, ctx->stream));
        launch_reduce_partials(S.partial.as<uint32_t>(), rows, s_pad, S.counts.as<uint32_t>(),
                               ctx->stream);
        const size_t sums_at = ((sizeof(uint32_t) * ns + 7) / 8) * 8;
        RESERVE(ctx->h_reg, sums_at + sizeof(double) * ns);
        uint8_t* hr = ctx->h_reg.as<uint8_t>();
        HIPCHK(hipMemcpyAsync(hr, S.counts.p, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(hr + sums_at, S.sum2.p, sizeof(double) * ns, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipGetLastError());
        if (const int wr = stream_wait_spin(ctx); wr != M3D_OK) return wr;
        std::memcpy(counts_out, hr, sizeof(uint32_t) * ns);
        std::memcpy(sums_out, hr + sums_at, sizeof(double) * ns);
    }
    return M3D_OK;
}

// sequential replay of the chunk in flight; counts / sums: one entry per survivor, in survivor order
int m3d_reg::replay(const uint32_t* counts_in, const double* sums_in) {
    HIPCHK(hipSetDevice(ctx->device));
    h_counts.assign(counts_in, counts_in + survivors.size());
    h_sum2.assign(sums_in, sums_in + survivors.size());

    // ---- sequential replay of the chunk
    uint32_t sv = 0;
    int k = 0;
    for (; k < n_exec; ++k) {
        const int it = itr + k;
        if (!(it < est_k_global)) break;  // `if (itr < est_k_global)` of the Open3D loop
        iters++;
        if (!pass[k]) continue;
        const uint32_t cnt = h_counts[sv];
        const double a_t = cnt ? h_sum2[sv] : 0.0;
        const double* T_dev = S.Ts.as<double>() + (size_t)sv * kRegTStride;
        sv++;
        const double fit = cnt ? (double)cnt / (double)n_src : 0.0;
        bool better = fit > best_fit;
        double rmse = 0.0;
        bool rmse_known = cnt == 0;  // empty correspondence set: rmse = 0
        if (!better && fit == best_fit && cnt == 0) {
            better = false;  // 0 < best_rmse never holds (both are the empty result)
        } else if (!better && fit == best_fit) {
            // IsBetterRANSACThan on equal fitness: rmse = sqrt(err2 / n) with the same n on both
            // sides, monotone in err2.  Order-free sums decide unless they are closer than the
            // summation-order bound 2 n u sum (x2 safety); then the serial-order sums decide.
            ties++;
            const double nu4 = 4.0 * (double)cnt * 1.1102230246251565e-16;
            if (a_t + nu4 * a_t < best_sum2 - nu4 * best_sum2) {
                better = true;
            } else if (a_t - nu4 * a_t > best_sum2 + nu4 * best_sum2) {
                better = false;
            } else {
                uint64_t c2;
                double e2;
                int r = exact_err2(R, T_dev, &c2, &e2);
                if (r != M3D_OK) return r;
                if (c2 != cnt) return fail(M3D_ERR_INTERNAL, "validation count mismatch");
                rmse = std::sqrt(e2 / (double)c2);
                rmse_known = true;
                exact_evals++;
                if (!best_rmse_known) {
                    r = exact_err2(R, best_T_dev, &c2, &e2);
                    if (r != M3D_OK) return r;
                    best_rmse = c2 ? std::sqrt(e2 / (double)c2) : 0.0;
                    best_rmse_known = true;
                    exact_evals++;
                }
                better = rmse < best_rmse;
            }
        }
        if (better) {
            best_fit = fit;
            best_cnt = cnt;
            best_rmse = rmse;
            best_rmse_known = rmse_known;
            best_sum2 = a_t;
            best_index = it;
            HIPCHK(hipMemcpyAsync(best_T_dev, T_dev, sizeof(double) * kRegTStride,
                                  hipMemcpyDeviceToDevice, ctx->stream));
            double ratio;
            const int r = corr_inlier_ratio(R, best_T_dev, &ratio);
            if (r != M3D_OK) return r;
            const double est_d = std::log(1.0 - confidence) / std::log(1.0 - std::pow(ratio, 3.0));
            est_k_local = est_d < (double)est_k_global ? ceil_to_int_x86(est_d) : est_k_local;
        }
        total_validation++;
        if (est_k_local < est_k_global) est_k_global = est_k_local;
    }
    if (k < n_exec) {
        // the loop went idle inside the chunk: nothing after it draws or runs
        this->finished = true;
        return M3D_OK;
    }
    itr += n_exec;
    chunk = std::min<size_t>(chunk * 2, 16384);
    return M3D_OK;
}

int m3d_reg::finish(double* T_out, m3d_reg_stats* stats) {
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(T_out, I4, sizeof(I4));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->best_index = -1;
    }
    if (this->trivial) return M3D_OK;
    HIPCHK(hipSetDevice(ctx->device));

    if (best_index >= 0) {
        HIPCHK(hipMemcpy(best_T_host, best_T_dev, sizeof(best_T_host), hipMemcpyDeviceToHost));
        // the reported inlier_rmse: a deterministic tree sum over the points in their original order (Open3D's own
        // value depends on its OpenMP reduction order; the serial-order sum is only formed when a fitness tie
        // needs it, and then best_rmse is that value)
        if (!best_rmse_known && stats) {   // (stats == NULL: the caller wants the pose only -- RANSACSolver::Solve's own return)
            uint64_t c2;
            double e2;
            const int r = tree_err2(R, best_T_dev, &c2, &e2);
            if (r != M3D_OK) return r;
            if (c2 != best_cnt) return fail(M3D_ERR_INTERNAL, "validation count mismatch");
            best_rmse = c2 ? std::sqrt(e2 / (double)c2) : 0.0;
        }
        std::memcpy(T_out, best_T_host, sizeof(best_T_host));
    }
    if (stats) {
        stats->fitness = best_fit;
        stats->inlier_rmse = best_rmse;
        stats->validations = total_validation;
        stats->iterations = iters;
        stats->best_index = best_index;
        stats->est_k = est_k_global;
        stats->ties = ties;
        stats->exact_rmse_evals = exact_evals;
        stats->lds_wave_hypotheses = stats->global_wave_hypotheses = 0;
        if (cache.stats) {   // (tile, hypothesis) pairs the candidate cache answered / handed to the list walk
            unsigned long long cs2[2] = {0, 0};
            HIPCHK(hipMemcpy(cs2, cache.stats, sizeof(cs2), hipMemcpyDeviceToHost));
            stats->lds_wave_hypotheses = cs2[0];
            stats->global_wave_hypotheses = cs2[1];
        }
        stats->nn_fp32_screen = g.nl32 ? 1 : 0;
        stats->nn_screen_fallbacks = 0;
        if (g.nl32_fallbacks) {
            unsigned long long fb = 0;
            HIPCHK(hipMemcpy(&fb, g.nl32_fallbacks, sizeof(fb), hipMemcpyDeviceToHost));
            stats->nn_screen_fallbacks = fb;
#ifdef M3D_REG_TRIP_STATS
            {   // the walk's trip counts (sorted_walk32): lane trips, wave maxima, histogram of the lanes' counts
                unsigned long long t[64];
                HIPCHK(hipMemcpy(t, g.nl32_fallbacks, sizeof(t), hipMemcpyDeviceToHost));
                fprintf(stderr, "walk trips: queries %llu lane-trips %llu (mean %.2f) wave-queries %llu sum of wave maxima %llu (mean %.2f); lanes by trips:",
                        t[8], t[9], (double)t[9] / (double)(t[8] ? t[8] : 1), t[10], t[11], (double)t[11] / (double)(t[10] ? t[10] : 1));
                for (int k = 0; k < 24; ++k) fprintf(stderr, " %d:%.3f", k, (double)t[16 + k] / (double)(t[8] ? t[8] : 1));
                fprintf(stderr, "\n");
                fprintf(stderr, "query fates: in grid %llu, outside the grid %llu; empty 3x3x3 list %llu, second phase entered %llu (matched there %llu), "
                                "matched in the end %llu; wave-queries %llu (lanes %llu): no lane matched %llu, some lane in the second phase %llu, "
                                "no lane with a list %llu\n",
                        t[40], t[41], t[42], t[43], t[44], t[45], t[46], t[50], t[47], t[48], t[49]);
            }
#endif
        }
    }
    if (stats) stats->ms_total = now_ms() - this->t_begin;
    return M3D_OK;
}

namespace {

// The session's argument checks, in front of any device work: < 0 = error (the reference throws), M3D_FALSE = the trivial
// session (Open3D returns the default RegistrationResult without looping: no device needed), M3D_OK = a real one.
int reg_check_args(const double* src, size_t n_src, const double* dst, size_t n_dst, const size_t* corr_src,
                   const size_t* corr_dst, size_t m, double threshold) {
    if (((!src && n_src) || (!dst && n_dst)) || (m && (!corr_src || !corr_dst)))
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (n_src < 3 || n_dst < 3)  // transform_estimation.cpp:130-133
        return fail(M3D_ERR_TOO_FEW_POINTS, "The number of points pair is less than 3.");
    if (n_src >= ((size_t)1 << 31) || n_dst >= ((size_t)1 << 31) || m >= ((size_t)1 << 31))
        return fail(M3D_ERR_INVALID_ARG, "too many points");
    for (size_t i = 0; i < m; ++i)
        if (corr_src[i] >= n_src || corr_dst[i] >= n_dst)
            return fail(M3D_ERR_INVALID_ARG, "correspondence index out of range");
    // Open3D: ransac_n < 3 || corres.size() < ransac_n || max_correspondence_distance <= 0 -> RegistrationResult()
    return (m < 3 || !(threshold > 0.0)) ? M3D_FALSE : M3D_OK;
}

// Caller holds the lane (no locks below; ctx may be null for a trivial session).  argument checks + uploads + grid; *rc_out < 0 on error (session NULL),
// M3D_OK otherwise.  csrc_in / cdst_in: the clouds already resident on this lane (borrowed), or null: uploaded here.
void reg_destroy_on(m3d_reg* q);
m3d_reg* reg_create_on(DeviceCtx* ctx, m3d_cloud* csrc_in, m3d_cloud* cdst_in, const double* src, size_t n_src,
                       const double* dst, size_t n_dst, const size_t* corr_src, const size_t* corr_dst, size_t m,
                       double threshold, int max_iter, double edge_length_threshold, double confidence,
                       const uint64_t* seed, int* rc_out, bool keep_orig = false) {
    *rc_out = M3D_OK;
    auto bad = [&](int code) -> m3d_reg* {
        *rc_out = code;
        return nullptr;
    };
    const int chk = reg_check_args(src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold);
    if (chk < 0) return bad(chk);
    if (chk == M3D_OK && !ctx) return bad(fail(M3D_ERR_DEVICE, "no lane"));
    m3d_reg* q = new m3d_reg();
    q->t_begin = now_ms();
    q->n_src = n_src;
    q->n_dst = n_dst;
    q->m = m;
    q->threshold = threshold;
    q->edge_length_threshold = edge_length_threshold;
    q->confidence = confidence;
    q->max_iter = max_iter;
    if (chk == M3D_FALSE) {
        q->trivial = true;
        return q;
    }
    if (csrc_in && cdst_in) {
        q->csrc = csrc_in;
        q->cdst = cdst_in;
        q->own_clouds = false;
    } else {
        q->csrc = m3d_cloud_create_on(ctx, src, nullptr, n_src, 0);
        q->cdst = q->csrc ? m3d_cloud_create_on(ctx, dst, nullptr, n_dst, 0) : nullptr;
        if (!q->csrc || !q->cdst) {
            if (q->csrc) m3d_cloud_destroy_on(q->csrc);
            delete q;
            return bad(M3D_ERR_DEVICE);
        }
    }
    q->ctx = ctx;
    q->keep_orig = keep_orig;
    const int rc = q->setup(src, dst, corr_src, corr_dst, seed);
    (void)hipStreamSynchronize(ctx->stream);
    if (rc != M3D_OK) {
        reg_destroy_on(q);
        return bad(rc);
    }
    return q;
}
void reg_destroy_on(m3d_reg* q) {
    if (!q) return;
    if (q->ctx) {
        (void)hipSetDevice(q->ctx->device);
        (void)hipStreamSynchronize(q->ctx->stream);
        q->S.release();
    }
    if (q->own_clouds) {
        if (q->csrc) m3d_cloud_destroy_on(q->csrc);
        if (q->cdst) m3d_cloud_destroy_on(q->cdst);
    }
    delete q;
}

}  // namespace

// compute_transformation_ransac on a lane the caller holds (m3d_registration_ransac, global_registration_on): the
// session's three steps in a loop, the lane kept from the uploads to the result.
void m3d::reg_session_release(m3d_reg* q) { reg_destroy_on(q); }

int m3d::registration_ransac_on(DeviceCtx* ctx, m3d_cloud* csrc, m3d_cloud* cdst, const double* src, size_t n_src,
                                const double* dst, size_t n_dst, const size_t* corr_src, const size_t* corr_dst, size_t m,
                                double threshold, int max_iter, double edge_length_threshold, double confidence,
                                const uint64_t* seed, double* T_out, m3d_reg_stats* stats, m3d_reg** session_out) {
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (session_out) *session_out = nullptr;
    std::memcpy(T_out, I4, sizeof(I4));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->best_index = -1;
    }
    int rc = M3D_OK;
    m3d_reg* q = reg_create_on(ctx, csrc, cdst, src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold, max_iter,
                               edge_length_threshold, confidence, seed, &rc, /*keep_orig=*/session_out != nullptr);
    if (!q) return rc;
    std::vector<uint32_t> counts;
    std::vector<double> sums;
    while (!q->trivial) {
        size_t ns = 0;
        rc = q->begin_chunk(&ns);
        if (rc != M3D_OK) break;   // M3D_FALSE: loop over; < 0: error
        counts.assign(std::max<size_t>(ns, 1), 0);
        sums.assign(std::max<size_t>(ns, 1), 0.0);
        rc = q->validate(0, ns, counts.data(), sums.data());
        if (rc != M3D_OK) break;
        rc = q->replay(counts.data(), sums.data());
        if (rc != M3D_OK) break;
    }
    if (q->trivial || rc == M3D_FALSE) rc = q->finish(T_out, stats);
    // (the caller goes on with the session's target grid: information_matrix_on; it releases the session -- reg_session_release)
    if (session_out && rc == M3D_OK && !q->trivial) *session_out = q;
    else reg_destroy_on(q);
    return rc;
}

extern "C" {

void m3d_reg_destroy(m3d_reg* q) {
    if (!q) return;
    if (q->ctx) {
        CtxLock lock(q->ctx);
        reg_destroy_on(q);
    } else {
        reg_destroy_on(q);
    }
}

m3d_reg* m3d_reg_create(const double* src, size_t n_src, const double* dst, size_t n_dst, const size_t* corr_src,
                        const size_t* corr_dst, size_t m, double threshold, int max_iter,
                        double edge_length_threshold, double confidence, const uint64_t* seed, int device) {
    int rc;
    const int chk = reg_check_args(src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold);
    if (chk < 0) return nullptr;
    if (chk == M3D_FALSE)
        return reg_create_on(nullptr, nullptr, nullptr, src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold, max_iter,
                             edge_length_threshold, confidence, seed, &rc);
    LaneLock lane(device);
    if (!lane.ctx) return nullptr;
    return reg_create_on(lane.ctx, nullptr, nullptr, src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold, max_iter,
                         edge_length_threshold, confidence, seed, &rc);
}

int m3d_reg_begin_chunk(m3d_reg* q, size_t* n_survivors) {
    if (!q || !n_survivors) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (q->trivial) {
        *n_survivors = 0;
        return M3D_FALSE;
    }
    CtxLock lock(q->ctx);
    return q->begin_chunk(n_survivors);
}

int m3d_reg_validate(m3d_reg* q, size_t s_begin, size_t s_end, uint32_t* counts, double* sums) {
    if (!q || q->trivial || (s_end > s_begin && (!counts || !sums))) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    CtxLock lock(q->ctx);
    return q->validate(s_begin, s_end, counts, sums);
}

int m3d_reg_replay(m3d_reg* q, const uint32_t* counts, const double* sums) {
    if (!q || q->trivial) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (!q->survivors.empty() && (!counts || !sums)) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    CtxLock lock(q->ctx);
    return q->replay(counts, sums);
}

int m3d_reg_finish(m3d_reg* q, double* T, m3d_reg_stats* stats) {
    if (!q || !T) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (q->trivial) return q->finish(T, stats);
    CtxLock lock(q->ctx);
    return q->finish(T, stats);
}

int m3d_registration_ransac(const double* src, size_t n_src, const double* dst, size_t n_dst,
                            const size_t* corr_src, const size_t* corr_dst, size_t m, double threshold,
                            int max_iter, double edge_length_threshold, double confidence,
                            const uint64_t* seed, int device, double* T_out, m3d_reg_stats* stats) {
    if (!T_out) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    const int chk = reg_check_args(src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold);
    if (chk < 0) return chk;
    if (chk == M3D_FALSE)   // the trivial session: identity, no device
        return registration_ransac_on(nullptr, nullptr, nullptr, src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold,
                                      max_iter, edge_length_threshold, confidence, seed, T_out, stats);
    LaneLock lane(device);
    if (!lane.ctx) return M3D_ERR_DEVICE;
    return registration_ransac_on(lane.ctx, nullptr, nullptr, src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold,
                                  max_iter, edge_length_threshold, confidence, seed, T_out, stats);
}

// The same loop with every chunk's validations sharded over the ranks of `comm`: sessions are seeded
// identically, so every rank draws the same triples and keeps the same survivors; rank r validates a contiguous run of
// whole 64-hypothesis groups (the validation kernel's unit); ONE all-gather per chunk carries (sum d^2, count) as
// 16-byte records (bit-exact transport); every rank replays the whole chunk.
int m3d_registration_ransac_sharded(const double* src, size_t n_src, const double* dst, size_t n_dst,
                                    const size_t* corr_src, const size_t* corr_dst, size_t m, double threshold,
                                    int max_iter, double edge_length_threshold, double confidence,
                                    const uint64_t* seed, int device, m3d_comm* comm, double* T_out,
                                    m3d_reg_stats* stats) {
    if (!comm)
        return m3d_registration_ransac(src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold, max_iter,
                                       edge_length_threshold, confidence, seed, device, T_out, stats);
    if (!T_out) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (comm->transport == m3d_comm::kRccl && comm->device != device)
        return fail(M3D_ERR_INVALID_ARG, "the RCCL communicator lives on another device");
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(T_out, I4, sizeof(I4));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->best_index = -1;
    }
    // one seed for all ranks: rank 0's when none was given
    uint64_t sd = 0;
    {
        std::random_device rd;
        const uint64_t mine = seed ? *seed : (uint64_t)rd();
        sd = mine;
        if (!seed && comm->world > 1) {
            std::vector<uint64_t> all((size_t)comm->world);
            LaneLock lane(device);
            DeviceCtx* ctx = lane.ctx;
            if (!ctx) return M3D_ERR_DEVICE;
            HIPCHK(hipSetDevice(ctx->device));
            const int rs = comm->allgather_host(&mine, all.data(), sizeof(uint64_t), ctx->stream);
            if (rs != M3D_OK) return rs;
            sd = all[0];
        }
    }
    int rc = M3D_OK;
    m3d_reg* q = nullptr;
    const int chk = reg_check_args(src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold);
    if (chk < 0) return chk;
    if (chk == M3D_FALSE) {
        q = reg_create_on(nullptr, nullptr, nullptr, src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold, max_iter,
                          edge_length_threshold, confidence, &sd, &rc);
    } else {
        LaneLock lane(device);
        if (!lane.ctx) return M3D_ERR_DEVICE;
        q = reg_create_on(lane.ctx, nullptr, nullptr, src, n_src, dst, n_dst, corr_src, corr_dst, m, threshold, max_iter,
                          edge_length_threshold, confidence, &sd, &rc);
    }
    if (!q) return rc;
    struct Rec {
        double sum;
        uint64_t count;
    };
    const size_t world = (size_t)comm->world, rank = (size_t)comm->rank;
    std::vector<uint32_t> counts, counts_all;
    std::vector<double> sums, sums_all;
    std::vector<Rec> mine, all;
    for (;;) {
        size_t ns = 0;
        rc = m3d_reg_begin_chunk(q, &ns);
        if (rc != M3D_OK) break;   // M3D_FALSE: loop over; < 0: error
        const size_t groups = (ns + 63) / 64, per = (groups + world - 1) / world;
        const size_t g0 = std::min(rank * per, groups), g1 = std::min(g0 + per, groups);
        const size_t s0 = g0 * 64, s1 = std::min(g1 * 64, ns), shard = per * 64;
        counts.assign(std::max<size_t>(shard, 1), 0);
        sums.assign(std::max<size_t>(shard, 1), 0.0);
        // A rank whose validation fails (a device error: the one step of the loop that is not the same host logic on
        // every rank) still takes part in the window's exchange, with a poisoned first record, so that its peers leave
        // the loop with an error instead of waiting for it in the all-gather forever.  Failures elsewhere
        // (begin_chunk, the exchange itself) happen on every rank alike or are fatal to the group, as with any collective.
        const int vrc = m3d_reg_validate(q, s0, std::max(s0, s1), counts.data(), sums.data());
        if (vrc != M3D_OK && !shard) {
            rc = vrc;
            break;
        }
        counts_all.assign(std::max<size_t>(ns, 1), 0);
        sums_all.assign(std::max<size_t>(ns, 1), 0.0);
        if (shard) {   // ns == 0 on every rank alike: nothing to exchange
            constexpr uint64_t kPoison = ~(uint64_t)0;
            mine.assign(shard, Rec{0.0, 0});
            for (size_t i = 0; i + s0 < s1; ++i) mine[i] = Rec{sums[i], counts[i]};
            if (vrc != M3D_OK) mine[0].count = kPoison;
            all.assign(shard * world, Rec{0.0, 0});
            {
                CtxLock lock(q->ctx);
                rc = hipSetDevice(q->ctx->device) == hipSuccess
                         ? comm->allgather_host(mine.data(), all.data(), sizeof(Rec) * shard, q->ctx->stream)
                         : fail(M3D_ERR_DEVICE, "hipSetDevice failed");
            }
            if (rc == M3D_OK && vrc != M3D_OK) rc = vrc;
            if (rc != M3D_OK) break;
            bool peer_failed = false;
            for (size_t r = 0; r < world; ++r) peer_failed = peer_failed || all[r * shard].count == kPoison;
            if (peer_failed) {
                rc = fail(M3D_ERR_DEVICE, "m3d_registration_ransac_sharded: another rank's validation failed");
                break;
            }
            for (size_t i = 0; i < ns; ++i) {   // rank-major slices of `shard` = survivor order
                counts_all[i] = (uint32_t)all[i].count;
                sums_all[i] = all[i].sum;
            }
        }
        rc = m3d_reg_replay(q, counts_all.data(), sums_all.data());
        if (rc != M3D_OK) break;
    }
    if (rc == M3D_FALSE) rc = m3d_reg_finish(q, T_out, stats);
    m3d_reg_destroy(q);
    return rc;
}


// Point-to-point ICP: Open3D RegistrationICP with TransformationEstimationPointToPoint, which both registration
// examples of the reference run on the RANSAC pose (examples/cpp/transform_estimation.cpp:82-86); SURVEY.md 8(f)
// N1.  [RECALL] of the Open3D 0.15.1 loop: oracle/misc3d_oracle_reg.c orc_registration_icp.  Reuses the
// registration path's target grid (+ original indices), the serial-order error sum, the Kabsch sums and the
// K3x3 umeyama assembly.  The moving cloud is transformed in place every iteration, as Open3D does.
int m3d_registration_icp(const double* src, size_t n_src, const double* dst, size_t n_dst,
                         double max_correspondence_distance, const double* T_init, int max_iteration,
                         double relative_fitness, double relative_rmse, int device, double* T_out,
                         m3d_icp_stats* stats, int64_t* correspondences) {
    const double t_begin = now_ms();
    if (!T_out || (!src && n_src) || (!dst && n_dst)) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double T[16];
    std::memcpy(T, T_init ? T_init : I4, sizeof(T));
    std::memcpy(T_out, T, sizeof(T));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (!(max_correspondence_distance > 0.0))   // Open3D: LogError("Invalid max_correspondence_distance.")
        return fail(M3D_ERR_INVALID_ARG, "Invalid max_correspondence_distance.");
    if (n_src >= ((size_t)1 << 31) || n_dst >= ((size_t)1 << 31)) return fail(M3D_ERR_INVALID_ARG, "too many points");
    if (n_src == 0 || n_dst == 0) {   // no correspondences: every update is the identity, fitness 0
        if (correspondences)
            for (size_t i = 0; i < n_src; ++i) correspondences[i] = -1;
        if (stats) {
            stats->iterations = max_iteration > 0 ? 1 : 0;   // the first repeat already meets both criteria
            stats->converged = max_iteration > 0;
        }
        return M3D_OK;
    }
    LaneLock lane(device);
    DeviceCtx* ctx = lane.ctx;
    if (!ctx) return M3D_ERR_DEVICE;
    m3d_cloud* csrc = m3d_cloud_create_on(ctx, src, nullptr, n_src, 0);
    if (!csrc) return M3D_ERR_DEVICE;
    m3d_cloud* cdst = m3d_cloud_create_on(ctx, dst, nullptr, n_dst, 0);
    if (!cdst) {
        m3d_cloud_destroy_on(csrc);
        return M3D_ERR_DEVICE;
    }
    Scratch S;
    DevBuf mx, my, mz, nn, d2;
    int rc;
    {
        rc = [&]() -> int {
            HIPCHK(hipSetDevice(ctx->device));
            const CloudView sv = csrc->view(), dv = cdst->view();
            const uint32_t n = sv.n;
            GridDesc g;
            const int rg = build_target_grid(ctx, S, dv, dst, n_dst, max_correspondence_distance, true, &g);
            if (rg != M3D_OK) return rg;
            const uint32_t nb = (n + 2047) / 2048;
            RESERVE(mx, sizeof(double) * n);
            RESERVE(my, sizeof(double) * n);
            RESERVE(mz, sizeof(double) * n);
            RESERVE(nn, sizeof(uint32_t) * n);
            RESERVE(d2, sizeof(double) * n);
            RESERVE(S.vals, sizeof(double) * n);
            RESERVE(S.block_counts, sizeof(uint32_t) * ((size_t)nb + 1));
            RESERVE(S.total, 16);
            RESERVE(S.sums, sizeof(double) * 32);
            RESERVE(S.partial_sum, sizeof(double) * 256 * 16);
            RESERVE(S.one_T, sizeof(double) * kRegTStride);
            RESERVE(ctx->h_small, 512);
            double* px = mx.as<double>();
            double* py = my.as<double>();
            double* pz = mz.as<double>();
            auto upload_T = [&](const double* M) -> int {
                HIPCHK(hipMemcpyAsync(S.one_T.p, M, sizeof(double) * 12, hipMemcpyHostToDevice, ctx->stream));
                return M3D_OK;
            };
            // pcd = source; if (!init.isIdentity()) pcd.Transform(init)
            if (std::memcmp(T, I4, sizeof(T)) != 0) {
                const int r = upload_T(T);
                if (r != M3D_OK) return r;
                launch_icp_transform(sv.x, sv.y, sv.z, n, S.one_T.as<double>(), px, py, pz, ctx->stream);
            } else {
                HIPCHK(hipMemcpyAsync(px, sv.x, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
                HIPCHK(hipMemcpyAsync(py, sv.y, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
                HIPCHK(hipMemcpyAsync(pz, sv.z, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
            }
            uint64_t cnt = 0;
            double e2 = 0.0;
            double hs[18];   // sums of the correspondence set for umeyama (filled by result())
            // GetRegistrationResultAndCorrespondences: nearest target point within the radius, count + error2
            auto result = [&]() -> int {
                launch_icp_nn(px, py, pz, n, g, S.cell_start.as<uint32_t>(), S.qx.as<double>(), S.qy.as<double>(),
                              S.qz.as<double>(), S.cell_orig.as<uint32_t>(), nn.as<uint32_t>(), d2.as<double>(),
                              ctx->stream);
                launch_icp_err(d2.as<double>(), n, g.r2, S.partial_sum.as<double>(), S.sums.as<double>() + 24, ctx->stream);
                // the sums of the NEXT ComputeTransformation are queued behind it at once (they read the count from
                // the device): one host wait per iteration instead of two; after the last iteration they go unused
                launch_icp_sums(px, py, pz, n, dv, nn.as<uint32_t>(), S.sums.as<double>() + 25,
                                S.partial_sum.as<double>(), S.sums.as<double>(), ctx->stream);
                RESERVE(ctx->h_small, 256);
                uint8_t* h = ctx->h_small.as<uint8_t>();
                HIPCHK(hipMemcpyAsync(h, S.sums.as<double>() + 24, 16, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipMemcpyAsync(h + 32, S.sums.p, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipGetLastError());
                if (const int wr = stream_wait_spin(ctx); wr != M3D_OK) return wr;   // (an iteration's one wait: polled, see corr_inlier_ratio)
                double es[2];
                std::memcpy(es, h, 16);
                std::memcpy(hs, h + 32, sizeof(hs));
                e2 = es[0];
                const uint32_t c = (uint32_t)es[1];
                cnt = c;
                return M3D_OK;
            };
            int r = result();
            if (r != M3D_OK) return r;
            double fit = (double)cnt / (double)n_src;
            double rm = cnt ? std::sqrt(e2 / (double)cnt) : 0.0;
            int it = 0;
            bool converged = false;
            for (; it < max_iteration; ++it) {
                double U[16];
                std::memcpy(U, I4, sizeof(U));
                if (cnt) {   // ComputeTransformation: Eigen::umeyama over the correspondence set, no scaling
                    const double* h = hs;
                    const double one_over_n = 1.0 / (double)cnt;
                    double ms[3], md[3], sig[9];
                    for (int k = 0; k < 3; ++k) {
                        ms[k] = h[k] * one_over_n;
                        md[k] = h[3 + k] * one_over_n;
                    }
                    for (int k = 0; k < 9; ++k) sig[k] = h[6 + k] * one_over_n;
                    const double src_var = ((h[15] + h[16]) + h[17]) * one_over_n;
                    umeyama_assemble(ms, md, sig, src_var, false, U);
                }
                // transformation = update * transformation
                double Tn[16];
                for (int rr = 0; rr < 4; ++rr)
                    for (int cc = 0; cc < 4; ++cc)
                        Tn[4 * rr + cc] = ((U[4 * rr] * T[cc] + U[4 * rr + 1] * T[4 + cc]) + U[4 * rr + 2] * T[8 + cc]) +
                                          U[4 * rr + 3] * T[12 + cc];
                std::memcpy(T, Tn, sizeof(T));
                r = upload_T(U);
                if (r != M3D_OK) return r;
                launch_icp_transform(px, py, pz, n, S.one_T.as<double>(), px, py, pz, ctx->stream);   // pcd.Transform(update)
                const double fit0 = fit, rm0 = rm;
                r = result();
                if (r != M3D_OK) return r;
                fit = (double)cnt / (double)n_src;
                rm = cnt ? std::sqrt(e2 / (double)cnt) : 0.0;
                if (std::fabs(fit0 - fit) < relative_fitness && std::fabs(rm0 - rm) < relative_rmse) {
                    ++it;
                    converged = true;
                    break;
                }
            }
            std::memcpy(T_out, T, sizeof(T));
            if (correspondences) {
                std::vector<uint32_t> hn(n);
                HIPCHK(hipMemcpy(hn.data(), nn.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
                for (uint32_t i = 0; i < n; ++i) correspondences[i] = hn[i] == 0xFFFFFFFFu ? -1 : (int64_t)hn[i];
            }
            if (stats) {
                stats->fitness = fit;
                stats->inlier_rmse = rm;
                stats->correspondences = cnt;
                stats->iterations = it;
                stats->converged = converged ? 1 : 0;
            }
            return M3D_OK;
        }();
        (void)hipStreamSynchronize(ctx->stream);
        S.release();
        mx.release(); my.release(); mz.release(); nn.release(); d2.release();
    }
    m3d_cloud_destroy_on(csrc);
    m3d_cloud_destroy_on(cdst);
    if (stats) stats->ms_total = now_ms() - t_begin;
    return rc;
}

// open3d::pipelines::registration::GetInformationMatrixFromPointClouds(source, target, max_dist, T): what
// ReconstructionPipeline::GlobalRegistration (src/pipeline.cpp:818-824) uses to accept or reject the RANSAC
// pose (info(5,5) / min(Ns, Nt) < 0.3 -> reject); SURVEY.md 8(f) N2.  info(5,5) is the correspondence count
// (exact); the other entries are moment sums of the matched target points (order-free tree sums).
int m3d_information_matrix(const double* src, size_t n_src, const double* dst, size_t n_dst,
                           double max_correspondence_distance, const double* T, int device, double* info,
                           uint64_t* n_correspondences) {
    if (!info || !T || (!src && n_src) || (!dst && n_dst)) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    for (int k = 0; k < 36; ++k) info[k] = 0.0;
    if (n_correspondences) *n_correspondences = 0;
    if (!(max_correspondence_distance > 0.0)) return fail(M3D_ERR_INVALID_ARG, "Invalid max_correspondence_distance.");
    if (n_src >= ((size_t)1 << 31) || n_dst >= ((size_t)1 << 31)) return fail(M3D_ERR_INVALID_ARG, "too many points");
    if (n_src == 0 || n_dst == 0) return M3D_OK;
    LaneLock lane(device);
    DeviceCtx* ctx = lane.ctx;
    if (!ctx) return M3D_ERR_DEVICE;
    m3d_cloud* csrc = m3d_cloud_create_on(ctx, src, nullptr, n_src, 0);
    if (!csrc) return M3D_ERR_DEVICE;
    m3d_cloud* cdst = m3d_cloud_create_on(ctx, dst, nullptr, n_dst, 0);
    if (!cdst) {
        m3d_cloud_destroy_on(csrc);
        return M3D_ERR_DEVICE;
    }
    const int rc = information_matrix_on(ctx, csrc, cdst, dst, n_dst, max_correspondence_distance, T, info, n_correspondences);
    m3d_cloud_destroy_on(csrc);
    m3d_cloud_destroy_on(cdst);
    return rc;
}
}  // extern "C"

// ... on a lane the caller holds, for two clouds resident on it (m3d_information_matrix, global_registration_on)
int m3d::information_matrix_on(DeviceCtx* ctx, m3d_cloud* csrc, m3d_cloud* cdst, const double* dst, size_t n_dst,
                               double max_correspondence_distance, const double* T, double* info,
                               uint64_t* n_correspondences, m3d_reg* session) {
    // session: a finished RANSAC session on the same two clouds whose target grid was built for the same radius with the points'
    // original indices kept (registration_ransac_on, session_out): its grid is searched instead of building one (a grid + its
    // neighbour lists for ONE pass of nearest-neighbour queries took 0.3 of the 0.4 ms this call took on 50 000-point fragments)
    const bool shared = session && session->ctx == ctx && session->keep_orig && !session->trivial && session->cdst == cdst &&
                        session->threshold == max_correspondence_distance && session->S.cell_orig.p;
    Scratch S;
    DevBuf mx, my, mz, nn, d2;
    for (int k = 0; k < 36; ++k) info[k] = 0.0;   // (the entries the sums below do not touch are zeros of the matrix)
    if (n_correspondences) *n_correspondences = 0;
    int rc;
    {
        rc = [&]() -> int {
            HIPCHK(hipSetDevice(ctx->device));
            const CloudView sv = csrc->view(), dv = cdst->view();
            const uint32_t n = sv.n;
            GridDesc g;
            if (shared) {
                g = session->g;
            } else {
                const int rg = build_target_grid(ctx, S, dv, dst, n_dst, max_correspondence_distance, true, &g, 4, true,
                                                 cdst->bb_known ? cdst->bb : nullptr);
                if (rg != M3D_OK) return rg;
            }
            Scratch& G = shared ? session->S : S;   // (the grid's arrays)
            RESERVE(mx, sizeof(double) * n);
            RESERVE(my, sizeof(double) * n);
            RESERVE(mz, sizeof(double) * n);
            RESERVE(nn, sizeof(uint32_t) * n);
            RESERVE(d2, sizeof(double) * n);
            RESERVE(S.sums, sizeof(double) * 32);
            RESERVE(S.partial_sum, sizeof(double) * 256 * 16);
            RESERVE(S.one_T, sizeof(double) * kRegTStride);
            RESERVE(S.block_counts, sizeof(uint32_t) * ((size_t)(n + 2047) / 2048 + 1));
            RESERVE(S.total, 16);
            RESERVE(S.vals, sizeof(double) * n);
            RESERVE(ctx->h_small, 512);
            HIPCHK(hipMemcpyAsync(S.one_T.p, T, sizeof(double) * 12, hipMemcpyHostToDevice, ctx->stream));
            launch_icp_transform(sv.x, sv.y, sv.z, n, S.one_T.as<double>(), mx.as<double>(), my.as<double>(),
                                 mz.as<double>(), ctx->stream);
            launch_icp_nn(mx.as<double>(), my.as<double>(), mz.as<double>(), n, g, G.cell_start.as<uint32_t>(),
                          G.qx.as<double>(), G.qy.as<double>(), G.qz.as<double>(), G.cell_orig.as<uint32_t>(),
                          nn.as<uint32_t>(), d2.as<double>(), ctx->stream);
            launch_compact_vals(d2.as<double>(), n, g.r2, S.block_counts.as<uint32_t>(), S.total.as<uint32_t>(),
                                S.vals.as<double>(), ctx->stream);   // only for the count
            launch_info_sums(n, dv, nn.as<uint32_t>(), S.partial_sum.as<double>(), S.sums.as<double>(), ctx->stream);
            uint8_t* h = ctx->h_small.as<uint8_t>();
            HIPCHK(hipMemcpyAsync(h, S.total.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(h + 8, S.sums.p, sizeof(double) * 9, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipGetLastError());
            if (const int wr = stream_wait_spin(ctx); wr != M3D_OK) return wr;
            uint32_t c;
            double m[9];
            std::memcpy(&c, h, 4);
            std::memcpy(m, h + 8, sizeof(m));
            const double cnt = (double)c;
            const double sx = m[0], sy = m[1], sz = m[2], xx = m[3], yy = m[4], zz = m[5], xy = m[6], xz = m[7], yz = m[8];
            auto set = [&](int a, int b, double v) {
                info[6 * a + b] = v;
                info[6 * b + a] = v;
            };
            set(0, 0, yy + zz);
            set(0, 1, -xy);
            set(0, 2, -xz);
            set(0, 4, -sz);
            set(0, 5, sy);
            set(1, 1, xx + zz);
            set(1, 2, -yz);
            set(1, 3, sz);
            set(1, 5, -sx);
            set(2, 2, xx + yy);
            set(2, 3, -sy);
            set(2, 4, sx);
            set(3, 3, cnt);
            set(4, 4, cnt);
            set(5, 5, cnt);
            if (n_correspondences) *n_correspondences = c;
            return M3D_OK;
        }();
        (void)hipStreamSynchronize(ctx->stream);
        S.release();
        mx.release(); my.release(); mz.release(); nn.release(); d2.release();
    }
    return rc;
}
extern "C" {

// misc3d::features::DetectBoundaryPoints, src/boundary_detection.cpp:68-113: the step the
// reference's showcase example runs right after fit_plane (examples/python/ransac_and_boundary.py:35-36).
int m3d_detect_boundary_points(const double* xyz, const double* normals, size_t n, int search, double radius,
                               int max_nn, double angle_threshold_deg, int device, size_t* indices, size_t* k_out) {
    if (!k_out || (!xyz && n) || (!indices && n)) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    *k_out = 0;
    if (n == 0) return fail(M3D_ERR_INVALID_ARG, "No PointCloud data.");   // :72-76 LogError
    if (search < 0 || search > 2) return fail(M3D_ERR_INVALID_ARG, "search: 0 = KNN, 1 = Radius, 2 = Hybrid");
    if ((search != 0 && !(radius > 0.0)) || (search != 1 && (max_nn < 1 || max_nn > kBoundaryMaxNb)))
        return fail(M3D_ERR_INVALID_ARG, "invalid search parameter (radius > 0, 1 <= max_nn <= 128)");
    if (n >= ((size_t)1 << 31)) return fail(M3D_ERR_INVALID_ARG, "too many points");
    LaneLock lane(device);
    DeviceCtx* ctx = lane.ctx;
    if (!ctx) return M3D_ERR_DEVICE;
    m3d_cloud* c = m3d_cloud_create_on(ctx, xyz, normals, n, 0);
    if (!c) return M3D_ERR_DEVICE;
    Scratch S;
    DevBuf flags;
    int rc;
    {
        rc = [&]() -> int {
            HIPCHK(hipSetDevice(ctx->device));
            const CloudView v = c->view();
            GridDesc g;
            double cell = radius;
            if (search == 0) {   // KNN: a cell that holds ~max_nn / 4 points on average (bounding-box density)
                double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (size_t i = 0; i < n; ++i)
                    for (int k = 0; k < 3; ++k)
                        if (std::isfinite(xyz[3 * i + k])) {
                            lo[k] = std::min(lo[k], xyz[3 * i + k]);
                            hi[k] = std::max(hi[k], xyz[3 * i + k]);
                        }
                double ext[3], emax = 0.0;
                for (int k = 0; k < 3; ++k) {
                    ext[k] = lo[k] <= hi[k] ? hi[k] - lo[k] : 0.0;
                    emax = std::max(emax, ext[k]);
                }
                double vol = 1.0;   // flat / degenerate extents count as 1 % of the largest
                for (int k = 0; k < 3; ++k) vol *= std::max(ext[k], 0.01 * emax);
                cell = emax > 0.0 ? std::cbrt(vol * (double)max_nn / (4.0 * (double)n)) : 1.0;
                if (!(cell > 0.0) || !std::isfinite(cell)) cell = 1.0;
            }
            const int rg = build_target_grid(ctx, S, v, xyz, n, cell, true, &g, /*K0=*/1, /*with_nl=*/false);
            if (rg != M3D_OK) return rg;
            RESERVE(flags, (size_t)v.n + 8);
            HIPCHK(hipMemsetAsync(flags.p, 0, (size_t)v.n + 8, ctx->stream));
            launch_boundary(v, g, S.cell_start.as<uint32_t>(), S.qx.as<double>(), S.qy.as<double>(), S.qz.as<double>(),
                            S.cell_orig.as<uint32_t>(), search, max_nn, angle_threshold_deg, flags.as<uint8_t>(),
                            flags.as<uint8_t>() + v.n, ctx->stream, S.total.as<uint32_t>());
            std::vector<uint8_t> hf((size_t)v.n + 8);
            HIPCHK(hipMemcpyAsync(hf.data(), flags.p, hf.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (hf[v.n])
                return fail(M3D_ERR_INVALID_ARG, "more than 128 neighbours within the radius (use KDTreeSearchParamHybrid)");
            size_t k = 0;
            for (uint32_t i = 0; i < v.n; ++i)
                if (hf[i]) indices[k++] = i;   // ascending (the reference's order depends on thread timing)
            *k_out = k;
            return M3D_OK;
        }();
        (void)hipStreamSynchronize(ctx->stream);
        S.release();
        flags.release();
    }
    m3d_cloud_destroy_on(c);
    return rc;
}

}  // extern "C"

extern "C" {
// test hook (include/misc3d_amd_bench.h): the checkers as this library's M3D_FP_ORDER compiles them, on the host
int m3d_bench_reg_checkers(const double* ps, const double* pd, const double* T, double edge_threshold,
                           double distance_threshold) {
    if (!ps || !pd || !T) return fail(M3D_ERR_INVALID_ARG, "null argument");
    return m3d::reg_checkers(ps, pd, T, edge_threshold, distance_threshold) ? 1 : 0;
}

}  // extern "C"
