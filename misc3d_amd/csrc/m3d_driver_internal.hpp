// m3d_driver_internal.hpp -- what the translation units of the host driver share (m3d_device.cpp: errors, lanes, block
// pools, resident clouds; m3d_fit.cpp: the RANSAC loop; m3d_refine.cpp: RefineModel; m3d_segmentation.cpp:
// SegmentPlaneIterative; m3d_multi.cpp: one process, several devices; m3d_bench_hooks.cpp).  Not part of any boundary.
#pragma once
#include "m3d_driver.hpp"
#include "m3d_comm.hpp"
#include "m3d_config.hpp"
#include "m3d_fp.hpp"
#include "m3d_mt19937.hpp"
#include "m3d_reg_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <random>
#include <thread>

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return m3d::fail(M3D_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define RESERVE(buf, bytes)                         \
    do {                                            \
        if (!(buf).reserve(bytes)) return M3D_ERR_DEVICE; \
    } while (0)

namespace m3d {

static inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }
static inline int minimal_sample(int kind) { return kind == M3D_PLANE ? 3 : (kind == M3D_SPHERE ? 4 : 2); }
static inline int num_params(int kind) { return kind == M3D_CYLINDER ? 7 : 4; }
static inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- m3d_device.cpp
const std::string& last_error_string();
bool is_library_pinned(const void* p, size_t bytes);   // inside a block m3d_host_alloc handed out?
void release_buffers(m3d_cloud* c);                    // every device buffer a cloud owns -> its lane's free list

// ---- m3d_fit.cpp
int stream_wait_spin(DeviceCtx* ctx);
int validate_fit_args(int kind, size_t n, bool has_normals, double prob);
uint64_t resolve_seed(const uint64_t* seed);
int finalize_deferred_refine(DeviceCtx* ctx);
int agree_seed(m3d_comm* comm, const uint64_t* seed, hipStream_t st, uint64_t* out);
int cloud_fit_locked(m3d_cloud* c, int kind, double thr, size_t max_iter, double prob, uint64_t seed, double* params,
                     size_t* inliers, size_t* n_inliers, m3d_stats* stats,
                     const std::function<int(int64_t)>* before_refine_wait = nullptr,
                     size_t* iterations_hint = nullptr /* in: iterations of a similar fit, out: of this one */,
                     m3d_comm* comm = nullptr);

// ---- m3d_refine.cpp
int compact_scratch(DeviceCtx* ctx, uint32_t nb, CompactScratch* out);
int exact_error(DeviceCtx* ctx, const CloudView& v, int kind, double thr,
                       const double* model_dev, uint64_t* count, double* error);
int approx_error_pair(DeviceCtx* ctx, const CloudView& v, int kind, double thr, const double* model_a,
                             const double* model_b, uint64_t* count_a, double* error_a, uint64_t* count_b, double* error_b);
int approx_error(DeviceCtx* ctx, const CloudView& v, int kind, double thr,
                        const double* model_dev, uint64_t* count, double* error);
bool plane_from_moments(const double* mean, const double* s, double* out);
int refine_slot(const DeviceCtx* ctx);
double* h_best_at(DeviceCtx* ctx);
uint8_t* h_total_at(DeviceCtx* ctx);
int issue_refine_compaction(DeviceCtx* ctx, const CloudView& flag_view, const uint32_t* orig_dev, int kind, double thr,
                            const double* model_dev, const double* lazy_in, void* total_host, bool fused = false,
                            uint64_t* idx_host = nullptr, const PartitionOut* part = nullptr);
int refine(DeviceCtx* ctx, const CloudView& flag_view, const CloudView& gather_view, const uint32_t* orig_dev, int kind, double thr,
           const double* model_dev, double* params_host /* in: best minimal model, out: refined */, size_t* inliers,
           size_t* n_inliers, int* general_fit_ok, int64_t expected_ni = -1, const std::function<int(int64_t)>* before_wait = nullptr,
           const double* lazy_in = nullptr /* pinned: the "in" value of params_host arrives with the wait */,
           const void* compaction_total = nullptr /* pinned: the compaction is already queued (on model_dev) */,
           bool fused = false /* model_dev is a minimal_fit_k record: moments ride on the compaction (needs lazy_in) */);

// m3d_segmentation.cpp
int segment_impl(const double* xyz, size_t n, double threshold, int max_iteration, double min_ratio,
                        const uint64_t* seed, int device, m3d_comm* comm, size_t max_clusters, double* planes,
                        size_t* cluster_offsets, size_t* cluster_indices, size_t* n_clusters,
                        double* cluster_points = nullptr /* n x 3: the xyz of cluster_indices[i] at 3 i (may be null) */);
// ---- m3d_segmentation.cpp (the removal of a fit's inliers stays inside it)
extern thread_local double g_seg_ms[6];   // the calling thread's last segmentation call (m3d_bench_last_segment_ms)


}  // namespace m3d
