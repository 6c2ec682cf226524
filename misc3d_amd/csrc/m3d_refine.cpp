// m3d_refine.cpp -- RefineModel (include/misc3d/common/ransac.h:534-549) and the exact / order-free error sums that decide
// fitness ties (ransac.h:595-596, 632-650): the ordered inlier list (compact_count_k + scan_blocks_k + compact_write_k, straight
// into the caller's page-locked buffer), GeneralFit from the fused moments, the serial-order sum on demand.
#include "m3d_driver_internal.hpp"

#pragma clang fp contract(off)

namespace m3d {


// Scratch of launch_compact (m3d_kernels.hpp, CompactScratch): one slot per compaction workgroup, zero when the buffer is
// (re)allocated and when the epoch counter starts over; every launch gets the next epoch.
int compact_scratch(DeviceCtx* ctx, uint32_t nb, CompactScratch* out) {
    const size_t need = sizeof(uint32_t) * ((size_t)nb + 1);
    const bool grow = ctx->block_counts.cap < need;
    if (grow) RESERVE(ctx->block_counts, need);
    if (ctx->compact_epoch >= kCompactEpochs) ctx->compact_epoch = 0;
    if (grow || ctx->compact_epoch == 0)
        HIPCHK(hipMemsetAsync(ctx->block_counts.p, 0, ctx->block_counts.cap, ctx->stream));
    out->slots = ctx->block_counts.as<uint32_t>();
    out->tag = ++ctx->compact_epoch << 12;
    return M3D_OK;
}

// EvaluateModel's (inlier_num, error) with the error summed in point order (ransac.h:632-640).
int exact_error(DeviceCtx* ctx, const CloudView& v, int kind, double thr,
                       const double* model_dev, uint64_t* count, double* error) {
    const uint32_t nb = (v.n + kCompactTile - 1) / kCompactTile;
    RESERVE(ctx->dist, sizeof(double) * (size_t)std::max<uint32_t>(v.n, 1));
    CompactScratch scratch;
    if (const int rc = compact_scratch(ctx, nb, &scratch); rc != M3D_OK) return rc;
    RESERVE(ctx->total, sizeof(uint32_t) * 4);
    RESERVE(ctx->sums, sizeof(double) * 32);
    RESERVE(ctx->h_small, 256);
    launch_compact(kind, v, model_dev, thr, 1, nullptr, nullptr, ctx->dist.as<double>(), nullptr,
                   nullptr, nullptr, nullptr, 0, scratch,
                   ctx->total.as<uint32_t>(), ctx->stream);
    launch_serial_sum(ctx->dist.as<double>(), ctx->total.as<uint32_t>(), ctx->sums.as<double>() + 16,
                      ctx->stream);
    uint8_t* h = ctx->h_small.as<uint8_t>();
    HIPCHK(hipMemcpyAsync(h, ctx->total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(h + 8, ctx->sums.as<double>() + 16, sizeof(double), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint32_t c32;
    std::memcpy(&c32, h, 4);
    std::memcpy(error, h + 8, 8);
    *count = c32;
    return M3D_OK;
}

// Order-free sums of the inlier distances (tree) + counts of the trial model and, when its sum is not known yet, of the
// incumbent (model_b != null): enough to decide most ties (see the tie rule in run_ransac).  ONE pass over the cloud for
// both, the results stored into pinned memory by the kernel's last workgroup: one launch and one wait per tie.
int approx_error_pair(DeviceCtx* ctx, const CloudView& v, int kind, double thr, const double* model_a,
                             const double* model_b, uint64_t* count_a, double* error_a, uint64_t* count_b, double* error_b) {
    const bool fresh = ctx->tie_scratch.cap == 0;
    RESERVE(ctx->tie_scratch, sizeof(double) * (kErrorSumScratchDoubles + 2));
    RESERVE(ctx->h_tie, 64);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(ctx->tie_scratch.as<double>() + kErrorSumScratchDoubles);
    if (fresh) HIPCHK(hipMemsetAsync(ticket, 0, sizeof(uint32_t), ctx->stream));
    double* h = ctx->h_tie.as<double>();
    launch_error_sum(kind, v, model_a, model_b, thr, ctx->tie_scratch.as<double>(), ticket, h, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *count_a = (uint64_t)h[0];
    *error_a = h[1];
    if (model_b) {
        *count_b = (uint64_t)h[2];
        *error_b = h[3];
    }
    return M3D_OK;
}
int approx_error(DeviceCtx* ctx, const CloudView& v, int kind, double thr,
                        const double* model_dev, uint64_t* count, double* error) {
    return approx_error_pair(ctx, v, kind, thr, model_dev, nullptr, count, error, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// GeneralFit closed forms (host; the sums come from sum_*_k)
// ------------------------------------------------------------------------------------------------
// PlaneEstimator::GeneralFit, ransac.h:190-211
bool plane_from_moments(const double* mean, const double* s, double* out) {
    const double xx = s[0], xy = s[1], xz = s[2], yy = s[3], yz = s[4], zz = s[5];
    const double det_x = yy * zz - yz * yz;
    const double det_y = xx * zz - xz * xz;
    const double det_z = xx * yy - xy * xy;
    double a, b, c;
    if (det_x > det_y && det_x > det_z) {
        a = det_x;
        b = xz * yz - xy * zz;
        c = xy * yz - xz * yy;
    } else if (det_y > det_z) {
        a = xz * yz - xy * zz;
        b = det_y;
        c = xy * xz - yz * xx;
    } else {
        a = xy * yz - xz * yy;
        b = xy * xz - yz * xx;
        c = det_z;
    }
    const double norm = std::sqrt((a * a + b * b) + c * c);
    if (norm < 1.0e-8) return false;
    a /= norm;
    b /= norm;
    c /= norm;
    out[0] = a;
    out[1] = b;
    out[2] = c;
    out[3] = -((a * mean[0] + b * mean[1]) + c * mean[2]);
    return true;
}

// SphereEstimator::GeneralFit, ransac.h:296-330: least squares of [2x 2y 2z 1] w = x^2+y^2+z^2.
// The reference's bdcSvd(FullU) needs an N_inl x N_inl matrix (its own TODO, ransac.h:318-319);
// here the same least-squares problem is solved from the CENTRED normal equations
//   4 S c' = 2 sum(p' q),  w3' = sum(q)/n,  q = |p'|^2,  p' = p - mean,
// then centre = mean + c', r = sqrt(|c'|^2 + w3').  Same minimiser, parameters agree to ~1e-12.
static bool sphere_from_moments(const double* mean, const double* s, double n, double* out) {
    double A[3][4] = {{4 * s[0], 4 * s[1], 4 * s[2], 2 * s[6]},
                      {4 * s[1], 4 * s[3], 4 * s[4], 2 * s[7]},
                      {4 * s[2], 4 * s[4], 4 * s[5], 2 * s[8]}};
    for (int col = 0; col < 3; ++col) {  // Gaussian elimination, partial pivoting
        int piv = col;
        for (int r = col + 1; r < 3; ++r)
            if (std::fabs(A[r][col]) > std::fabs(A[piv][col])) piv = r;
        if (piv != col)
            for (int k = 0; k < 4; ++k) std::swap(A[piv][k], A[col][k]);
        if (A[col][col] == 0.0) continue;
        for (int r = col + 1; r < 3; ++r) {
            const double f = A[r][col] / A[col][col];
            for (int k = col; k < 4; ++k) A[r][k] -= f * A[col][k];
        }
    }
    double c[3];
    for (int r = 2; r >= 0; --r) {
        double acc = A[r][3];
        for (int k = r + 1; k < 3; ++k) acc -= A[r][k] * c[k];
        c[r] = A[r][r] != 0.0 ? acc / A[r][r] : 0.0;
    }
    const double w3 = s[9] / n;
    out[0] = mean[0] + c[0];
    out[1] = mean[1] + c[1];
    out[2] = mean[2] + c[2];
    out[3] = std::sqrt(((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) + w3);
    return true;
}


// RefineModel, ransac.h:534-549.  flag_view: cloud the distances are evaluated on; gather_view +
// orig: when the flags are computed on a compacted cloud (segmentation) the inlier list holds
// ORIGINAL indices and the GeneralFit sums gather from the original cloud (same values, same order).
// expected_ni >= 0: the inlier count is already known from the scoring pass (the usual case).  Then nothing
// has to wait for the compaction's own total: the GeneralFit sums run on the main stream while the index list
// travels to the host on the copy stream, and the total is only CHECKED at the end (a mismatch falls back to
// the synchronous order; the callers treat it as an internal error anyway).
// First stage of RefineModel: the ordered inlier list of `model_dev` (+ the model record to the pinned `lazy_in`).
// total_host (pinned) receives the inlier count.  Separate from refine() so that a probability-1 fit can queue it
// on the device's own prediction of the winner right behind the last scoring launch (run_ransac).
// fused: the model record carries the provisional centre of GeneralFit's sums (a record written by minimal_fit_k), so
// the compaction's counting pass accumulates the moments as well and no pass over the inlier list follows.
// idx_host: the caller's page-locked index list; the compaction writes it directly (the 8 bytes per inlier cross the
// host link while the kernel runs instead of in a copy command the host issues after it has woken up).
// where the compaction puts the ordered inlier list on the device (DeviceCtx::idx_out_override)
static uint64_t* idx_dev(DeviceCtx* ctx) { return ctx->idx_out_override ? ctx->idx_out_override : ctx->idx.as<uint64_t>(); }
// the pinned words RefineModel's kernels write, per slot (DeviceCtx::defer_refine; slot 0 otherwise)
int refine_slot(const DeviceCtx* ctx) { return ctx->defer_refine ? ctx->refine_slot : 0; }
double* h_best_at(DeviceCtx* ctx) { return ctx->h_best.as<double>() + (size_t)refine_slot(ctx) * kModelStride; }
uint8_t* h_total_at(DeviceCtx* ctx) { return ctx->h_pick.as<uint8_t>() + 64 + 8 * refine_slot(ctx); }
static double* h_moments_at(DeviceCtx* ctx) { return ctx->h_moments.as<double>() + (size_t)refine_slot(ctx) * kFusedMomentDoubles; }

int issue_refine_compaction(DeviceCtx* ctx, const CloudView& flag_view, const uint32_t* orig_dev, int kind,
                                   double thr, const double* model_dev, const double* lazy_in, void* total_host,
                                   bool fused, uint64_t* idx_host, const PartitionOut* part) {
    const uint32_t n = flag_view.n;
    const uint32_t nb = (n + kCompactTile - 1) / kCompactTile;
    RESERVE(ctx->idx, sizeof(uint64_t) * (size_t)std::max<uint32_t>(n, 1));
    CompactScratch scratch;
    if (const int rc = compact_scratch(ctx, nb, &scratch); rc != M3D_OK) return rc;
    RESERVE(ctx->total, sizeof(uint32_t) * 4);
    fused = fused && kind != M3D_CYLINDER;
    if (fused) {
        RESERVE(ctx->moment_partial, sizeof(double) * 16 * (size_t)std::max<uint32_t>(nb, 1));
        RESERVE(ctx->h_moments, sizeof(double) * 2 * kFusedMomentDoubles);
    }
    launch_compact(kind, flag_view, model_dev, thr, 0, orig_dev,
                   idx_dev(ctx), nullptr,
                   nullptr, nullptr, nullptr, nullptr, 0, scratch,
                   ctx->total.as<uint32_t>(), ctx->stream, const_cast<double*>(lazy_in),
                   fused ? ctx->moment_partial.as<double>() : nullptr, fused ? h_moments_at(ctx) : nullptr,
                   idx_host, static_cast<uint32_t*>(total_host) /* pinned: the kernel writes the total there itself */, part);
    ctx->compaction_fused = fused;
    ctx->compaction_idx_host = idx_host;
    return M3D_OK;
}

int refine(DeviceCtx* ctx, const CloudView& flag_view, const CloudView& gather_view,
                  const uint32_t* orig_dev, int kind, double thr, const double* model_dev,
                  double* params_host /* in: best minimal model, out: refined */, size_t* inliers,
                  size_t* n_inliers, int* general_fit_ok, int64_t expected_ni,
                  const std::function<int(int64_t)>* before_wait,
                  const double* lazy_in /* pinned: the "in" value of params_host arrives with the wait */,
                  const void* compaction_total /* pinned: the compaction is already queued (on model_dev) */,
                  bool fused /* model_dev is a minimal_fit_k record: moments ride on the compaction (needs lazy_in) */) {
    const uint32_t n = flag_view.n;
    fused = fused && lazy_in != nullptr;
    RESERVE(ctx->sums, sizeof(double) * 32);
    RESERVE(ctx->sum_partial, sizeof(double) * kSumPartialDoubles);
    RESERVE(ctx->h_sums, sizeof(double) * kGeneralFitHostDoubles);
    RESERVE(ctx->h_small, 256);
    uint8_t* h = ctx->h_small.as<uint8_t>();
    const uint8_t* h_total = compaction_total ? static_cast<const uint8_t*>(compaction_total) : h;
    if (!compaction_total) {
        uint64_t* idx_host = inliers && fused && !ctx->idx_out_override &&
                                     is_library_pinned(inliers, sizeof(uint64_t) * (size_t)std::max<uint32_t>(n, 1))
                                 ? reinterpret_cast<uint64_t*>(inliers) : nullptr;
        const PartitionOut* part = ctx->partition_hook && orig_dev ? (*ctx->partition_hook)(expected_ni) : nullptr;
        const int rc = issue_refine_compaction(ctx, flag_view, orig_dev, kind, thr, model_dev, lazy_in, h, fused, idx_host, part);
        if (rc != M3D_OK) return rc;
    }
    const bool have_moments = fused && ctx->compaction_fused;
    const bool idx_on_host = inliers && ctx->compaction_idx_host == reinterpret_cast<uint64_t*>(inliers);
    if (expected_ni >= 0 && (uint64_t)expected_ni <= n) {
        const uint32_t ni_e = (uint32_t)expected_ni;
        const bool need_fit_e = kind != M3D_CYLINDER && ni_e >= (kind == M3D_PLANE ? 3u : 4u);
        if (!(compaction_total && ctx->ev_compact_early)) HIPCHK(hipEventRecord(ctx->ev_compact, ctx->stream));
        ctx->ev_compact_early = false;
        // page-locked destination (m3d_host_alloc): the index list leaves NOW, on the copy stream, under the sums
        // (or has been written by the compaction itself: idx_on_host)
        const bool early_copy = !idx_on_host && inliers && ni_e && is_library_pinned(inliers, sizeof(uint64_t) * (size_t)ni_e);
        if (early_copy) {
            HIPCHK(hipStreamWaitEvent(copy_stream_of(ctx), ctx->ev_compact, 0));
            HIPCHK(hipMemcpyAsync(inliers, (void*)idx_dev(ctx),
                                  sizeof(uint64_t) * (size_t)ni_e, hipMemcpyDeviceToHost, copy_stream_of(ctx)));
        }
        if (need_fit_e && !have_moments) {
            launch_general_fit_sums(gather_view, idx_dev(ctx), ni_e, ctx->sum_partial.as<double>(),
                                    ctx->h_sums.as<double>(), ctx->stream);
        }
        // work the caller wants queued behind these kernels before the host waits (segmentation: the removal of
        // these very inliers), so that ONE wait covers both
        bool hooked = false;
        if (before_wait) {
            const int hr = (*before_wait)(expected_ni);
            before_wait = nullptr;
            hooked = true;
            if (hr != M3D_OK) return hr;
        }
        // last: a copy into the caller's (pageable) buffer keeps the host busy until it is done
        if (inliers && ni_e && !early_copy && !idx_on_host) {
            HIPCHK(hipStreamWaitEvent(copy_stream_of(ctx), ctx->ev_compact, 0));
            HIPCHK(hipMemcpyAsync(inliers, (void*)idx_dev(ctx),
                                  sizeof(uint64_t) * (size_t)ni_e, hipMemcpyDeviceToHost, copy_stream_of(ctx)));
        }
        HIPCHK(hipGetLastError());
        // everything RefineModel reads is complete at ev_compact when the moments rode on the compaction (or no fit is
        // due): what the hook queued behind it (segmentation: the removal of these inliers, tens of microseconds of
        // kernels) is not waited for -- the caller goes on preparing the next round under it
        if (hooked && (have_moments || !need_fit_e)) {
            HIPCHK(hipEventSynchronize(ctx->ev_compact));
        } else {
            const int wrc = stream_wait_spin(ctx);
            if (wrc != M3D_OK) return wrc;
        }
        // (the copy stream is waited for when THIS call put the list on it -- and not even then when the caller collects
        // its lists at the end: DeviceCtx::defer_copy_sync)
        const bool list_on_copy_stream = inliers && ni_e && !idx_on_host;
        if (list_on_copy_stream && !(ctx->defer_copy_sync && ctx->idx_out_override)) HIPCHK(hipStreamSynchronize(copy_stream_of(ctx)));
        if (lazy_in) std::memcpy(params_host, lazy_in, sizeof(double) * kModelStride);
        uint32_t ni_chk;
        std::memcpy(&ni_chk, h_total, 4);
        if (ni_chk != ni_e)   // should not happen: redo in the order that does not rely on the expectation
            return refine(ctx, flag_view, gather_view, orig_dev, kind, thr, model_dev, params_host, inliers, n_inliers,
                          general_fit_ok, -1, nullptr, nullptr, nullptr);
        *n_inliers = ni_e;
        *general_fit_ok = 1;
        if (kind != M3D_CYLINDER) {
            if (!need_fit_e) {
                *general_fit_ok = 0;  // MinimalCheck, ransac.h:166-169, 298-301
            } else {
                double sums[14];
                double mean[3];
                if (have_moments) {
                    // raw moments about the record's provisional centre -> mean + centred moments (m3d_kernels.hip)
                    const double* rec = lazy_in;
                    const double c0[3] = {kind == M3D_PLANE ? rec[4] : rec[0], kind == M3D_PLANE ? rec[5] : rec[1],
                                          kind == M3D_PLANE ? rec[6] : rec[2]};
                    moments_about_mean(h_moments_at(ctx), c0, (double)ni_e, mean, sums + 4);
                } else {
                    general_fit_sums_finish(ctx->h_sums.as<double>(), sums);
                    for (int k = 0; k < 3; ++k) mean[k] = sums[k] / (double)ni_e;
                }
                double out[4];
                const bool ok = kind == M3D_PLANE ? plane_from_moments(mean, sums + 4, out)
                                                  : sphere_from_moments(mean, sums + 4, (double)ni_e, out);
                if (ok)
                    std::memcpy(params_host, out, sizeof(out));  // model refined in place
                else
                    *general_fit_ok = 0;  // model left as the best minimal model (ransac.h:204-207)
            }
        }
        return M3D_OK;
    }
    HIPCHK(hipGetLastError());
    if (before_wait) {
        const int hr = (*before_wait)(-1);
        before_wait = nullptr;
        if (hr != M3D_OK) return hr;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (lazy_in) std::memcpy(params_host, lazy_in, sizeof(double) * kModelStride);
    uint32_t ni;
    std::memcpy(&ni, h_total, 4);
    *n_inliers = ni;
    *general_fit_ok = 1;
    const bool need_fit = kind != M3D_CYLINDER;  // cylinder GeneralFit is a no-op, ransac.h:427-433
    const uint32_t min_pts = kind == M3D_PLANE ? 3 : 4;
    if (need_fit) {
        if (ni < min_pts) {
            *general_fit_ok = 0;  // MinimalCheck, ransac.h:166-169, 298-301
        } else {
            launch_general_fit_sums(gather_view, idx_dev(ctx), ni, ctx->sum_partial.as<double>(),
                                    ctx->h_sums.as<double>(), ctx->stream);
        }
    }
    if (inliers && ni && !idx_on_host)
        HIPCHK(hipMemcpyAsync(inliers, (void*)idx_dev(ctx), sizeof(uint64_t) * (size_t)ni,
                              hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (need_fit && *general_fit_ok) {
        double sums[14];
        general_fit_sums_finish(ctx->h_sums.as<double>(), sums);
        const double mean[3] = {sums[0] / (double)ni, sums[1] / (double)ni, sums[2] / (double)ni};
        double out[4];
        bool ok;
        if (kind == M3D_PLANE)
            ok = plane_from_moments(mean, sums + 4, out);
        else
            ok = sphere_from_moments(mean, sums + 4, (double)ni, out);
        if (ok)
            std::memcpy(params_host, out, sizeof(out));  // model refined in place
        else
            *general_fit_ok = 0;  // model left as the best minimal model (ransac.h:204-207)
    }
    return M3D_OK;
}

}  // namespace m3d

using namespace m3d;

extern "C" {

int m3d_cloud_exact_error(m3d_cloud* c, int kind, double threshold, const double* model,
                          uint64_t* count, double* error) {
    if (!c || kind < 0 || kind > 2 || !model || !count || !error)
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    DeviceCtx* ctx = c->ctx;
    CtxLock lock(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    RESERVE(ctx->small, 256);
    double tmp[kModelStride] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::memcpy(tmp, model, sizeof(double) * num_params(kind));
    HIPCHK(hipMemcpyAsync(ctx->small.p, tmp, sizeof(tmp), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // tmp is a stack buffer
    return exact_error(ctx, c->view(), kind, threshold, ctx->small.as<double>(), count, error);
}

int m3d_cloud_refine_expect(m3d_cloud* c, int kind, double threshold, double* params, int64_t expected_inliers,
                            size_t* inliers, size_t* n_inliers) {
    if (!c || kind < 0 || kind > 2 || !params || !n_inliers)
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    DeviceCtx* ctx = c->ctx;
    CtxLock lock(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    RESERVE(ctx->small, 256);
    RESERVE(ctx->h_small, 256);
    double tmp[kModelStride] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::memcpy(tmp, params, sizeof(double) * num_params(kind));
    // staged through pinned memory (bytes 192.. of h_small; refine() uses the first 128): no host wait before the launches
    std::memcpy(ctx->h_small.as<uint8_t>() + 192, tmp, sizeof(tmp));
    HIPCHK(hipMemcpyAsync(ctx->small.p, ctx->h_small.as<uint8_t>() + 192, sizeof(tmp), hipMemcpyHostToDevice, ctx->stream));
    int gf = 1;
    const CloudView v = c->view();
    const int rc = refine(ctx, v, c->base_view(), c->orig(), kind, threshold, ctx->small.as<double>(), tmp, inliers,
                          n_inliers, &gf, expected_inliers);
    if (rc != M3D_OK) return rc;
    std::memcpy(params, tmp, sizeof(double) * num_params(kind));
    return gf ? M3D_OK : M3D_FALSE;
}
int m3d_cloud_refine(m3d_cloud* c, int kind, double threshold, double* params, size_t* inliers,
                     size_t* n_inliers) {
    return m3d_cloud_refine_expect(c, kind, threshold, params, -1, inliers, n_inliers);
}


}  // extern "C"
