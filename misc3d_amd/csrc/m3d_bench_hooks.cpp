// m3d_bench_hooks.cpp -- measurement hooks of include/misc3d_amd_bench.h that need nothing of the fit's internals
// (m3d_bench_time_score / m3d_bench_plane_upper_bounds live next to issue_chunk in m3d_fit.cpp).  Not part of the boundary.
#include "m3d_driver_internal.hpp"

#pragma clang fp contract(off)

using namespace m3d;

extern "C" {

int m3d_bench_last_segment_ms(double out[6]) {
    if (!out) return fail(M3D_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < 6; ++k) out[k] = g_seg_ms[k];
    return M3D_OK;
}

int m3d_bench_cloud_setup_ms(const m3d_cloud* c, double out[5]) {
    if (!c || !out) return fail(M3D_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < 5; ++k) out[k] = c->setup_ms[k];
    return M3D_OK;
}

int m3d_bench_fp64_issue_rate(int device, double ms_target, double* tops, double* ms_measured) {
    if (!tops) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    LaneLock lane(device);
    DeviceCtx* ctx = lane.ctx;
    if (!ctx) return M3D_ERR_DEVICE;
    HIPCHK(hipSetDevice(ctx->device));
    RESERVE(ctx->small, 256);
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: every SIMD holds 8 waves
    // one wave issues 16 * iters instructions of 4 cycles; a SIMD interleaves its 8 waves
    auto run = [&](int iters, float* ms) -> int {
        HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
        launch_fp64_issue_probe(ctx->small.as<double>(), blocks, iters, ctx->stream);
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
        return M3D_OK;
    };
    float ms = 0;
    int rc = run(256, &ms);   // warm-up + calibration
    if (rc != M3D_OK) return rc;
    rc = run(2048, &ms);
    if (rc != M3D_OK) return rc;
    const double per_iter = (double)ms / 2048.0;
    const int iters = (int)std::min(4.0e6, std::max(1024.0, (ms_target > 0 ? ms_target : 2.0) / std::max(per_iter, 1e-9)));
    rc = run(iters, &ms);
    if (rc != M3D_OK) return rc;
    const double ops = (double)blocks * 256.0 * 16.0 * (double)iters;
    *tops = ops / ((double)ms * 1e-3) / 1e12;
    if (ms_measured) *ms_measured = ms;
    return M3D_OK;
}

int m3d_bench_mfma_probe(int device, const double* xyz512, const double box[6], double max_abs, const double* records, size_t n_h,
                         double* out_q, double* out_h, float* out_off) {
    if (!xyz512 || !box || !records || !n_h || !out_q || !out_h || !out_off || n_h > (1u << 20)) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
#ifndef M3D_EXPERIMENTAL
    (void)device;
    (void)max_abs;
    return fail(M3D_ERR_INVALID_ARG, "the MFMA screen is compiled with -DM3D_EXPERIMENTAL only (m3d_bench_experimental() == 0)");
#else
    LaneLock lane(device);
    DeviceCtx* ctx = lane.ctx;
    if (!ctx) return M3D_ERR_DEVICE;
    HIPCHK(hipSetDevice(ctx->device));
    struct Bufs {   // (a test hook: its scratch does not outlive the call)
        DevBuf pts, box, rec, q, h, off;
        ~Bufs() {
            pts.release(); box.release(); rec.release(); q.release(); h.release(); off.release();
        }
    } bufs;
    DevBuf &d_pts = bufs.pts, &d_box = bufs.box, &d_rec = bufs.rec, &d_q = bufs.q, &d_h = bufs.h, &d_off = bufs.off;
    RESERVE(d_pts, sizeof(double) * 512 * 3);
    RESERVE(d_box, sizeof(double) * 6);
    RESERVE(d_rec, sizeof(double) * kModelStride * n_h);
    RESERVE(d_q, sizeof(double) * 1024 * n_h);
    RESERVE(d_h, sizeof(double) * 3 * n_h);
    RESERVE(d_off, sizeof(float) * 512 * 3);
    HIPCHK(hipMemcpyAsync(d_pts.p, xyz512, sizeof(double) * 512 * 3, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_box.p, box, sizeof(double) * 6, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_rec.p, records, sizeof(double) * kModelStride * n_h, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(d_q.p, 0xFF, sizeof(double) * 1024 * n_h, ctx->stream));
    HIPCHK(hipMemsetAsync(d_h.p, 0xFF, sizeof(double) * 3 * n_h, ctx->stream));
    launch_mfma_probe(d_pts.as<double>(), d_box.as<double>(), max_abs, d_rec.as<double>(), (uint32_t)n_h, d_q.as<double>(),
                      d_h.as<double>(), d_off.as<float>(), ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_q, d_q.p, sizeof(double) * 1024 * n_h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(out_h, d_h.p, sizeof(double) * 3 * n_h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(out_off, d_off.p, sizeof(float) * 512 * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return M3D_OK;
#endif
}

int m3d_bench_experimental(void) { return kExperimentalBuild ? 1 : 0; }


}  // extern "C"
