// m3d_normals.hip -- misc3d::common::EstimateNormalsFromMap on gfx950 (the upstream
// producer of the normals fit_cylinder needs).
//
// Replaces src/normal_estimation.cpp:64-207 (CalcNormalsFromPointMap + SumDense):
//   nm_moments_k   :82-101   nine moment images + validity mask of the zero-padded map
//   nm_box_sum_k   :36-62    (2k+1)^2 sliding-window sums, in SumDense's summation order
//   nm_normals_k   :133-178  covariance, smallest eigenvector (J3x3, m3d_eig3.hpp), orientation to the view point
// Every step is a pass over W x H doubles: HBM-bound image work.  Layout: the padded images are stored
// COLUMN-major (idx = c * H + r).  SumDense runs a serial recurrence along each row, so the parallel
// dimension is the row index; with column-major storage the lanes of a wave (consecutive rows) read
// consecutive addresses in every step of the recurrence.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>

#include "m3d_driver.hpp"
#include "m3d_eig3.hpp"

#pragma clang fp contract(off)

namespace m3d {

constexpr int kNmImages = 10;   // x y z xx xy xz yy yz zz mask

// one thread per pixel of the (unpadded) map; r fastest, so the column-major writes are coalesced
__global__ void nm_moments_k(const double* __restrict__ xyz, uint32_t w, uint32_t h, uint32_t k, size_t WH,
                             uint32_t H, double* __restrict__ img) {
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= (size_t)w * h) return;
    const uint32_t r = (uint32_t)(t % h), c = (uint32_t)(t / h);
    const double* p = xyz + ((size_t)r * w + c) * 3;
    const double x = p[0], y = p[1], z = p[2];
    if (!(z == z)) return;   // :90 -- invalid pixels stay zero in every image
    const size_t idx = (size_t)(c + k) * H + (r + k);
    img[idx] = x;
    img[WH + idx] = y;
    img[2 * WH + idx] = z;
    img[3 * WH + idx] = x * x;
    img[4 * WH + idx] = x * y;
    img[5 * WH + idx] = x * z;
    img[6 * WH + idx] = y * y;
    img[7 * WH + idx] = y * z;
    img[8 * WH + idx] = z * z;
    img[9 * WH + idx] = 1.0;
}

// SumDense (:36-62) for padded row r (thread) of image blockIdx.y: first window summed row by row, left to
// right; every further column starts from its left neighbour and adds, row by row, (entering - leaving).
// The recurrence is a serial fp64 chain ((2k+1) dependent adds per column), so the kernel is latency-bound:
// the differences of the next DEPTH columns are kept in flight in registers while the chain runs, and a
// wave per workgroup spreads the few (rows x 10) threads over as many CUs as possible.
template <int K, int DEPTH>
__global__ __launch_bounds__(64) void nm_box_sum_k(const double* __restrict__ img, uint32_t W, uint32_t H,
                                                    size_t WH, double* __restrict__ sum) {
    constexpr uint32_t k = K;
    constexpr int N = 2 * K + 1;
    const uint32_t r = k + blockIdx.x * 64u + threadIdx.x;
    if (r >= H - k) return;
    const double* __restrict__ d = img + (size_t)blockIdx.y * WH + (r - k);   // row r - k of column 0
    double* __restrict__ o = sum + (size_t)blockIdx.y * WH + r;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j)        // rows r - k .. r + k (outer), columns 0 .. 2k (inner)
        for (int c0 = 0; c0 < N; ++c0) acc += d[(size_t)c0 * H + j];
    o[(size_t)k * H] = acc;
    const uint32_t c_end = W - k;
    if (k + 1 >= c_end) return;
    // column c: entering = column c + k, leaving = column c - k - 1 (pointers advance by H per column)
    const uint32_t last = c_end - 1;
    double df[DEPTH][N];   // ring: differences (entering - leaving) of columns c .. c + DEPTH - 1
    auto fetch = [&](int slot, uint32_t c) {
        const uint32_t cc = c < last ? c : last;   // clamped loads past the end are never consumed
        const double* __restrict__ pin = d + (size_t)(cc + k) * H;
        const double* __restrict__ pout = d + (size_t)(cc - k - 1) * H;
#pragma unroll
        for (int j = 0; j < N; ++j) df[slot][j] = pin[j] - pout[j];
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) fetch(s, k + 1 + s);
    double* __restrict__ op = o + (size_t)(k + 1) * H;
    for (uint32_t c = k + 1; c < c_end; c += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
#pragma unroll
            for (int j = 0; j < N; ++j) acc += df[s][j];   // past the end: garbage in, never stored
            if (c + s < c_end) op[(size_t)s * H] = acc;
            fetch(s, c + s + DEPTH);   // refill this slot for column c + s + DEPTH
        }
        op += (size_t)DEPTH * H;
    }
}

// any window size (no register staging)
__global__ __launch_bounds__(256) void nm_box_sum_any_k(const double* __restrict__ img, uint32_t W, uint32_t H,
                                                         uint32_t k, size_t WH, double* __restrict__ sum) {
    const uint32_t r = k + blockIdx.x * 256u + threadIdx.x;
    if (r >= H - k) return;
    const double* __restrict__ d = img + (size_t)blockIdx.y * WH;
    double* __restrict__ o = sum + (size_t)blockIdx.y * WH;
    double acc = 0.0;
    for (uint32_t r0 = r - k; r0 <= r + k; ++r0)
        for (uint32_t c0 = 0; c0 <= 2 * k; ++c0) acc += d[(size_t)c0 * H + r0];
    o[(size_t)k * H + r] = acc;
    for (uint32_t c = k + 1; c < W - k; ++c) {
        for (uint32_t r0 = r - k; r0 <= r + k; ++r0)
            acc += d[(size_t)(c + k) * H + r0] - d[(size_t)(c - k - 1) * H + r0];
        o[(size_t)c * H + r] = acc;
    }
}

__global__ void nm_normals_k(const double* __restrict__ img, const double* __restrict__ sum, uint32_t w, uint32_t h,
                             uint32_t k, size_t WH, uint32_t H, double vx, double vy, double vz,
                             double* __restrict__ normals) {
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= (size_t)w * h) return;
    const uint32_t r = (uint32_t)(t % h), c = (uint32_t)(t / h);
    const size_t idx = (size_t)(c + k) * H + (r + k);
    double* out = normals + ((size_t)r * w + c) * 3;
    if (img[9 * WH + idx] == 0.0) {   // the reference leaves these uninitialised (:133-136); NaN here
        const double nan = u2f(0x7FF8000000000000ull);
        out[0] = out[1] = out[2] = nan;
        return;
    }
    const double scale = 1.0 / sum[9 * WH + idx];
    const double hx = sum[idx] * scale, hy = sum[WH + idx] * scale, hz = sum[2 * WH + idx] * scale;
    double C[9];
    C[0] = sum[3 * WH + idx] * scale - hx * hx;
    C[1] = sum[4 * WH + idx] * scale - hx * hy;
    C[2] = sum[5 * WH + idx] * scale - hx * hz;
    C[4] = sum[6 * WH + idx] * scale - hy * hy;
    C[5] = sum[7 * WH + idx] * scale - hy * hz;
    C[8] = sum[8 * WH + idx] * scale - hz * hz;
    C[3] = C[1];
    C[6] = C[2];
    C[7] = C[5];
    double n[3];
    j3x3_smallest_eigvec(C, n);
    const double dd = ((vx - img[idx]) * n[0] + (vy - img[WH + idx]) * n[1]) + (vz - img[2 * WH + idx]) * n[2];
    if (dd < 0) {
        n[0] *= -1;
        n[1] *= -1;
        n[2] *= -1;
    }
    out[0] = n[0];
    out[1] = n[1];
    out[2] = n[2];
}

}  // namespace m3d

using namespace m3d;

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(M3D_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" int m3d_normals_from_map(const double* xyz, uint32_t w, uint32_t h, uint32_t k, const double* view_point,
                                    int device, double* normals, double* ms_device) {
    if (((!xyz || !normals) && w && h) || !view_point) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (ms_device) *ms_device = 0.0;
    if (w == 0 || h == 0) return M3D_OK;
    if ((uint64_t)w * h >= ((uint64_t)1 << 31) || k > 4096) return fail(M3D_ERR_INVALID_ARG, "map too large");
    LaneLock lane(device);
    DeviceCtx* ctx = lane.ctx;
    if (!ctx) return M3D_ERR_DEVICE;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t W = w + 2 * k, H = h + 2 * k;
    const size_t WH = (size_t)W * H, n = (size_t)w * h;
    DevBuf d_xyz, d_img, d_sum, d_nrm;
    auto done = [&](int r) {
        d_xyz.release(); d_img.release(); d_sum.release(); d_nrm.release();
        return r;
    };
    if (!d_xyz.reserve(sizeof(double) * 3 * n) || !d_img.reserve(sizeof(double) * kNmImages * WH) ||
        !d_sum.reserve(sizeof(double) * kNmImages * WH) || !d_nrm.reserve(sizeof(double) * 3 * n))
        return done(M3D_ERR_DEVICE);
    bool ok = hipMemcpyAsync(d_xyz.p, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
              hipMemsetAsync(d_img.p, 0, sizeof(double) * kNmImages * WH, ctx->stream) == hipSuccess &&
              hipMemsetAsync(d_sum.p, 0, sizeof(double) * kNmImages * WH, ctx->stream) == hipSuccess &&
              hipEventRecord(ctx->ev0, ctx->stream) == hipSuccess;
    if (ok) {
        const uint32_t nb = (uint32_t)((n + 255) / 256);
        nm_moments_k<<<nb, 256, 0, ctx->stream>>>(d_xyz.as<double>(), w, h, k, WH, H, d_img.as<double>());
        const dim3 grid64((h + 63) / 64, kNmImages), grid((h + 255) / 256, kNmImages);
        const double* im = d_img.as<double>();
        double* sm = d_sum.as<double>();
        switch (k) {   // window sizes with a compiled recurrence (python default k = 5, examples use 3)
            case 1: nm_box_sum_k<1, 8><<<grid64, 64, 0, ctx->stream>>>(im, W, H, WH, sm); break;
            case 2: nm_box_sum_k<2, 8><<<grid64, 64, 0, ctx->stream>>>(im, W, H, WH, sm); break;
            case 3: nm_box_sum_k<3, 8><<<grid64, 64, 0, ctx->stream>>>(im, W, H, WH, sm); break;
            case 4: nm_box_sum_k<4, 6><<<grid64, 64, 0, ctx->stream>>>(im, W, H, WH, sm); break;
            case 5: nm_box_sum_k<5, 4><<<grid64, 64, 0, ctx->stream>>>(im, W, H, WH, sm); break;
            case 6: nm_box_sum_k<6, 4><<<grid64, 64, 0, ctx->stream>>>(im, W, H, WH, sm); break;
            case 7: nm_box_sum_k<7, 4><<<grid64, 64, 0, ctx->stream>>>(im, W, H, WH, sm); break;
            default: nm_box_sum_any_k<<<grid, 256, 0, ctx->stream>>>(im, W, H, k, WH, sm); break;
        }
        nm_normals_k<<<nb, 256, 0, ctx->stream>>>(d_img.as<double>(), d_sum.as<double>(), w, h, k, WH, H, view_point[0],
                                                 view_point[1], view_point[2], d_nrm.as<double>());
        ok = hipEventRecord(ctx->ev1, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(normals, d_nrm.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!ok) return done(fail(M3D_ERR_DEVICE, "m3d_normals_from_map: HIP error"));
    float ms = 0;
    if (ms_device && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) *ms_device = ms;
    return done(M3D_OK);
}
