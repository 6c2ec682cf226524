// m3d_bound.hip -- histogram upper bounds for plane hypotheses (round 4).
//
// Bound-and-prune (keep_mask_k) drops a hypothesis whose touched tiles cannot hold the incumbent's count: 512 points per
// tile the box tests could not exclude.  On a cloud with a dominant plane that bound is useless for the hypotheses that
// matter -- ~1400 of C2's 10 000 have all three samples on the plane, touch the same ~1000 tiles and are then counted point
// by point, although all but a few hundred of them end tens of thousands of inliers below the incumbent.  A tile that lies
// on a surface is THIN along one direction, and a plane hypothesis cuts a slab out of that direction:
//
//   tile_frames_k   once per resident cloud (lazily, at its first long plane fit): one wave per tile of the sorted copy
//                   finds a robust local frame (c; e, u, v) -- least-quantile search over 64 candidate planes through
//                   triples of the tile's own points, then three rounds of trimmed PCA -- and stores, for w_p = e . (p - c),
//                   a cumulative HISTOGRAM of the tile's points over 128 bins of w together with U >= |u . (p - c)|,
//                   V >= |v . (p - c)|, W >= |w_p| and R >= the residual of the decomposition.
//   plane_bound_k   per fit window, after the keep masks: lane = surviving hypothesis, tiles stream through scalar loads.
//                   For every touched (tile, hypothesis) pair, with S(p) = n . p + d the plane value,
//                       S(p) = S(c) + (n . e) w_p + (n . u) u_p + (n . v) v_p + n . res_p            (an identity),
//                   so |S(p)| < T forces (n . e) w_p into an interval of half width T + a around -S(c),
//                   a = |n . u| U + |n . v| V + |n|_1 R + rounding terms: the histogram's mass over the bins that interval
//                   meets is an UPPER BOUND of the pair's inlier count (every rounding of the evaluation is inside a; the bins
//                   are monotone in w and taken one further out on both sides).  Summed per hypothesis: ubsum[h].
//                   The workgroup that finishes a block of 64 hypotheses last (a ticket per block) applies the keep rule:
//                   keep bit off where ubsum[h] < best count of EARLIER hypotheses -- keep_mask_k's rule with a bound that is
//                   within ~10 % of the true count for hypotheses near a surface instead of 512 per tile.
//
// A hypothesis dropped here has count <= ubsum < best count of hypotheses before it: it can neither beat nor tie the incumbent
// in the sequential replay (ransac.h:595-596); its record is reported as 0 like any pruned hypothesis'.  Nothing else
// changes: the survivors are counted by the same kernels.  Dead points (tombstones) only lower counts: the bound stays valid.
// Frames belong to the tiles they were built on: a sorted copy that is re-partitioned drops them (SortedView::frames = null).
#include "m3d_cull_kernels.hpp"

#include <cstdlib>

#include "m3d_bound_fp.hpp"
#include "m3d_config.hpp"
#include "m3d_eig3.hpp"
#include "m3d_fp.hpp"

#pragma clang fp contract(off)

namespace m3d {

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

constexpr int kFrameLevels = 12;   // residual levels of the candidate search: factor 2 each, from extent / 2^11 to extent

__global__ __launch_bounds__(64) void tile_frames_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                     const double* __restrict__ sz, uint32_t n_tiles,
                                                     double* __restrict__ frames, uint16_t* __restrict__ cum) {
    __shared__ float pf[3][kTilePoints];
    __shared__ uint32_t hist[kBoundBins + 2];
    const int lane = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    constexpr int P = kTilePoints / 64;
    double px[P], py[P], pz[P];
    bool fin[P];
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, mabs = 0.0, nf = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const size_t i = (size_t)tile * kTilePoints + j * 64 + lane;
        px[j] = sx[i];
        py[j] = sy[i];
        pz[j] = sz[i];
        fin[j] = (px[j] * 0.0 == 0.0) && (py[j] * 0.0 == 0.0) && (pz[j] * 0.0 == 0.0);
        if (fin[j]) {
            lo[0] = fmin(lo[0], px[j]); hi[0] = fmax(hi[0], px[j]);
            lo[1] = fmin(lo[1], py[j]); hi[1] = fmax(hi[1], py[j]);
            lo[2] = fmin(lo[2], pz[j]); hi[2] = fmax(hi[2], pz[j]);
            mabs = fmax(mabs, fmax(fabs(px[j]), fmax(fabs(py[j]), fabs(pz[j]))));
            nf += 1.0;
        }
    }
    for (int k = 0; k < 3; ++k) {
        lo[k] = wave_min(lo[k]);
        hi[k] = wave_max(hi[k]);
    }
    mabs = wave_max(mabs);
    nf = wave_sum(nf);
    double* __restrict__ fr = frames + (size_t)tile * kFrameStride;
    uint16_t* __restrict__ cm = cum + (size_t)tile * kCumStride;
    const double ext = fmax(hi[0] - lo[0], fmax(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(nf >= 16.0) || !(ext > 1e-290) || !(ext < 1e30) || !(mabs < 1e30)) {   // (wave-uniform) no frame: every touched pair counts 512
        if (lane < kFrameStride) fr[lane] = 0.0;
        return;
    }
    const double bc[3] = {0.5 * lo[0] + 0.5 * hi[0], 0.5 * lo[1] + 0.5 * hi[1], 0.5 * lo[2] + 0.5 * hi[2]};
    const float qnan = __uint_as_float(0x7FC00000u);
#pragma unroll
    for (int j = 0; j < P; ++j) {
        pf[0][j * 64 + lane] = fin[j] ? (float)(px[j] - bc[0]) : qnan;
        pf[1][j * 64 + lane] = fin[j] ? (float)(py[j] - bc[1]) : qnan;
        pf[2][j * 64 + lane] = fin[j] ? (float)(pz[j] - bc[2]) : qnan;
    }
    for (int i = lane; i < kBoundBins + 2; i += 64) hist[i] = 0u;
    __syncthreads();
    // ---- least-quantile search: lane = candidate plane through three of the tile's points (pseudo-random positions along the
    // Hilbert order: a tile that is part surface, part clutter consists of runs), scored by the smallest residual level
    // that holds a fifth of the points, then by how many it holds
    const uint32_t i0 = ((uint32_t)lane * 8u + 3u) & 511u;
    const uint32_t i1 = (i0 + 29u + (((uint32_t)lane * 2654435761u) >> 16) % 170u) & 511u;
    const uint32_t i2 = (i1 + 31u + ((((uint32_t)lane * 40503u + 77u) * 2246822519u) >> 16) % 170u) & 511u;
    const float q0[3] = {pf[0][i0], pf[1][i0], pf[2][i0]};
    const float ax = pf[0][i1] - q0[0], ay = pf[1][i1] - q0[1], az = pf[2][i1] - q0[2];
    const float bx = pf[0][i2] - q0[0], by = pf[1][i2] - q0[1], bz = pf[2][i2] - q0[2];
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float len = __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
    const float extf = (float)ext;
    const bool cand_ok = len > 1e-6f * extf * extf && len * 0.0f == 0.0f;
    const float il = cand_ok ? 1.0f / len : 0.0f;
    nx *= il;
    ny *= il;
    nz *= il;
    const float d0 = nx * q0[0] + ny * q0[1] + nz * q0[2];
    int ex;
    (void)__builtin_frexp(ext, &ex);                  // ext = f 2^ex, f in [0.5, 1): ilogb(ext) = ex - 1
    const int emin = (ex - 1) - kFrameLevels + 1;     // level l: residuals in [2^(emin + l), 2^(emin + l + 1)); 0 also takes everything below
    unsigned long long clo = 0ull, chi = 0ull;        // six 10-bit counters each (a count is at most 512)
    for (int i = 0; i < kTilePoints; ++i) {
        const float r = __builtin_fabsf(__builtin_fmaf(nx, pf[0][i], __builtin_fmaf(ny, pf[1][i], nz * pf[2][i])) - d0);
        const int eb = (int)((__float_as_uint(r) >> 23) & 0xFFu) - 127;   // (NaN: 128 -> the top level, which holds every point anyway)
        const int lev = min(max(eb - emin, 0), kFrameLevels - 1);
        const bool up = lev >= 6;
        const unsigned long long one = 1ull << (10 * (up ? lev - 6 : lev));
        clo += up ? 0ull : one;
        chi += up ? one : 0ull;
    }
    const uint32_t need = max(16u, (uint32_t)(0.2 * nf));
    uint32_t acc = 0, lev_s = kFrameLevels - 1, cnt_s = 0;
    bool found = false;
#pragma unroll
    for (int l = 0; l < kFrameLevels; ++l) {
        acc += (uint32_t)(((l >= 6 ? chi : clo) >> (10 * (l >= 6 ? l - 6 : l))) & 1023ull);
        if (!found && acc >= need) {
            found = true;
            lev_s = (uint32_t)l;
            cnt_s = acc;
        }
    }
    uint32_t key = (cand_ok && found) ? ((lev_s << 16) | ((1023u - min(cnt_s, 1023u)) << 6) | (uint32_t)lane) : 0xFFFFFFFFu;
    for (int off = 32; off > 0; off >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, off, 64));
    double e[3] = {0.0, 0.0, 1.0}, c[3] = {bc[0], bc[1], bc[2]}, band = INFINITY;
    if (key != 0xFFFFFFFFu) {   // (wave-uniform)
        const int kb = (int)(key & 63u);
        e[0] = (double)__shfl(nx, kb, 64);
        e[1] = (double)__shfl(ny, kb, 64);
        e[2] = (double)__shfl(nz, kb, 64);
        c[0] = bc[0] + (double)__shfl(q0[0], kb, 64);
        c[1] = bc[1] + (double)__shfl(q0[1], kb, 64);
        c[2] = bc[2] + (double)__shfl(q0[2], kb, 64);
        band = 3.0 * __builtin_ldexp(1.0, emin + (int)(key >> 16) + 1);
    }
    // ---- trimmed PCA: the points within `band` of the current plane give the next one; band = 3.5 rms afterwards
    double s = 0.25 * ext;
    for (int it = 0; it < 3; ++it) {
        double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double dx = px[j] - c[0], dy = py[j] - c[1], dz = pz[j] - c[2];
            const double w = (dx * e[0] + dy * e[1]) + dz * e[2];
            if (fin[j] && fabs(w) < band) {
                m[0] += 1.0;
                m[1] += dx; m[2] += dy; m[3] += dz;
                m[4] += dx * dx; m[5] += dx * dy; m[6] += dx * dz;
                m[7] += dy * dy; m[8] += dy * dz; m[9] += dz * dz;
            }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) m[k] = wave_sum(m[k]);
        if (!(m[0] >= 8.0)) break;   // (wave-uniform)
        const double in = 1.0 / m[0];
        const double mx = m[1] * in, my = m[2] * in, mz = m[3] * in;
        const double A[9] = {m[4] * in - mx * mx, m[5] * in - mx * my, m[6] * in - mx * mz,
                             m[5] * in - mx * my, m[7] * in - my * my, m[8] * in - my * mz,
                             m[6] * in - mx * mz, m[8] * in - my * mz, m[9] * in - mz * mz};
        double en[3];
        j3x3_smallest_eigvec(A, en);
        if (!((en[0] + en[1] + en[2]) * 0.0 == 0.0)) break;   // (degenerate covariance: keep the frame in hand)
        e[0] = en[0]; e[1] = en[1]; e[2] = en[2];
        c[0] += mx; c[1] += my; c[2] += mz;
        const double q = (e[0] * (A[0] * e[0] + A[1] * e[1] + A[2] * e[2]) + e[1] * (A[3] * e[0] + A[4] * e[1] + A[5] * e[2])) +
                         e[2] * (A[6] * e[0] + A[7] * e[1] + A[8] * e[2]);
        s = sqrt(fmax(q, 0.0));
        s = fmax(s, 1e-7 * ext);
        band = 3.5 * s;
    }
    // ---- the frame's other two directions (any orthonormal completion serves the identity)
    const int kmin = (fabs(e[0]) <= fabs(e[1]) && fabs(e[0]) <= fabs(e[2])) ? 0 : (fabs(e[1]) <= fabs(e[2]) ? 1 : 2);
    const double a3[3] = {kmin == 0 ? 1.0 : 0.0, kmin == 1 ? 1.0 : 0.0, kmin == 2 ? 1.0 : 0.0};
    double u[3] = {e[1] * a3[2] - e[2] * a3[1], e[2] * a3[0] - e[0] * a3[2], e[0] * a3[1] - e[1] * a3[0]};
    const double ul = sqrt((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);
    u[0] /= ul; u[1] /= ul; u[2] /= ul;
    const double v[3] = {e[1] * u[2] - e[2] * u[1], e[2] * u[0] - e[0] * u[2], e[0] * u[1] - e[1] * u[0]};
    const double wlo = -5.0 * s, invd = (double)kBoundBins / (10.0 * s);
    double U = 0.0, V = 0.0, W = 0.0, R = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        if (!fin[j]) continue;
        const double dx = px[j] - c[0], dy = py[j] - c[1], dz = pz[j] - c[2];
        const double w = (dx * e[0] + dy * e[1]) + dz * e[2];
        const double uu = (dx * u[0] + dy * u[1]) + dz * u[2];
        const double vv = (dx * v[0] + dy * v[1]) + dz * v[2];
        const double rx = ((dx - w * e[0]) - uu * u[0]) - vv * v[0];
        const double ry = ((dy - w * e[1]) - uu * u[1]) - vv * v[1];
        const double rz = ((dz - w * e[2]) - uu * u[2]) - vv * v[2];
        U = fmax(U, fabs(uu));
        V = fmax(V, fabs(vv));
        W = fmax(W, fabs(w));
        R = fmax(R, fmax(fabs(rx), fmax(fabs(ry), fabs(rz))));
        atomicAdd(&hist[bound_bin(w, wlo, invd)], 1u);
    }
    U = wave_max(U);
    V = wave_max(V);
    W = wave_max(W);
    R = wave_max(R);
    __syncthreads();
    {   // cum[k] = points with bin < k, k = 0 .. kBoundBins + 2
        const uint32_t h0 = hist[2 * lane], h1 = hist[2 * lane + 1];
        uint32_t incl = h0 + h1;
        const uint32_t own = incl;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= off) incl += t;
        }
        const uint32_t excl = incl - own;
        cm[2 * lane] = (uint16_t)excl;
        cm[2 * lane + 1] = (uint16_t)(excl + h0);
        if (lane == 63) cm[kBoundBins + 2] = (uint16_t)incl;
    }
    // the residual as computed carries its own rounding (a dozen operations on numbers of size <= mabs + ext)
    const double Rm = R + 1e-13 * (mabs + ext);
    // (plane_bound_k evaluates in fp32: sizes it can hold without leaving the normal range)
    const bool ok = (U + V + W + Rm) < 1e12 && invd > 0.0 && invd < 1e30 && ext > 1e-12 && mabs < 1e12;
    if (lane == 0) {
        fr[0] = c[0]; fr[1] = c[1]; fr[2] = c[2];
        fr[3] = e[0]; fr[4] = e[1]; fr[5] = e[2];
        fr[6] = u[0]; fr[7] = u[1]; fr[8] = u[2];
        fr[9] = v[0]; fr[10] = v[1]; fr[11] = v[2];
        fr[12] = U; fr[13] = V; fr[14] = Rm; fr[15] = wlo; fr[16] = invd; fr[17] = W; fr[18] = nf;
        fr[19] = ok ? 1.0 : 0.0;
    }
}

void launch_tile_frames(const SortedView& s, double* frames, uint16_t* cum, hipStream_t st) {
    if (s.n_tiles) tile_frames_k<<<s.n_tiles, 64, 0, st>>>(s.x, s.y, s.z, s.n_tiles, frames, cum);
}

// One workgroup (eight waves) = 64 hypotheses of the survivor list (written by the keep kernels: emit_survivors; the blocks of 64
// are dealt out along block x, grid-stride) x a range of tiles (block y), the waves taking the tiles in turn.  The frames and
// histograms and fp32 boxes of the range go to LDS in one cooperative sweep, the 64 hypotheses' records in another (eight lanes
// to a record), and the loop reads LDS only; a tile none of the 64 hypotheses touches is skipped.
//
// Arithmetic and its margins: m3d_bound_fp.hpp (plane_pair_ub: the value at the tile's centre in fp64, the rest in fp32 with every
// number pushed outwards; tests/cpp/test_plane_bound.cpp runs the same code on the host against exact counts).
constexpr int kBoundWaves = 8;   // (4 waves x 16 tiles each: step 0.2594-0.2621 ms; 8 x 8: 0.2570-0.2572)
template <int KIND /* 0 plane, 1 sphere, 2 cylinder (cyl_pair_ub: the shell as a slab per tile) */, int kBoundTpw /* tiles per wave */>
__global__ __launch_bounds__(64 * kBoundWaves) void plane_bound_k(const double* __restrict__ frames, const uint16_t* __restrict__ cum,
                                                                   uint32_t n_tiles, double max_abs,
                                                                   const double* __restrict__ score,
                                                                   const unsigned long long* __restrict__ masks, uint32_t n_groups,
                                                                   const double* __restrict__ boxes, const float* __restrict__ cull32,
                                                                   uint32_t* __restrict__ surv_count,
                                                                   const uint32_t* __restrict__ surv,
                                                                   uint32_t* __restrict__ ubsum, const uint32_t* __restrict__ best_count,
                                                                   unsigned long long* __restrict__ keep,
                                                                   uint32_t* __restrict__ tickets /* [0]: finished blocks; [1 + block] */,
                                                                   uint32_t max_list) {
    constexpr int kBoundTpb = kBoundWaves * kBoundTpw;     // tiles per workgroup
    __shared__ float bx_s[kBoundTpb][6];                   // the tile's fp32 box (cull_tiles32_k's)
    __shared__ double c_s[kBoundTpb][3];
    __shared__ float f_s[kBoundTpb][kFrameStride];         // slots 3 .. 19 of the frame in fp32 (U, V, R, W rounded up)
    __shared__ __attribute__((aligned(8))) uint16_t cm_s[kBoundTpb][kCumStride];
    __shared__ uint32_t wsum[kBoundWaves][64];
    __shared__ double rec_s[64][9];    // (+ 1: the lanes' rows fall into different banks)
    __shared__ float q_s[64][13];   // the hypothesis' fp32 box-test record: 8 words (planes) / 11 (cylinders); odd stride: the lanes' rows fall into different banks
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t total = surv_count[0];
    if (total > max_list) {   // (uniform) the old rule kept most of the window: no incumbent worth the name, or a cloud without
        // structure -- the bound would be computed for thousands of hypotheses to drop none (a launch of ~7 us per 1000
        // of them); the list is discarded, every keep bit stays.  (Workgroup (0, 0) resets the count while others may still be
        // reading it: harmless ONLY because every outcome of that read returns at once -- `total` too long: here; 0: the next
        // line.  Nothing may be done with `total` in front of these two exits: ADVICE r4.)
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) surv_count[0] = 0u;
        return;
    }
    if (blockIdx.x * 64u >= total) return;   // (workgroup-uniform)
    const uint32_t t0 = blockIdx.y * (uint32_t)kBoundTpb, nt = min((uint32_t)kBoundTpb, n_tiles - t0);
    {
        const double* __restrict__ src = frames + (size_t)t0 * kFrameStride;
        for (uint32_t i = threadIdx.x; i < nt * (uint32_t)kFrameStride; i += 64u * kBoundWaves) {
            const uint32_t tl = i / (uint32_t)kFrameStride, k = i % (uint32_t)kFrameStride;
            const double v = src[i];
            if (k < 3u) c_s[tl][k] = v;
            f_s[tl][k] = frame_to_f32(v, k);
        }
        if (KIND != 0) {   // spheres, cylinders: the tile's bounding radius about its centre, once per tile (slot 0: the centre's fp32 copy is not used)
            __syncthreads();
            if (threadIdx.x < nt) f_s[threadIdx.x][0] = cyl_tile_rho(f_s[threadIdx.x]);
        }
        if (cull32)
            for (uint32_t i = threadIdx.x; i < nt * 6u; i += 64u * kBoundWaves)
                bx_s[i / 6u][i % 6u] = reinterpret_cast<const float*>(boxes + (size_t)(t0 + i / 6u) * kBoxStride + 8)[i % 6u];
        const uint32_t* __restrict__ csrc = reinterpret_cast<const uint32_t*>(cum + (size_t)t0 * kCumStride);
        uint32_t* cdst = reinterpret_cast<uint32_t*>(&cm_s[0][0]);
        for (uint32_t i = threadIdx.x; i < nt * (uint32_t)(kCumStride / 2); i += 64u * kBoundWaves) cdst[i] = csrc[i];
    }
    for (uint32_t first = blockIdx.x * 64u; first < total; first += gridDim.x * 64u) {   // (workgroup-uniform)
        const uint32_t nb = min(64u, total - first);
        const bool has = (uint32_t)lane < nb;
        const uint32_t h = surv[first + (has ? (uint32_t)lane : 0u)];
        // the 64 hypotheses' records reach the eight waves through LDS, fetched ONCE and eight lanes to a record: every wave
        // reading its own lane's record took 13 load instructions of 64 cache lines each, four times over (3.6 M line
        // requests per launch)
        __syncthreads();   // (the previous block's records and sums have been read)
        {
            const uint32_t k = threadIdx.x & 7u;   // word k of the record of survivors threadIdx.x / 8, + threads / 8, ...
#pragma unroll
            for (uint32_t si = threadIdx.x >> 3; si < 64u; si += (64u * kBoundWaves) >> 3) {
                const uint32_t hh = surv[first + min(si, nb - 1u)];
                rec_s[si][k] = score[(size_t)hh * kModelStride + k];
                if (cull32) {
                    q_s[si][k] = cull32[(size_t)(hh >> 1) * 24u + (hh & 1u) + 2u * k];
                    if (KIND == 2 && k < 3u) q_s[si][8u + k] = cull32[(size_t)(hh >> 1) * 24u + (hh & 1u) + 2u * (8u + k)];
                }
            }
        }
        __syncthreads();   // (records -- and, the first time round, the frames -- are in LDS)
        PlaneBoundRec pr;
        CylBoundRec cr;
        if (KIND == 0) {
            pr = plane_bound_record(rec_s[lane], max_abs);
            pr.ok = pr.ok && has;
        } else {
            cr = KIND == 1 ? sphere_bound_record(rec_s[lane], max_abs) : cyl_bound_record(rec_s[lane], max_abs);
            cr.ok = cr.ok && has;
        }
        const unsigned long long bit = 1ull << (h & 63u);
        const unsigned long long* __restrict__ mrow = masks + (size_t)(h >> 6);
        // touched or not: the box test itself, from the hypothesis' fp32 record and the tile's fp32 box (the arithmetic of
        // cull32_one<0>, m3d_cull_kernels.hip: the bits cull_tiles32_k wrote) -- the mask words are 8 useful bytes per 128-byte
        // line for this kernel's lanes; without fp32 records (m3d_config.cull_fp32 = 0) it reads them
        bool tch[kBoundTpw];
        float q[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (cull32) {   // (kernel argument: uniform)
#pragma unroll
            for (int k = 0; k < (KIND == 2 ? 11 : 8); ++k) q[k] = q_s[lane][k];
        } else {
#pragma unroll
            for (int i = 0; i < kBoundTpw; ++i) {
                const uint32_t tl = (uint32_t)(wave + kBoundWaves * i);
                tch[i] = (has && tl < nt) ? (mrow[(size_t)(t0 + tl) * n_groups] & bit) != 0ull : false;
            }
        }
        uint32_t ub = 0;
#pragma unroll
        for (int i = 0; i < kBoundTpw; ++i) {
            const int tl = wave + kBoundWaves * i;
            if (cull32 && KIND == 0) {
                const float* bx = bx_s[tl];
                const float sv = __builtin_fmaf(q[0], bx[0], __builtin_fmaf(q[1], bx[1], __builtin_fmaf(q[2], bx[2], q[3])));
                const float hx = __builtin_fmaxf(bx[3], 0.0f), hy = __builtin_fmaxf(bx[4], 0.0f), hz = __builtin_fmaxf(bx[5], 0.0f);
                const float rv = __builtin_fmaf(q[4], hx, __builtin_fmaf(q[5], hy, __builtin_fmaf(q[6], hz, q[7])));
                tch[i] = has && (uint32_t)tl < nt && bx[3] >= 0.0f && !(rv - __builtin_fabsf(sv) < 0.0f);
            } else if (cull32 && KIND == 2) {
                // cull32_one<2> (m3d_cull_kernels.hip), operation for operation: the bit cull_tiles32_k wrote for this (tile,
                // hypothesis) -- round 6: until now the cylinders read the mask words (16 gathers of 8 useful bytes per line and
                // wave, a chain of dependent loads per block of 64 survivors: 61 us per C3 window)
                const float* bx = bx_s[tl];
                const bool live = bx[3] >= 0.0f;
                const float hx = live ? bx[3] : 0.0f, hy = live ? bx[4] : 0.0f, hz = live ? bx[5] : 0.0f;
                const float rb = __builtin_sqrtf(__builtin_fmaf(hz, hz, __builtin_fmaf(hy, hy, hx * hx))) * 1.000001f;
                const float d1 = __builtin_fmaf(q[0], bx[0], __builtin_fmaf(q[1], bx[1], __builtin_fmaf(q[2], bx[2], q[3])));
                const float d2 = __builtin_fmaf(q[4], bx[0], __builtin_fmaf(q[5], bx[1], __builtin_fmaf(q[6], bx[2], q[7])));
                const float tt = __builtin_fmaf(d2, d2, d1 * d1);
                const float dist = __builtin_sqrtf(tt);
                const float rt = q[8] * rb;
                const float t1 = (q[9] + rt) - dist;
                const float t2 = (dist + rt) - q[10];
                const float t = __uint_as_float(__float_as_uint(t1) | __float_as_uint(t2));
                tch[i] = has && (uint32_t)tl < nt && live && !(t < 0.0f);
            }
            if (__ballot(tch[i]) == 0ull) continue;   // (wave-uniform)
            const uint32_t u_t = KIND == 0 ? plane_pair_ub(pr, c_s[tl], f_s[tl], cm_s[tl])   // (wave-uniform addresses: broadcast reads)
                                           : cyl_pair_ub(cr, c_s[tl], f_s[tl], f_s[tl][0], cm_s[tl]);
            ub += tch[i] ? u_t : 0u;
        }
        wsum[wave][lane] = ub;
        __syncthreads();
        if (wave == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < kBoundWaves; ++w) tot += wsum[w][lane];
            if (has && tot) atomicAdd(&ubsum[h], tot);
            // the keep rule, by whichever workgroup of this block of 64 hypotheses finishes last (a ticket per block: nobody
            // waits): keep bit off where the bound stays below the best count of earlier hypotheses
            __threadfence();
            const uint32_t blk = first / 64u;
            uint32_t old = 0;
            if (lane == 0) old = atomicAdd(&tickets[1u + blk], 1u);
            old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
            if (old == gridDim.y - 1u) {   // (wave-uniform)
                __threadfence();
                const uint32_t best = best_count[0];
                const uint32_t sum = has ? atomicAdd(&ubsum[h], 0u) : 0u;   // (the other workgroups' adds, at the memory side)
                if (has && best != 0u && sum < best) atomicAnd(&keep[h >> 6], ~bit);
                uint32_t done = 0;
                if (lane == 0) {
                    tickets[1u + blk] = 0u;
                    done = atomicAdd(&tickets[0], 1u);
                    if (done == (total + 63u) / 64u - 1u) {   // the last block of the launch: the list is consumed
                        tickets[0] = 0u;
                        surv_count[0] = 0u;
                    }
                }
            }
        }
    }
}

// plane_bound_k's sums for CYLINDERS whose box tests left a word per hypothesis and 64 tiles (cull_hyp32_k, `touched`): round 6.
// plane_bound_k gives 64 survivors to a workgroup of eight waves that share them and split the tiles, and repeats the box test of every
// (survivor, tile): on a C3 window (47 952 hypotheses, ~13 000 survivors, 1954 tiles) 104 us -- 13 of them the box tests (31 VALU
// instructions with a correctly rounded sqrt, 19 more for the box's radius), and per 64 survivors and 128 tiles a workgroup's staging of
// frames and histograms, the records through the LDS, the bound records in fp64 by all eight waves, the cross-wave sum, the ticket.
// Here a WAVE owns 64 survivors and walks a block of kCylTpb tiles whose frames the workgroup has staged once for its 512 survivors;
// a lane's touched tiles are the bits of ONE word, every record is prepared once per tile block, nothing crosses waves: 104 -> 78 us,
// which is cyl_pair_ub's ~100 instructions on every (wave of survivors, tile) some lane touches -- all of them: a survivor touches
// a tenth of the tiles.  The same cyl_pair_ub on the same pairs; the sums are integer additions: order-free.
constexpr int kCylTpb = 64;    // tiles per workgroup (one word per lane; 128: two)
__global__ __launch_bounds__(64 * kBoundWaves) void cyl_bound_words_k(const double* __restrict__ frames, const uint16_t* __restrict__ cum,
                                                                       uint32_t n_tiles, double max_abs, const double* __restrict__ score,
                                                                       uint32_t* __restrict__ surv_count, const uint32_t* __restrict__ surv,
                                                                       uint32_t* __restrict__ ubsum, const uint32_t* __restrict__ best_count,
                                                                       unsigned long long* __restrict__ keep, uint32_t* __restrict__ tickets,
                                                                       uint32_t max_list, const unsigned long long* __restrict__ touched,
                                                                       uint32_t touched_stride) {
    __shared__ double c_s[kCylTpb][3];
    __shared__ float f_s[kCylTpb][kFrameStride];
    __shared__ __attribute__((aligned(8))) uint16_t cm_s[kCylTpb][kCumStride];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t total = surv_count[0];
    if (total > max_list) {   // (uniform; plane_bound_k's rule and its reasons)
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) surv_count[0] = 0u;
        return;
    }
    if (blockIdx.x * (64u * kBoundWaves) >= total) return;   // (workgroup-uniform)
    const uint32_t t0 = blockIdx.y * (uint32_t)kCylTpb, nt = min((uint32_t)kCylTpb, n_tiles - t0);
    {
        const double* __restrict__ src = frames + (size_t)t0 * kFrameStride;
        for (uint32_t i = threadIdx.x; i < nt * (uint32_t)kFrameStride; i += 64u * kBoundWaves) {
            const uint32_t tl = i / (uint32_t)kFrameStride, k = i % (uint32_t)kFrameStride;
            const double v = src[i];
            if (k < 3u) c_s[tl][k] = v;
            f_s[tl][k] = frame_to_f32(v, k);
        }
        __syncthreads();
        if (threadIdx.x < nt) f_s[threadIdx.x][0] = cyl_tile_rho(f_s[threadIdx.x]);   // (slot 0: the centre's fp32 copy is not used)
        const uint32_t* __restrict__ csrc = reinterpret_cast<const uint32_t*>(cum + (size_t)t0 * kCumStride);
        uint32_t* cdst = reinterpret_cast<uint32_t*>(&cm_s[0][0]);
        for (uint32_t i = threadIdx.x; i < nt * (uint32_t)(kCumStride / 2); i += 64u * kBoundWaves) cdst[i] = csrc[i];
        __syncthreads();
    }
    // (no barrier below: every wave goes its own way)
    for (uint32_t first = (blockIdx.x * kBoundWaves + wave) * 64u; first < total; first += gridDim.x * (64u * kBoundWaves)) {   // (wave-uniform)
        const uint32_t nb = min(64u, total - first);
        const bool has = (uint32_t)lane < nb;
        const uint32_t h = surv[first + (has ? (uint32_t)lane : 0u)];
        double rec[8];
        {
            const double2* __restrict__ rp = reinterpret_cast<const double2*>(score + (size_t)h * kModelStride);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double2 v = rp[k];
                rec[2 * k] = v.x;
                rec[2 * k + 1] = v.y;
            }
        }
        CylBoundRec cr = cyl_bound_record(rec, max_abs);
        cr.ok = cr.ok && has;
        unsigned long long w0 = has ? touched[(size_t)(t0 / 64u) * touched_stride + h] : 0ull;
        unsigned long long w1 = (has && t0 + 64u < n_tiles) ? touched[(size_t)(t0 / 64u + 1u) * touched_stride + h] : 0ull;
        uint32_t ub = 0;
        // (measured and not kept: every lane popping ITS next touched tile -- one pair per lane and trip, the tile's frame gathered from the
        //  LDS per lane: ~20 scattered LDS reads per trip cost more than the idle lanes of a tile-uniform trip, 127 against 91 us)
        for (uint32_t tl = 0; tl < nt; ++tl) {   // (wave-uniform: the tile's frame and histogram are broadcast reads)
            const bool mine = (((tl < 64u ? w0 : w1) >> (tl & 63u)) & 1ull) != 0ull;
            if (__builtin_amdgcn_ballot_w64(mine) == 0ull) continue;
            const uint32_t u_t = cyl_pair_ub(cr, c_s[tl], f_s[tl], f_s[tl][0], cm_s[tl]);
            ub += mine ? u_t : 0u;
        }
        if (has && ub) atomicAdd(&ubsum[h], ub);
        // the keep rule, by whichever wave of this block of 64 hypotheses finishes last over the tile blocks (plane_bound_k's tail)
        __threadfence();
        const uint32_t blk = first / 64u;
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(&tickets[1u + blk], 1u);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        if (old == gridDim.y - 1u) {   // (wave-uniform)
            __threadfence();
            const uint32_t best = best_count[0];
            const uint32_t sum = has ? atomicAdd(&ubsum[h], 0u) : 0u;   // (the other workgroups' adds, at the memory side)
            if (has && best != 0u && sum < best) atomicAnd(&keep[h >> 6], ~(1ull << (h & 63u)));
            if (lane == 0) {
                tickets[1u + blk] = 0u;
                const uint32_t done = atomicAdd(&tickets[0], 1u);
                if (done == (total + 63u) / 64u - 1u) {   // the last block of the launch: the list is consumed
                    tickets[0] = 0u;
                    surv_count[0] = 0u;
                }
            }
        }
    }
}

void launch_plane_bound(int kind, const SortedView& s, const double* score, const unsigned long long* masks, unsigned long long* keep,
                        uint32_t n_groups, uint32_t group_begin, uint32_t group_end, uint32_t* ubsum,
                        const uint32_t* best_count, uint32_t* surv_count, const uint32_t* surv, uint32_t* tickets,
                        const float* cull32, hipStream_t st, bool always, const unsigned long long* touched, uint32_t touched_stride) {
    group_end = std::min(group_end, n_groups);
    if (!s.frames || !s.frame_cum || !s.n_tiles || !surv || !surv_count || !tickets || group_begin >= group_end) return;
    const uint32_t window = group_end - group_begin;
    constexpr uint32_t gdiv = 6;   // one block of 64 survivors per 6 groups of the window to begin with (4 .. 8 equal, measured r4)
    const uint32_t gx = std::min<uint32_t>(window, std::max<uint32_t>(4u, (window + gdiv - 1) / gdiv));
    // tiles per wave (12 .. 32 measured slower: profiles/r04_plane_bound.txt).  Round 5 (profiles/r05_c2_front_end.txt): 4 -- twice
    // the workgroups, half the tile loop -- 19.5 -> 31.5 us, and by ablation the tile loop is 1.5 us of the launch's 19, the adds /
    // fence / ticket behind it 6.5, the rest dependent loads of data the kernel before has just written (~2 us a round trip)
    // Cylinders: 16 -- their survivor lists are long (a quarter of a C3 window: every hypothesis with both samples on the surface),
    // so a workgroup's chain of dependent loads is worth twice the tiles: 8 / 16 / 32 tiles per wave 80 / 60 / 75 us per window.
    constexpr int tpw = 8, tpw_cyl = 16;
    const uint32_t tpb = (uint32_t)(kBoundWaves * (kind == 0 ? tpw : tpw_cyl));
    const uint32_t max_list = always ? 0xFFFFFFFFu : window * 32u;   // half of the window's hypotheses
    if (kind == 2 && touched && s.radius < 1e18) {   // a wave per 64 survivors: a workgroup per 8 groups of the window covers a list of every hypothesis
        const dim3 gw((window + kBoundWaves - 1) / kBoundWaves, (s.n_tiles + kCylTpb - 1) / kCylTpb);
        cyl_bound_words_k<<<gw, 64 * kBoundWaves, 0, st>>>(s.frames, s.frame_cum, s.n_tiles, s.max_abs, score, surv_count, surv, ubsum, best_count,
                                                           keep, tickets, max_list, touched, touched_stride);
        return;
    }
    const dim3 g(gx, (s.n_tiles + tpb - 1) / tpb), b(64 * kBoundWaves);
    if (s.radius >= 1e18 || kind == 1) cull32 = nullptr;   // (no fp32 boxes: launch_cull_mask's condition; spheres: the box tests' mask words are read)
    auto go = [&](auto kernel) {
        kernel<<<g, b, 0, st>>>(s.frames, s.frame_cum, s.n_tiles, s.max_abs, score, masks, n_groups, s.boxes, cull32, surv_count, surv,
                                ubsum, best_count, keep, tickets, max_list);
    };
    if (kind == 0) go(plane_bound_k<0, tpw>);
    else if (kind == 1) go(plane_bound_k<1, tpw_cyl>);
    else go(plane_bound_k<2, tpw_cyl>);
}

}  // namespace m3d
