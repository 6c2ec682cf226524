// m3d_reg_cache.hip -- the validation's CANDIDATE CACHE (round 6): the nearest-neighbour search of a hypothesis' 200 000
// queries answered from registers instead of per-lane gathers.
//
//   replaces (as a filter in front of reg_validate_k): the kd-tree query of Open3D's GetRegistrationResultAndCorrespondences
//   under registration::RANSACSolver::Solve, /root/reference/src/transform_estimation.cpp:154-161
//
// Why.  reg_validate_k answers a query with a walk over a per-cell list: ~20 per-lane gather instructions and ~600 VALU
// instructions per 64 queries; three rounds of counters (profiles/r03..r05_pmc_reg_validate.txt, r05_reg_validate_findings.txt)
// say its time goes with the NUMBER of gather instructions (L1 busy 0.88) and nothing else.  But the poses that reach the
// validation are near-copies of one another (they passed the checkers), and a wave keeps its 64 source points for all the
// hypotheses of its split: under a REFERENCE pose A (the incumbent) source point p sits at xa = A p, and under a hypothesis B
// at x = B p, delta = |x - xa| of a few millimetres.  So, once per reference pose and source point:
//   * the K = 32 target points nearest to xa (fewer where the cloud is sparse), as fp32 offsets from xa, and
//   * a radius R with a CERTIFICATE: every target point that is NOT in the list lies at least R from xa
// (reg_cache_build_k: the target grid's (2B+1)^3 block around xa covers the ball of radius B h, a bisection on the radius
// finds the largest R whose ball holds <= K points).  Then for any pose B every unlisted target point is at least R - delta
// from x, and if the list's minimum is below that, it is the minimum over the WHOLE target -- what the kd-tree returns.
//
// reg_validate_cached_k holds the 32 x 3 offsets of its lane's source point in 96 VGPRs across the whole hypothesis loop
// and evaluates them with packed fp32 arithmetic, no memory access at all; the winner's distance is then formed in fp64
// from the winner's fp64 coordinates with reg_validate_k's own expression (one 32-byte gather per query), so counts AND
// sums are the same bits.  Certificates per query (s_j = fp32 squared distance to candidate j, m1 <= m2 the two smallest,
// E(s) the rounding bound below, t = R (1 - 2^-20) - |u| (1 + 2^-20), u = fl32(x - xa)):
//   coverage   t > 0 and m1 + E(m1) < t^2          the nearest target point is in the list
//   identity   m2 - E(m2) > m1 + E(m1)             candidate j1 is the strictly nearest of the list, also in fp64
//   no match   t^2 >= r^2 (1 + 2^-20) and m1 - E(m1) >= r^2 (1 + 2^-20)     nothing within the search radius at all
// A (tile, hypothesis) pair whose 256 queries all hold a certificate is written like reg_validate_k writes it; any other
// pair is flagged in `redo` and reg_validate_k walks it afterwards -- the result is the walk's, bit for bit, either way.
//
// Rounding bound.  c_j = fl32(q_j - xa) and u carry a relative 2^-24 each, |c_j|, |u| <= R: every coordinate difference
// is within e = 2^-22 R of the truth; the three products and two sums add 3 2^-24 s; packing the candidate's slot into
// the five low mantissa bits of s (so that one v_min3 tree carries the index along) 2^-18 s:
//   |s - d^2| <= 2 e (|dx| + |dy| + |dz|) + 3 e^2 + (2^-22 + 2^-18) s <= 3.5 2^-22 R sqrt(s) + ...  <=  2^-23 R^2 + 2^-17 s =: E(s)
// (AM-GM: R sqrt(s) <= R^2 / 16 + 4 s).  For R = 13 mm and a neighbour at 2.5 mm that is 1.4e-8 m of distance: 1.4 queries in
// 10^5 are near-ties, 0.4 % of the 256-query tiles.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "m3d_reg_kernels.hpp"

namespace m3d {

namespace {

__device__ __forceinline__ bool cache_cell_of(const GridDesc& g, double x, double y, double z, int lo_pad, int* ix, int* iy,
                                              int* iz) {   // (m3d_reg_kernels.hip: cell_of)
    const double fx = (x - g.ox) * g.inv_h, fy = (y - g.oy) * g.inv_h, fz = (z - g.oz) * g.inv_h;
    if (!(fx >= (double)lo_pad && fx < (double)(g.nx - lo_pad) && fy >= (double)lo_pad && fy < (double)(g.ny - lo_pad) &&
          fz >= (double)lo_pad && fz < (double)(g.nz - lo_pad)))
        return false;
    *ix = (int)fx;
    *iy = (int)fy;
    *iz = (int)fz;
    return true;
}

// target points of the (2B+1)^3 block around cell (ix, iy, iz), clipped to the table: f(index into qx / qy / qz)
template <class F>
__device__ __forceinline__ void for_block(const GridDesc& g, const uint32_t* __restrict__ cell_start, int ix, int iy, int iz,
                                          int B, F&& f) {
    const int x0 = max(ix - B, 0), x1 = min(ix + B, (int)g.nx - 1);
    for (int dz = -B; dz <= B; ++dz) {
        const int z = iz + dz;
        if (z < 0 || z >= (int)g.nz) continue;
        for (int dy = -B; dy <= B; ++dy) {
            const int y = iy + dy;
            if (y < 0 || y >= (int)g.ny) continue;
            const uint32_t row = ((uint32_t)z * g.ny + (uint32_t)y) * g.nx;
            const uint32_t b = cell_start[row + (uint32_t)x0], e = cell_start[row + (uint32_t)x1 + 1u];
            for (uint32_t c = b; c < e; ++c) f(c);
        }
    }
}

}  // namespace

// One thread per (sorted) source point.  R < 0: the slot holds no query (NaN padding, a point with a non-finite coordinate):
// it never matches under any pose, like in reg_validate_k.
__global__ __launch_bounds__(256) void reg_cache_build_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                         const double* __restrict__ sz, const double* __restrict__ T,
                                                         GridDesc g, const uint32_t* __restrict__ cell_start,
                                                         const double* __restrict__ qx, const double* __restrict__ qy,
                                                         const double* __restrict__ qz, RegCache c) {
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const size_t i = (size_t)tile * 256u + tid;
    const double x = sx[i], y = sy[i], z = sz[i];
    // (the pose exactly as reg_validate_k applies it)
    const double px = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    const double py = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    const double pz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
    const double h = 1.0 / g.inv_h;
    float2* __restrict__ cx = c.cx + (size_t)tile * (kRegCacheK / 2) * 256u + tid;   // [tile][pair][lane]
    float2* __restrict__ cy = c.cy + (size_t)tile * (kRegCacheK / 2) * 256u + tid;
    float2* __restrict__ cz = c.cz + (size_t)tile * (kRegCacheK / 2) * 256u + tid;
    double4* __restrict__ c64 = c.c64 + (size_t)tile * kRegCacheK * 256u + tid;      // [tile][slot][lane]
    const float far = 1e18f;   // an empty slot: s = 3e36, never below any t^2
    // slot k of this lane: component k & 1 of pair k / 2
    auto put = [&](uint32_t k, float vx, float vy, float vz, const double4& q) {
        reinterpret_cast<float*>(cx + (size_t)(k >> 1) * 256u)[k & 1u] = vx;
        reinterpret_cast<float*>(cy + (size_t)(k >> 1) * 256u)[k & 1u] = vy;
        reinterpret_cast<float*>(cz + (size_t)(k >> 1) * 256u)[k & 1u] = vz;
        c64[(size_t)k * 256u] = q;
    };
    uint32_t slot = 0;
    double R = -1.0;
    c.xa[i] = px;
    c.ya[i] = py;
    c.za[i] = pz;
    if (fabs(px) < INFINITY && fabs(py) < INFINITY && fabs(pz) < INFINITY) {   // (NaN fails the comparisons)
        int ix, iy, iz;
        const int K = g.K;
        if (!cache_cell_of(g, px, py, pz, K, &ix, &iy, &iz)) {
            // at least K + 1 cells outside the target's bounding box on some axis (the table carries 2K + 1 pad cells per side)
            R = (double)(K + 1) * h * (1.0 - 1e-5);
        } else {
            // largest radius in (lo, cap] whose open ball holds <= K target points: three rounds of a nine-way search over
            // the block (every pass over the block's ~125 cells is a chain of dependent loads: four passes instead of a
            // bisection's eleven), the block of half-width B covering the ball of radius B h (1 - 1e-5) around any point of its
            // centre cell (the cell assignment rounds by < 1e-6 h)
            auto radius = [&](int B, double lo) {
                const double cap = (double)B * h * (1.0 - 1e-5);
                double hi = cap;
                for (int round = 0; round < 3; ++round) {
                    double lim[9];
                    uint32_t n[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        lim[j] = j == 8 ? hi : lo + (hi - lo) * ((double)(j + 1) / 9.0);
                        n[j] = 0;
                    }
                    for_block(g, cell_start, ix, iy, iz, B, [&](uint32_t k) {
                        const double dx = px - qx[k], dy = py - qy[k], dz = pz - qz[k];
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
                        for (int j = 0; j < 9; ++j) n[j] += d2 < lim[j] * lim[j] ? 1u : 0u;
                    });
                    if (n[8] <= (uint32_t)kRegCacheK) return hi;   // (first round: the whole block's ball; later: hi itself qualifies)
                    double nlo = lo, nhi = lim[0];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (n[j] <= (uint32_t)kRegCacheK) {
                            nlo = lim[j];
                            nhi = lim[j + 1];
                        }
                    lo = nlo;
                    hi = nhi;
                }
                return lo;   // (only ever set to a radius whose count was <= K; the caller's lo qualifies by construction)
            };
            int B = 2;
            R = radius(B, 0.0);
            if (R == (double)B * h * (1.0 - 1e-5) && K + 2 > B) {   // a sparse neighbourhood: out to the search radius and a bit
                B = K + 2;
                R = radius(B, R);
            }
            const double R2 = R * R;
            for_block(g, cell_start, ix, iy, iz, B, [&](uint32_t k) {
                const double ex = px - qx[k], ey = py - qy[k], ez = pz - qz[k];   // (the search.s own expression decides membership)
                if (((ex * ex + ey * ey) + ez * ez) < R2 && slot < (uint32_t)kRegCacheK) {
                    put(slot, (float)(qx[k] - px), (float)(qy[k] - py), (float)(qz[k] - pz), make_double4(qx[k], qy[k], qz[k], 0.0));
                    ++slot;
                }
            });
            R *= 1.0 - 1e-6;
            // a radius below a thousandth of a cell (more than K coincident target points): no certificate -- and the bound
            // E(s) >= 2^-23 R^2 then stays above what a flushed fp32 product loses (the host admits cells of 1e-12 and more)
            if (R < h * 0x1p-10) R = 0.0;
        }
    }
    for (uint32_t k = slot; k < (uint32_t)kRegCacheK; ++k) put(k, far, far, far, make_double4(0.0, 0.0, 0.0, 0.0));
    // rounded towards zero: the certificate may only get smaller
    float Rf = (float)R;
    if (R > 0.0 && (double)Rf > R) Rf = __uint_as_float(__float_as_uint(Rf) - 1u);
    c.R[i] = Rf;
}

void launch_reg_cache_build(const CloudView& src_sorted, const double* T_dev, const GridDesc& g, const uint32_t* cell_start,
                            const double* qx, const double* qy, const double* qz, const RegCache& c, hipStream_t s) {
    const uint32_t n_tiles = src_sorted.n_pad / kRegTile;
    if (!n_tiles) return;
    static_assert(kRegTile == 256, "one workgroup of the builder per validation tile");
    reg_cache_build_k<<<n_tiles, 256, 0, s>>>(src_sorted.x, src_sorted.y, src_sorted.z, T_dev, g, cell_start, qx, qy, qz, c);
}

// ------------------------------------------------------------------------------------------------
// The filter.  Same grid, block -> (tile, split) map, phases and hypothesis loop as reg_validate_k.
// ------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cache_phase_tile(uint32_t tile_local, uint32_t res_mask) {   // (reg_validate_k's phase_tile)
    const uint32_t k = (uint32_t)__popc(res_mask);
    uint32_t m = res_mask;
    for (uint32_t j = tile_local % k; j > 0; --j) m &= m - 1u;
    return (tile_local / k) * 8u + (uint32_t)(__ffs(m) - 1);
}

__global__ __launch_bounds__(256) void reg_validate_cached_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                             const double* __restrict__ sz, const double* __restrict__ Ts,
                                                             uint32_t s_pad, uint32_t s_per_split, double r2, RegCache c,
                                                             uint32_t* __restrict__ partial_cnt,
                                                             double* __restrict__ partial_sum, uint32_t res_mask,
                                                             uint32_t n_tiles_total, const uint8_t* __restrict__ keep,
                                                             uint32_t n_tiles_launch, uint32_t n_split,
                                                             uint8_t* __restrict__ redo,
                                                             unsigned long long* __restrict__ stats) {
    __shared__ uint32_t red[4][64];
    __shared__ double reds[4][64];
    __shared__ uint32_t redf[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t xcd = blockIdx.x % 8u, jq = blockIdx.x / 8u;
    const uint32_t tile_local = (jq / n_split) * 8u + xcd, split = jq % n_split;
    if (tile_local >= n_tiles_launch) return;
    const uint32_t tile = cache_phase_tile(tile_local, res_mask);
    if (tile >= n_tiles_total) return;
    const size_t base = (size_t)tile * kRegTile + (size_t)wave * 64 + lane;
    const double x = sx[base], y = sy[base], z = sz[base];
    const double xa = c.xa[base], ya = c.ya[base], za = c.za[base];
    const float R = c.R[base];
    const bool query = R >= 0.0f;   // (false: padding, a non-finite point -- no match under any pose)
    f32x2_t CX[kRegCacheK / 2], CY[kRegCacheK / 2], CZ[kRegCacheK / 2];
    {
        const size_t cb = (size_t)tile * (kRegCacheK / 2) * 256u + threadIdx.x;
#pragma unroll
        for (int k = 0; k < kRegCacheK / 2; ++k) {
            const float2 a = c.cx[cb + (size_t)k * 256u], b = c.cy[cb + (size_t)k * 256u], d = c.cz[cb + (size_t)k * 256u];
            CX[k] = f32x2_t{a.x, a.y};
            CY[k] = f32x2_t{b.x, b.y};
            CZ[k] = f32x2_t{d.x, d.y};
        }
    }
    const double4* __restrict__ c64 = c.c64 + (size_t)tile * kRegCacheK * 256u + threadIdx.x;
    const float Rlo = R * (1.0f - 0x1p-20f);
    const float ER = R * R * 0x1p-23f;                         // E(s) = ER + 2^-17 s
    const float r2hi = (float)r2 * (1.0f + 0x1p-19f);          // >= r^2 (1 + 2^-20), whatever the conversion rounded
    const uint32_t s0 = split * s_per_split, s1 = min(s0 + s_per_split, s_pad);
    uint32_t n_ok = 0, n_redo = 0;
    for (uint32_t sb = s0; sb < s1; sb += 64) {
        uint32_t acc = 0, acc_fail = 0;
        double acc_sum = 0.0;
        const unsigned long long keep_mask = keep ? __ballot(keep[sb + (uint32_t)lane] != 0) : ~0ull;
        double tn[12];
        {
            const double* __restrict__ T = Ts + (size_t)sb * kRegTStride;
#pragma unroll
            for (int k = 0; k < 12; ++k) tn[k] = T[k];
        }
        for (uint32_t ss = 0; ss < 64u; ++ss) {
            double t[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) t[k] = tn[k];
            {   // (the record behind the last one exists: Ts holds s_pad + 1 records)
                const double* __restrict__ T = Ts + (size_t)(sb + ss + 1u) * kRegTStride;
#pragma unroll
                for (int k = 0; k < 12; ++k) tn[k] = T[k];
            }
            uint32_t cnt = 0, fail = 0;
            double sum = 0.0;
            const bool run = (keep_mask >> ss) & 1ull;
            if (t[0] == t[0] && run) {   // padding records are NaN (wave-uniform branch)
                const double px = ((t[0] * x + t[1] * y) + t[2] * z) + t[3];
                const double py = ((t[4] * x + t[5] * y) + t[6] * z) + t[7];
                const double pz = ((t[8] * x + t[9] * y) + t[10] * z) + t[11];
                const float ux = (float)(px - xa), uy = (float)(py - ya), uz = (float)(pz - za);
                const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
                // what every UNLISTED target point keeps from the query, rounded down (v_sqrt_f32: 1 ulp, inside the 2^-20);
                // a pose that carries the query beyond R -- or a NaN / inf pose -- leaves 0: no information
                const float tt = __builtin_fmaxf(Rlo - __builtin_amdgcn_sqrtf(uu) * (1.0f + 0x1p-20f), 0.0f);
                const f32x2_t U_x = {ux, ux}, U_y = {uy, uy}, U_z = {uz, uz};
                // the two smallest s (m1 <= m2) with the winner's slot in the five low mantissa bits -- as INTEGERS: non-negative
                // floats order like their bit patterns, and v_min_u32 / v_max_u32 / v_min3_u32 need no canonicalised inputs
                uint32_t m1 = 0x7F800000u, m2 = 0x7F800000u;
#pragma unroll
                for (int k = 0; k < kRegCacheK / 2; ++k) {
                    const f32x2_t dx = CX[k] - U_x, dy = CY[k] - U_y, dz = CZ[k] - U_z;
                    const f32x2_t s = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    const uint32_t a = (__float_as_uint(s.x) & ~31u) | (uint32_t)(2 * k);
                    const uint32_t b = (__float_as_uint(s.y) & ~31u) | (uint32_t)(2 * k + 1);
                    const uint32_t lo = min(a, b), hi = max(a, b);
                    m2 = min(min(max(m1, lo), m2), hi);   // (m1 <= m2, lo <= hi): the second smallest of the four
                    m1 = min(m1, lo);
                    // (a chain, not a tree: left alone the compiler re-associates the 32 minima into trees that keep every s
                    // alive -- 214 VGPRs, two waves per SIMD; a wave issues one VALU instruction per 4 cycles either way)
                    asm volatile("" : "+v"(m1), "+v"(m2));
                }
                const uint32_t j1 = m1 & 31u;
                const float f1 = __uint_as_float(m1), f2 = __uint_as_float(m2);   // (a NaN pose: the bit patterns are NaNs, every test below fails)
                const float e1 = __builtin_fmaf(f1, 0x1p-17f, ER), e2 = __builtin_fmaf(f2, 0x1p-17f, ER);
                const float t2 = tt * tt;
                const bool covered = f1 + e1 < t2;                       // the nearest target point is in the list
                const bool unique = f2 - e2 > f1 + e1;                   // ... and it is candidate j1, in fp64 as well
                const bool nothing = t2 >= r2hi && f1 - e1 >= r2hi;      // no target point within the search radius
                double d2 = INFINITY;
                if (covered) {
                    const double4 w = c64[(size_t)j1 * 256u];
                    const double ddx = px - w.x, ddy = py - w.y, ddz = pz - w.z;
                    d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                }
                const bool exact = (covered && unique) || (!covered && nothing);
                // No certificate: the nearest target point is a listed one (at least f1 - e1 away, squared) or an unlisted one (at
                // least tt away), so min(f1 - e1, tt^2) bounds its squared distance from below -- the query MAY be an inlier unless
                // that already reaches the radius, and if it is one it adds at least that much to the sum.  Counted into the same
                // records: an upper bound of the count, a lower bound of the sum -- what bound-and-prune needs; the pair is
                // flagged, and a hypothesis that survives the pruning has its flagged pairs walked (launch_reg_validate).
                const float lb2 = __builtin_fmaxf(__builtin_fminf(f1 - e1, t2) * (1.0f - 0x1p-20f), 0.0f);
                const bool maybe = !(lb2 >= r2hi);   // (NaN: maybe)
                const bool f = query && (exact ? (covered && d2 < r2) : maybe);
                fail = __ballot(query && !exact) != 0ull ? 1u : 0u;
                cnt = (uint32_t)__popcll(__ballot(f));
                sum = f ? (exact ? d2 : (double)(lb2 == lb2 ? lb2 : 0.0f)) : 0.0;
                for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            }
            acc = ((uint32_t)lane == ss) ? cnt : acc;
            acc_sum = ((uint32_t)lane == ss) ? sum : acc_sum;
            acc_fail = ((uint32_t)lane == ss) ? fail : acc_fail;
        }
        red[wave][lane] = acc;
        reds[wave][lane] = acc_sum;
        redf[wave][lane] = acc_fail;
        __syncthreads();
        if (wave == 0) {
            const bool bad = (redf[0][lane] | redf[1][lane] | redf[2][lane] | redf[3][lane]) != 0u;
            const bool live = (keep_mask >> lane) & 1ull;
            redo[(size_t)tile * s_pad + sb + lane] = bad ? 1 : 0;
            partial_cnt[(size_t)tile * s_pad + sb + lane] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            partial_sum[(size_t)tile * s_pad + sb + lane] = (reds[0][lane] + reds[1][lane]) + (reds[2][lane] + reds[3][lane]);
            n_ok += (uint32_t)__popcll(__ballot(live && !bad));
            n_redo += (uint32_t)__popcll(__ballot(live && bad));
        }
        __syncthreads();
    }
    if (stats && threadIdx.x == 0) {
        atomicAdd(stats, (unsigned long long)n_ok);
        atomicAdd(stats + 1, (unsigned long long)n_redo);
    }
}

void launch_reg_validate_cached(const CloudView& src, const double* Ts, uint32_t s_pad, uint32_t per_split, uint32_t nsplit,
                                uint32_t slots, double r2, const RegCache& c, uint32_t* partial_cnt, double* partial_sum,
                                uint32_t res_mask, uint32_t n_tiles, const uint8_t* keep, uint32_t tiles, uint8_t* redo,
                                hipStream_t s) {
    reg_validate_cached_k<<<slots * 8 * nsplit, 256, 0, s>>>(src.x, src.y, src.z, Ts, s_pad, per_split, r2, c, partial_cnt,
                                                            partial_sum, res_mask, n_tiles, keep, tiles, nsplit, redo, c.stats);
}

}  // namespace m3d
