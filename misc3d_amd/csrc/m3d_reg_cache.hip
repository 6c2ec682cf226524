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
// at x = B p, delta = |x - xa| of a few millimetres.  So, once per reference pose and source point (reg_cache_build_k):
//   * the 32 target points nearest to xa as fp32 offsets from xa (in general: RINGS of kRegCacheK, kRegCacheTiers of them), and
//   * per ring a radius R_t with a CERTIFICATE: every target point that is NOT in rings 0..t lies at least R_t from xa
// (the target grid's (2B+1)^3 block around xa covers the ball of radius B h; a nine-way search on the radius finds the
// largest R_t whose ball holds <= 32 (t + 1) points).  Then for any pose B every unlisted target point is at least R_t - delta
// from x, and if the lists' minimum is below that, it is the minimum over the WHOLE target -- what the kd-tree returns.
//
// reg_validate_cached_k holds the 32 x 3 offsets of its lane's source point in 96 VGPRs across the whole hypothesis loop and
// evaluates them with packed fp32 arithmetic, no memory access at all (with several rings: ring by ring, a wave stopping as soon
// as all its queries are settled).  The winner's distance is formed in fp64 from the winner's fp64
// coordinates with reg_validate_k's own expression (one 32-byte gather per query), so counts AND sums are the same bits.
// Certificates per query (s_j = fp32 squared distance to candidate j, m1 <= m2 the two smallest so far, E(s) the rounding
// bound below, t = R (1 - 2^-20) - |u| (1 + 2^-20), u = fl32(x - xa)):
//   coverage   t > 0 and m1 + E(m1) < t^2          the nearest target point is in the lists
//   identity   m2 - E(m2) > m1 + E(m1)             candidate j1 is the strictly nearest of the lists, also in fp64
//   no match   t^2 >= r^2 (1 + 2^-20) and m1 - E(m1) >= r^2 (1 + 2^-20)     nothing within the search radius at all
// A query without a certificate still yields BOUNDS: its nearest target point is a listed one (squared distance >= m1 - E(m1))
// or an unlisted one (>= t), so it may be an inlier unless min(m1 - E, t^2) already reaches the radius, and adds at least that
// much to the sum if it is.  The (tile, hypothesis) record then holds an upper bound of the count and a lower bound of the
// sum and is flagged in `redo`; bound-and-prune is exact on bounds, and the flagged pairs of the hypotheses that survive
// it are walked by reg_validate_k (launch_reg_validate) -- the final records are the walk's, bit for bit, either way.
//
// Rounding bound.  c_j = fl32(q_j - xa) and u carry a relative 2^-24 each, |c_j|, |u| <= R: every coordinate difference
// is within e = 2^-22 R of the truth; the three products and two sums add 3 2^-24 s; packing the candidate's slot into
// the seven low mantissa bits of s (so that the minimum carries the index along) 2^-16 s:
//   |s - d^2| <= 2 e (|dx| + |dy| + |dz|) + 3 e^2 + (2^-22 + 2^-16) s <= 3.5 2^-22 R sqrt(s) + ...  <=  2^-23 R^2 + 2^-15 s =: E(s)
// (AM-GM: R sqrt(s) <= R^2 / 16 + 4 s).  For R = 13 mm and a neighbour at 2.5 mm that is 4e-8 m of distance: 4 queries in
// 10^5 are near-ties, 1 % of the 256-query tiles.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "m3d_reg_cache_fp.hpp"
#include "m3d_reg_kernels.hpp"

namespace m3d {

namespace {

constexpr int kTierK = kRegCacheK;          // candidates per tier
constexpr int kTiers = kRegCacheTiers;
constexpr int kSlots = kTierK * kTiers;     // <= 128: the slot rides in seven mantissa bits
static_assert(kSlots <= 128 && kSlots % 2 == 0 && kTierK % 2 == 0, "slot packing");
// (every ring lives in registers: kSlots candidates in 3 kSlots VGPRs)

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

// The target points of the (2B+1)^3 block around cell (ix, iy, iz), clipped to the table, dealt over the 64 lanes of ONE
// wave: f(active, index into qx / qy / qz) is called by all lanes together (it may ballot).  A block is (2B+1)^2 x-rows
// of 2B+1 contiguous cells: lanes fetch the rows' bounds (64 rows at a time), a wave prefix sum numbers the points, and
// item t finds its row by a binary search over the prefix (ds_bpermute).
template <class F>
__device__ __forceinline__ void for_block_items(const GridDesc& g, const uint32_t* __restrict__ cell_start, int ix, int iy, int iz,
                                                int B, int lane, F&& f) {
    const int side = 2 * B + 1, nrows = side * side;
    const int x0 = max(ix - B, 0), x1 = min(ix + B, (int)g.nx - 1);
    for (int r0 = 0; r0 < nrows; r0 += 64) {
        const int r = r0 + lane;
        uint32_t b = 0, e = 0;
        if (r < nrows) {
            const int z = iz + r / side - B, y = iy + r % side - B;
            if (z >= 0 && z < (int)g.nz && y >= 0 && y < (int)g.ny) {
                const uint32_t row = ((uint32_t)z * g.ny + (uint32_t)y) * g.nx;
                b = cell_start[row + (uint32_t)x0];
                e = cell_start[row + (uint32_t)x1 + 1u];
            }
        }
        const uint32_t cnt = e - b;
        uint32_t incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        const uint32_t excl = incl - cnt;
        for (uint32_t t0 = 0; t0 < total; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            int lo = 0, hi = 63;   // the first row whose inclusive prefix exceeds t
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi) >> 1;
                if (__shfl(incl, mid, 64) > t) hi = mid;
                else lo = mid + 1;
            }
            const uint32_t rb = __shfl(b, lo, 64), rex = __shfl(excl, lo, 64);
            const bool act = t < total;
            f(act, act ? rb + (t - rex) : 0u);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Cell rings: ring[cell] = Chebyshev distance, in cells, to the nearest OCCUPIED cell of the target grid (0: occupied; 255:
// none within `rounds` cells).  A query in a cell of ring rho has (rho - 1) whole empty cells between it and every target
// point along some axis: every target point is at least (rho - 1) h away -- a lower bound where the lists have none (a pose
// that carries the query beyond their radius), and for rho - 1 >= K (the grid's cells per search radius) the certificate that
// NOTHING lies within the radius: what reg_validate_k finds out by scanning (2K + 1)^3 cells.  One round = the minimum over
// the 3x3x3 block, separably (x, y, z), plus one.
// ------------------------------------------------------------------------------------------------
__global__ void ring_init_k(const uint32_t* __restrict__ cell_start, uint32_t ncell, uint8_t* __restrict__ ring) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c < ncell) ring[c] = cell_start[c + 1] > cell_start[c] ? 0 : 255;
}
// out[c] = min of in over c and its two neighbours along the axis of stride `stride` (n cells long); LAST: the round's third
// pass -- out[c] = min(prev[c], that minimum + 1) (saturating below 255)
template <bool LAST>
__global__ void ring_min3_k(const uint8_t* __restrict__ in, const uint8_t* prev, uint8_t* out /* LAST: may be prev (a thread reads and writes its own cell only) */,
                            uint32_t ncell, uint32_t stride, uint32_t n) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t k = (c / stride) % n;
    uint32_t m = in[c];
    if (k > 0) m = min(m, (uint32_t)in[c - stride]);
    if (k + 1 < n) m = min(m, (uint32_t)in[c + stride]);
    out[c] = LAST ? (uint8_t)min((uint32_t)prev[c], min(m + 1u, 255u)) : (uint8_t)m;
}
// ring: the result (ncell bytes); tmp: 2 x ncell bytes of scratch
void launch_reg_rings(const GridDesc& g, const uint32_t* cell_start, uint8_t* ring, uint8_t* tmp, int rounds, hipStream_t s) {
    const uint32_t ncell = g.nx * g.ny * g.nz, nb = (ncell + 255) / 256;
    uint8_t *a = tmp, *b = tmp + ncell;
    ring_init_k<<<nb, 256, 0, s>>>(cell_start, ncell, ring);
    for (int r = 0; r < rounds; ++r) {
        ring_min3_k<false><<<nb, 256, 0, s>>>(ring, nullptr, a, ncell, 1u, g.nx);
        ring_min3_k<false><<<nb, 256, 0, s>>>(a, nullptr, b, ncell, g.nx, g.ny);
        ring_min3_k<true><<<nb, 256, 0, s>>>(b, ring, ring, ncell, g.nx * g.ny, g.nz);
    }
}

// One WAVE per (sorted) source point.  R < 0 in every tier: the slot holds no query (NaN padding, a point with a non-finite
// coordinate): it never matches under any pose, like in reg_validate_k.
__global__ __launch_bounds__(256) void reg_cache_build_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                         const double* __restrict__ sz, uint32_t n_pad,
                                                         const double* __restrict__ T, GridDesc g,
                                                         const uint32_t* __restrict__ cell_start,
                                                         const double* __restrict__ qx, const double* __restrict__ qy,
                                                         const double* __restrict__ qz, RegCache c) {
    const int lane = threadIdx.x & 63;
    const size_t i = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i >= n_pad) return;   // (whole wave)
    const uint32_t tile = (uint32_t)(i / 256u), tid = (uint32_t)(i % 256u);
    const double x = sx[i], y = sy[i], z = sz[i];
    // (the pose exactly as reg_validate_k applies it)
    const double px = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    const double py = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    const double pz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
    const double h = 1.0 / g.inv_h;
    float2* __restrict__ cx = c.cx + (size_t)tile * (kSlots / 2) * 256u + tid;   // [tile][pair][lane of the tile]
    float2* __restrict__ cy = c.cy + (size_t)tile * (kSlots / 2) * 256u + tid;
    float2* __restrict__ cz = c.cz + (size_t)tile * (kSlots / 2) * 256u + tid;
    double4* __restrict__ c64 = c.c64 + (size_t)tile * kSlots * 256u + tid;      // [tile][slot][lane of the tile]
    const float far = 1e18f;   // an empty slot: s = 3e36, never below any t^2
    auto put = [&](uint32_t k, float vx, float vy, float vz, const double4& q) {   // slot k: component k & 1 of pair k / 2
        reinterpret_cast<float*>(cx + (size_t)(k >> 1) * 256u)[k & 1u] = vx;
        reinterpret_cast<float*>(cy + (size_t)(k >> 1) * 256u)[k & 1u] = vy;
        reinterpret_cast<float*>(cz + (size_t)(k >> 1) * 256u)[k & 1u] = vz;
        c64[(size_t)k * 256u] = q;
    };
    auto dist2 = [&](uint32_t k) {
        const double dx = px - qx[k], dy = py - qy[k], dz = pz - qz[k];
        return (dx * dx + dy * dy) + dz * dz;
    };
    uint32_t count[kTiers];
    double R[kTiers];
    for (int t = 0; t < kTiers; ++t) {
        count[t] = 0;
        R[t] = -1.0;
    }
    if (lane == 0) {
        c.xa[i] = px;
        c.ya[i] = py;
        c.za[i] = pz;
    }
    if (fabs(px) < INFINITY && fabs(py) < INFINITY && fabs(pz) < INFINITY) {   // (NaN fails the comparisons; wave-uniform)
        int ix, iy, iz;
        const int K = g.K;
        if (!cache_cell_of(g, px, py, pz, K, &ix, &iy, &iz)) {
            // at least K + 1 cells outside the target's bounding box on some axis (the table carries 2K + 1 pad cells per side)
            for (int t = 0; t < kTiers; ++t) R[t] = (double)(K + 1) * h * (1.0 - 1e-5);
        } else {
            // largest radius in (lo, cap] whose open ball holds <= want target points: three rounds of a nine-way search, every
            // round one pass over the block; the block of half-width B covers the ball of radius B h (1 - 1e-5) around any
            // point of its centre cell (the cell assignment rounds by < 1e-6 h)
            // (*n_out: the points inside the radius returned; n_lo: inside lo)
            auto radius = [&](int B, double lo, uint32_t n_lo, uint32_t want, uint32_t* n_out) {
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
                    for_block_items(g, cell_start, ix, iy, iz, B, lane, [&](bool act, uint32_t k) {
                        const double d2 = act ? dist2(k) : INFINITY;
#pragma unroll
                        for (int j = 0; j < 9; ++j) n[j] += (uint32_t)__popcll(__ballot(d2 < lim[j] * lim[j]));
                    });
                    if (n[8] <= want) {   // (first round: the whole block's ball; later: hi itself qualifies)
                        *n_out = n[8];
                        return hi;
                    }
                    double nlo = lo, nhi = lim[0];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (n[j] <= want) {
                            nlo = lim[j];
                            nhi = lim[j + 1];
                            n_lo = n[j];
                        }
                    lo = nlo;
                    hi = nhi;
                }
                *n_out = n_lo;
                return lo;   // (only ever set to a radius whose count was <= want; the caller's lo qualifies by construction)
            };
            // inside[t]: the target points inside R[t].  The slots are filled CUMULATIVELY -- the points of ring t (R[t-1] <= d < R[t])
            // from slot inside[t-1] on: a ring may hold more than 32 points when the tiers before it hold fewer, but the points
            // inside R[t] always fit the first 32 (t + 1) slots, which is what the certificate of tier t needs evaluated
            uint32_t inside[kTiers];
            int B = 3;
            R[0] = radius(B, 0.0, 0u, (uint32_t)kTierK, &inside[0]);
            if (R[0] == (double)B * h * (1.0 - 1e-5) && K + 2 > B) {   // a sparse neighbourhood: out to the search radius and a bit
                B = K + 2;
                R[0] = radius(B, R[0], inside[0], (uint32_t)kTierK, &inside[0]);
            }
            const double capB = (double)B * h * (1.0 - 1e-5);
            for (int t = 1; t < kTiers; ++t) {
                inside[t] = inside[t - 1];
                R[t] = R[t - 1] == capB ? capB : radius(B, R[t - 1], inside[t - 1], (uint32_t)(kTierK * (t + 1)), &inside[t]);
            }
            for (int t = 0; t < kTiers; ++t) count[t] = t ? inside[t - 1] : 0u;   // (running slot of ring t)
            for_block_items(g, cell_start, ix, iy, iz, B, lane, [&](bool act, uint32_t k) {
                const double d2 = act ? dist2(k) : INFINITY;   // (the search's own expression decides membership)
                int tier = -1;
#pragma unroll
                for (int t = kTiers - 1; t >= 0; --t) tier = d2 < R[t] * R[t] ? t : tier;
#pragma unroll
                for (int t = 0; t < kTiers; ++t) {
                    const unsigned long long m = __ballot(tier == t);
                    if (tier == t) {
                        const uint32_t pos = count[t] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (pos < (uint32_t)kSlots)
                            put(pos, (float)(qx[k] - px), (float)(qy[k] - py), (float)(qz[k] - pz), make_double4(qx[k], qy[k], qz[k], 0.0));
                    }
                    count[t] += (uint32_t)__popcll(m);
                }
            });
            for (int t = 0; t < kTiers; ++t) {
                R[t] *= 1.0 - 1e-6;
                // a radius below a thousandth of a cell (more than 32 coincident target points): no certificate -- and the bound
                // E(s) >= 2^-23 R^2 then stays above what a flushed fp32 product loses (the host admits cells of 1e-12 and more)
                if (R[t] < h * 0x1p-10) R[t] = 0.0;
            }
        }
    }
    // (count[kTiers - 1] has run up to the number of listed points: the slots behind it are empty)
    for (uint32_t k = count[kTiers - 1] + (uint32_t)lane; k < (uint32_t)kSlots; k += 64u) put(k, far, far, far, make_double4(0.0, 0.0, 0.0, 0.0));
    for (int t = 0; t < kTiers; ++t)
        if (lane == t) {
            // rounded towards zero: the certificate may only get smaller
            float Rf = (float)R[t];
            if (R[t] > 0.0 && (double)Rf > R[t]) Rf = __uint_as_float(__float_as_uint(Rf) - 1u);
            c.R[(size_t)t * n_pad + i] = Rf;
        }
}

void launch_reg_cache_build(const CloudView& src_sorted, const double* T_dev, const GridDesc& g, const uint32_t* cell_start,
                            const double* qx, const double* qy, const double* qz, const RegCache& c, hipStream_t s) {
    static_assert(kRegTile == 256, "layouts of the cache");
    if (!src_sorted.n_pad) return;
    reg_cache_build_k<<<(src_sorted.n_pad + 3) / 4, 256, 0, s>>>(src_sorted.x, src_sorted.y, src_sorted.z, src_sorted.n_pad, T_dev, g,
                                                                 cell_start, qx, qy, qz, c);
}

// ------------------------------------------------------------------------------------------------
// The filter.  Same grid, block -> (tile, split) map, phases and hypothesis loop as reg_validate_k.
// ------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cache_phase_tile(uint32_t tile_local, uint32_t res_mask) {   // (reg_validate_k's phase_tile)
    const uint32_t k = (uint32_t)__popc(res_mask);
    uint32_t m = res_mask;
    for (uint32_t j = tile_local % k; j > 0; --j) m &= m - 1u;
    return (tile_local / k) * 32u + (uint32_t)(__ffs(m) - 1);
}

// two candidates (a pair of slots) against the query: the two smallest s so far (m1 <= m2) with the winner's slot in the seven
// low mantissa bits -- as INTEGERS: non-negative floats order like their bit patterns, and v_min_u32 / v_max_u32 / v_min3_u32
// need no canonicalised inputs
__device__ __forceinline__ void cache_visit2(const f32x2_t cx, const f32x2_t cy, const f32x2_t cz, const f32x2_t U_x,
                                             const f32x2_t U_y, const f32x2_t U_z, uint32_t slot0, uint32_t& m1, uint32_t& m2) {
    const f32x2_t dx = cx - U_x, dy = cy - U_y, dz = cz - U_z;
    const f32x2_t s = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
    const uint32_t a = (__float_as_uint(s.x) & ~kCacheSlotMask) | slot0;   // (cache_packed_s, two candidates per instruction)
    const uint32_t b = (__float_as_uint(s.y) & ~kCacheSlotMask) | (slot0 + 1u);
    const uint32_t lo = min(a, b), hi = max(a, b);
    m2 = min(min(max(m1, lo), m2), hi);   // (m1 <= m2, lo <= hi): the second smallest of the four
    m1 = min(m1, lo);
    // (a chain, not a tree: left alone the compiler re-associates the minima into trees that keep every s alive -- 214 VGPRs,
    // two waves per SIMD; a wave issues one VALU instruction per 4 cycles either way)
    asm volatile("" : "+v"(m1), "+v"(m2));
}

// (three waves per SIMD: 168 VGPRs -- left alone the allocator takes 169 and the kernel falls to two)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void reg_validate_cached_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                             const double* __restrict__ sz, const double* __restrict__ Ts,
                                                             uint32_t s_pad, uint32_t s_per_split, GridDesc g, RegCache c,
                                                             uint32_t n_pad, uint32_t* __restrict__ partial_cnt,
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
    float Rt[kTiers];
#pragma unroll
    for (int t = 0; t < kTiers; ++t) Rt[t] = c.R[(size_t)t * n_pad + base];
    const bool query = Rt[0] >= 0.0f;   // (false: padding, a non-finite point -- no match under any pose)
    const size_t cb = (size_t)tile * (kSlots / 2) * 256u + threadIdx.x;
    f32x2_t CX[kSlots / 2], CY[kSlots / 2], CZ[kSlots / 2];   // every tier stays in registers
#pragma unroll
    for (int k = 0; k < kSlots / 2; ++k) {
        const float2 a = c.cx[cb + (size_t)k * 256u], b = c.cy[cb + (size_t)k * 256u], d = c.cz[cb + (size_t)k * 256u];
        CX[k] = f32x2_t{a.x, a.y};
        CY[k] = f32x2_t{b.x, b.y};
        CZ[k] = f32x2_t{d.x, d.y};
    }
    const double4* __restrict__ c64 = c.c64 + (size_t)tile * kSlots * 256u + threadIdx.x;
    const double r2 = g.r2;
    const float r2hi = (float)r2 * (1.0f + 0x1p-19f);          // >= r^2 (1 + 2^-20), whatever the conversion rounded
    // a cell edge rounded DOWN, less the rounding of the cell assignment (c.ring)
    float h_lo = (float)((1.0 / g.inv_h) * (1.0 - 1e-5));
    if ((double)h_lo > (1.0 / g.inv_h) * (1.0 - 1e-5)) h_lo = __uint_as_float(__float_as_uint(h_lo) - 1u);
    const uint32_t s0 = split * s_per_split, s1 = min(s0 + s_per_split, s_pad);
    uint32_t n_ok = 0, n_flag = 0;
    unsigned long long n_tier = 0, n_ring = 0;   // wave-queries that went past tier 0 / looked the cell rings up (this wave's)
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
                const float du = __builtin_amdgcn_sqrtf(uu) * (1.0f + 0x1p-20f);   // |u| rounded up (v_sqrt_f32: 1 ulp, inside the 2^-20)
                const f32x2_t U_x = {ux, ux}, U_y = {uy, uy}, U_z = {uz, uz};
                uint32_t m1 = 0x7F800000u, m2 = 0x7F800000u;
                bool exact = !query, inl = false;
                float lb2 = 0.0f;
                double d2 = INFINITY;
                // the certificates of a tier (radius R) on the minima so far, for the lanes that hold none yet
                auto certify = [&](float R) {   // (m3d_reg_cache_fp.hpp: cache_certify, the code tests/cpp/test_reg_cache.cpp checks)
                    if (exact) return;
                    const CacheVerdict v = cache_certify(m1, m2, R, du, r2hi);
                    if (v.winner) {
                        const double4 w = c64[(size_t)(m1 & kCacheSlotMask) * 256u];
                        const double ddx = px - w.x, ddy = py - w.y, ddz = pz - w.z;
                        d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                        inl = d2 < r2;
                        exact = true;
                    } else if (v.nothing) {
                        exact = true;
                    } else {
                        lb2 = v.lb2;
                    }
                };
                // (a wave all of whose queries the pose has carried beyond the outermost radius learns nothing from the lists)
                if (__ballot(query && du < Rt[kTiers - 1]) != 0ull) {
#pragma unroll
                    for (int tier = 0; tier < kTiers; ++tier) {
                        if (tier > 0) {
                            if (__ballot(!exact) == 0ull) break;   // (wave-uniform: every query is settled by the nearer rings)
                            n_tier++;
                        }
#pragma unroll
                        for (int k = tier * (kTierK / 2); k < (tier + 1) * (kTierK / 2); ++k)
                            cache_visit2(CX[k], CY[k], CZ[k], U_x, U_y, U_z, (uint32_t)(2 * k), m1, m2);
                        certify(Rt[tier]);
                    }
                }
                // The cell rings for what the lists could not settle: (ring - 1) whole empty cells lie between the query and every
                // target point -- at least K of them: nothing within the search radius (exact); fewer: a lower bound
                if (c.ring && __ballot(!exact) != 0ull) {   // (wave-uniform)
                    n_ring++;
                    if (!exact) {
                        int ix, iy, iz;
                        uint32_t rho = 255u;   // outside the table: 2K + 1 pad cells and more from every target point
                        if (cache_cell_of(g, px, py, pz, 0, &ix, &iy, &iz))
                            rho = c.ring[((size_t)(uint32_t)iz * g.ny + (uint32_t)iy) * g.nx + (uint32_t)ix];
                        if (px != px || py != py || pz != pz) rho = 0u;   // (a NaN pose says nothing)
                        if (rho >= (uint32_t)g.K + 1u) {
                            exact = true;
                            inl = false;
                        } else if (rho >= 2u) {
                            const float l = (float)(rho - 1u) * h_lo;
                            lb2 = __builtin_fmaxf(lb2, l * l * (1.0f - 0x1p-20f));
                        }
                    }
                }
                // A flagged query MAY be an inlier unless its bound already reaches the radius, and adds at least the bound to the
                // sum if it is one: an upper bound of the count, a lower bound of the sum -- what bound-and-prune needs.
                const bool f = query && (exact ? inl : !(lb2 >= r2hi));
                fail = __ballot(!exact) != 0ull ? 1u : 0u;
                cnt = (uint32_t)__popcll(__ballot(f));
                sum = f ? (exact ? d2 : (double)lb2) : 0.0;
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
            n_flag += (uint32_t)__popcll(__ballot(live && bad));
        }
        __syncthreads();
    }
    if (stats) {
        if (threadIdx.x == 0) {
            atomicAdd(stats, (unsigned long long)n_ok);
            atomicAdd(stats + 1, (unsigned long long)n_flag);
        }
        if (lane == 0 && n_tier) atomicAdd(stats + 2, n_tier);
        if (lane == 0 && n_ring) atomicAdd(stats + 3, n_ring);
    }
}

void launch_reg_validate_cached(const CloudView& src, const double* Ts, uint32_t s_pad, uint32_t per_split, uint32_t nsplit,
                                uint32_t slots, const GridDesc& g, const RegCache& c, uint32_t* partial_cnt, double* partial_sum,
                                uint32_t res_mask, uint32_t n_tiles, const uint8_t* keep, uint32_t tiles, uint8_t* redo,
                                hipStream_t s) {
    reg_validate_cached_k<<<slots * 8 * nsplit, 256, 0, s>>>(src.x, src.y, src.z, Ts, s_pad, per_split, g, c, src.n_pad, partial_cnt,
                                                            partial_sum, res_mask, n_tiles, keep, tiles, nsplit, redo, c.stats);
}

}  // namespace m3d
