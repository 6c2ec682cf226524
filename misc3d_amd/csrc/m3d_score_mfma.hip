// m3d_score_mfma.hip -- score_mfma_k: the inlier counting of the culled path with the SCREEN ON THE MATRIX PIPE.
//
// EvaluateModel (include/misc3d/common/ransac.h:626-641) asks, per hypothesis and point, one question: is the distance
// below the threshold.  score_screen_k (m3d_cull_kernels.hip) answers it with 3.6 packed-fp32 VALU instructions per pair and
// sends the pairs its rounding bound cannot decide to the exact fp64 code.  On gfx950 every such instruction occupies its
// SIMD for ~4.15 cycles (tools/ubench/valu_rates.hip) -- but a v_mfma_f32_32x32x16_f16 between them costs ~10 cycles of
// issue and then works beside the VALU.  The plane's verdict is the sign of q = T^2 - S^2, a quadratic form in the tile's
// offsets, i.e. a contraction over ten monomials: the matrix pipe evaluates it for 32 points x 32 hypotheses per pair of
// MFMAs (split-fp16 operands, 30 of K = 32 slots: m3d_fp.hpp, "QUADRIC records"), and the VALU is left with 1.5 instructions
// per pair -- the shift that collects the sign bit and half a v_min3 for "was any point too close to call".  Decisions are the
// fp64 code's, bit for bit: a (tile, hypothesis) pair with a point inside the bound is recounted by tile_count.
//
//   * A operand (points): per 32-point block and lane 8 halfs x 2 K-steps, made from the tile's fp32 offsets
//     (SortedView::tile_f32) when the wave starts: X = x~ sigma, four products + the linear z per lane (the two half-waves
//     hold different monomials of the same point), each cut into an fp16 pair by v_cvt_pk_f16_f32: 128 VGPRs for the tile.
//   * B operand (hypotheses): lane k prepares hypothesis k's coefficients FOR THIS TILE in fp64 (plane_quadric_record), cuts
//     them and hands the upper K-halves to lane k +- 32 with v_permlane32_swap: two sub-batches of 32 columns per 64 lanes.
//   * per sub-batch: 16 blocks x (2 MFMA + 16 v_alignbit + 8 v_min3); the lane's 256 sign bits are counted by v_bcnt every
//     32, the two half-waves' counts meet through one more swap; one vector atomic per 32 hypotheses.
// A workgroup is one wave and takes up to 64 groups of 64 hypotheses of its tile: the operand build (~2000 cycles) wants
// hundreds of surviving hypotheses behind it.
#include "m3d_cull_kernels.hpp"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

#include "m3d_config.hpp"
#include "m3d_fp.hpp"
#include "m3d_tile_count.hpp"

#pragma clang fp contract(off)

namespace m3d {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kMfmaMaxGroups = 64;   // 64-hypothesis groups per workgroup (the id list: 8 KB of LDS)

// two fp32 -> one register of two fp16, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t pk16(float a, float b) {
    const h16x2 h = __builtin_convertvector(f32x2{a, b}, h16x2);
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float lo16(uint32_t u) { return (float)__builtin_bit_cast(h16x2, u)[0]; }
__device__ __forceinline__ float hi16(uint32_t u) { return (float)__builtin_bit_cast(h16x2, u)[1]; }
// (m0, m1) -> H = (m0h, m1h), L = (m0l, m1l): the remainders are exact in fp32
__device__ __forceinline__ void split_pair(float m0, float m1, uint32_t& H, uint32_t& L) {
    H = pk16(m0, m1);
    L = pk16(m0 - lo16(H), m1 - hi16(H));
}
// fp64 value -> fp16 pieces (first piece = the nearest fp16 of the fp32 rounding; the remainders are exact in fp64)
__device__ __forceinline__ void pieces2(double v, float& h, float& l) {
    const _Float16 p = (_Float16)(float)v;
    h = (float)p;
    l = (float)(v - (double)h);
}
__device__ __forceinline__ void pieces3(double v, float& p1, float& p2, float& p3) {
    p1 = (float)(_Float16)(float)v;
    const double r1 = v - (double)p1;
    p2 = (float)(_Float16)(float)r1;
    p3 = (float)(r1 - (double)p2);
}

// The A operand of one 32-point block for this lane (row = lane % 32; `upper` = lane >= 32 holds K-slots 8..15 of each step):
//   step 0: (P0h, P1h | P0h, P1h | P0l, P1l | Zh, Zl)      lower half-wave: P0 = xx, P1 = yy;     upper: P0 = xz, P1 = yz
//   step 1: (P2h, P3h | P2h, P3h | P2l, P3l | 2048, 2048)   lower half-wave: P2 = zz, P3 = xy;     upper: P2 = 16 x, P3 = 16 y
// with Z = 16 z (the linear monomial); the B operand pairs (bh, bl, bh) with the first three registers (b_record).
__device__ __forceinline__ void a_block(float x, float y, float z, float sig, bool upper, uint32_t cc, u32x4& a0, u32x4& a1) {
    const float X = x * sig, Y = y * sig, Z = z * sig;   // (powers of two: exact)
    const float v0 = upper ? Z : X, v1 = upper ? Z : Y;
    const float w2a = upper ? X : Z, w2b = upper ? (float)kMfmaLin : Z;
    const float w3a = upper ? Y : X, w3b = upper ? (float)kMfmaLin : Y;
    const float P0 = X * v0, P1 = Y * v1, P2 = w2a * w2b, P3 = w3a * w3b;
    const float ZL = Z * (float)kMfmaLin;
    uint32_t H01, L01, H23, L23;
    split_pair(P0, P1, H01, L01);
    split_pair(P2, P3, H23, L23);
    const float zh = (float)(_Float16)ZL;
    const uint32_t ZZ = pk16(ZL, ZL - zh);
    a0 = u32x4{H01, H01, L01, ZZ};
    a1 = u32x4{H23, H23, L23, cc};
}

// The B operand pieces of one hypothesis, as its own lane holds them before the exchange: four octets (step, K-half)
//   step 0 lower: (xx_h, yy_h | xx_l, yy_l | xx_h, yy_h | z_h, z_h)      step 0 upper: (xz.., yz.. | .. | .. | z_l, 0)
//   step 1 lower: (zz_h, xy_h | zz_l, xy_l | zz_h, xy_h | k1, k2)         step 1 upper: (x.., y.. | .. | .. | k3, 0)
struct BPieces {
    u32x4 s0l, s0u, s1l, s1u;
    float hs;
};
__device__ __forceinline__ BPieces b_record(const double* __restrict__ rec, const double (&box)[6], double max_abs, int p, bool live) {
    double b[10], hs;
    plane_quadric_record(rec, box, max_abs, p, b, &hs);
    float h[9], l[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) pieces2(b[i], h[i], l[i]);
    float k1, k2, k3;
    pieces3(b[9], k1, k2, k3);
    BPieces o;
    // monomial order of b: xx, yy, zz, xy, xz, yz, x, y, z
    auto octet = [&](int i, int j, uint32_t last) {
        const uint32_t hh = pk16(h[i], h[j]), ll = pk16(l[i], l[j]);
        return u32x4{hh, ll, hh, last};
    };
    o.s0l = octet(0, 1, pk16(h[8], h[8]));
    o.s0u = octet(4, 5, pk16(l[8], 0.0f));
    o.s1l = octet(2, 3, pk16(k1, k2));
    o.s1u = octet(6, 7, pk16(k3, 0.0f));
    if (!live) {   // no hypothesis in this lane: a zero column (its verdicts are not read)
        o.s0l = o.s0u = o.s1l = o.s1u = u32x4{0u, 0u, 0u, 0u};
        hs = 0.0;
    }
    o.hs = (float)hs;
    return o;
}

// lower half-wave of `a` keeps its own values, the upper half-wave receives the lower half-wave's `b`; and the reverse for
// the second result (v_permlane32_swap: a[32..63] <-> b[0..31])
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap32(u32x4& a, u32x4& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t x = a[i], y = b[i];
        swap32(x, y);
        a[i] = x;
        b[i] = y;
    }
}

template <int KIND>
__global__ __launch_bounds__(64) void score_mfma_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                    const double* __restrict__ sz, const double* __restrict__ boxes,
                                                    double max_abs, const double* __restrict__ score,
                                                    const unsigned long long* __restrict__ masks,
                                                    const unsigned long long* __restrict__ keep, uint32_t n_groups,
                                                    uint32_t groups_per_block /* <= kMfmaMaxGroups */,
                                                    uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                    uint32_t* __restrict__ pair_rep, uint32_t group_begin, uint32_t group_end,
                                                    const float* __restrict__ tile_f32) {
    static_assert(KIND == 0, "planes only (sphere / cylinder: two quadric columns, not built yet)");
    __shared__ uint16_t ids[kMfmaMaxGroups * 64];
    const uint32_t tile = blockIdx.x;
    uint32_t* __restrict__ counts = counts_rep + (size_t)(tile % kCountReplicas) * rep_stride;
    const uint32_t g0 = group_begin + blockIdx.y * groups_per_block;
    const int lane = threadIdx.x;
    const bool upper = lane >= 32;
    unsigned long long mm = 0;
    if ((uint32_t)lane < groups_per_block && g0 + lane < group_end) mm = masks[(size_t)tile * n_groups + g0 + lane] & keep[g0 + lane];
    if (!__ballot(mm != 0)) return;
    // ---- the surviving hypotheses' ids (relative to g0), ascending: lane l expands its own word behind the words before it
    uint32_t total;
    {
        const uint32_t pc = (uint32_t)__popcll(mm);
        uint32_t incl = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= off) incl += t;
        }
        total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t at = incl - pc;
        unsigned long long w = mm;
        while (w) {
            ids[at++] = (uint16_t)((uint32_t)lane * 64u + (uint32_t)__builtin_ctzll(w));
            w &= w - 1ull;
        }
    }
    __syncthreads();
    if (lane == 0) atomicAdd(&pair_rep[(tile + blockIdx.y * 67u) % (uint32_t)kPairMain], total);
    double box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) box[k] = boxes[(size_t)tile * kBoxStride + k];   // (wave-uniform: scalar loads)
    const int p = mfma_tile_exp(box);
    const bool tile_screened = boxes[(size_t)tile * kBoxStride + 6] != 0.0 && p != kMfmaNoTile;
    const double* __restrict__ score0 = score + (size_t)g0 * 64u * kModelStride;
    const size_t base = (size_t)tile * kTilePoints + lane;
    constexpr int P = kTilePoints / 64;
    // the exact count of one (tile, hypothesis) pair (score_screen_k's): the fp64 points come back from memory (L2)
    auto exact_count = [&](uint32_t id) -> uint32_t {
        double rec[kModelStride];
        const double* __restrict__ rp = score0 + (size_t)id * kModelStride;   // id wave-uniform -> scalar loads
#pragma unroll
        for (int k = 0; k < kModelStride; ++k) rec[k] = rp[k];
        uint32_t c = 0;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            double x[P / 2], y[P / 2], z[P / 2];
#pragma unroll
            for (int j = 0; j < P / 2; ++j) {
                x[j] = sx[base + 64 * (half * (P / 2) + j)];
                y[j] = sy[base + 64 * (half * (P / 2) + j)];
                z[j] = sz[base + 64 * (half * (P / 2) + j)];
            }
            c += tile_count<KIND, P / 2>(rec, x, y, z);
        }
        return c;
    };
    if (!tile_screened) {   // (wave-uniform) NaN padding / non-finite offsets: every pair of the tile through the exact code
        for (uint32_t i = 0; i < total; ++i) {
            const uint32_t id = ids[i];
            const uint32_t e = exact_count(id);
            if (lane == 0) {
                atomicAdd(&pair_rep[(uint32_t)kPairMain + tile % (uint32_t)(kPairLead - kPairMain)], 1u);
                if (e) atomicAdd(&counts[g0 * 64u + id], e);
            }
        }
        return;
    }
    // ---- the tile's A operands: block blk = points blk * 32 .. + 31 = row blk / 2, lanes (blk % 2) * 32 .. of the tile;
    // tile_f32 holds [coordinate][row pair j][lane] x (row 2 j, row 2 j + 1)
    u32x4 A0[16], A1[16];
    {
        const float sig = __builtin_ldexpf(1.0f, p);
        const uint32_t cc = upper ? pk16((float)kMfmaConstA, (float)kMfmaConstA) : pk16((float)kMfmaConstA, (float)kMfmaConstA);
        const f32x2* __restrict__ t2 = reinterpret_cast<const f32x2*>(tile_f32 + (size_t)tile * kTileF32Floats) + (lane & 31);
        constexpr int Q = kTilePoints / 128;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const f32x2 xv = t2[(0 * Q + j) * 64 + hb * 32], yv = t2[(1 * Q + j) * 64 + hb * 32], zv = t2[(2 * Q + j) * 64 + hb * 32];
                // rows 2 j (component x) and 2 j + 1 (component y); block = 2 row + hb
                a_block(xv.x, yv.x, zv.x, sig, upper, cc, A0[4 * j + hb], A1[4 * j + hb]);
                a_block(xv.y, yv.y, zv.y, sig, upper, cc, A0[4 * j + 2 + hb], A1[4 * j + 2 + hb]);
            }
        }
    }
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t b0 = 0; b0 < total; b0 += 64u) {
        const bool live = b0 + (uint32_t)lane < total;
        const int my = live ? (int)ids[b0 + lane] : 0;   // lane k: k-th id of the batch
        BPieces bp = b_record(score0 + (size_t)my * kModelStride, box, max_abs, p, live);
        // columns 0..31 = hypotheses of lanes 0..31 (sub-batch 0), then those of lanes 32..63 (sub-batch 1)
        swap32(bp.s0l, bp.s0u);
        swap32(bp.s1l, bp.s1u);
        uint32_t hq0 = __float_as_uint(bp.hs), hq1 = hq0;
        swap32(hq0, hq1);
        uint32_t result[2];
        unsigned long long undecided[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            if (sb == 1 && b0 + 32u >= total) {   // (wave-uniform) no hypothesis in the second half of the batch
                result[1] = 0;
                undecided[1] = 0;
                break;
            }
            const h16x8 B0 = __builtin_bit_cast(h16x8, sb ? bp.s0u : bp.s0l), B1 = __builtin_bit_cast(h16x8, sb ? bp.s1u : bp.s1l);
            const float hq = __uint_as_float(sb ? hq1 : hq0);
            uint32_t bits = 0, outside = 0;
            float mn = __builtin_inff();
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A0[blk]), B0, zero16, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A1[blk]), B1, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; ++j) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(acc[j]), 31);   // q < 0: outside
#pragma unroll
                for (int j = 0; j < 16; j += 2) mn = __builtin_fminf(__builtin_fminf(mn, __builtin_fabsf(acc[j])), __builtin_fabsf(acc[j + 1]));
                if (blk & 1) outside += (uint32_t)__popc(bits);
            }
            // the two half-waves hold the two halves of the column's 512 points
            uint32_t oa = outside, ob = outside;
            swap32(oa, ob);
            result[sb] = (uint32_t)kTilePoints - (oa + ob);
            undecided[sb] = __ballot(!(mn >= hq));   // (h = NaN: the record is not screened)
        }
        // hypotheses the screen could not decide: the exact code (rare)
        uint32_t park = 0;   // lane n < 32: sub-batch 0's hypothesis n; lane 32 + n: sub-batch 1's
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            uint32_t und = ((uint32_t)undecided[sb] | (uint32_t)(undecided[sb] >> 32));
            const uint32_t n_here = min(32u, total - min(total, b0 + 32u * (uint32_t)sb));
            und &= n_here >= 32u ? 0xFFFFFFFFu : ((1u << n_here) - 1u);
            const uint32_t r = result[sb];
            park = (upper == (sb == 1)) ? r : park;   // (both half-waves hold the column's total)
            while (und) {   // wave-uniform
                const uint32_t n = (uint32_t)__builtin_ctz(und);
                und &= und - 1u;
                const uint32_t e = exact_count((uint32_t)__builtin_amdgcn_readlane(my, (int)(32u * (uint32_t)sb + n)));
                if (lane == 0) atomicAdd(&pair_rep[(uint32_t)kPairMain + tile % (uint32_t)(kPairLead - kPairMain)], 1u);
                park = ((uint32_t)lane == 32u * (uint32_t)sb + n) ? e : park;
            }
        }
        if (live && park) atomicAdd(&counts[g0 * 64u + (uint32_t)my], park);
    }
}

// ---- test probe (m3d_bench_mfma_probe): the screen's value for ONE tile of 512 points and n_h plane records, as the
// production kernel's own device functions produce it -- q_pipe / Sigma_h per (hypothesis, point) and the unscaled band h --
// so that a test can hold them against q~ = T^2 - S~^2 evaluated in exact arithmetic (tests/test_gpu_mfma_screen.py).
__global__ __launch_bounds__(64) void mfma_probe_k(const double* __restrict__ pts /* 512 x 3 */, const double* __restrict__ boxp /* 6 */,
                                                   double max_abs, const double* __restrict__ recs /* n_h x kModelStride */,
                                                   uint32_t n_h, double* __restrict__ out_q /* n_h x 512 */,
                                                   double* __restrict__ out_h /* n_h x 2: h, Sigma_h */,
                                                   float* __restrict__ out_off /* 512 x 3: the fp32 offsets */) {
    const int lane = threadIdx.x;
    const bool upper = lane >= 32;
    double box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) box[k] = boxp[k];
    const int p = mfma_tile_exp(box);
    if (p == kMfmaNoTile) return;
    const float sig = __builtin_ldexpf(1.0f, p);
    const uint32_t cc = pk16((float)kMfmaConstA, (float)kMfmaConstA);
    u32x4 A0[16], A1[16];
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {   // block blk = points 32 blk .. 32 blk + 31
        const int pt = blk * 32 + (lane & 31);
        const float x = (float)(pts[pt * 3 + 0] - box[0]), y = (float)(pts[pt * 3 + 1] - box[1]), z = (float)(pts[pt * 3 + 2] - box[2]);
        if (!upper) {
            out_off[pt * 3 + 0] = x;
            out_off[pt * 3 + 1] = y;
            out_off[pt * 3 + 2] = z;
        }
        a_block(x, y, z, sig, upper, cc, A0[blk], A1[blk]);
    }
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t b0 = 0; b0 < n_h; b0 += 64u) {
        const bool live = b0 + (uint32_t)lane < n_h;
        const uint32_t my = live ? b0 + (uint32_t)lane : 0u;
        BPieces bp = b_record(recs + (size_t)my * kModelStride, box, max_abs, p, live);
        if (live) {
            double b[10], hs, sg = 0.0;
            plane_quadric_record(recs + (size_t)my * kModelStride, box, max_abs, p, b, &hs, &sg);
            out_h[2 * my] = hs / sg;   // (NaN stays NaN)
            out_h[2 * my + 1] = sg;
        }
        swap32(bp.s0l, bp.s0u);
        swap32(bp.s1l, bp.s1u);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const h16x8 B0 = __builtin_bit_cast(h16x8, sb ? bp.s0u : bp.s0l), B1 = __builtin_bit_cast(h16x8, sb ? bp.s1u : bp.s1l);
            const uint32_t hyp = b0 + 32u * (uint32_t)sb + (uint32_t)(lane & 31);
            double sg = 1.0;
            if (hyp < n_h) {
                double b[10], hs;
                plane_quadric_record(recs + (size_t)hyp * kModelStride, box, max_abs, p, b, &hs, &sg);
            }
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A0[blk]), B0, zero16, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A1[blk]), B1, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = (j & 3) + 8 * (j >> 2) + (upper ? 4 : 0);
                    if (hyp < n_h) out_q[(size_t)hyp * 512 + blk * 32 + row] = (double)acc[j] / sg;
                }
            }
        }
    }
}
void launch_mfma_probe(const double* pts, const double* box, double max_abs, const double* recs, uint32_t n_h, double* out_q,
                       double* out_h, float* out_off, hipStream_t st) {
    mfma_probe_k<<<1, 64, 0, st>>>(pts, box, max_abs, recs, n_h, out_q, out_h, out_off);
}

bool launch_score_mfma(int kind, const SortedView& s, const double* score, const unsigned long long* masks,
                       const unsigned long long* keep, uint32_t n_groups, uint32_t* counts_rep, uint32_t rep_stride,
                       uint32_t* pair_rep, hipStream_t st, uint32_t group_begin, uint32_t group_end, hipEvent_t ev_start,
                       hipEvent_t ev_stop) {
    if (kind != 0 || !s.tile_f32 || s.has_dead || config().score_mfma == 0 || config().score_fp32_screen == 0) return false;
    group_end = std::min(group_end, n_groups);
    if (!s.n_tiles || group_begin >= group_end) return true;
    const uint32_t window = group_end - group_begin;
    const uint32_t gpb = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)config().score_mfma_groups, kMfmaMaxGroups));
    const dim3 g(s.n_tiles, (window + gpb - 1) / gpb), b(64);
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(score_mfma_k<0>, g, b, 0, st, ev_start, ev_stop, 0, s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep,
                              n_groups, gpb, counts_rep, rep_stride, pair_rep, group_begin, group_end, (const float*)s.tile_f32);
    else
        score_mfma_k<0><<<g, b, 0, st>>>(s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep, n_groups, gpb, counts_rep, rep_stride,
                                         pair_rep, group_begin, group_end, (const float*)s.tile_f32);
    return true;
}

}  // namespace m3d
