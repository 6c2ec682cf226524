// m3d_score_mfma.hip -- score_mfma_k: the inlier counting of the culled path with the SCREEN ON THE MATRIX PIPE.
//
// EvaluateModel (include/misc3d/common/ransac.h:626-641) asks, per hypothesis and point, one question: is the distance
// below the threshold.  score_screen_k (m3d_cull_kernels.hip) answers it with 3.6 packed-fp32 VALU instructions per pair and
// sends the pairs its rounding bound cannot decide to the exact fp64 code.  On gfx950 every such instruction occupies its
// SIMD for ~4.15 cycles (tools/ubench/valu_rates.hip) -- but a v_mfma_f32_32x32x16_f16 between them costs ~10 cycles of
// issue and then works beside the VALU.  Here the matrix pipe evaluates, for 32 points x 32 hypotheses per instruction, the
// plane's two one-sided values u1 = T - S and u2 = T + S (split-fp16 operands, 12 of K = 16 slots: m3d_fp.hpp, "Records of the
// MFMA screen"); inside <=> t = u1 u2 > 0, and the VALU is left with 2.5 instructions per pair -- one VOP2 multiply, the shift
// that collects t's sign bit, half a v_min3 for "was any |t| too small to call".  Decisions are the fp64 code's, bit for bit: a
// (tile, hypothesis) pair with a point inside the bound is recounted by tile_count.
//
//   * A workgroup = one tile x up to 64 groups of 64 hypotheses, FOUR waves.  A operand (points): per 32-point block and lane
//     8 halfs, made from the tile's fp32 offsets (SortedView::tile_f32): X = x~ sigma cut into fp16 pairs by v_cvt_pk_f16_f32;
//     the lower half-wave holds the pieces of x, y, z, the upper one z's third slot and the constant's.  Each wave builds four
//     of the sixteen blocks into LDS (16 KB per tile); every wave then reads a block's operand back (one ds_read_b128) right
//     before the MFMAs that take it.
//   * the surviving hypotheses of the workgroup's groups are compacted into one id list (wave 0: lane = mask word); the waves
//     take its batches of 64 in turn, so the four are balanced whatever the masks look like.
//   * B operands (hypotheses, two columns each): lane k prepares hypothesis k's constants FOR THIS TILE in fp64
//     (plane_mfma_record), cuts them and hands the upper K-halves to lane k +- 32 with v_permlane32_swap: two sub-batches of
//     32 hypotheses per 64 lanes.  The records of the wave's NEXT batch are requested before the current one is evaluated.
//   * per sub-batch: 16 blocks x (2 MFMA + 16 v_mul + 16 v_alignbit + 8 v_min3), the MFMAs of block i + 1 issued before the
//     post-processing of block i; the lane's 256 sign bits are counted by v_bcnt every 32, the two half-waves' counts meet
//     through one more swap; one vector atomic per 32 hypotheses.
//
// STATUS (round 4, MI355X, C2 = 1 M points x 10 000 planes; profiles/r04_score_mfma.txt): OPT-IN (m3d_config.score_mfma = 1).
// Counts are identical (the parity suite runs this path), the loop is what the micro-benchmark promised -- the launch takes
// 0.087 ms against score_screen_k's 0.100 when the undecided pairs are ignored -- but its band is ~4 x the packed screen's
// (25.5 u (r + T + |D|) against 8 u M_l: the pieces' and the pipe's rounding), 1.7 % of the pairs instead of 0.23 % go to the
// exact code, and those concentrate in the tiles of the winning plane: the workgroups that own them run 0.2 ms longer than
// everybody else and the launch takes 0.31 ms.  What would make it the default is spelled out in DESIGN.md 4 ("MFMA screen").
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

// The A operand of one 32-point block for this lane (row = lane % 32; `upper` = lane >= 32 holds K-slots 8..15):
//   lower half-wave: (xh, yh | xh, yh | xl, yl | zh, zl)        upper half-wave: (zh, 2048 | 2048, 2048 | 0, 0 | 0, 0)
// against the B column  (bxh, byh | bxl, byl | bxh, byh | bzh, bzh)   and   (bzl, k1 | k2, k3 | 0, 0 | 0, 0)   (b_record).
constexpr uint32_t kC16 = 0x6800u;   // 2048 as fp16
__device__ __forceinline__ u32x4 a_block(float x, float y, float z, float sig, bool upper) {
    const float X = x * sig, Y = y * sig, Z = z * sig;   // (powers of two: exact)
    uint32_t H, L;
    split_pair(X, Y, H, L);
    const float zh = (float)(_Float16)Z;
    const uint32_t ZZ = pk16(Z, Z - zh);
    const uint32_t zc = (ZZ & 0xFFFFu) | (kC16 << 16), cc = kC16 | (kC16 << 16);
    return u32x4{upper ? zc : H, upper ? cc : H, upper ? 0u : L, upper ? 0u : ZZ};
}

// The B operand pieces of one hypothesis, as its own lane holds them before the exchange: per column (u1 = T - S, u2 = T + S)
// a lower octet (K-slots 0..7) and an upper one (8..15)
struct BPieces {
    u32x4 c1l, c1u, c2l, c2u;
    float hs;
};
__device__ __forceinline__ BPieces b_record(const double (&rec)[5], const double (&box)[6], double max_abs, int p, bool live) {
    double k[2], hs;
    plane_mfma_record(rec, box, max_abs, p, k, &hs);
    // 1024 alpha: rounded to fp32 (part of the bound), then cut in two
    const float af[3] = {(float)(rec[0] * kMfmaCoef), (float)(rec[1] * kMfmaCoef), (float)(rec[2] * kMfmaCoef)};
    uint32_t hh, ll;
    split_pair(af[0], af[1], hh, ll);
    const float zh = (float)(_Float16)af[2], zl = af[2] - zh;
    float k1[3], k2[3];
    pieces3(k[0], k1[0], k1[1], k1[2]);
    pieces3(k[1], k2[0], k2[1], k2[2]);
    BPieces o;
    const uint32_t zz = pk16(zh, zh);
    constexpr uint32_t neg = 0x80008000u;
    o.c2l = u32x4{hh, ll, hh, zz};
    o.c1l = u32x4{hh ^ neg, ll ^ neg, hh ^ neg, zz ^ neg};
    o.c2u = u32x4{pk16(zl, k2[0]), pk16(k2[1], k2[2]), 0u, 0u};
    o.c1u = u32x4{pk16(-zl, k1[0]), pk16(k1[1], k1[2]), 0u, 0u};
    const bool screened = hs == hs;   // (h = NaN: the record is not screened -- keep its column finite)
    if (!live || !screened) o.c1l = o.c1u = o.c2l = o.c2u = u32x4{0u, 0u, 0u, 0u};
    if (!live) hs = 0.0;   // no hypothesis in this lane: zero columns, their verdicts are not read
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

// one sub-batch: 32 hypotheses (B1, B2) against the tile's 16 blocks; As = the workgroup's A operands in LDS.
// out: the lane's number of OUTSIDE points among its 256, the smallest |t|
__device__ __forceinline__ void sub_batch(const u32x4* __restrict__ As, int lane, const h16x8 B1, const h16x8 B2, uint32_t& outside, float& mn_out) {
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t bits = 0, cnt = 0;
    float mn = __builtin_inff();
    // The operands never change while the workgroup runs, and an MFMA depends on nothing but its operands: left alone, the
    // compiler reads all sixteen blocks up front (64 registers) and issues all 32 MFMAs before the first v_mul (512 registers
    // of results).  The lane's LDS index is therefore laundered through an empty asm that also takes the sign string of the
    // block finished last: block i + 2's operand cannot be read, and its MFMAs cannot be issued, before block i - 1 is done.
    uint32_t li = (uint32_t)lane;
    asm volatile("" : "+v"(li));
    h16x8 a = __builtin_bit_cast(h16x8, As[li]);
    f32x16 u1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B1, zero16, 0, 0, 0);
    f32x16 u2n = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B2, zero16, 0, 0, 0);
    h16x8 an = __builtin_bit_cast(h16x8, As[64 + li]);
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        const f32x16 u1 = u1n, u2 = u2n;
        if (blk < 15) {   // the next block's MFMAs go first: the matrix pipe works under this block's VALU instructions
            u1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, B1, zero16, 0, 0, 0);
            u2n = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, B2, zero16, 0, 0, 0);
        }
        if (blk < 14) {   // ... and the operand of the block after it is requested (bits: the block BEFORE this one is done)
            asm volatile("" : "+v"(li) : "v"(bits));
            an = __builtin_bit_cast(h16x8, As[(blk + 2) * 64 + li]);
        }
        __builtin_amdgcn_sched_barrier(0);
        float t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            t[j] = u1[j] * u2[j];
            bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[j]), 31);   // t < 0: outside
        }
#pragma unroll
        for (int j = 0; j < 16; j += 2) mn = __builtin_fminf(__builtin_fminf(mn, __builtin_fabsf(t[j])), __builtin_fabsf(t[j + 1]));
        if (blk & 1) cnt += (uint32_t)__popc(bits);
        asm volatile("" : "+v"(mn));   // (a minimum is associative: left alone, the compiler keeps all 256 products for one tree at the end)
        __builtin_amdgcn_sched_barrier(0);
    }
    outside = cnt;
    mn_out = mn;
}

template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void score_mfma_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                     const double* __restrict__ sz, const double* __restrict__ boxes,
                                                     double max_abs, const double* __restrict__ score,
                                                     const unsigned long long* __restrict__ masks,
                                                     const unsigned long long* __restrict__ keep, uint32_t n_groups,
                                                     uint32_t groups_per_block /* <= kMfmaMaxGroups */,
                                                     uint32_t* __restrict__ counts_rep, uint32_t rep_stride,
                                                     uint32_t* __restrict__ pair_rep, uint32_t group_begin, uint32_t group_end,
                                                     const float* __restrict__ tile_f32) {
    static_assert(KIND == 0, "planes only");
    __shared__ uint16_t ids[kMfmaMaxGroups * 64];
    __shared__ u32x4 As[16 * 64];   // the tile's A operands: [block][lane]
    __shared__ uint32_t s_total;
    const uint32_t tile = blockIdx.x;
    uint32_t* __restrict__ counts = counts_rep + (size_t)(tile % kCountReplicas) * rep_stride;
    const uint32_t g0 = group_begin + blockIdx.y * groups_per_block;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (the wave's index: uniform, in an SGPR)
    const bool upper = lane >= 32;
    // every wave looks at the workgroup's mask words itself: a uniform exit without a barrier
    unsigned long long mm = 0;
    if ((uint32_t)lane < groups_per_block && g0 + lane < group_end) mm = masks[(size_t)tile * n_groups + g0 + lane] & keep[g0 + lane];
    if (!__ballot(mm != 0)) return;
    double box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) box[k] = boxes[(size_t)tile * kBoxStride + k];   // (wave-uniform: scalar loads)
    const int p = mfma_tile_exp(box);
    const bool tile_screened = boxes[(size_t)tile * kBoxStride + 6] != 0.0 && p != kMfmaNoTile;
    // ---- wave 0: the surviving hypotheses' ids (relative to g0), ascending: lane l expands its own word behind the words before it
    if (wave == 0) {
        const uint32_t pc = (uint32_t)__popcll(mm);
        uint32_t incl = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_total = incl;
        uint32_t at = incl - pc;
        unsigned long long w = mm;
        while (w) {
            ids[at++] = (uint16_t)((uint32_t)lane * 64u + (uint32_t)__builtin_ctzll(w));
            w &= w - 1ull;
        }
    }
    // ---- the tile's A operands: block = 32 consecutive lanes of one row of the tile; wave w builds the blocks of row pair w;
    // tile_f32 holds [coordinate][row pair j][lane] x (row 2 j, row 2 j + 1)
    if (tile_screened) {
        const float sig = __builtin_ldexpf(1.0f, p);
        const f32x2* __restrict__ t2 = reinterpret_cast<const f32x2*>(tile_f32 + (size_t)tile * kTileF32Floats) + (lane & 31);
        constexpr int Q = kTilePoints / 128;
        const int j = wave;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const f32x2 xv = t2[(0 * Q + j) * 64 + hb * 32], yv = t2[(1 * Q + j) * 64 + hb * 32], zv = t2[(2 * Q + j) * 64 + hb * 32];
            As[(4 * j + hb) * 64 + lane] = a_block(xv.x, yv.x, zv.x, sig, upper);       // row 2 j
            As[(4 * j + 2 + hb) * 64 + lane] = a_block(xv.y, yv.y, zv.y, sig, upper);   // row 2 j + 1
        }
    }
    __syncthreads();
    const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_total);
    if (threadIdx.x == 0) atomicAdd(&pair_rep[(tile + blockIdx.y * 67u) % (uint32_t)kPairMain], total);
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
    // ---- batches of 64 ids, the waves in turn; the NEXT batch's records are in flight while this one is evaluated
    auto fetch = [&](uint32_t b0, int& my, double (&rec)[5]) {
        const bool live = b0 + (uint32_t)lane < total;
        my = live ? (int)ids[b0 + lane] : 0;   // lane k: k-th id of the batch (b0 >= total: id 0, a valid address)
        const double* __restrict__ rp = score0 + (size_t)my * kModelStride;
#pragma unroll
        for (int k = 0; k < 5; ++k) rec[k] = rp[k];
    };
    int my_n;
    double rec_n[5];
    fetch((uint32_t)wave * 64u, my_n, rec_n);
    for (uint32_t b0 = (uint32_t)wave * 64u; b0 < total; b0 += 256u) {
        const bool live = b0 + (uint32_t)lane < total;
        const int my = my_n;
        uint32_t park = 0;            // lane k: the count of the batch's k-th hypothesis
        unsigned long long und = 0;   // bit k: it goes to the exact code
        if (tile_screened) {          // (uniform)
            BPieces bp = b_record(rec_n, box, max_abs, p, live);
            fetch(b0 + 256u, my_n, rec_n);
            // columns 0..31 = hypotheses of lanes 0..31 (sub-batch 0), then those of lanes 32..63 (sub-batch 1)
            swap32(bp.c1l, bp.c1u);
            swap32(bp.c2l, bp.c2u);
            uint32_t hq0 = __float_as_uint(bp.hs), hq1 = hq0;
            swap32(hq0, hq1);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                if (sb == 1 && b0 + 32u >= total) break;   // (wave-uniform) no hypothesis in the second half of the batch
                const h16x8 B1 = __builtin_bit_cast(h16x8, sb ? bp.c1u : bp.c1l), B2 = __builtin_bit_cast(h16x8, sb ? bp.c2u : bp.c2l);
                const float hq = __uint_as_float(sb ? hq1 : hq0);
                uint32_t outside;
                float mn;
                sub_batch(As, lane, B1, B2, outside, mn);
                // the two half-waves hold the two halves of the column's 512 points
                uint32_t oa = outside, ob = outside;
                swap32(oa, ob);
                park = (upper == (sb == 1)) ? (uint32_t)kTilePoints - (oa + ob) : park;   // (both half-waves hold the column's total)
                const unsigned long long u = __ballot(!(mn >= hq));   // (h = NaN: the record is not screened)
                und |= (unsigned long long)((uint32_t)u | (uint32_t)(u >> 32)) << (32 * sb);
            }
            und &= __ballot(live);
        } else {   // NaN padding / non-finite offsets: every pair of the tile through the exact code
            fetch(b0 + 256u, my_n, rec_n);
            und = __ballot(live);
        }
        while (und) {   // wave-uniform; rare on screened tiles
            const uint32_t n = (uint32_t)__builtin_ctzll(und);
            und &= und - 1ull;
            const uint32_t e = exact_count((uint32_t)__builtin_amdgcn_readlane(my, (int)n));
            if (lane == 0) atomicAdd(&pair_rep[(uint32_t)kPairMain + tile % (uint32_t)(kPairLead - kPairMain)], 1u);
            park = ((uint32_t)lane == n) ? e : park;
        }
        if (live && park) atomicAdd(&counts[g0 * 64u + (uint32_t)my], park);
    }
}

// ---- test probe (m3d_bench_mfma_probe): the screen's values for ONE tile of 512 points and n_h plane records, as the
// production kernel's own device functions produce them -- u1 / Sigma_h, u2 / Sigma_h per (hypothesis, point), the band on
// their product and E_p -- so that a test can hold them against T -+ S~ evaluated in exact arithmetic
// (tests/test_gpu_mfma_screen.py).
__global__ __launch_bounds__(64) void mfma_probe_k(const double* __restrict__ pts /* 512 x 3 */, const double* __restrict__ boxp /* 6 */,
                                                   double max_abs, const double* __restrict__ recs /* n_h x kModelStride */,
                                                   uint32_t n_h, double* __restrict__ out_u /* n_h x 512 x 2 */,
                                                   double* __restrict__ out_h /* n_h x 3: h (on t, unscaled), Sigma_h, E_p */,
                                                   float* __restrict__ out_off /* 512 x 3: the fp32 offsets */) {
    const int lane = threadIdx.x;
    const bool upper = lane >= 32;
    double box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) box[k] = boxp[k];
    const int p = mfma_tile_exp(box);
    if (p == kMfmaNoTile) return;
    const float sig = __builtin_ldexpf(1.0f, p);
    u32x4 A[16];
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {   // block blk = points 32 blk .. 32 blk + 31
        const int pt = blk * 32 + (lane & 31);
        const float x = (float)(pts[pt * 3 + 0] - box[0]), y = (float)(pts[pt * 3 + 1] - box[1]), z = (float)(pts[pt * 3 + 2] - box[2]);
        if (!upper) {
            out_off[pt * 3 + 0] = x;
            out_off[pt * 3 + 1] = y;
            out_off[pt * 3 + 2] = z;
        }
        A[blk] = a_block(x, y, z, sig, upper);
    }
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t b0 = 0; b0 < n_h; b0 += 64u) {
        const bool live = b0 + (uint32_t)lane < n_h;
        const uint32_t my = live ? b0 + (uint32_t)lane : 0u;
        double rec[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) rec[k] = recs[(size_t)my * kModelStride + k];
        BPieces bp = b_record(rec, box, max_abs, p, live);
        const double sg = kMfmaCoef * __builtin_ldexp(1.0, p);   // the pipe's unit
        if (live) {
            double k2[2], hs, ep = 0.0;
            plane_mfma_record(rec, box, max_abs, p, k2, &hs, &ep);
            out_h[3 * my] = (hs / sg) / sg;   // (NaN stays NaN)
            out_h[3 * my + 1] = sg;
            out_h[3 * my + 2] = ep;
        }
        swap32(bp.c1l, bp.c1u);
        swap32(bp.c2l, bp.c2u);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const h16x8 B1 = __builtin_bit_cast(h16x8, sb ? bp.c1u : bp.c1l), B2 = __builtin_bit_cast(h16x8, sb ? bp.c2u : bp.c2l);
            const uint32_t hyp = b0 + 32u * (uint32_t)sb + (uint32_t)(lane & 31);
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                const h16x8 a = __builtin_bit_cast(h16x8, A[blk]);
                const f32x16 u1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B1, zero16, 0, 0, 0);
                const f32x16 u2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B2, zero16, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = (j & 3) + 8 * (j >> 2) + (upper ? 4 : 0);
                    if (hyp < n_h) {
                        out_u[((size_t)hyp * 512 + blk * 32 + row) * 2] = (double)u1[j] / sg;
                        out_u[((size_t)hyp * 512 + blk * 32 + row) * 2 + 1] = (double)u2[j] / sg;
                    }
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
    const dim3 g(s.n_tiles, (window + gpb - 1) / gpb), b(256);
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(score_mfma_k<0>, g, b, 0, st, ev_start, ev_stop, 0, s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep,
                              n_groups, gpb, counts_rep, rep_stride, pair_rep, group_begin, group_end, (const float*)s.tile_f32);
    else
        score_mfma_k<0><<<g, b, 0, st>>>(s.x, s.y, s.z, s.boxes, s.max_abs, score, masks, keep, n_groups, gpb, counts_rep, rep_stride,
                                         pair_rep, group_begin, group_end, (const float*)s.tile_f32);
    return true;
}

}  // namespace m3d
