// m3d_match_kernels.hip -- exact nearest neighbour in descriptor space with an fp32 screening pass.
//
// Replaces the query loops of registration::NearestSearch (src/correspondence_matching.cpp:13-44)
// for FPFH-sized descriptors (dim 33).  The result is the EXACT fp64 nearest neighbour of the brute
// force kernel nn_k (same serial-order accumulation, ties -> lowest index); the speed comes from not
// evaluating all N x M pairs in fp64:
//   1. nn32_scan_k     fp32 (packed v_pk_fma_f32) |a|^2 + |b|^2 - 2 a.b over all pairs; per query the running
//                      minimum m and a ring of the rows seen with d32 <= m + 2 E_q.  E_q bounds |d32 - d_exact|
//                      (input rounding to fp32 + fp32 arithmetic, ~60 u (|a|^2 + |b|^2) worst case; used:
//                      E_q = (2 dim + 16) * 2^-24 * 1.5 * (|a|^2 + max|b|^2)).  The exact argmin j* satisfies
//                      d32(j*) <= d(j*) + E <= d(j~) + E <= m_final + 2E <= m_running + 2E, so it (and every
//                      exact tie) enters the ring.
//   2. nn64_verify_k   exact fp64 distances of the ring entries inside the final window, reference accumulation order.
//   3. nn_exact_one_k  a query whose ring evicted a possibly valid entry, or whose fp32 bound is not finite
//                      (adversarial data), is redone exactly by a whole wave.
#include "m3d_reg_kernels.hpp"

#include <algorithm>
#include <cstdlib>

#include "m3d_fp.hpp"
#include "m3d_match_scan.hpp"

#pragma clang fp contract(off)

namespace m3d {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// fp32 row: kScreenDimP floats = 33 values, one zero pad (even count for packed FMAs), |row|^2 of the
// ROUNDED values at [34], one more pad -> 144 B rows, 16 B aligned for wide scalar loads.
constexpr int kDotW = 34;
constexpr int kNormSlot = 34;

__global__ void to_f32_k(const double* __restrict__ f, uint32_t n, float* __restrict__ out,
                         float* __restrict__ norm2) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
    for (int k = 0; k < kDotW; ++k) {
        const float v = k < 33 ? (float)f[(size_t)i * 33 + k] : 0.0f;
        out[(size_t)i * kScreenDimP + k] = v;
        acc = __builtin_fmaf(v, v, acc);
    }
    out[(size_t)i * kScreenDimP + kNormSlot] = acc;
    out[(size_t)i * kScreenDimP + kNormSlot + 1] = 0.0f;
    norm2[i] = acc;
}

struct Row32 {
    f32x2 v[kDotW / 2];
    float n2;
};

__device__ __forceinline__ Row32 load_row(const float* __restrict__ d) {
    Row32 r;
#pragma unroll
    for (int k = 0; k < kDotW / 2; ++k) r.v[k] = {d[2 * k], d[2 * k + 1]};
    r.n2 = d[kNormSlot];
    return r;
}

// d32(q, j) = (|q|^2 + |b_j|^2) - 2 q.b_j; two independent packed-FMA chains; one query per lane
// (row in VGPRs), database rows wave-uniform (scalar loads, prefetched one row ahead by the callers).
__device__ __forceinline__ float d32_of(const Row32& q, const Row32& d) {
    f32x2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k + 1 < kDotW / 2; k += 2) {
        a0 = __builtin_elementwise_fma(q.v[k], d.v[k], a0);
        a1 = __builtin_elementwise_fma(q.v[k + 1], d.v[k + 1], a1);
    }
    a0 = __builtin_elementwise_fma(q.v[kDotW / 2 - 1], d.v[kDotW / 2 - 1], a0);
    const f32x2 a = a0 + a1;
    return __builtin_fmaf(-2.0f, a.x + a.y, q.n2 + d.n2);
}

// One pass over a slice of the database: running fp32 minimum m and every row with d32 <= m + 2E at
// the time it is seen (a superset of the rows within 2E of the FINAL minimum, because m only falls),
// kept in a ring of kRing (index, d32) entries.  An entry pushed out of the ring is remembered only
// through the smallest evicted d32: if that is above the final window the evicted entries were all
// stale, otherwise the query takes the exact fallback.
// Two queries per lane (rows in VGPRs): every database row fetched by the scalar unit feeds
// 2 x 64 distance evaluations.  Block b covers queries [512 b, 512 b + 512): lane t holds 512 b + t and
// 512 b + 256 + t.
__global__ __launch_bounds__(256) void nn32_scan_k(const float* __restrict__ q, uint32_t nq,
                                                    const float* __restrict__ db, uint32_t ndb,
                                                    uint32_t db_per_split, float e_coeff, float max_dn2,
                                                    uint2* __restrict__ ring, uint32_t* __restrict__ ring_count,
                                                    float* __restrict__ part_min, float* __restrict__ evict_min) {
    const uint32_t ia = blockIdx.x * 512u + threadIdx.x, ib = ia + 256u;
    const uint32_t iia = ia < nq ? ia : nq - 1, iib = ib < nq ? ib : nq - 1;
    const Row32 qa = load_row(q + (size_t)iia * kScreenDimP);
    const Row32 qb = load_row(q + (size_t)iib * kScreenDimP);
    const uint32_t j0 = blockIdx.y * db_per_split, j1 = min(ndb, j0 + db_per_split);
    // 2E (+ slack for the fp32 evaluation of the window itself); not finite -> the screen proves nothing
    const float two_ea = 2.0f * e_coeff * (qa.n2 + max_dn2) * 1.000001f + 1e-30f;
    const float two_eb = 2.0f * e_coeff * (qb.n2 + max_dn2) * 1.000001f + 1e-30f;
    const bool live_a = two_ea < INFINITY && ia < nq, live_b = two_eb < INFINITY && ib < nq;
    uint2* __restrict__ ring_a = ring + ((size_t)blockIdx.y * nq + iia) * kRing;
    uint2* __restrict__ ring_b = ring + ((size_t)blockIdx.y * nq + iib) * kRing;
    ScanState sa, sb;
    if (j0 < j1) {
        Row32 cur = load_row(db + (size_t)j0 * kScreenDimP);
        for (uint32_t j = j0; j < j1; ++j) {
            const Row32 nxt = load_row(db + (size_t)(j + 1) * kScreenDimP);   // spare row behind the last one
            const float da = d32_of(qa, cur);
            const float dbv = d32_of(qb, cur);
            scan_step(sa, da, two_ea, live_a, j, ring_a);
            scan_step(sb, dbv, two_eb, live_b, j, ring_b);
            cur = nxt;
        }
    }
    if (ia < nq) {
        const size_t o = (size_t)blockIdx.y * nq + ia;
        ring_count[o] = sa.cnt;
        part_min[o] = two_ea < INFINITY ? sa.best : -INFINITY;   // -inf forces the fallback in the verify kernel
        evict_min[o] = sa.ev;
    }
    if (ib < nq) {
        const size_t o = (size_t)blockIdx.y * nq + ib;
        ring_count[o] = sb.cnt;
        part_min[o] = two_eb < INFINITY ? sb.best : -INFINITY;
        evict_min[o] = sb.ev;
    }
}

// exact distances of the surviving candidates: serial-order fp64 accumulation (nanoflann
// L2_Simple_Adaptor order); among equal distances the lowest database index wins, as in a serial scan.
// Eight lanes per query: the slices' rings are taken in turn by the lanes, (distance, index) minimum across the eight (one
// thread per query walked up to splits x kRing dependent gathers: 0.51 ms per 200 000 queries, now 0.2).
// |q - row|^2 in the reference's order (k = 0 .. 32, one product and one addition each), the query's row in registers and the
// database row's 33 loads issued eleven at a time: the generic loop re-read the query from memory for every candidate and waited
// for every pair of loads (round 6: nn64_verify_k 228 -> 116 us per launch, nn64_verify_rev_k 200 -> 96 us: the call 6.4 -> 6.1 ms)
__device__ __forceinline__ double exact_d2_33(const double (&qr)[33], const double* __restrict__ row) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double dv[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) dv[k] = row[11 * c + k];
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const double df = qr[11 * c + k] - dv[k];
            acc += df * df;
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void nn64_verify_k(const double* __restrict__ q, uint32_t nq, const double* __restrict__ db, int dim,
                              const uint2* __restrict__ ring, const uint32_t* __restrict__ ring_count,
                              const float* __restrict__ part_min, const float* __restrict__ evict_min,
                              uint32_t splits, float e_coeff, float e_abs, float max_dn2_v, const float* __restrict__ max_dn2_p,
                              const float* __restrict__ qn2, uint32_t* __restrict__ nn, uint32_t* __restrict__ overflow_list,
                              uint32_t* __restrict__ overflow_count, uint32_t q_base, const uint32_t* __restrict__ perm) {
    // (perm: the rings carry positions in the packed database, perm[position] = row -- match_order_part; null: rows)
    // (q, qn2, nn and the per-query arrays address the launch's nq queries, the first of which is query q_base of the whole matrix --
    // what the overflow list carries; max |row|^2 of the database: the device cell when there is one, else the host's value)
    const uint32_t i = blockIdx.x * 32u + (threadIdx.x >> 3), sub = threadIdx.x & 7u;
    if (i >= nq) return;   // (whole groups of eight lanes)
    const float max_dn2 = max_dn2_p ? *max_dn2_p : max_dn2_v;
    float m = INFINITY;
    bool fallback = false;
    for (uint32_t s = sub; s < splits; s += 8) {
        const float pm = part_min[(size_t)s * nq + i];
        if (pm == -INFINITY) fallback = true;
        m = fminf(m, pm);
    }
    for (int off = 4; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
    const float win = m + (2.0f * (e_coeff * (qn2[i] + max_dn2) + e_abs) * 1.000001f + 1e-30f);
    if (!(win < INFINITY) && m < INFINITY) fallback = true;
    for (uint32_t s = sub; s < splits; s += 8)
        if (evict_min[(size_t)s * nq + i] <= win) fallback = true;   // a possibly valid candidate was evicted
    // (the eight lanes of a query sit in one wave: the group's verdict through its ballot bits)
    const unsigned long long fb = __ballot(fallback);
    const uint32_t grp = (threadIdx.x & 63u) >> 3;
    if ((fb >> (8u * grp)) & 0xFFull) {
        if (sub == 0) overflow_list[atomicAdd(overflow_count, 1u)] = q_base + i;
        return;
    }
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    const bool d33 = dim == 33;   // (uniform)
    double qr[33];
    if (d33) {
#pragma unroll
        for (int k = 0; k < 33; ++k) qr[k] = q[(size_t)i * 33 + k];
    }
    for (uint32_t s = sub; s < splits; s += 8) {
        const size_t o = (size_t)s * nq + i;
        const uint32_t c = ring_count[o];
        const uint32_t live = c < (uint32_t)kRing ? c : (uint32_t)kRing;
        const uint32_t first = c < (uint32_t)kRing ? 0u : c % kRing;   // oldest retained entry
        for (uint32_t t = 0; t < live; ++t) {
            const uint2 e = ring[o * kRing + (first + t) % kRing];
            if (!(__uint_as_float(e.y) <= win)) continue;   // stale: was only near an earlier running minimum
            const uint32_t j = perm ? perm[e.x] : e.x;
            double acc = 0.0;
            if (d33) {
                acc = exact_d2_33(qr, db + (size_t)j * 33);
            } else {
                for (int k = 0; k < dim; ++k) {
                    const double df = q[(size_t)i * dim + k] - db[(size_t)j * dim + k];
                    acc += df * df;
                }
            }
            if (acc < bd || (acc == bd && j < bi)) {   // equal distances: the lowest index (first found by a serial scan)
                bd = acc;
                bi = j;
            }
        }
    }
    for (int off = 4; off > 0; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    if (sub == 0) nn[i] = bi;
}

// one workgroup of sixteen waves per overflowed query: exact brute force, threads stride the database, (distance, index)
// lexicographic minimum across the workgroup.  (One wave per query until round 5: 25 ms for ONE query against 200 000 rows -- a
// single row whose candidates overflow turned a 7 ms match into a 32 ms one.)
constexpr int kExactThreads = 1024;
__global__ __launch_bounds__(kExactThreads) void nn_exact_one_k(const double* __restrict__ q, const double* __restrict__ db,
                                                                 uint32_t ndb, int dim, const uint32_t* __restrict__ list,
                                                                 uint32_t* __restrict__ nn) {
    __shared__ double wd[kExactThreads / 64];
    __shared__ uint32_t wi[kExactThreads / 64];
    const uint32_t i = list[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t j = threadIdx.x; j < ndb; j += kExactThreads) {
        double acc = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double df = q[(size_t)i * dim + k] - db[(size_t)j * dim + k];
            acc += df * df;
        }
        if (acc < bd) {   // (a thread's rows come in ascending order: the first of equal distances stays)
            bd = acc;
            bi = j;
        }
    }
    auto take = [&](double od, uint32_t oi) {
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    };
    for (int off = 32; off > 0; off >>= 1) take(__shfl_xor(bd, off, 64), (uint32_t)__shfl_xor((int)bi, off, 64));
    if (lane == 0) {
        wd[wave] = bd;
        wi[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kExactThreads / 64; ++w) take(wd[w], wi[w]);
        nn[i] = bi;
    }
}

// ... and spread over the chip: kExactSlices workgroups per overflowed query, each over a slice of the rows; the partial minima
// (distance, index) meet in nn_exact_reduce_k.  Same distances bit for bit (a row's sum is one thread's, in the order of k), same
// tie rule (the lowest index among equal distances).  One query against 200 000 rows: 1 ms on one CU, ~20 us on 64.
constexpr uint32_t kExactSlices = 64;
__global__ __launch_bounds__(256) void nn_exact_part_k(const double* __restrict__ q, const double* __restrict__ db, uint32_t ndb,
                                                        int dim, const uint32_t* __restrict__ list, double* __restrict__ part_d,
                                                        uint32_t* __restrict__ part_i) {
    __shared__ double wd[4];
    __shared__ uint32_t wi[4];
    const uint32_t i = list[blockIdx.x];
    const uint32_t per = (ndb + kExactSlices - 1) / kExactSlices, j0 = blockIdx.y * per, j1 = min(ndb, j0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t j = j0 + threadIdx.x; j < j1; j += 256u) {
        double acc = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double df = q[(size_t)i * dim + k] - db[(size_t)j * dim + k];
            acc += df * df;
        }
        if (acc < bd) {
            bd = acc;
            bi = j;
        }
    }
    auto take = [&](double od, uint32_t oi) {
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    };
    for (int off = 32; off > 0; off >>= 1) take(__shfl_xor(bd, off, 64), (uint32_t)__shfl_xor((int)bi, off, 64));
    if (lane == 0) {
        wd[wave] = bd;
        wi[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) take(wd[w], wi[w]);
        part_d[(size_t)blockIdx.x * kExactSlices + blockIdx.y] = bd;
        part_i[(size_t)blockIdx.x * kExactSlices + blockIdx.y] = bi;
    }
}
__global__ __launch_bounds__(64) void nn_exact_reduce_k(const double* __restrict__ part_d, const uint32_t* __restrict__ part_i,
                                                         const uint32_t* __restrict__ list, uint32_t* __restrict__ nn) {
    static_assert(kExactSlices == 64, "one lane per slice");
    double bd = part_d[(size_t)blockIdx.x * kExactSlices + threadIdx.x];
    uint32_t bi = part_i[(size_t)blockIdx.x * kExactSlices + threadIdx.x];
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    if (threadIdx.x == 0) nn[list[blockIdx.x]] = bi;
}
size_t nn_exact_scratch_bytes(uint32_t n_overflow) { return (size_t)n_overflow * kExactSlices * (sizeof(double) + sizeof(uint32_t)); }
void launch_nn_exact(const double* q, const double* db, uint32_t ndb, int dim, const uint32_t* list, uint32_t n_overflow, uint32_t* nn,
                     void* scratch /* nn_exact_scratch_bytes(n_overflow), or null: one workgroup per query */, hipStream_t s) {
    if (!n_overflow) return;
    if (!scratch || ndb < 16384u) {
        nn_exact_one_k<<<n_overflow, kExactThreads, 0, s>>>(q, db, ndb, dim, list, nn);
        return;
    }
    double* pd = static_cast<double*>(scratch);
    uint32_t* pi = reinterpret_cast<uint32_t*>(pd + (size_t)n_overflow * kExactSlices);
    nn_exact_part_k<<<dim3(n_overflow, kExactSlices), 256, 0, s>>>(q, db, ndb, dim, list, pd, pi);
    nn_exact_reduce_k<<<n_overflow, 64, 0, s>>>(pd, pi, list, nn);
}

__global__ void max_f32_k(const float* __restrict__ v, uint32_t n, float* __restrict__ out) {
    __shared__ float sm[256];
    float m = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, v[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// ------------------------------------------------------------------------------------------------
// MFMA screen (v_mfma_f32_32x32x16_f16): the same candidate rings as nn32_scan_k, ~6x fewer cycles per
// pair.  Round 4 (M3D_MATCH_HI_ONLY = 1, the default: m3d_match_scan.hpp): the screen contracts the HI halves only --
// k 0..32 (-2 a_hi) b_hi, k 33..38 the norms' pieces, K = 48: three MFMA steps per tile pair instead of seven -- with the
// bound's coefficient at 1.1e-3 instead of 1e-4; 200 k x 200 k x 33: scan 11.3 -> 7.4 ms, call 16.4 -> 12.2 ms, the same
// 94 279 pairs, no fallback (profiles/r04_match_kernel_stats.txt).  The full split below is the M3D_MATCH_HI_ONLY = 0 build:
// d(a, b) = |a|^2 + |b|^2 - 2 a.b is ONE K = 112 contraction of fp16 operands with fp32 accumulate:
//   k   0.. 32   (-2 a_hi) * b_hi          a = a_hi + a_lo (+ 2^-22 |a|): split fp16, data pre-scaled by a power
//   k  33.. 65   (-2 a_lo) * b_hi          of two so that max |v| <= 2048 (exact in both directions)
//   k  66.. 98   (-2 a_hi) * b_lo
//   k  99..101   |a|^2 / 2^15 in three fp16 pieces * 2^15          (norms of the REPRESENTED rows, from fp64)
//   k 102..104   2^15 * |b|^2 / 2^15 in three pieces
//   k 105..111   0
// Products of two fp16 are exact in fp32; what is lost is a_lo b_lo (<= 2^-22 |a||b|), the split residue
// and the fp32 accumulation: |d16 - d_exact| <= kMfmaECoeff (|a|^2 + |b|^2) with a wide margin (the bound
// only has to be valid; the verification is exact).  Database rows are the A operand (M), queries the B
// operand (N): in the 32x32 accumulator a lane owns ONE query (column lane & 31) and 16 of the 32 rows
// (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), so the running minimum / ring logic is per lane, with the
// two half-waves of a query treated as two more database slices.  Operands are pre-packed in fragment
// order (pack_f16_k): one coalesced 16-byte load per lane and K-step.
// ------------------------------------------------------------------------------------------------
// role 0: database rows (A operand), role 1: queries (B operand).  out: [tile][step][lane] h8.
// norm2[i] = |represented row|^2 in scaled units (fp32, for the window); rows >= n of the last tile get a huge norm in
// both roles (never a candidate, never inside a threshold).
__global__ void pack_f16_k(const double* __restrict__ f, uint32_t n, uint32_t n_tiles, double scale, int role,
                           _Float16* __restrict__ out, float* __restrict__ norm2) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_tiles * 32u) return;
    _Float16 row[kMfmaK];
    for (int k = 0; k < kMfmaK; ++k) row[k] = (_Float16)0.0f;
    double nrm = 0.0;
    if (i < n) {
        for (int k = 0; k < 33; ++k) {
            const double v = f[(size_t)i * 33 + k] * scale;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (double)hi);
            const double rep = (double)hi + (double)lo;
            nrm += rep * rep;
            if (role == 0) {
                row[k] = (_Float16)(-2.0 * (double)hi);
                if (!kMfmaHiOnly) {
                    row[33 + k] = (_Float16)(-2.0 * (double)lo);
                    row[66 + k] = (_Float16)(-2.0 * (double)hi);
                }
            } else {
                row[k] = hi;
                if (!kMfmaHiOnly) {
                    row[33 + k] = hi;
                    row[66 + k] = lo;
                }
            }
        }
    }
    double pn = (i < n) ? nrm / (double)kMfmaC : 65504.0;   // (both roles: m3d_match_mfma.hip, mfma_post)
    const int own = role == 0 ? kMfmaNormAt : kMfmaNormAt + 3, other = role == 0 ? kMfmaNormAt + 3 : kMfmaNormAt;
    for (int k = 0; k < 3; ++k) {
        const _Float16 piece = (_Float16)pn;
        row[own + k] = piece;
        pn -= (double)piece;
        row[other + k] = (_Float16)kMfmaC;
    }
    if (i < n) norm2[i] = (float)nrm;
    const uint32_t t = i / 32u, r = i % 32u;
    for (int k = 0; k < kMfmaK; ++k) {
        const uint32_t st = k / 16, kk = k % 16, lane = r + 32u * (kk / 8), j = kk % 8;
        out[(((size_t)t * kMfmaSteps + st) * 64 + lane) * 8 + j] = row[k];
    }
}

// max |v| over an n x 33 matrix (non-finite values propagate as +inf so the caller can bail out)
__global__ void max_abs_k(const double* __restrict__ f, size_t count, double* __restrict__ out) {
    __shared__ double sm[256];
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const double v = fabs(f[i]);
        m = (v > m || v != v) ? (v != v ? INFINITY : v) : m;
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

void launch_to_f32_33(const double* f, uint32_t n, float* out32, float* norm2, float* max_norm2, hipStream_t s) {
    if (!n) return;
    to_f32_k<<<(n + 255) / 256, 256, 0, s>>>(f, n, out32, norm2);
    max_f32_k<<<1, 256, 0, s>>>(norm2, n, max_norm2);
}

// Workspace (device): ring splits x nq x kRing uint2, ring_count / part_min / evict_min splits x nq,
// overflow_list nq u32, overflow_count 1 u32 (zeroed here).  max_dn2 = max |b|^2 over the database (host value).
// d32 must carry one spare row behind row ndb-1 (prefetch).  *h_overflow = queries that took the exact fallback.
hipError_t launch_nn_screened33(const double* q, const float* q32, const float* qn, uint32_t nq, const double* db,
                                const float* d32, uint32_t ndb, float max_dn2, uint32_t splits, uint2* ring,
                                uint32_t* ring_count, float* part_min, float* evict_min, uint32_t* overflow_list,
                                uint32_t* overflow_count, uint32_t* nn, uint32_t* h_overflow, hipStream_t s) {
    constexpr int DIM = 33;
    *h_overflow = 0;
    if (!nq || !ndb) return hipSuccess;
    (void)hipMemsetAsync(overflow_count, 0, sizeof(uint32_t), s);
    const uint32_t per = (ndb + splits - 1) / splits;
    const float e_coeff = (2.0f * DIM + 16.0f) * 5.9604645e-08f * 1.5f;
    nn32_scan_k<<<dim3((nq + 511) / 512, splits), 256, 0, s>>>(q32, nq, d32, ndb, per, e_coeff, max_dn2, ring,
                                                              ring_count, part_min, evict_min);
    nn64_verify_k<<<(nq + 31) / 32, 256, 0, s>>>(q, nq, db, DIM, ring, ring_count, part_min, evict_min, splits,
                                                   e_coeff, 0.0f, max_dn2, nullptr, qn, nn, overflow_list, overflow_count, 0u, nullptr);
    hipError_t e = hipMemcpyAsync(h_overflow, overflow_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (*h_overflow) nn_exact_one_k<<<*h_overflow, kExactThreads, 0, s>>>(q, db, ndb, DIM, overflow_list, nn);
    return hipGetLastError();
}

// ---- cross-check (src/correspondence_matching.cpp:64-78) on the device ------------------------------------------------------------
// pair (i, nn_ab[i]) is kept when nn_ba[nn_ab[i]] == i; pairs in the order of i, as the reference's loop pushes them.  Three small
// launches (count per 1024 queries, offsets, write); the pairs go straight to page-locked host memory (u32 x 2, ~0.75 MB for 200 000
// queries) instead of both index arrays (1.6 MB) + a host loop whose branch mispredicts every other query (0.3 ms together).
constexpr uint32_t kMutualPerThread = 4, kMutualPerBlock = 256 * kMutualPerThread;
__device__ __forceinline__ uint32_t mutual_flags(const uint32_t* __restrict__ nn_ab, const uint32_t* __restrict__ nn_ba, uint32_t na,
                                                 uint32_t nb, uint32_t i0, uint32_t (&j)[kMutualPerThread]) {
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t k = 0; k < kMutualPerThread; ++k) {
        const uint32_t i = i0 + k;
        j[k] = i < na ? nn_ab[i] : 0xFFFFFFFFu;
        if (j[k] < nb && nn_ba[j[k]] == i) bits |= 1u << k;
    }
    return bits;
}
__global__ __launch_bounds__(256) void mutual_count_k(const uint32_t* __restrict__ nn_ab, const uint32_t* __restrict__ nn_ba, uint32_t na,
                                                       uint32_t nb, uint32_t* __restrict__ block_count) {
    __shared__ uint32_t wsum[4];
    uint32_t j[kMutualPerThread];
    uint32_t c = __popc(mutual_flags(nn_ab, nn_ba, na, nb, blockIdx.x * kMutualPerBlock + threadIdx.x * kMutualPerThread, j));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// exclusive offsets in place, the total to *total (one workgroup; a few hundred blocks)
__global__ __launch_bounds__(256) void mutual_offsets_k(uint32_t* __restrict__ block_count, uint32_t n_blocks, uint32_t* __restrict__ total) {
    __shared__ uint32_t part[256];
    const uint32_t per = (n_blocks + 255u) / 256u, b0 = threadIdx.x * per, b1 = min(n_blocks, b0 + per);
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; ++b) sum += block_count[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 256; ++t) {
            const uint32_t v = part[t];
            part[t] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t v = block_count[b];
        block_count[b] = run;
        run += v;
    }
}
__global__ __launch_bounds__(256) void mutual_write_k(const uint32_t* __restrict__ nn_ab, const uint32_t* __restrict__ nn_ba, uint32_t na,
                                                       uint32_t nb, const uint32_t* __restrict__ block_offset, uint2* __restrict__ out) {
    __shared__ uint32_t wsum[4];
    uint32_t j[kMutualPerThread];
    const uint32_t i0 = blockIdx.x * kMutualPerBlock + threadIdx.x * kMutualPerThread;
    const uint32_t bits = mutual_flags(nn_ab, nn_ba, na, nb, i0, j);
    const uint32_t c = __popc(bits), lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = c;   // inclusive prefix over the wave's lanes
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63u) wsum[wave] = incl;
    __syncthreads();
    uint32_t at = block_offset[blockIdx.x] + incl - c;
    for (uint32_t w = 0; w < wave; ++w) at += wsum[w];
#pragma unroll
    for (uint32_t k = 0; k < kMutualPerThread; ++k)
        if (bits & (1u << k)) out[at++] = make_uint2(i0 + k, j[k]);
}
void launch_mutual_pairs(const uint32_t* nn_ab, const uint32_t* nn_ba, uint32_t na, uint32_t nb, uint32_t* block_scratch, uint32_t* total,
                         uint2* out, hipStream_t s) {
    const uint32_t blocks = mutual_blocks(na);
    if (!blocks) return;
    mutual_count_k<<<blocks, 256, 0, s>>>(nn_ab, nn_ba, na, nb, block_scratch);
    mutual_offsets_k<<<1, 256, 0, s>>>(block_scratch, blocks, total);
    mutual_write_k<<<blocks, 256, 0, s>>>(nn_ab, nn_ba, na, nb, block_scratch, out);
}
uint32_t mutual_blocks(uint32_t na) { return (na + kMutualPerBlock - 1) / kMutualPerBlock; }

// ---- MFMA path launchers -------------------------------------------------------------------------
uint32_t mfma_tiles(uint32_t n) { return (n + 31u) / 32u; }
// tiles padded so that a wave's two query tiles always exist
uint32_t mfma_query_tiles(uint32_t n) { return ((n + 511u) / 512u) * 16u; }

// many workgroups, one atomicMax on the bit pattern per workgroup (the values are >= 0: their order is the integers';
// a NaN is skipped, as fmaxf does).  The one-workgroup kernel took 113 us for 200 000 values.
__global__ void max_f32_wide_k(const float* __restrict__ v, uint32_t n, uint32_t* __restrict__ out_bits) {
    __shared__ float sm[256];
    float m = 0.0f;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) m = fmaxf(m, v[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint(fmaxf(sm[0], 0.0f)));
}
void launch_max_f32(const float* v, uint32_t n, float* out, hipStream_t s, bool accumulate) {
    if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float), s);
    if (n) max_f32_wide_k<<<std::min<uint32_t>((n + 255) / 256, 256), 256, 0, s>>>(v, n, reinterpret_cast<uint32_t*>(out));
}
void launch_max_abs(const double* f, size_t count, double* partial /* kMaxAbsPartials */, hipStream_t s) {
    // (256 workgroups read 52.8 MB at 1.1 TB/s: 48 us per matrix; 2048 of them at 3.5 TB/s)
    max_abs_k<<<kMaxAbsPartials, 256, 0, s>>>(f, count, partial);
}
// pack_f16_k for BOTH roles in one pass, a wave per tile of 32 rows (hi-only operands, K = 48).  pack_f16_k gives every row to
// one thread: 264-byte strides on the way in, 48 two-byte stores per row on the way out -- 108 us per 200 000 rows and layout,
// four launches per call.  Here the tile's 32 x 33 doubles arrive as ONE contiguous 8.4 KB block (coalesced), are split in LDS,
// and every lane stores its MFMA fragment -- eight consecutive K-slots of one row -- as one 16-byte word per step and layout.
// The same values bit for bit: hi = RN16(v scale), the norm summed over k = 0 .. 32 in order from hi + lo in fp64.
#if M3D_MATCH_HI_ONLY
// perm (match_order_part): row r of tile t is row perm[32 t + r] - perm_base of f, only the database layout is written, no norms
__global__ __launch_bounds__(256) void pack_f16_both_k(const double* __restrict__ f, uint32_t n, uint32_t tiles_a, uint32_t tiles_b,
                                                        double scale, h8* __restrict__ out_a, h8* __restrict__ out_b,
                                                        float* __restrict__ norm2, const uint32_t* __restrict__ perm,
                                                        uint32_t perm_base) {
    static_assert(kMfmaHiOnly && kMfmaK == 48 && kMfmaNormAt == 33, "the cooperative packer writes the hi-only layout");
    __shared__ _Float16 hi16[4][32][kMfmaK];   // [wave][row][K-slot]: data 0 .. 32, the row's norm pieces 33 .. 35, zero behind
    __shared__ double sq[4][32][34];            // (hi + lo)^2 per element (34: the rows of a half-wave fall on different banks)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t t = blockIdx.x * 4u + wave;
    if (t >= max(tiles_a, tiles_b)) return;   // (wave-uniform; no barrier below: a wave only talks to itself)
    const size_t first = (size_t)t * 32u * 33u, total = (size_t)n * 33u;
    for (uint32_t e = lane; e < 32u * 33u; e += 64u) {
        const uint32_t r = e / 33u, k = e % 33u;
        _Float16 hi = (_Float16)0.0f;
        double rep2 = 0.0;
        const uint32_t src = perm ? perm[t * 32u + r] - perm_base : t * 32u + r;   // (one tile's 32 entries: broadcast loads)
        if (perm ? src < n : first + e < total) {
            const double v = f[perm ? (size_t)src * 33u + k : first + e] * scale;
            hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (double)hi);
            const double rep = (double)hi + (double)lo;
            rep2 = rep * rep;
        }
        hi16[wave][r][k] = hi;
        sq[wave][r][k] = rep2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < 32u) {   // lane r: the norm of row r, summed in pack_f16_k's order
        const uint32_t i = perm ? perm[t * 32u + lane] - perm_base : t * 32u + lane;
        double nrm = 0.0;
        for (int k = 0; k < 33; ++k) nrm += sq[wave][lane][k];
        if (i < n && !perm) norm2[i] = (float)nrm;
        // the norm's three fp16 pieces (padding rows carry 65504 in both roles: as a database row never the nearest, as a query beyond
        // every threshold of the reverse search -- the scan's hot predicates then need no "is this a real row / query" term)
        double pa = i < n ? nrm / (double)kMfmaC : 65504.0, pb = pa;
        for (int k = 0; k < 3; ++k) {
            const _Float16 qa = (_Float16)pa, qb = (_Float16)pb;
            hi16[wave][lane][33 + k] = qa;        // role 0's own pieces
            hi16[wave][lane][36 + k] = qb;        // role 1's own pieces (equal to role 0's for a real row)
            pa -= (double)qa;
            pb -= (double)qb;
        }
        for (int k = 39; k < kMfmaK; ++k) hi16[wave][lane][k] = (_Float16)0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t r = lane & 31u, hb = lane >> 5;
    const bool real = (perm ? perm[t * 32u + r] - perm_base : t * 32u + r) < n;   // (a padding row's data slots are +0 in both layouts)
    const _Float16 c = (_Float16)kMfmaC;
#pragma unroll
    for (int st = 0; st < kMfmaSteps; ++st) {
        h8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = st * 16 + (int)hb * 8 + j;
            const _Float16 v = hi16[wave][r][k];
            // role 0 (database rows): -2 hi, own norm pieces at 33 .. 35, 2^15 at 36 .. 38;  role 1 (queries): hi, 2^15 at 33 .. 35, own
            // pieces at 36 .. 38
            a[j] = k < 33 ? (real ? (_Float16)(-2.0 * (double)v) : (_Float16)0.0f) : (k < 36 ? v : (k < 39 ? c : (_Float16)0.0f));
            b[j] = k < 33 ? v : (k < 36 ? c : (k < 39 ? v : (_Float16)0.0f));
        }
        if (t < tiles_a) out_a[((size_t)t * kMfmaSteps + st) * 64 + lane] = a;
        if (t < tiles_b && !perm) out_b[((size_t)t * kMfmaSteps + st) * 64 + lane] = b;
    }
}
#endif
void launch_pack_f16_both(const double* f, uint32_t n, double scale, void* out_a, void* out_b, float* norm2, hipStream_t s) {
    const uint32_t ta = mfma_tiles(n), tb = mfma_query_tiles(n), tm = std::max(ta, tb);
    if (!tm) return;
#if M3D_MATCH_HI_ONLY
    pack_f16_both_k<<<(tm + 3) / 4, 256, 0, s>>>(f, n, ta, tb, scale, reinterpret_cast<h8*>(out_a), reinterpret_cast<h8*>(out_b), norm2,
                                                 nullptr, 0u);
#else
    launch_pack_f16(f, n, ta, scale, 0, out_a, norm2, s);
    launch_pack_f16(f, n, tb, scale, 1, out_b, norm2, s);
#endif
}
void launch_pack_f16(const double* f, uint32_t n, uint32_t n_tiles, double scale, int role, void* out, float* norm2,
                     hipStream_t s) {
    if (!n_tiles) return;
    pack_f16_k<<<(n_tiles * 32u + 255) / 256, 256, 0, s>>>(f, n, n_tiles, scale, role,
                                                           reinterpret_cast<_Float16*>(out), norm2);
}

// ---- both directions of ANNMatcher::Match in ONE scan (m3d_match_scan.hpp, RevOut) ------------------------------
// thresholds of the reverse search from its warm-up minima: thr[j] = min over slices + 2 E_j (E as in the forward
// windows, with the roles exchanged); a row without a usable bound gets thr = -inf and a counter past the cap, which
// sends it to the exact fallback.  Rows >= n of the last tile: -inf.
__global__ void rev_threshold_k(const float* __restrict__ premin, uint32_t slices, uint32_t n, uint32_t n_pad,
                                const float* __restrict__ n2, const float* __restrict__ max_other_n2_p, float* __restrict__ thr,
                                uint32_t* __restrict__ cnt, bool first_set, const uint32_t* __restrict__ perm, uint32_t perm_base) {
    // (perm: thr[p] is the threshold of the row at position p of the ordered part, perm[p] - perm_base of the part's rows)
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pad) return;
    const uint32_t j = perm ? perm[p] - perm_base : p;
    if (j >= n) {
        thr[p] = -INFINITY;
        return;
    }
    float m = INFINITY;
    for (uint32_t s = 0; s < slices; ++s) m = fminf(m, premin[(size_t)s * n + j]);
    const float two_e = 2.0f * (kMfmaECoeff * (n2[j] + *max_other_n2_p) + kMfmaEAbs) * 1.000001f + 1e-30f;
    const float t = m + two_e;
    const bool usable = t < INFINITY && m == m;
    thr[p] = usable ? t : -INFINITY;
    // (a later set of thresholds -- the same minima under the norm bound of more queries -- keeps the counters of the scans so far)
    if (first_set) cnt[j] = usable ? 0u : 0x80000000u;
    else if (!usable) atomicOr(cnt + j, 0x80000000u);
}
// thr4[g] = max of thr[4g .. 4g + 3]: what the scan's fast path tests the minimum of a run of four rows against
__global__ void rev_threshold4_k(const float* __restrict__ thr, uint32_t n_groups, float* __restrict__ thr4) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_groups) return;
    const float4 t = reinterpret_cast<const float4*>(thr)[g];
    thr4[g] = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
}

// The rows of a part ordered by threshold within chunks of kOrderChunk consecutive rows (one workgroup per chunk, a bitonic
// network in the LDS; descending, equal thresholds by row: the order is a function of the thresholds alone).  Why: the scan tests a
// lane's minimum over 16 rows against the LARGEST of their thresholds; in the caller's order the thresholds of neighbouring rows
// differ by octaves (they follow the local density of the descriptors) and the test let 11.4 M of 39 M side-tiles into the slow
// path for 2.4 M candidates (200 k x 200 k, tools/gpu/scan_sorted_probe.py); ordered, 2.4 M.  Within chunks only: ordered as a
// whole, the dense rows come last and every query's running minimum keeps improving to the end (ring appends 3.5 M -> 9.2 M).
constexpr uint32_t kOrderChunk = 1024;
__global__ __launch_bounds__(256) void rev_order_k(float* __restrict__ thr, uint32_t n_pad, uint32_t* __restrict__ perm, uint32_t perm_base) {
    __shared__ unsigned long long key[kOrderChunk];
    const uint32_t c0 = blockIdx.x * kOrderChunk;
    for (uint32_t i = threadIdx.x; i < kOrderChunk; i += 256u) {
        unsigned long long k = ~0ull;   // (past the part's end: behind everything)
        if (c0 + i < n_pad) {
            const uint32_t b = __float_as_uint(thr[c0 + i]);
            const uint32_t asc = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);   // the floats' order as unsigned integers
            k = ((unsigned long long)(~asc) << 32) | i;
        }
        key[i] = k;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= kOrderChunk; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < kOrderChunk / 2; t += 256u) {
                const uint32_t lo = 2u * t - (t & (stride - 1u)), hi = lo + stride;
                const bool up = (lo & size) == 0u;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a > b) == up) {
                    key[lo] = b;
                    key[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < kOrderChunk; i += 256u) {
        if (c0 + i >= n_pad) continue;
        const unsigned long long k = key[i];
        const uint32_t asc = ~(uint32_t)(k >> 32);
        thr[c0 + i] = __uint_as_float(asc ^ ((asc >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        perm[c0 + i] = perm_base + c0 + (uint32_t)k;
    }
}

// the scan's per-(slice, query) candidate lists sorted by database row: entry (row, d16) of query q -> slot of row
__global__ void rev_bin_k(const uint2* __restrict__ list, const uint32_t* __restrict__ list_cnt, uint32_t nq, size_t lists,
                          uint32_t* __restrict__ cnt, uint2* __restrict__ cand, uint32_t q_base, const uint32_t* __restrict__ perm) {
    const size_t o = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (o >= lists) return;
    const uint32_t c = list_cnt[o];   // <= kRevLane (a full list sends the rest straight to the rows' slots)
    const uint32_t q = q_base + (uint32_t)(o % nq);
    for (uint32_t t = 0; t < c; ++t) {
        const uint2 e = list[o * kRevLane + t];
        const uint32_t row = perm[e.x];   // (the lists carry positions in the packed database)
        const uint32_t slot = atomicAdd(cnt + row, 1u) & 0x7FFFFFFFu;
        if (slot < (uint32_t)kRevCap) cand[(size_t)row * kRevCap + slot] = make_uint2(q, e.y);
    }
}

// exact distances of a row's candidates (serial-order fp64 accumulation, lowest index among equal distances: what
// NearestSearch(dst, src) returns for query j).  Eight lanes per row: first the smallest screen value m of the list,
// then only the candidates inside the final window m + 2 E_j are evaluated exactly (the list was collected against the
// looser threshold of the sample), the lanes taking them in turn; (distance, index) minimum across the eight.
// An overflowed or empty list takes the exact fallback.
__global__ __launch_bounds__(256) void nn64_verify_rev_k(const double* __restrict__ q, uint32_t nq, const double* __restrict__ db,
                                                          int dim, const uint32_t* __restrict__ cnt,
                                                          const uint2* __restrict__ cand, const float* __restrict__ qn2,
                                                          const float* __restrict__ max_dn2_p, uint32_t* __restrict__ nn,
                                                          uint32_t* __restrict__ overflow_list,
                                                          uint32_t* __restrict__ overflow_count) {
    const uint32_t j = blockIdx.x * 32u + (threadIdx.x >> 3), sub = threadIdx.x & 7u;
    if (j >= nq) return;   // (whole groups of eight lanes)
    const uint32_t c = cnt[j];
    if (c == 0u || c > (uint32_t)kRevCap) {   // (bit 31: flagged by rev_threshold_k)
        if (sub == 0) overflow_list[atomicAdd(overflow_count, 1u)] = j;
        return;
    }
    const uint2* __restrict__ my = cand + (size_t)j * kRevCap;
    float m = INFINITY;
    for (uint32_t t = sub; t < c; t += 8) m = fminf(m, __uint_as_float(my[t].y));
    for (int off = 4; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
    const float win = m + (2.0f * (kMfmaECoeff * (qn2[j] + *max_dn2_p) + kMfmaEAbs) * 1.000001f + 1e-30f);
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    const bool d33 = dim == 33;   // (uniform)
    double qr[33];
    if (d33) {
#pragma unroll
        for (int k = 0; k < 33; ++k) qr[k] = q[(size_t)j * 33 + k];
    }
    for (uint32_t t = sub; t < c; t += 8) {
        const uint2 e = my[t];
        if (!(__uint_as_float(e.y) <= win)) continue;
        const uint32_t i = e.x;
        double acc = 0.0;
        if (d33) {
            acc = exact_d2_33(qr, db + (size_t)i * 33);
        } else {
            for (int k = 0; k < dim; ++k) {
                const double df = q[(size_t)j * dim + k] - db[(size_t)i * dim + k];
                acc += df * df;
            }
        }
        if (acc < bd || (acc == bd && i < bi)) {
            bd = acc;
            bi = i;
        }
    }
    for (int off = 4; off > 0; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    if (sub == 0) nn[j] = bi;
}

// src -> dst AND dst -> src nearest neighbours from one pass over the 32 x 32 product tiles (MatchWork, m3d_reg_kernels.hpp).
//   forward: queries = a (qB_a, role 1), database = b (dA_b, role 0): rings of 2 * splits slices + nn64_verify_k;
//   reverse: the warm-up minima of b's rows over the first tiles of a (qB_b against dA_a) give the per-row thresholds
//            the main scan tests its accumulators against; nn64_verify_rev_k evaluates the collected candidates.
// The steps are launched one by one by the host (m3d_registration.cpp: match_mfma): a call whose matrices come from the host
// runs them per slice of the queries and part of the database while the next slice is on the link.
static size_t mfma_tile_entries() { return (size_t)kMfmaSteps * 64; }   // h8 entries of a packed tile
void match_pack(const MatchWork& w, int side, uint32_t row0, uint32_t rows, double scale, hipStream_t s) {
    const double* f = (side == 0 ? w.a : w.b) + (size_t)row0 * 33;
    h8* dA = reinterpret_cast<h8*>(side == 0 ? w.dA_a : w.dA_b) + (size_t)(row0 / 32u) * mfma_tile_entries();
    h8* qB = reinterpret_cast<h8*>(side == 0 ? w.qB_a : w.qB_b) + (size_t)(row0 / 32u) * mfma_tile_entries();
    float* n2 = (side == 0 ? w.an2 : w.bn2) + row0;
    launch_pack_f16_both(f, rows, scale, dA, qB, n2, s);
    launch_max_f32(n2, rows, side == 0 ? w.max_an2 : w.max_bn2, s, true);
}
void match_forward_warm(const MatchWork& w, uint32_t q0, uint32_t nq, hipStream_t s) {
    const h8* qB = reinterpret_cast<const h8*>(w.qB_a) + (size_t)(q0 / 32u) * mfma_tile_entries();
    const uint32_t n_tiles = mfma_tiles(w.nb);
    // the first 1/16 of the database (same grid as the main pass: every slice of it takes a share)
    const uint32_t warm = std::min<uint32_t>(n_tiles, std::max<uint32_t>(w.splits, n_tiles / 16));
    launch_nn16_warm(qB, w.an2 + q0, nq, w.dA_b, w.nb, warm, w.splits, w.max_bn2, w.premin + (size_t)2 * w.splits * q0, s);
}
void match_reverse_thresholds(const MatchWork& w, uint32_t row0, uint32_t rows, int set, hipStream_t s) {
    const uint32_t a_tiles = mfma_tiles(w.na), b_tiles = mfma_tiles(w.nb);
    float* premin = w.rev_premin + (size_t)2 * w.splits_r * row0;
    if (set == 0) {
        // reverse warm-up: the rows against the first 1/8 of a (1/16: twice the candidates and slow-path detours of the
        // main scan for half the warm-up: 17.3 against 16.9 ms on 200 k x 200 k; 1/4: 17.0)
        const uint32_t warm_r = std::min<uint32_t>(a_tiles, std::max<uint32_t>(w.splits_r, a_tiles / 8));
        const h8* qB = reinterpret_cast<const h8*>(w.qB_b) + (size_t)(row0 / 32u) * mfma_tile_entries();
        launch_nn16_warm(qB, w.bn2 + row0, rows, w.dA_a, w.na, warm_r, w.splits_r, w.max_an2, premin, s);
    }
    float* thr = w.rthr + (size_t)set * 40u * b_tiles;   // (a set: 32 row + 8 run thresholds per tile of b)
    float* thr4 = thr + (size_t)b_tiles * 32u;
    const uint32_t pad = ((rows + 31u) / 32u) * 32u;
    // set 0: in the rows' order (match_order_part orders them next); later sets: straight into the order set 0 found
    rev_threshold_k<<<(pad + 255) / 256, 256, 0, s>>>(premin, 2 * w.splits_r, rows, pad, w.bn2 + row0, w.max_an2, thr + row0,
                                                       w.rcnt + row0, set == 0, set == 0 ? nullptr : w.rperm + row0, row0);
    if (set != 0) rev_threshold4_k<<<(pad / 4u + 255) / 256, 256, 0, s>>>(thr + row0, pad / 4u, thr4 + row0 / 4u);
}
void match_order_part(const MatchWork& w, uint32_t row0, uint32_t rows, hipStream_t s) {
    const uint32_t b_tiles = mfma_tiles(w.nb), pad = ((rows + 31u) / 32u) * 32u;
    float* thr = w.rthr;   // (set 0)
    float* thr4 = thr + (size_t)b_tiles * 32u;
    rev_order_k<<<(pad + kOrderChunk - 1) / kOrderChunk, 256, 0, s>>>(thr + row0, pad, w.rperm + row0, row0);
    rev_threshold4_k<<<(pad / 4u + 255) / 256, 256, 0, s>>>(thr + row0, pad / 4u, thr4 + row0 / 4u);
    h8* dA = reinterpret_cast<h8*>(w.dA_b) + (size_t)(row0 / 32u) * mfma_tile_entries();
    const uint32_t ta = pad / 32u;
    pack_f16_both_k<<<(ta + 3) / 4, 256, 0, s>>>(w.b + (size_t)row0 * 33, rows, ta, 0u, w.scale, dA, nullptr, nullptr, w.rperm + row0, row0);
}
void match_scan(const MatchWork& w, uint32_t q0, uint32_t nq, uint32_t split0, uint32_t splits, uint32_t tile_end, int set,
                hipStream_t s) {
    const uint32_t b_tiles = mfma_tiles(w.nb);
    const size_t base = (size_t)2 * w.splits * q0;   // per-(slice, query) arrays of the queries from q0: [slice][nq] behind those of the queries before
    RevOut rev;
    rev.thr = w.rthr + (size_t)set * 40u * b_tiles;
    rev.thr4 = rev.thr + (size_t)b_tiles * 32u;
    rev.cnt = w.rcnt;
    rev.cand = w.rcand;
    rev.list = w.rlist + base * kRevLane;
    rev.list_cnt = w.rlist_cnt + base;
    rev.q_base = q0;
    rev.perm = w.rperm;
    const h8* qB = reinterpret_cast<const h8*>(w.qB_a) + (size_t)(q0 / 32u) * mfma_tile_entries();
    launch_nn16_scan(qB, w.an2 + q0, nq, w.dA_b, w.nb, std::min(tile_end, b_tiles), w.plan, split0, splits, 2 * w.splits, w.max_bn2,
                     w.premin + base, w.ring + base * kRing, w.ring_count + base, w.part_min + base, w.evict_min + base, s, &rev);
}
void match_verify_forward(const MatchWork& w, uint32_t q0, uint32_t nq, hipStream_t s) {
    const size_t base = (size_t)2 * w.splits * q0;
    nn64_verify_k<<<(nq + 31) / 32, 256, 0, s>>>(w.a + (size_t)q0 * 33, nq, w.b, 33, w.ring + base * kRing, w.ring_count + base,
                                                   w.part_min + base, w.evict_min + base, 2 * w.splits, kMfmaECoeff, kMfmaEAbs, 0.0f,
                                                   w.max_bn2, w.an2 + q0, w.nn_ab + q0, w.overflow_list, w.overflow_count, q0, w.rperm);
}
void match_reverse_bin(const MatchWork& w, uint32_t q0, uint32_t nq, hipStream_t s) {
    const size_t base = (size_t)2 * w.splits * q0, lists = (size_t)2 * w.splits * nq;
    rev_bin_k<<<(uint32_t)((lists + 255) / 256), 256, 0, s>>>(w.rlist + base * kRevLane, w.rlist_cnt + base, nq, lists, w.rcnt, w.rcand, q0, w.rperm);
}
void match_verify_reverse(const MatchWork& w, hipStream_t s) {
    nn64_verify_rev_k<<<(w.nb + 31) / 32, 256, 0, s>>>(w.b, w.nb, w.a, 33, w.rcnt, w.rcand, w.bn2, w.max_an2, w.nn_ba,
                                                        w.overflow_list_r, w.overflow_count_r);
}
hipError_t match_exact_fallbacks(const MatchWork& w, const uint32_t* h_overflow /* [2] */, void* scratch, hipStream_t s) {
    launch_nn_exact(w.a, w.b, w.nb, 33, w.overflow_list, h_overflow[0], w.nn_ab, scratch, s);
    launch_nn_exact(w.b, w.a, w.na, 33, w.overflow_list_r, h_overflow[1], w.nn_ba, scratch, s);   // (stream order: the scratch is free again)
    return hipGetLastError();
}

}  // namespace m3d
