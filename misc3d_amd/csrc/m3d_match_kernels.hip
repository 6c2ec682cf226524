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

#include "m3d_fp.hpp"

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
struct ScanState {
    float best = INFINITY, win = INFINITY, ev = INFINITY;
    uint32_t cnt = 0;
};

__device__ __forceinline__ void scan_step(ScanState& st, float dv, float two_e, bool live, uint32_t j,
                                          uint2* __restrict__ my) {
    if (dv <= st.win && live) {   // also taken while win == +inf
        const uint32_t slot = st.cnt % kRing;
        if (st.cnt >= (uint32_t)kRing) st.ev = fminf(st.ev, __uint_as_float(my[slot].y));
        my[slot] = make_uint2(j, __float_as_uint(dv));
        st.cnt++;
        if (dv < st.best) {
            st.best = dv;
            st.win = dv + two_e;
        }
    }
}

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
__global__ void nn64_verify_k(const double* __restrict__ q, uint32_t nq, const double* __restrict__ db, int dim,
                              const uint2* __restrict__ ring, const uint32_t* __restrict__ ring_count,
                              const float* __restrict__ part_min, const float* __restrict__ evict_min,
                              uint32_t splits, float e_coeff, float e_abs, float max_dn2,
                              const float* __restrict__ qn2, uint32_t* __restrict__ nn, uint32_t* __restrict__ overflow_list,
                              uint32_t* __restrict__ overflow_count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nq) return;
    float m = INFINITY;
    bool fallback = false;
    for (uint32_t s = 0; s < splits; ++s) {
        const float pm = part_min[(size_t)s * nq + i];
        if (pm == -INFINITY) fallback = true;
        m = fminf(m, pm);
    }
    const float win = m + (2.0f * (e_coeff * (qn2[i] + max_dn2) + e_abs) * 1.000001f + 1e-30f);
    if (!(win < INFINITY) && m < INFINITY) fallback = true;
    for (uint32_t s = 0; s < splits; ++s)
        if (evict_min[(size_t)s * nq + i] <= win) fallback = true;   // a possibly valid candidate was evicted
    if (fallback) {
        overflow_list[atomicAdd(overflow_count, 1u)] = i;
        return;
    }
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t s = 0; s < splits; ++s) {
        const size_t o = (size_t)s * nq + i;
        const uint32_t c = ring_count[o];
        const uint32_t live = c < (uint32_t)kRing ? c : (uint32_t)kRing;
        const uint32_t first = c < (uint32_t)kRing ? 0u : c % kRing;   // oldest retained entry
        for (uint32_t t = 0; t < live; ++t) {
            const uint2 e = ring[o * kRing + (first + t) % kRing];
            if (!(__uint_as_float(e.y) <= win)) continue;   // stale: was only near an earlier running minimum
            const uint32_t j = e.x;
            double acc = 0.0;
            for (int k = 0; k < dim; ++k) {
                const double df = q[(size_t)i * dim + k] - db[(size_t)j * dim + k];
                acc += df * df;
            }
            if (acc < bd || (acc == bd && j < bi)) {   // equal distances: the lowest index (first found by a serial scan)
                bd = acc;
                bi = j;
            }
        }
    }
    nn[i] = bi;
}

// one wave per overflowed query: exact brute force, lanes stride the database, (distance, index)
// lexicographic minimum across lanes.
__global__ __launch_bounds__(64) void nn_exact_one_k(const double* __restrict__ q, const double* __restrict__ db,
                                                      uint32_t ndb, int dim, const uint32_t* __restrict__ list,
                                                      uint32_t* __restrict__ nn) {
    const uint32_t i = list[blockIdx.x];
    const int lane = threadIdx.x;
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t j = lane; j < ndb; j += 64) {
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
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    if (lane == 0) nn[i] = bi;
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
// pair.  d(a, b) = |a|^2 + |b|^2 - 2 a.b is ONE K = 112 contraction of fp16 operands with fp32 accumulate:
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
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMfmaK = 112, kMfmaSteps = 7;
constexpr float kMfmaECoeff = 1.0e-4f;
// Absolute part of the bound (scaled^2 units; data scaled to max |v| in [1024, 2048)): fp16 underflow.  Values far
// below the largest one lose their lo piece to the subnormal spacing 2^-24 -- or to zero if the matrix core
// flushes subnormal inputs (<= 6.1e-5 per element): 2 * 6.1e-5 * sum(|a_k| + |b_k|) <= 16.5, plus two norm
// remainders <= 6.1e-5 * 2^15 = 2 each.  32 covers it; for well-scaled data it is noise next to the relative
// part (~1e3), for badly scaled data (one huge row) it correctly sends everything to the exact path.
constexpr float kMfmaEAbs = 32.0f;
constexpr float kMfmaC = 32768.0f;   // 2^15

// role 0: database rows (A operand), role 1: queries (B operand).  out: [tile][step][lane] h8.
// norm2[i] = |represented row|^2 in scaled units (fp32, for the window); rows >= n of the last tile: role 0
// gets a huge norm (never a candidate), role 1 zeros.
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
                row[33 + k] = (_Float16)(-2.0 * (double)lo);
                row[66 + k] = (_Float16)(-2.0 * (double)hi);
            } else {
                row[k] = hi;
                row[33 + k] = hi;
                row[66 + k] = lo;
            }
        }
    }
    double pn = (i < n) ? nrm / (double)kMfmaC : (role == 0 ? 65504.0 : 0.0);
    const int own = role == 0 ? 99 : 102, other = role == 0 ? 102 : 99;
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

__device__ __forceinline__ void mfma_post(const f32x16& acc, ScanState& st, float two_e, bool live, uint32_t row0,
                                          uint32_t ndb, uint2* __restrict__ my) {
    // min(a, b) = med3(a, b, -inf): v_med3_f32 needs no NaN canonicalisation of its inputs (fminf does, which
    // would triple the VALU work of this steady-state path).  A NaN entry cannot be a candidate anyway.
    const float ninf = -INFINITY;
    float t[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] = __builtin_amdgcn_fmed3f(acc[2 * r], acc[2 * r + 1], ninf);
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = __builtin_amdgcn_fmed3f(t[2 * r], t[2 * r + 1], ninf);
    const float tmin = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(t[0], t[1], ninf),
                                               __builtin_amdgcn_fmed3f(t[2], t[3], ninf), ninf);
    if (tmin <= st.win && live) {   // rare once the running minimum has settled
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = row0 + (uint32_t)((r & 3) + 8 * (r >> 2));
            scan_step(st, acc[r], two_e, row < ndb, row, my);
        }
    }
}

// One wave: 64 queries (two 32-column B tiles held in registers for the whole scan) against a slice of
// the database, one 32-row A tile at a time (next tile's fragments prefetched).
__global__ __launch_bounds__(256) void nn16_scan_k(const h8* __restrict__ qB, const float* __restrict__ qn2,
                                                    uint32_t nq, const h8* __restrict__ dA, uint32_t ndb,
                                                    uint32_t tiles_per_split, float max_dn2,
                                                    uint2* __restrict__ ring, uint32_t* __restrict__ ring_count,
                                                    float* __restrict__ part_min, float* __restrict__ evict_min) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t qt0 = (blockIdx.x * 4u + wave) * 2u;   // first of this wave's two query tiles
    const uint32_t half = lane >> 5;
    h8 b0[kMfmaSteps], b1[kMfmaSteps];
#pragma unroll
    for (int s = 0; s < kMfmaSteps; ++s) {
        b0[s] = qB[((size_t)qt0 * kMfmaSteps + s) * 64 + lane];
        b1[s] = qB[((size_t)(qt0 + 1) * kMfmaSteps + s) * 64 + lane];
    }
    const uint32_t qa = qt0 * 32u + (lane & 31), qb = qa + 32u;
    const float na = qa < nq ? qn2[qa] : 0.0f, nb = qb < nq ? qn2[qb] : 0.0f;
    const float two_ea = 2.0f * (kMfmaECoeff * (na + max_dn2) + kMfmaEAbs) * 1.000001f + 1e-30f;
    const float two_eb = 2.0f * (kMfmaECoeff * (nb + max_dn2) + kMfmaEAbs) * 1.000001f + 1e-30f;
    const bool live_a = two_ea < INFINITY && qa < nq, live_b = two_eb < INFINITY && qb < nq;
    // slice id = 2 * blockIdx.y + half: the two half-waves of a query see disjoint rows
    const uint32_t slice = blockIdx.y * 2u + half;
    uint2* __restrict__ ring_a = ring + ((size_t)slice * nq + (qa < nq ? qa : nq - 1)) * kRing;
    uint2* __restrict__ ring_b = ring + ((size_t)slice * nq + (qb < nq ? qb : nq - 1)) * kRing;
    ScanState sa, sb;
    const uint32_t n_tiles = (ndb + 31u) / 32u;
    const uint32_t t0 = blockIdx.y * tiles_per_split, t1 = min(n_tiles, t0 + tiles_per_split);
    if (t0 < t1) {
        h8 cur[kMfmaSteps];
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) cur[s] = dA[((size_t)t0 * kMfmaSteps + s) * 64 + lane];
        for (uint32_t t = t0; t < t1; ++t) {
            h8 nxt[kMfmaSteps];
            const uint32_t tn = t + 1 < t1 ? t + 1 : t;
#pragma unroll
            for (int s = 0; s < kMfmaSteps; ++s) nxt[s] = dA[((size_t)tn * kMfmaSteps + s) * 64 + lane];
            f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
            for (int s = 0; s < kMfmaSteps; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], b0[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], b1[s], acc1, 0, 0, 0);
            }
            const uint32_t row0 = t * 32u + 4u * half;
            mfma_post(acc0, sa, two_ea, live_a, row0, ndb, ring_a);
            mfma_post(acc1, sb, two_eb, live_b, row0, ndb, ring_b);
#pragma unroll
            for (int s = 0; s < kMfmaSteps; ++s) cur[s] = nxt[s];
        }
    }
    if (qa < nq) {
        const size_t o = (size_t)slice * nq + qa;
        ring_count[o] = sa.cnt;
        part_min[o] = two_ea < INFINITY ? sa.best : -INFINITY;
        evict_min[o] = sa.ev;
    }
    if (qb < nq) {
        const size_t o = (size_t)slice * nq + qb;
        ring_count[o] = sb.cnt;
        part_min[o] = two_eb < INFINITY ? sb.best : -INFINITY;
        evict_min[o] = sb.ev;
    }
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
    nn64_verify_k<<<(nq + 255) / 256, 256, 0, s>>>(q, nq, db, DIM, ring, ring_count, part_min, evict_min, splits,
                                                   e_coeff, 0.0f, max_dn2, qn, nn, overflow_list, overflow_count);
    hipError_t e = hipMemcpyAsync(h_overflow, overflow_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (*h_overflow) nn_exact_one_k<<<*h_overflow, 64, 0, s>>>(q, db, ndb, DIM, overflow_list, nn);
    return hipGetLastError();
}

// ---- MFMA path launchers -------------------------------------------------------------------------
uint32_t mfma_tiles(uint32_t n) { return (n + 31u) / 32u; }
// tiles padded so that a wave's two query tiles always exist
uint32_t mfma_query_tiles(uint32_t n) { return ((n + 511u) / 512u) * 16u; }

void launch_max_f32(const float* v, uint32_t n, float* out, hipStream_t s) { max_f32_k<<<1, 256, 0, s>>>(v, n, out); }
void launch_max_abs(const double* f, size_t count, double* partial /* 256 */, hipStream_t s) {
    max_abs_k<<<256, 256, 0, s>>>(f, count, partial);
}
void launch_pack_f16(const double* f, uint32_t n, uint32_t n_tiles, double scale, int role, void* out, float* norm2,
                     hipStream_t s) {
    if (!n_tiles) return;
    pack_f16_k<<<(n_tiles * 32u + 255) / 256, 256, 0, s>>>(f, n, n_tiles, scale, role,
                                                           reinterpret_cast<_Float16*>(out), norm2);
}

// Same workspace as launch_nn_screened33 with 2 * splits slices.  qB: queries packed with role 1
// (mfma_query_tiles(nq) tiles), dA: database packed with role 0 (mfma_tiles(ndb) tiles); qn: scaled |q|^2.
hipError_t launch_nn_mfma33(const double* q, const void* qB, const float* qn, uint32_t nq, const double* db,
                            const void* dA, uint32_t ndb, float max_dn2, uint32_t splits, uint2* ring,
                            uint32_t* ring_count, float* part_min, float* evict_min, uint32_t* overflow_list,
                            uint32_t* overflow_count, uint32_t* nn, uint32_t* h_overflow, hipStream_t s) {
    constexpr int DIM = 33;
    *h_overflow = 0;
    if (!nq || !ndb) return hipSuccess;
    (void)hipMemsetAsync(overflow_count, 0, sizeof(uint32_t), s);
    const uint32_t n_tiles = mfma_tiles(ndb);
    const uint32_t per = (n_tiles + splits - 1) / splits;
    nn16_scan_k<<<dim3((nq + 255) / 256, splits), 256, 0, s>>>(reinterpret_cast<const h8*>(qB), qn, nq,
                                                              reinterpret_cast<const h8*>(dA), ndb, per, max_dn2,
                                                              ring, ring_count, part_min, evict_min);
    nn64_verify_k<<<(nq + 255) / 256, 256, 0, s>>>(q, nq, db, DIM, ring, ring_count, part_min, evict_min, 2 * splits,
                                                   kMfmaECoeff, kMfmaEAbs, max_dn2, qn, nn, overflow_list,
                                                   overflow_count);
    hipError_t e = hipMemcpyAsync(h_overflow, overflow_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (*h_overflow) nn_exact_one_k<<<*h_overflow, 64, 0, s>>>(q, db, ndb, DIM, overflow_list, nn);
    return hipGetLastError();
}

}  // namespace m3d
