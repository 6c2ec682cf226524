// m3d_match_scan.hpp -- pieces shared by the two screening translation units of the matcher
// (m3d_match_kernels.hip: fp32 VALU screen, packing, verification; m3d_match_mfma.hip: MFMA screen, built with
// -fno-honor-nans so that the per-tile minimum needs no NaN canonicalisation).
#pragma once
#include "m3d_reg_kernels.hpp"

namespace m3d {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// M3D_MATCH_HI_ONLY: the screen contracts the HI halves only -- K = 48, three MFMA steps per tile pair
// instead of seven -- and pays with a wider bound: what is left out is 2 (a_hi . b_lo + a_lo . b_hi + a_lo . b_lo) with
// |x_lo| <= 2^-11 |x| per element, i.e. at most 2^-9 |a| |b| <= 2^-10 (|a|^2 + |b|^2) = 9.8e-4, plus the 1e-4 of the full-width
// screen's own rounding = 1.08e-3: kMfmaECoeff 1.2e-3 (11 % above the derived bound; 1.1e-3 left 2 % -- ADVICE r4).  The lo
// halves then only matter to the verification kernels, which are fp64 anyway.
#ifndef M3D_MATCH_HI_ONLY
#define M3D_MATCH_HI_ONLY 1
#endif
constexpr bool kMfmaHiOnly = M3D_MATCH_HI_ONLY != 0;
constexpr int kMfmaK = kMfmaHiOnly ? 48 : 112, kMfmaSteps = kMfmaK / 16;
constexpr int kMfmaNormAt = kMfmaHiOnly ? 33 : 99;   // first K-slot of the norms' six pieces
constexpr float kMfmaECoeff = kMfmaHiOnly ? 1.2e-3f : 1.0e-4f;
// Absolute part of the bound (scaled^2 units; data scaled to max |v| in [1024, 2048)): fp16 underflow.  Values far
// below the largest one lose their lo piece to the subnormal spacing 2^-24 -- or to zero if the matrix core
// flushes subnormal inputs (<= 6.1e-5 per element): 2 * 6.1e-5 * sum(|a_k| + |b_k|) <= 16.5, plus two norm
// remainders <= 6.1e-5 * 2^15 = 2 each.  32 covers it; for well-scaled data it is noise next to the relative
// part (~1e3), for badly scaled data (one huge row) it correctly sends everything to the exact path.
constexpr float kMfmaEAbs = 32.0f;
constexpr float kMfmaC = 32768.0f;   // 2^15

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

// The REVERSE search riding on the same product tiles (ANNMatcher::Match runs both directions,
// src/correspondence_matching.cpp:59-62): the scan that finds every query's nearest database row also sees every
// distance the opposite search needs -- database row j against all queries.  Per database row a threshold
// thr[j] = u_j + 2 E_j, u_j = the screen's minimum over a SAMPLE of the queries (a warm-up pass with the roles swapped);
// every (query, row) pair of the main pass with d16 <= thr[row] is a candidate of row j.  The exact nearest query of
// row j and every exact tie are among them: d16(i*, j) <= d(i*, j) + E <= d(i_s, j) + E <= u_j + 2E for the sample's
// best i_s.  thr = -inf: nothing is collected for that row (padding rows; rows whose bound is not finite).
// A lane of the scan owns one query and keeps its candidates in a list of its own -- kRevLane (row, d16) entries per
// (slice, query), a plain store each: an atomic slot counter per ROW inside the scan cost the wave a memory round trip
// per candidate and doubled the scan's time -- and rev_bin_k sorts the lists by row afterwards (the atomics, all
// independent, in a kernel of their own).  A lane whose list is full (a query near very many rows: a few lists in a
// thousand) appends to the row's slots directly.  A row with more than kRevCap candidates takes the exact fallback
// (with thresholds from a 1/8 sample the count is geometric with mean 8: (7/8)^256 = 1e-15 for unclustered data).
constexpr int kRevCap = kMatchRevCap;   // candidates kept per database row (m3d_reg_kernels.hpp)
constexpr int kRevLane = 8;             // candidates a (slice, query) list holds
struct RevOut {
    const float* thr = nullptr;   // mfma_tiles(ndb) * 32 thresholds (scaled^2 units)
    const float* thr4 = nullptr;  // mfma_tiles(ndb) * 8: the largest threshold of every run of four rows
    uint32_t* cnt = nullptr;      // ndb counters: low 31 bits candidates binned so far, bit 31 = take the exact fallback
    uint2* cand = nullptr;        // ndb x kRevCap slots (query, d16 bits): filled by rev_bin_k, and by full lists directly
    uint2* list = nullptr;        // slices x nq x kRevLane entries (row, d16 bits)
    uint32_t* list_cnt = nullptr; // slices x nq
    const uint32_t* perm = nullptr;   // position in the packed database -> row (what cnt / cand are indexed by); lists and rings carry positions
    uint32_t q_base = 0;          // index of the launch's first query in the whole query matrix (what cand's entries carry)
};

// the main pass over splits split0 .. split0 + splits - 1 of the database (plan; tiles below tile_end), for the nq queries
// whose per-query arrays the pointers address; premin = init_slices x nq warm-up minima (launch_nn16_warm); max_dn2: device cell
void launch_nn16_scan(const void* qB, const float* qn, uint32_t nq, const void* dA, uint32_t ndb, uint32_t tile_end,
                      const SplitPlan& plan, uint32_t split0, uint32_t splits, uint32_t init_slices, const float* max_dn2,
                      const float* premin, uint2* ring, uint32_t* ring_count, float* part_min, float* evict_min,
                      hipStream_t s, const RevOut* rev = nullptr);
// the warm-up pass alone (running minimum over the first `warm_tiles` tiles of the database, per (slice, query)):
// part_min receives 2 * splits x nq values
void launch_nn16_warm(const void* qB, const float* qn, uint32_t nq, const void* dA, uint32_t ndb, uint32_t warm_tiles,
                      uint32_t splits, const float* max_dn2, float* part_min, hipStream_t s);

}  // namespace m3d
