// m3d_match_scan.hpp -- pieces shared by the two screening translation units of the matcher
// (m3d_match_kernels.hip: fp32 VALU screen, packing, verification; m3d_match_mfma.hip: MFMA screen, built with
// -fno-honor-nans so that the per-tile minimum needs no NaN canonicalisation).
#pragma once
#include "m3d_reg_kernels.hpp"

namespace m3d {

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

void launch_nn16_scan(const void* qB, const float* qn, uint32_t nq, const void* dA, uint32_t ndb,
                      uint32_t tiles_per_split, uint32_t splits, float max_dn2, float* premin /* 2 splits nq */,
                      uint2* ring, uint32_t* ring_count, float* part_min, float* evict_min, hipStream_t s);

}  // namespace m3d
