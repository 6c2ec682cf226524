// m3d_reg_cache_fp.hpp -- the arithmetic of the validation's candidate cache (m3d_reg_cache.hip), shared by the kernel and the
// host check tests/cpp/test_reg_cache.cpp (which holds it against an exact nearest-neighbour search over ALL target points):
// the squared distance to a candidate in fp32, its rounding bound, what the unlisted points keep from a query, and the three
// certificates.  Derivation of the bound: m3d_reg_cache.hip's header.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define M3D_RC_HD __host__ __device__ __forceinline__
#else
#define M3D_RC_HD inline
#endif

namespace m3d {

constexpr uint32_t kCacheSlotMask = 127u;   // the candidate's slot rides in the seven low mantissa bits of s

M3D_RC_HD uint32_t cache_f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
M3D_RC_HD float cache_u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// s = |c - u|^2 as the kernel forms it (three subtractions, a product, two fused multiply-adds; the kernel does two candidates
// per instruction with v_pk_*: the same IEEE operations per candidate), with the slot packed in
M3D_RC_HD uint32_t cache_packed_s(float cx, float cy, float cz, float ux, float uy, float uz, uint32_t slot) {
    const float dx = cx - ux, dy = cy - uy, dz = cz - uz;
    const float s = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    return (cache_f2u(s) & ~kCacheSlotMask) | slot;
}
// E(s) = 2^-23 R^2 + 2^-15 s >= |s - d^2| for every candidate within R of xa and every query within R of xa
M3D_RC_HD float cache_err(float s, float R) {
#ifdef M3D_CACHE_NO_SLACK
    (void)R;
    return s * 0.0f;
#else
    return __builtin_fmaf(s, 0x1p-15f, R * R * 0x1p-23f);
#endif
}
// what every UNLISTED target point keeps from the query (du = |u| rounded up), rounded down; never negative, 0 for a NaN
M3D_RC_HD float cache_reach(float R, float du) {
#ifdef M3D_CACHE_INFLATE_R
    R *= 1.1f;   // (the mutation the host check must catch)
#endif
    return __builtin_fmaxf(R * (1.0f - 0x1p-20f) - du, 0.0f);
}
struct CacheVerdict {
    bool winner;    // the nearest target point is candidate (m1 & kCacheSlotMask): evaluate it in fp64
    bool nothing;   // no target point within the search radius
    float lb2;      // neither: a lower bound of the nearest target point's squared distance (the query MAY be an inlier unless lb2 >= r2hi)
};
// m1 <= m2: the two smallest packed s over the rings evaluated so far, R: the outermost of those rings' radii,
// r2hi >= r^2 (1 + 2^-20)
M3D_RC_HD CacheVerdict cache_certify(uint32_t m1, uint32_t m2, float R, float du, float r2hi) {
    const float tt = cache_reach(R, du);
    const float f1 = cache_u2f(m1), f2 = cache_u2f(m2);   // (a NaN pose: NaN patterns, every test below fails)
    const float e1 = cache_err(f1, R), e2 = cache_err(f2, R);
    const float t2 = tt * tt;
    const bool covered = f1 + e1 < t2;                       // the nearest target point is in the lists
    const bool unique = f2 - e2 > f1 + e1;                   // ... and it is candidate j1, in fp64 as well
    const bool nothing = t2 >= r2hi && f1 - e1 >= r2hi;      // no target point within the search radius
    CacheVerdict v;
    v.winner = covered && unique;
    v.nothing = !covered && nothing;
    const float l = __builtin_fmaxf(__builtin_fminf(f1 - e1, t2) * (1.0f - 0x1p-20f), 0.0f);
    v.lb2 = l == l ? l : 0.0f;
    return v;
}

}  // namespace m3d
