// m3d_match_mfma.hip -- the split-fp16 MFMA screen of the matcher (design: m3d_match_kernels.hip header and
// DESIGN.md "Matcher").  Separate translation unit because it is compiled with -fno-honor-nans: the inputs
// are finite by construction (the host only takes this path for finite, rescaled data), and without that
// promise every fminf of the per-tile minimum costs two extra v_max canonicalisations.
#include "m3d_match_scan.hpp"

#include <algorithm>

#pragma clang fp contract(off)

namespace m3d {

#ifdef M3D_SCAN_STATS   // (diagnostic build: how often the slow paths are entered)
__device__ unsigned long long g_scan_stats[8];
#define SCAN_STAT(i) atomicAdd(&g_scan_stats[i], 1ull)
__global__ void scan_stats_print_k() {
    printf("scan stats: side-tiles with a ring entry %llu, with a rev entry %llu, rev runs entered %llu, rev candidates (lane-rows) %llu, "
           "ring appends (lane-rows) %llu, side-tiles where tmin <= max of the 4 run thresholds %llu\n",
           g_scan_stats[0], g_scan_stats[1], g_scan_stats[2], g_scan_stats[3], g_scan_stats[4], g_scan_stats[5]);
    for (int i = 0; i < 8; ++i) g_scan_stats[i] = 0;
}
#else
#define SCAN_STAT(i) ((void)0)
#endif
// Per tile and query tile: 16 distances per lane.  The running minimum is updated unconditionally; rows are
// appended when they lie within the window of the minimum INCLUDING this tile (still a superset of the final
// window).  A wave carries 128 rings, so early in a scan some lane has a new record in most tiles: the append path is
// entered per wave-uniform branch and, inside, only the rows that any lane needs are touched (one v_cmp + scalar
// branch per row), instead of 16 divergent blocks.
// group_min: the minima of the lane's four runs of four rows (acc[4g .. 4g + 3] = rows 8g + 4 half + 0..3 of the tile):
// the first two levels of the 16-value minimum, kept because the reverse search tests them against the runs'
// thresholds (two instructions per run: v_min + v_min3 under -fno-honor-nans).
__device__ __forceinline__ void group_min(const f32x16& acc, float (&g)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = fminf(fminf(acc[4 * k], acc[4 * k + 1]), fminf(acc[4 * k + 2], acc[4 * k + 3]));
}
// "does any lane ...": the lane mask straight from the compare (__any / __ballot take an int: the flag is first materialised in a
// VGPR and compared again -- two VALU instructions and their latency in front of every wave-uniform branch of the scan)
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// The hot predicates are single compares, so that their lane masks come straight out of v_cmp (a compound condition is first
// materialised per lane and compared again): a lane without a live query carries two_e = -inf -- its window is -inf, nothing is
// inside --, a padding query's packed norm is 65504 x 2^15 (pack_f16_both_k: every distance of it is beyond any threshold), a
// padding row of the database likewise (beyond any window; the append still checks row < ndb).
template <bool MIN_ONLY>
__device__ __forceinline__ void mfma_post(const f32x16& acc, const float (&g)[4], ScanState& st, float two_e, uint32_t row0,
                                          uint32_t ndb, uint2* __restrict__ my) {
    const float tmin = fminf(fminf(g[0], g[1]), fminf(g[2], g[3]));
    st.best = fminf(st.best, tmin);
    if (MIN_ONLY) return;
    st.win = st.best + two_e;
    // One tile pair in five gets here (some lane of the 64 has a row inside its window: a new record of its running minimum, mostly),
    // nearly always for ONE row of one lane: first the four runs' minima, then the four rows of a run that some lane needs
    // (16 row tests per entry before: 150 instructions, a third of the scan's VALU and most of its SALU work)
    if (any_lane(tmin <= st.win)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (any_lane(g[k] <= st.win)) {
#pragma unroll
                for (int r = 4 * k; r < 4 * k + 4; ++r) {
                    const bool hit = acc[r] <= st.win;
                    if (any_lane(hit)) {
                        const uint32_t row = row0 + (uint32_t)((r & 3) + 8 * (r >> 2));
                        if (hit && row < ndb) {
                            const uint32_t slot = st.cnt % kRing;
                            if (st.cnt >= (uint32_t)kRing) st.ev = fminf(st.ev, __uint_as_float(my[slot].y));
                            my[slot] = make_uint2(row, __float_as_uint(acc[r]));
                            st.cnt++;
                        }
                    }
                }
            }
        }
    }
}

// The reverse search's share of a tile (m3d_match_scan.hpp, RevOut).  Fast path: the four run minima against the four
// run thresholds (thr4 = the largest of the run's four row thresholds: conservative) -- 4 compares per side and tile.
// Behind the wave-uniform branch, per run that some lane hit: the four rows' own thresholds (one ds_read_b128), a
// ballot and scalar branch per row, a plain store into the lane's own list per candidate.
template <int G>
__device__ __forceinline__ void rev_run(const f32x16& acc, const float4& th, uint32_t row0, uint2* __restrict__ my,
                                        uint32_t& cnt, const RevOut& rev, uint32_t q) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = k & 2 ? (k & 1 ? th.w : th.z) : (k & 1 ? th.y : th.x);
        const float v = acc[4 * G + k];
        const bool hit = v <= t;   // (a padding query's distances are beyond every threshold, a padding row's threshold is -inf)
        if (any_lane(hit)) {
            if (hit) {
                SCAN_STAT(3);
                const uint32_t row = row0 + (uint32_t)(k + 8 * G);
                if (cnt < (uint32_t)kRevLane) {
                    my[cnt++] = make_uint2(row, __float_as_uint(v));
                } else {   // list full: straight into the row's slots (what rev_bin_k does with the listed ones)
                    const uint32_t orig = rev.perm[row];
                    const uint32_t slot = atomicAdd(rev.cnt + orig, 1u) & 0x7FFFFFFFu;
                    if (slot < (uint32_t)kRevCap) rev.cand[(size_t)orig * kRevCap + slot] = make_uint2(rev.q_base + q, __float_as_uint(v));
                }
            }
        }
    }
}

// ---- the interleaved loop's pieces (M3D_SCAN_INTERLEAVE) ------------------------------------------------------------------
// What a side's fast path leaves behind: the run minima and five lane masks (SGPR pairs).  Everything branch-free, so that
// it can stand BETWEEN the matrix instructions of the other side's chain.
struct FastOut {
    float g[4];
    uint64_t ring = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
};
template <bool MIN_ONLY, bool REV>
__device__ __forceinline__ void post_fast_b(FastOut& f, ScanState& st, float two_e, const float4& t4) {
    const float tmin = fminf(fminf(f.g[0], f.g[1]), fminf(f.g[2], f.g[3]));
    st.best = fminf(st.best, tmin);
    if (!MIN_ONLY) {
        st.win = st.best + two_e;
        f.ring = __builtin_amdgcn_ballot_w64(tmin <= st.win);
    }
    if (REV) {
#if M3D_SCAN_REV_TILE_TEST
        // ONE test per side: the rows come ordered by threshold (rev_order_k), the largest of a lane's sixteen is hardly above the
        // others -- 2.8 M side-tiles of 39 M pass it for 2.4 M with a candidate (four run tests: 2.5 M)
        f.r0 = __builtin_amdgcn_ballot_w64(tmin <= t4.x);   // (t4.x: the caller's maximum of the four run thresholds)
#else
        f.r0 = __builtin_amdgcn_ballot_w64(f.g[0] <= t4.x);
        f.r1 = __builtin_amdgcn_ballot_w64(f.g[1] <= t4.y);
        f.r2 = __builtin_amdgcn_ballot_w64(f.g[2] <= t4.z);
        f.r3 = __builtin_amdgcn_ballot_w64(f.g[3] <= t4.w);
#endif
#ifdef M3D_SCAN_STATS
        if (any_lane(tmin <= fmaxf(fmaxf(t4.x, t4.y), fmaxf(t4.z, t4.w))) && (threadIdx.x & 63) == 0) SCAN_STAT(5);
#endif
    }
}
// the slow paths of a side: mfma_post's append and the reverse search's rows, behind ONE wave-uniform branch
template <bool REV>
__device__ __forceinline__ void post_slow(const f32x16& acc, const FastOut& f, ScanState& st, uint32_t row0, uint32_t ndb,
                                          uint2* __restrict__ rg, const float* __restrict__ rows, uint2* __restrict__ rl,
                                          uint32_t& rc, const RevOut& rev, uint32_t q) {
#ifdef M3D_SCAN_ABL_NOSLOW
    return;   // (timing only: wrong results)
#endif
#ifdef M3D_SCAN_STATS
    if (REV && (threadIdx.x & 63) == 0) {
        if (f.ring) SCAN_STAT(0);
        if (f.r0 | f.r1 | f.r2 | f.r3) SCAN_STAT(1);
        for (int k = 0; k < 4; ++k)
            if (k == 0 ? f.r0 : k == 1 ? f.r1 : k == 2 ? f.r2 : f.r3) SCAN_STAT(2);
    }
#endif
    if ((f.ring | f.r0 | f.r1 | f.r2 | f.r3) == 0ull) return;
#ifdef M3D_SCAN_ABL_NORING
    if (false) {
#else
    if (f.ring) {
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (any_lane(f.g[k] <= st.win)) {
#pragma unroll
                for (int r = 4 * k; r < 4 * k + 4; ++r) {
                    const bool hit = acc[r] <= st.win;
                    if (any_lane(hit)) {
                        const uint32_t row = row0 + (uint32_t)((r & 3) + 8 * (r >> 2));
                        if (hit && row < ndb) {
                            SCAN_STAT(4);
                            const uint32_t slot = st.cnt % kRing;
                            if (st.cnt >= (uint32_t)kRing) st.ev = fminf(st.ev, __uint_as_float(rg[slot].y));
                            rg[slot] = make_uint2(row, __float_as_uint(acc[r]));
                            st.cnt++;
                        }
                    }
                }
            }
        }
    }
#ifdef M3D_SCAN_ABL_NOREV
    if (false) {
#else
    if (REV) {
#endif
        if (f.r0) rev_run<0>(acc, *reinterpret_cast<const float4*>(rows), row0, rl, rc, rev, q);
        if (f.r1) rev_run<1>(acc, *reinterpret_cast<const float4*>(rows + 8), row0, rl, rc, rev, q);
        if (f.r2) rev_run<2>(acc, *reinterpret_cast<const float4*>(rows + 16), row0, rl, rc, rev, q);
        if (f.r3) rev_run<3>(acc, *reinterpret_cast<const float4*>(rows + 24), row0, rl, rc, rev, q);
    }
}

// One wave: 64 queries (two 32-column B tiles held in registers for the whole scan) against a slice of
// the database.  The four waves of a block walk the SAME slice, so the A tiles are staged through LDS once
// per block (double-buffered, kStageTiles tiles per stage, one barrier per stage) instead of being
// fetched from L2 by every wave.
// tiles per stage = database tiles between two barriers of the workgroup: 2: nn16_scan_k<false, true> 7.50 ms on 200 k x 200 k,
// 3: 7.19, 4: 6.86, 6: 7.96 (five staging registers per thread)
#ifndef M3D_MATCH_STAGE_TILES
#define M3D_MATCH_STAGE_TILES 8
#endif
constexpr int kStageTiles = M3D_MATCH_STAGE_TILES;
// M3D_MATCH_SW_PIPELINE=1 (round 5, measured and refuted: 6.87 -> 8.33-8.40 ms on 200 k x 200 k, with and without scheduling
// barriers, unrolled by two and by four): tile u + 1's MFMAs issued before tile u's post-processing, on a second accumulator
// set.  A wave issues in order: its VALU work cannot start before the last of the six MFMAs -- two dependent chains of three --
// has ISSUED, so nothing overlaps inside the wave and the second accumulator set only lengthens live ranges (146 -> 156 VGPRs).
// (Also measured: a start offset per workgroup against lockstep phases of a SIMD's three waves: 6.865 ms, nothing.)  Off; not compiled.
#ifndef M3D_MATCH_SW_PIPELINE
#define M3D_MATCH_SW_PIPELINE 0
#endif
#ifndef M3D_SCAN_INTERLEAVE
#define M3D_SCAN_INTERLEAVE 1
#endif
#ifndef M3D_SCAN_REV_TILE_TEST
#define M3D_SCAN_REV_TILE_TEST 0
#endif
// queries per workgroup = 64 x waves: every wave of a workgroup reads the same staged tiles, so the staging traffic per query goes
// with 1 / waves (four waves: 14.7 GB from L2 per 200 k x 200 k scan, a fifth of the scan's time)
#ifndef M3D_MATCH_BLOCK_WAVES
#define M3D_MATCH_BLOCK_WAVES 8
#endif
constexpr int kBlockWaves = M3D_MATCH_BLOCK_WAVES, kBlockThreads = 64 * kBlockWaves;
constexpr int kStageEntries = kStageTiles * kMfmaSteps * 64;   // h8 entries per stage (12 KB)
static_assert(kStageEntries % kBlockThreads == 0, "every thread copies the same number of entries per stage");
// Staging (round 5): the tiles go from global memory STRAIGHT into the LDS (global_load_lds_dwordx4: lane l's 16 bytes land at
// M0 + 16 l -- tools/ubench/lds_dma_check.hip), three buffers deep: the copy of stage s + 2 is issued when stage s begins.  Through
// registers and two buffers the copy had ONE stage (~0.7 us) to arrive -- less than a trip to L2 and back under load: 0.6 ms of the
// 3.2 ms the MFMAs + minima alone take -- and held 12 VGPRs.  Issued as inline assembly: the compiler orders every LDS read behind
// ALL outstanding copies it knows of (it cannot tell the buffers apart), which would make each stage wait for the copies just issued;
// the waits are written by hand (a wave's copies complete in issue order, so "at most the next stage's copies outstanding" means this
// stage's have landed; other memory operations in flight only make the wait longer, never shorter).
constexpr int kStageBufs = 3;
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// (M0 carries the LDS base of the copy; the statement puts back what it found there, so M0 is no clobber)
__device__ __forceinline__ void lds_copy16(const void* g, uint32_t lds_wave_base) {   // lane l: 16 bytes from g -> LDS lds_wave_base + 16 l
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_wave_base)
                 : "memory");
}
__device__ __forceinline__ void lds_copy4(const void* g, uint32_t lds_wave_base) {    // lane l: 4 bytes -> LDS lds_wave_base + 4 l
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_wave_base)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_copies_but() {   // until at most N of this wave's memory operations are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MIN_ONLY = true: warm-up pass over the first tiles of the database, running minimum only (no rings);
// its per-(slice, query) minima seed the main pass (init_min / init_slices), so that a ring starts with a
// bound close to its final minimum and "new record" events -- which cost a wave-wide detour each, and a wave
// carries 128 rings -- become rare instead of happening in most tiles.
template <bool MIN_ONLY, bool REV>
__global__ __launch_bounds__(kBlockThreads) void nn16_scan_k(const h8* __restrict__ qB, const float* __restrict__ qn2,
                                                    uint32_t nq, const h8* __restrict__ dA, uint32_t ndb,
                                                    uint32_t tile_end, SplitPlan plan, uint32_t split0,
                                                    const float* __restrict__ max_dn2_p,
                                                    const float* __restrict__ init_min, uint32_t init_slices,
                                                    uint2* __restrict__ ring, uint32_t* __restrict__ ring_count,
                                                    float* __restrict__ part_min, float* __restrict__ evict_min,
                                                    RevOut rev) {
    // (nq queries starting at rev.q_base of the whole query matrix: every per-query pointer is the slice's own; splits
    // split0 .. split0 + gridDim.y - 1 of the database: a launch may cover a part of either side, MatchWork in m3d_reg_kernels.hpp)
    const float max_dn2 = *max_dn2_p;   // the largest |row|^2 of the database rows packed so far (>= this launch's rows')
    __shared__ h8 stage[kStageBufs][kStageEntries];
    __shared__ __attribute__((aligned(16))) float sthr[kStageBufs][kStageTiles * 32];   // REV: the staged tiles' row thresholds
    __shared__ __attribute__((aligned(16))) float sthr4[kStageBufs][kStageTiles * 8];   // ... and run thresholds, [tile][half][run]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t qt0 = (blockIdx.x * (uint32_t)kBlockWaves + wave) * 2u;   // first of this wave's two query tiles
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
    const float win_ea = live_a ? two_ea : -INFINITY, win_eb = live_b ? two_eb : -INFINITY;   // (mfma_post: no window for a dead lane)
    // slice id = 2 * blockIdx.y + half: the two half-waves of a query see disjoint rows
    const uint32_t slice = (split0 + blockIdx.y) * 2u + half;
    uint2* __restrict__ ring_a = ring + ((size_t)slice * nq + (qa < nq ? qa : nq - 1)) * kRing;
    uint2* __restrict__ ring_b = ring + ((size_t)slice * nq + (qb < nq ? qb : nq - 1)) * kRing;
    // REV: this lane's candidate lists of the reverse search (one per query and slice, like the rings)
    uint2* __restrict__ rl_a = REV ? rev.list + ((size_t)slice * nq + (qa < nq ? qa : nq - 1)) * kRevLane : nullptr;
    uint2* __restrict__ rl_b = REV ? rev.list + ((size_t)slice * nq + (qb < nq ? qb : nq - 1)) * kRevLane : nullptr;
    uint32_t rc_a = 0, rc_b = 0;
    ScanState sa, sb;
    if (!MIN_ONLY && init_min) {
        for (uint32_t k = 0; k < init_slices; ++k) {
            if (qa < nq) sa.best = fminf(sa.best, init_min[(size_t)k * nq + qa]);
            if (qb < nq) sb.best = fminf(sb.best, init_min[(size_t)k * nq + qb]);
        }
    }
    const uint32_t t0 = plan.begin(split0 + blockIdx.y), t1 = min(tile_end, plan.end(split0 + blockIdx.y));
    if (t0 < t1) {   // block-uniform
        // entry e of a stage = fragment (tile, step, lane) in packed order: consecutive in memory, 64 entries of a wave = 1 KB of the LDS
        constexpr int kPerThread = kStageEntries / kBlockThreads;
        constexpr int kCopies = kPerThread + (REV ? 2 : 0);   // copy instructions a wave issues per stage (the same for every wave)
        const uint32_t last_entry = (t1 - 1) * (uint32_t)(kMfmaSteps * 64) + (kMfmaSteps * 64 - 1);
        const uint32_t wave_first = (uint32_t)(tid & ~63);
        auto fetch = [&](uint32_t t_first, int buf) {
#pragma unroll
            for (int k = 0; k < kPerThread; ++k) {
                const uint32_t e = (uint32_t)tid + (uint32_t)kBlockThreads * k;
                const uint32_t g = min(t_first * (uint32_t)(kMfmaSteps * 64) + e, last_entry);   // clamp: stay inside the slice
                lds_copy16(dA + g, __builtin_amdgcn_readfirstlane(lds_offset(&stage[buf][wave_first + (uint32_t)kBlockThreads * k])));
            }
            if (REV) {
                // the stage's row thresholds (kStageTiles * 32 of them: the later waves copy what the first ones copy -- every wave
                // issues the same number of copies, which is what its wait counts) and run thresholds (the first half-wave of each
                // wave; slot [tile u][half][run g] <- run 2 g + half of tile u)
                const uint32_t i = (uint32_t)tid & (kStageTiles * 32u - 1u);
                lds_copy4(rev.thr + min(t_first * 32u + i, t1 * 32u - 1u), __builtin_amdgcn_readfirstlane(lds_offset(&sthr[buf][i & ~63u])));
                if (lane < kStageTiles * 8) {
                    const uint32_t k = (uint32_t)lane, u = k >> 3, hf = (k >> 2) & 1u, g = k & 3u;
                    lds_copy4(rev.thr4 + min((t_first + u) * 8u + 2u * g + hf, t1 * 8u - 1u), lds_offset(&sthr4[buf][0]));
                }
            }
        };
        static_assert((kStageTiles * 32) % 64 == 0 && ((kStageTiles * 32) & (kStageTiles * 32 - 1)) == 0 && kStageTiles * 32 <= kBlockThreads &&
                          kStageTiles * 8 <= 64,
                      "the threshold copies: whole waves of row thresholds, one (half-)wave of run thresholds");
        fetch(t0, 0);
        if (t0 + kStageTiles < t1) fetch(t0 + kStageTiles, 1);
        int buf = 0;
        for (uint32_t t = t0; t < t1; t += kStageTiles) {
            // this stage's copies have landed when at most the next stage's are outstanding; the barrier publishes them -- and says that
            // every wave is done with the stage before this one, whose buffer the copy of the stage after next may now overwrite
            if (t + kStageTiles < t1) wait_copies_but<kCopies>();
            else wait_copies_but<0>();
            __syncthreads();
            if (t + 2 * kStageTiles < t1) fetch(t + 2 * kStageTiles, buf >= 1 ? buf - 1 : kStageBufs - 1);
            const uint32_t in_stage = min((uint32_t)kStageTiles, t1 - t);
            // one side after the other: the run minima of a side are dead before the other side's are formed
            auto side = [&](const f32x16& acc, ScanState& st, float two_e, uint2* __restrict__ rg, uint32_t q,
                            uint2* __restrict__ rl, uint32_t& rc, uint32_t u, const float4& t4) {
                const uint32_t row0 = (t + u) * 32u + 4u * half;
                float g4[4];
                group_min(acc, g4);
                mfma_post<MIN_ONLY>(acc, g4, st, two_e, row0, ndb, rg);
                if (REV) {
                    // (one lane mask per run, straight from its compare; their union decides the branch on the scalar unit)
                    const uint64_t m0 = __builtin_amdgcn_ballot_w64(g4[0] <= t4.x), m1 = __builtin_amdgcn_ballot_w64(g4[1] <= t4.y),
                                   m2 = __builtin_amdgcn_ballot_w64(g4[2] <= t4.z), m3 = __builtin_amdgcn_ballot_w64(g4[3] <= t4.w);
                    if ((m0 | m1 | m2 | m3) != 0ull) {
                        const float* rows = &sthr[buf][u * 32u + 4u * half];
                        if (m0) rev_run<0>(acc, *reinterpret_cast<const float4*>(rows), row0, rl, rc, rev, q);
                        if (m1) rev_run<1>(acc, *reinterpret_cast<const float4*>(rows + 8), row0, rl, rc, rev, q);
                        if (m2) rev_run<2>(acc, *reinterpret_cast<const float4*>(rows + 16), row0, rl, rc, rev, q);
                        if (m3) rev_run<3>(acc, *reinterpret_cast<const float4*>(rows + 24), row0, rl, rc, rev, q);
                    }
                }
            };
            // the six MFMAs of database tile u of the stage (both query tiles of the wave)
            auto multiply = [&](uint32_t u, f32x16& a0, f32x16& a1) {
                h8 cur[kMfmaSteps];
#pragma unroll
                for (int s = 0; s < kMfmaSteps; ++s) cur[s] = stage[buf][(u * kMfmaSteps + s) * 64 + lane];
                a0 = f32x16{0};
                a1 = f32x16{0};
#pragma unroll
                for (int s = 0; s < kMfmaSteps; ++s) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], b0[s], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], b1[s], a1, 0, 0, 0);
                }
            };
            // the tile's run thresholds: the same for both sides
            auto thresholds = [&](uint32_t u) {
                float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (REV) {
                    t4 = *reinterpret_cast<const float4*>(&sthr4[buf][u * 8u + 4u * half]);
                }
                return t4;
            };
#if M3D_MATCH_SW_PIPELINE
            // (the refuted variant: see M3D_MATCH_SW_PIPELINE above)
            f32x16 accA0, accA1, accB0, accB1;
            multiply(0u, accA0, accA1);
            for (uint32_t u = 0; u < in_stage; u += 2u) {   // two tiles per trip: the sets alternate without moves, the code stays two copies
                const bool odd = u + 1u < in_stage;   // (workgroup-uniform)
                if (odd) {
                    multiply(u + 1u, accB0, accB1);
                    // (no scheduling barrier)
                }
                side(accA0, sa, win_ea, ring_a, qa, rl_a, rc_a, u, thresholds(u));
                side(accA1, sb, win_eb, ring_b, qb, rl_b, rc_b, u, thresholds(u));
                if (odd) {
                    if (u + 2u < in_stage) {
                        multiply(u + 2u, accA0, accA1);
                        // (no scheduling barrier)
                    }
                    side(accB0, sa, win_ea, ring_a, qa, rl_a, rc_a, u + 1u, thresholds(u + 1u));
                    side(accB1, sb, win_eb, ring_b, qb, rl_b, rc_b, u + 1u, thresholds(u + 1u));
                }
            }
#elif M3D_SCAN_INTERLEAVE
            // One side's chain of three matrix instructions with the OTHER side's fast path between them: a wave issues in order,
            // so VALU work overlaps its own matrix instructions only where it stands between them in the program
            // (tools/ubench/mfma_chain.hip: a dependent chain issues at the pipe's rate; six v_min after each of its MFMAs cost
            // a tenth of what they cost behind the six).  posted = the other side's accumulators, complete since its own chain.
            auto chain_under = [&](const h8 (&c)[kMfmaSteps], const h8 (&b)[kMfmaSteps], f32x16& out, const f32x16& posted,
                                   FastOut& f, ScanState& st, float two_e, const float4& t4) {
                static_assert(kMfmaSteps == 3, "three matrix instructions per chain");
                __builtin_amdgcn_sched_barrier(0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], b[0], f32x16{0}, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                group_min(posted, f.g);
                __builtin_amdgcn_sched_barrier(0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[1], b[1], out, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                post_fast_b<MIN_ONLY, REV>(f, st, two_e, t4);
                __builtin_amdgcn_sched_barrier(0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[2], b[2], out, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto fragments = [&](uint32_t u, h8 (&c)[kMfmaSteps]) {
#pragma unroll
                for (int s = 0; s < kMfmaSteps; ++s) c[s] = stage[buf][(u * kMfmaSteps + s) * 64 + lane];
            };
            {
                f32x16 acc0, acc1;
                h8 cur[kMfmaSteps];
                fragments(0u, cur);
                float4 t4 = thresholds(0u);
                acc0 = f32x16{0};
#pragma unroll
                for (int s = 0; s < kMfmaSteps; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], b0[s], acc0, 0, 0, 0);
                uint32_t u = 0;
                FastOut fa, fb;
                for (;;) {
                    // side b's chain of tile u under side a's fast path of tile u
                    chain_under(cur, b1, acc1, acc0, fa, sa, win_ea, t4);
                    // tile u + 1's fragments and thresholds: requested now, they arrive under side a's slow path and side b's ...
                    const uint32_t un = min(u + 1u, (uint32_t)kStageTiles - 1u);   // (past the stage's end: a tile nobody uses)
                    fragments(un, cur);   // (tile u's are dead: side b's chain has issued)
                    const float4 t4n = thresholds(un);
                    post_slow<REV>(acc0, fa, sa, (t + u) * 32u + 4u * half, ndb, ring_a, &sthr[buf][u * 32u + 4u * half], rl_a, rc_a, rev, qa);
                    if (u + 1u >= in_stage) break;   // (workgroup-uniform)
                    // side a's chain of tile u + 1 under side b's fast path of tile u
                    chain_under(cur, b0, acc0, acc1, fb, sb, win_eb, t4);
                    post_slow<REV>(acc1, fb, sb, (t + u) * 32u + 4u * half, ndb, ring_b, &sthr[buf][u * 32u + 4u * half], rl_b, rc_b, rev, qb);
                    t4 = t4n;
                    ++u;
                }
                // the stage's last tile, side b: nothing left to stand under
                group_min(acc1, fb.g);
                post_fast_b<MIN_ONLY, REV>(fb, sb, win_eb, t4);
                post_slow<REV>(acc1, fb, sb, (t + u) * 32u + 4u * half, ndb, ring_b, &sthr[buf][u * 32u + 4u * half], rl_b, rc_b, rev, qb);
            }
#else
            for (uint32_t u = 0; u < in_stage; ++u) {
                f32x16 acc0, acc1;
                const float4 t4 = thresholds(u);
                multiply(u, acc0, acc1);
                side(acc0, sa, win_ea, ring_a, qa, rl_a, rc_a, u, t4);
                side(acc1, sb, win_eb, ring_b, qb, rl_b, rc_b, u, t4);
            }
#endif
            buf = buf + 1 == kStageBufs ? 0 : buf + 1;
        }
    }
    if (MIN_ONLY) {
        if (qa < nq) part_min[(size_t)slice * nq + qa] = sa.best;
        if (qb < nq) part_min[(size_t)slice * nq + qb] = sb.best;
        return;
    }
    if (REV) {
        if (qa < nq) rev.list_cnt[(size_t)slice * nq + qa] = rc_a;
        if (qb < nq) rev.list_cnt[(size_t)slice * nq + qb] = rc_b;
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

void launch_nn16_warm(const void* qB, const float* qn, uint32_t nq, const void* dA, uint32_t ndb, uint32_t warm_tiles,
                      uint32_t splits, const float* max_dn2, float* part_min, hipStream_t s) {
    const dim3 grid((nq + kBlockThreads - 1) / kBlockThreads, splits);
    SplitPlan even;
    even.per = (warm_tiles + splits - 1) / splits;
    even.full = splits;
    nn16_scan_k<true, false><<<grid, kBlockThreads, 0, s>>>(reinterpret_cast<const h8*>(qB), qn, nq, reinterpret_cast<const h8*>(dA), ndb,
                                                  warm_tiles, even, 0u, max_dn2, nullptr, 0, nullptr, nullptr, part_min,
                                                  nullptr, RevOut());
}

void launch_nn16_scan(const void* qB, const float* qn, uint32_t nq, const void* dA, uint32_t ndb, uint32_t tile_end,
                      const SplitPlan& plan, uint32_t split0, uint32_t splits, uint32_t init_slices, const float* max_dn2,
                      const float* premin, uint2* ring, uint32_t* ring_count, float* part_min, float* evict_min,
                      hipStream_t s, const RevOut* rev) {
    const h8* q8 = reinterpret_cast<const h8*>(qB);
    const h8* d8 = reinterpret_cast<const h8*>(dA);
    const dim3 grid((nq + kBlockThreads - 1) / kBlockThreads, splits);
    if (rev) {
        nn16_scan_k<false, true><<<grid, kBlockThreads, 0, s>>>(q8, qn, nq, d8, ndb, tile_end, plan, split0, max_dn2, premin,
                                                      init_slices, ring, ring_count, part_min, evict_min, *rev);
#ifdef M3D_SCAN_STATS
        scan_stats_print_k<<<1, 1, 0, s>>>();
#endif
    }
    else
        nn16_scan_k<false, false><<<grid, kBlockThreads, 0, s>>>(q8, qn, nq, d8, ndb, tile_end, plan, split0, max_dn2, premin,
                                                       init_slices, ring, ring_count, part_min, evict_min, RevOut());
}

}  // namespace m3d
