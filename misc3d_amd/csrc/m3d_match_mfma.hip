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
    printf("scan stats: side-tiles with a ring entry %llu, with a rev entry %llu, rev candidates (lane-rows) %llu, ring appends (lane-rows) %llu\n",
           g_scan_stats[0], g_scan_stats[1], g_scan_stats[3], g_scan_stats[4]);
    for (int i = 0; i < 8; ++i) g_scan_stats[i] = 0;
}
#else
#define SCAN_STAT(i) ((void)0)
#endif
// Per tile and query tile: 16 distances per lane (acc[4g .. 4g + 3] = rows 8g + 4 half + 0..3 of the tile).
// "does any lane ...": the lane mask straight from the compare (__any / __ballot take an int: the flag is first materialised in a
// VGPR and compared again -- two VALU instructions and their latency in front of every wave-uniform branch of the scan)
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// the minima of the lane's four runs of four rows: what the slow paths test first (v_min + v_min3 per run under -fno-honor-nans)
__device__ __forceinline__ void group_min(const f32x16& acc, float (&g)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = fminf(fminf(acc[4 * k], acc[4 * k + 1]), fminf(acc[4 * k + 2], acc[4 * k + 3]));
}

// ---- a side's FAST path: ten VALU instructions, no branch -- so that it can stand BETWEEN the matrix instructions of the
// other side's chain (nn16_scan_k) --, leaving the lane's minimum and two lane masks (SGPR pairs) behind.
// Round 6; before: 26 instructions and two to four branches per side (run minima, the running minimum and its window every tile,
// four run tests of the reverse search).  What made the single tests sharp enough:
//   forward  tmin <= win, the window of the running minimum BEFORE this tile: a new record (tmin < best <= win) passes it like a
//            row inside the window does, so best and win need an update in the slow path only;
//   reverse  tmin <= t16, the largest threshold of the lane's sixteen rows: the rows come ordered by threshold within chunks
//            of 1024 (rev_order_k), so t16 is hardly above the others -- 2.7 M side-tiles of 39 M pass for 2.4 M with a candidate
//            (in the caller's order the four run tests let 11.4 M pass).
// The hot predicates are single compares, their lane masks come straight out of v_cmp: a lane without a live query carries
// two_e = -inf (its window is -inf), a padding query's packed norm is 65504 x 2^15 (pack_f16_both_k: every distance of it is
// beyond any threshold), a padding row of the database likewise (beyond any window; the append still checks row < ndb).
struct FastOut {
    float tmin = INFINITY;
    uint64_t ring = 0, rev = 0;
};
__device__ __forceinline__ void post_fast_a(const f32x16& a, float (&m)[5]) {   // fifteen of the sixteen values: five v_min3
#pragma unroll
    for (int k = 0; k < 5; ++k) m[k] = fminf(fminf(a[3 * k], a[3 * k + 1]), a[3 * k + 2]);
}
template <bool MIN_ONLY, bool REV>
__device__ __forceinline__ void post_fast_b(const f32x16& a, const float (&m)[5], FastOut& f, ScanState& st, float t16) {
    f.tmin = fminf(fminf(fminf(m[0], m[1]), a[15]), fminf(fminf(m[2], m[3]), m[4]));   // v_min3, v_min3, v_min
    if (MIN_ONLY) {
        st.best = fminf(st.best, f.tmin);
        return;
    }
    f.ring = __builtin_amdgcn_ballot_w64(f.tmin <= st.win);
    if (REV) f.rev = __builtin_amdgcn_ballot_w64(f.tmin <= t16);
}
// The reverse search's rows of one run (m3d_match_scan.hpp, RevOut): the four rows' own thresholds, a ballot and scalar branch
// per row, a plain store into the lane's own list per candidate.
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
                const uint32_t row = row0 + (uint32_t)(k + 8 * G);   // (a POSITION in the ordered database: rev_bin_k translates)
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
// ---- a side's SLOW paths behind ONE wave-uniform branch (6 % + 8 % of the side-tiles).  Forward: the running minimum and its
// window INCLUDING this tile (still a superset of the final window), then the runs and rows that some lane needs (one v_cmp +
// scalar branch each: nearly always ONE row of one lane).  Reverse: the four run thresholds, then rev_run.
// t4 / rows: the tile's run and row thresholds of this lane's half, in the LDS.
template <bool REV>
__device__ __forceinline__ void post_slow(const f32x16& acc, const FastOut& f, ScanState& st, float two_e, uint32_t row0, uint32_t ndb,
                                          uint2* __restrict__ rg, const float* __restrict__ t4, const float* __restrict__ rows,
                                          uint2* __restrict__ rl, uint32_t& rc, const RevOut& rev, uint32_t q) {
#ifdef M3D_SCAN_STATS
    if (REV && (threadIdx.x & 63) == 0) {
        if (f.ring) SCAN_STAT(0);
        if (f.rev) SCAN_STAT(1);
    }
#endif
    if ((f.ring | f.rev) == 0ull) return;
    float g[4];
    group_min(acc, g);
    if (f.ring) {
        st.best = fminf(st.best, f.tmin);   // (a lane outside the mask: tmin > win >= best, nothing changes)
        st.win = st.best + two_e;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (any_lane(g[k] <= st.win)) {
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
    if (REV && f.rev) {
        const float4 th = *reinterpret_cast<const float4*>(t4);
        if (any_lane(g[0] <= th.x)) { rev_run<0>(acc, *reinterpret_cast<const float4*>(rows), row0, rl, rc, rev, q); }
        if (any_lane(g[1] <= th.y)) { rev_run<1>(acc, *reinterpret_cast<const float4*>(rows + 8), row0, rl, rc, rev, q); }
        if (any_lane(g[2] <= th.z)) { rev_run<2>(acc, *reinterpret_cast<const float4*>(rows + 16), row0, rl, rc, rev, q); }
        if (any_lane(g[3] <= th.w)) { rev_run<3>(acc, *reinterpret_cast<const float4*>(rows + 24), row0, rl, rc, rev, q); }
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
// (Round 5, measured and refuted: tile u + 1's six MFMAs issued before tile u's post-processing on a second accumulator set --
//  6.87 -> 8.33 ms: a wave issues in order, VALU work BEHIND the MFMAs cannot start before the last of them has issued.  Round 6
//  put the VALU work BETWEEN them instead: chain_under below.)
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
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void nn16_scan_k(const h8* __restrict__ qB, const float* __restrict__ qn2,
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
    // (the window of the seeded minimum; unseeded: +inf, the first tile enters the slow path and sets both)
    if (!MIN_ONLY) {
        sa.win = sa.best + win_ea;
        sb.win = sb.best + win_eb;
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
            // One side's chain of three matrix instructions with the OTHER side's fast path between them: a wave issues in order,
            // so VALU work overlaps its own matrix instructions only where it stands between them in the program
            // (tools/ubench/mfma_chain.hip: a dependent chain issues at the pipe's rate; six v_min after each of its MFMAs cost
            // a tenth of what they cost behind all six).  posted = the other side's accumulators, complete since its own chain.
            // (Measured and not kept: every fragment register loaded again right behind its own MFMA with what the NEXT chain needs in
            //  that place -- three MFMAs of lead for the LDS round trip, no second register set, but every tile read twice: 3.94 ms
            //  against 3.82 on 200 k x 200 k, same box, twice.  The tile's fragments are requested once, behind its second chain.
            //  s_setprio 1 / 3 for the length of a chain: 4.01 / 3.96 against 3.94 ms.)
            auto fragment = [&](uint32_t u, int s) { return stage[buf][(u * kMfmaSteps + s) * 64 + lane]; };
            // (t16n: tile `next`'s largest threshold, requested in front of the chain and folded behind its second MFMA; null: not wanted)
            auto chain_under = [&](h8 (&c)[kMfmaSteps], const h8 (&b)[kMfmaSteps], f32x16& out, const f32x16& posted, FastOut& f,
                                   ScanState& st, float t16, uint32_t next, float* t16n, bool last_use) {
                static_assert(kMfmaSteps == 3, "three matrix instructions per chain");
                float m[5];
                float4 t4n = make_float4(0.f, 0.f, 0.f, 0.f);
                if (REV && t16n) t4n = *reinterpret_cast<const float4*>(&sthr4[buf][next * 8u + 4u * half]);
                __builtin_amdgcn_sched_barrier(0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], b[0], f32x16{0}, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                post_fast_a(posted, m);
                __builtin_amdgcn_sched_barrier(0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[1], b[1], out, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                post_fast_b<MIN_ONLY, REV>(posted, m, f, st, t16);
                if (REV && t16n) *t16n = fmaxf(fmaxf(t4n.x, t4n.y), fmaxf(t4n.z, t4n.w));
                __builtin_amdgcn_sched_barrier(0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[2], b[2], out, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (last_use) {   // (tile `next`'s fragments: the tile's second chain has issued)
#pragma unroll
                    for (int k = 0; k < kMfmaSteps; ++k) c[k] = fragment(next, k);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // the largest threshold of this lane's sixteen rows of tile u (post_fast_b)
            auto threshold16 = [&](uint32_t u) {
                if (!REV) return 0.0f;
                const float4 t4 = *reinterpret_cast<const float4*>(&sthr4[buf][u * 8u + 4u * half]);
                return fmaxf(fmaxf(t4.x, t4.y), fmaxf(t4.z, t4.w));
            };
            auto slow = [&](const f32x16& acc, const FastOut& f, ScanState& st, float two_e, uint32_t u, uint2* __restrict__ rg,
                            uint2* __restrict__ rl, uint32_t& rc, uint32_t q) {
                if (MIN_ONLY) return;
#ifdef M3D_SCAN_ABL_NOSLOW
                if ((f.ring ^ f.rev) != 0x5DEADBEEF1234567ull) return;   // (timing only: wrong results)
#endif
                post_slow<REV>(acc, f, st, two_e, (t + u) * 32u + 4u * half, ndb, rg, &sthr4[buf][u * 8u + 4u * half],
                               &sthr[buf][u * 32u + 4u * half], rl, rc, rev, q);
            };
            f32x16 acc0, acc1;
            h8 cur[kMfmaSteps];
#pragma unroll
            for (int s = 0; s < kMfmaSteps; ++s) cur[s] = fragment(0u, s);
            float t16 = threshold16(0u);
            acc0 = f32x16{0};
#pragma unroll
            for (int s = 0; s < kMfmaSteps; ++s) {   // the stage's first chain: nothing to stand over yet
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], b0[s], acc0, 0, 0, 0);
            }
            uint32_t u = 0;
            FastOut fa, fb;
            for (;;) {
                const uint32_t un = min(u + 1u, (uint32_t)kStageTiles - 1u);   // (past the stage's end: a tile nobody uses)
                // side b's chain of tile u under side a's fast path of tile u; the fragments behind it: tile u + 1's
                float t16n = 0.0f;
                chain_under(cur, b1, acc1, acc0, fa, sa, t16, un, &t16n, true);
                slow(acc0, fa, sa, win_ea, u, ring_a, rl_a, rc_a, qa);
                if (u + 1u >= in_stage) break;   // (workgroup-uniform)
                // side a's chain of tile u + 1 under side b's fast path of tile u
                chain_under(cur, b0, acc0, acc1, fb, sb, t16, un, nullptr, false);
                slow(acc1, fb, sb, win_eb, u, ring_b, rl_b, rc_b, qb);
                t16 = t16n;
                ++u;
            }
            {   // the stage's last tile, side b: nothing left to stand under
                float m[5];
                post_fast_a(acc1, m);
                post_fast_b<MIN_ONLY, REV>(acc1, m, fb, sb, t16);
                slow(acc1, fb, sb, win_eb, u, ring_b, rl_b, rc_b, qb);
            }
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
