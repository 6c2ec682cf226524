// m3d_match.cpp -- host driver of the mutual nearest-neighbour matcher (round 6: split out of m3d_registration.cpp).
//
//   m3d_match_mutual_nn      registration::ANNMatcher::Match          src/correspondence_matching.cpp:52-84
//
// The search is exact for both of the reference's methods (FLANN's exact kd-tree, Annoy's approximate forest: the exact answer
// is what is matched): an fp16 MFMA screen over the hi halves of the descriptors (one scan serves both directions), exact fp64
// verification of the screen's candidates, brute-force fall-backs spread over the chip; uploads in slices under the scan when
// the call is alone on its device (MatchWork, m3d_reg_kernels.hpp; DESIGN.md section 4 "The matcher in round 5").
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/misc3d_amd_bench.h"
#include "m3d_config.hpp"
#include "m3d_driver.hpp"
#include "m3d_reg_kernels.hpp"

namespace m3d {
int stream_wait_spin(DeviceCtx* ctx);   // (m3d_fit.cpp: the end of the stream's work, polled in page-locked memory)
}

#pragma clang fp contract(off)

using namespace m3d;

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(M3D_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define RESERVE(buf, bytes)                               \
    do {                                                  \
        if (!(buf).reserve(bytes)) return M3D_ERR_DEVICE; \
    } while (0)

namespace {

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

thread_local unsigned g_match_path = 0;         // m3d_bench_match_last_path's bits
thread_local uint64_t g_match_fallbacks = 0;   // queries of the CALLING THREAD's last m3d_match_mutual_nn that took the exact fallback
                                               // (thread-local: concurrent calls on different devices used to race on one global)

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

uint64_t m3d_match_last_fallbacks(void) { return g_match_fallbacks; }
unsigned m3d_bench_match_last_path(void) { return g_match_path; }

int m3d_match_mutual_nn(const double* feat_src, size_t n_src, const double* feat_dst, size_t n_dst, int dim,
                        int method, int n_trees, int device, size_t* out_src, size_t* out_dst, size_t* k_out) {
    (void)method;   // FLANN (exact kd-tree) and ANNOY (approximate forest) both map to the exact search
    (void)n_trees;
    if (!k_out || dim <= 0 || dim > 1024 || ((!feat_src || !out_src || !out_dst) && n_src) || (!feat_dst && n_dst))
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    *k_out = 0;
    if (n_src == 0 || n_dst == 0) return M3D_OK;
    if (n_src >= ((size_t)1 << 31) || n_dst >= ((size_t)1 << 31)) return fail(M3D_ERR_INVALID_ARG, "too many points");
    LaneLock lane(device);
    if (!lane.ctx) return M3D_ERR_DEVICE;
    return match_mutual_nn_on(lane.ctx, MatchSide{feat_src, nullptr, -1.0}, n_src, MatchSide{feat_dst, nullptr, -1.0}, n_dst, dim,
                              out_src, out_dst, k_out);
}
}  // extern "C"

// -DM3D_MATCH_TIMELINE: the host's clock at the stations of one m3d_match_mutual_nn call, on stderr (tools/match_timeline.sh)
#ifdef M3D_MATCH_TIMELINE
namespace {
thread_local double tl_t[32];
thread_local const char* tl_name[32];
thread_local int tl_n = 0;
}  // namespace
#define MATCH_MARK(name) do { if (tl_n < 32) { tl_t[tl_n] = now_ms(); tl_name[tl_n++] = name; } } while (0)
#define MATCH_TIMELINE_BEGIN() do { tl_n = 0; MATCH_MARK("entry"); } while (0)
#define MATCH_TIMELINE_END()                                                                                                       \
    do {                                                                                                                         \
        MATCH_MARK("blocks returned");                                                                                           \
        for (int i_ = 1; i_ < tl_n; ++i_)                                                                                        \
            fprintf(stderr, "match timeline  %-40s +%.3f ms  (%.3f)\n", tl_name[i_], tl_t[i_] - tl_t[i_ - 1], tl_t[i_] - tl_t[0]); \
    } while (0)
#else
#define MATCH_MARK(name) do { } while (0)
#define MATCH_TIMELINE_BEGIN() do { } while (0)
#define MATCH_TIMELINE_END() do { } while (0)
#endif

// ---- the matcher's MFMA screen, step by step (MatchWork, m3d_reg_kernels.hpp) ------------------------------------------------------
namespace {
// enough (query block x database split) workgroups to fill the chip
uint32_t splits_for(uint32_t nq, uint32_t ndb, uint32_t q_per_block, uint32_t rows_per_unit, uint32_t want_blocks = 2048) {
    const uint32_t blocks = (nq + q_per_block - 1) / q_per_block;
    const uint32_t s = std::max<uint32_t>(1, (want_blocks + blocks - 1) / blocks);
    return std::min<uint32_t>(s, std::max<uint32_t>(1, ndb / rows_per_unit));
}
struct Cut {
    uint32_t row0, rows;
};
constexpr uint32_t kSliceMinRows = 65536;   // matrices below this are uploaded whole (the slices' launches would not fill the chip)

// The two nearest-neighbour searches of ANNMatcher::Match (src/correspondence_matching.cpp:52-62) on the split-fp16 screen.
// a / b: the matrices -- the caller's arrays (uploaded here, to up_a / up_b) or resident ones.  Returns 0 (nn_ab / nn_ba written),
// 1 when the data does not take this screen (zeros / NaN / inf / a range fp16 scaling cannot hold: both matrices are on the device,
// ordered before the main stream's next work, and the caller runs the fp32 screen), < 0 on an error.
//
// sliced = false: both matrices first, then the kernels in their order on the lane's main stream.
// sliced = true: 105 MB of descriptors take 1.9 ms over the link while the kernels wait.  The queries go up in two
// slices, the database in `parts` of whole splits; the order on the link is A0, B0, B1 .. , A1, and every (query slice, part) block of
// the scan starts when its two sides are packed, on two compute streams in turn (so that one block's last workgroups run beside
// the next one's first).  What looks global is not:
//   * the fp16 scale needs max |v| of both matrices -- taken from A0 and B0 with one bit of headroom, every later slice's maximum
//     checked at the end (violated: the call is redone unsliced on the matrices now resident; a power-of-two scale changes no
//     result, only what the screen lets through);
//   * the screen's bound uses the largest |row|^2 of the other side -- a device cell that grows as slices are packed: a window
//     needs the bound of the rows its running minimum has SEEN (the parts so far), a threshold of the reverse search the bound of
//     the queries scanned against it (one set of thresholds per query slice, from the same warm-up minima); the verification
//     kernels run last and read the final cells.
// pairs != null: the cross-check (launch_mutual_pairs) is queued behind the verification kernels BEFORE the call's one wait, on the
// expectation that nothing takes the exact fall-back (the rule: C4 none, fragment pairs none) -- pairs->done then says that the pairs are
// in the lane's page-locked block; otherwise the caller queues it once more behind the fall-backs.  One host round trip less per call
// (~50 us: 7 % of a resident fragment pair's match).
struct SpecPairs {
    uint32_t* block_scratch = nullptr;
    uint32_t* host = nullptr;   // h_match: [0] the count, from [16] the pairs
    bool done = false;
};
int match_mfma(DeviceCtx* ctx, const MatchSide& side_a, uint32_t na, const MatchSide& side_b, uint32_t nb, double* up_a, double* up_b,
               bool sliced, uint32_t* nn_ab, uint32_t* nn_ba, uint64_t* fallbacks, SpecPairs* pairs = nullptr) {
    if (pairs) pairs->done = false;
    const uint32_t QA = sliced && !side_a.dev ? 2u : 1u, PB = sliced && !side_b.dev ? 4u : 1u;
    sliced = QA * PB > 1;
    hipStream_t s = ctx->stream, s2 = s, cs = s;
    if (sliced) {
        s2 = pre_stream_of(ctx);
        cs = copy_stream_of(ctx);
        if (!s2 || !cs || !aux_event_of(ctx, 2 * (QA + PB))) return M3D_ERR_DEVICE;
    }
    DevBuf part_min, ns2, nd2, ring, ring_count, evict, over_list, scal, pA_s, pA_d, pB_s, pB_d, premin, rev_premin, rthr, rcnt, rcand,
        rlist, rlist_cnt, over_list_r, mabs, rperm;
    bool in_flight = false;   // work queued on s2 / cs that s has not been made to wait for
    auto done = [&](int r) {
        if (r != 0 && in_flight) {   // a failure half-way: nothing of this call may still run when its blocks go back
            (void)hipStreamSynchronize(cs);
            (void)hipStreamSynchronize(s2);
            (void)hipStreamSynchronize(s);
        }
        for (DevBuf* b : {&part_min, &ns2, &nd2, &ring, &ring_count, &evict, &over_list, &scal, &pA_s, &pA_d, &pB_s, &pB_d, &premin,
                          &rev_premin, &rthr, &rcnt, &rcand, &rlist, &rlist_cnt, &over_list_r, &mabs, &rperm})
            b->release();
        return r;
    };
    // ~3 blocks fit a CU (146 VGPRs): aim for >= 10 rounds of 768 blocks so the last round costs little
    MatchWork w;
    w.na = na;
    w.nb = nb;
    w.a = side_a.dev ? side_a.dev : up_a;
    w.b = side_b.dev ? side_b.dev : up_b;
    const uint32_t a_tiles = mfma_tiles(na), b_tiles = mfma_tiles(nb);
    // ... but a split should hold an eighth of a mid-sized database at least (1024 .. 6144 rows): with the 512-query workgroups of
    // round 5 a 50 000 x 50 000 search was cut into 42 splits of 37 tiles -- prologue, 84 warm-up minima per query, 84 slices for
    // the verification to read -- and took 1.85 ms; with 8 splits 1.39 (35 000: 1.25 -> 1.11, 70 000: 2.43 -> 2.35; 10 000 and
    // 100 000 upwards unchanged)
    const auto rows_per_split = [](uint32_t ndb) { return std::min(6144u, std::max(1024u, ndb / 8u)); };
    w.splits = splits_for(na, nb, 256, rows_per_split(nb), 8192);
    w.splits_r = splits_for(nb, na, 256, rows_per_split(na), 8192);
    if (PB > 1) w.splits = (w.splits + PB - 1) / PB * PB;
    // the last full-length share of the tiles in four splits of 1/2, 1/4, 1/8, 1/8 (SplitPlan): splits - 3 shares in all
    w.plan.tail = w.splits >= 8 ? 4u : 0u;
    w.plan.full = w.splits - w.plan.tail;
    const uint32_t shares = w.plan.full + (w.plan.tail ? 1u : 0u);
    w.plan.per = (b_tiles + shares - 1) / shares;
    // (an eighth of a split a whole number of tiles; sliced: parts begin on a multiple of 16 tiles = 512 rows -- the packed query
    // layout's padding --, also a part that begins with tail split k, per - (per >> k) tiles into the tail)
    const uint32_t spp = w.splits / PB;   // splits per part
    uint32_t align = PB > 1 ? 16u : 8u;
    for (uint32_t k = 1; PB > 1 && k < w.plan.tail; ++k)
        if ((w.plan.full + k) % spp == 0) align = 16u << k;
    w.plan.per = (w.plan.per + align - 1) / align * align;
    // cuts: query slices on multiples of 512 rows, database parts of whole splits
    std::vector<Cut> qa, pb;
    {
        // the first query slice a quarter of the queries (it holds the reverse warm-up's sample, the first eighth): with halves the
        // scan starts 0.24 ms later, with a sixth the chip runs out of work before the second slice has arrived (200 k x 200 k:
        // 9.55 / 9.3 / 9.5 ms for 1/2, 1/4, 1/6)
        const uint32_t first = QA > 1 ? std::min(na, ((na + 3u) / 4u + 511u) / 512u * 512u) : na;
        qa.push_back(Cut{0, first});
        if (first < na) qa.push_back(Cut{first, na - first});
        for (uint32_t p = 0; p < PB; ++p) {
            const uint32_t r0 = std::min<uint64_t>((uint64_t)w.plan.begin(p * spp) * 32u, nb);
            const uint32_t r1 = std::min<uint64_t>((uint64_t)w.plan.end((p + 1) * spp - 1) * 32u, nb);
            pb.push_back(Cut{r0, r1 - r0});   // (rows 0: the part's splits are empty; their scan only writes the slices' defaults)
        }
    }
    const size_t part = (size_t)2 * w.splits * na, tile_bytes = (size_t)kMfmaRowHalfs * 2 * 32;
    const size_t n_cuts = qa.size() + pb.size();
    if (!part_min.reserve(sizeof(float) * part) || !ns2.reserve(sizeof(float) * na) || !nd2.reserve(sizeof(float) * nb) ||
        !ring.reserve(sizeof(uint2) * (size_t)kRing * part) || !ring_count.reserve(sizeof(uint32_t) * part) ||
        !evict.reserve(sizeof(float) * part) || !over_list.reserve(sizeof(uint32_t) * na) || !scal.reserve(256) ||
        !pA_s.reserve(tile_bytes * a_tiles) || !pA_d.reserve(tile_bytes * b_tiles) || !pB_s.reserve(tile_bytes * mfma_query_tiles(na)) ||
        !pB_d.reserve(tile_bytes * mfma_query_tiles(nb)) || !premin.reserve(sizeof(float) * part) ||
        !rev_premin.reserve(sizeof(float) * (size_t)2 * w.splits_r * nb) || !rthr.reserve(sizeof(float) * 40 * (size_t)b_tiles * qa.size()) ||
        !rcnt.reserve(sizeof(uint32_t) * nb) || !rcand.reserve(sizeof(uint2) * (size_t)kMatchRevCap * nb) ||
        !rlist.reserve(sizeof(uint2) * (size_t)kMatchRevLane * part) || !rlist_cnt.reserve(sizeof(uint32_t) * part) ||
        !over_list_r.reserve(sizeof(uint32_t) * nb) || !mabs.reserve(sizeof(double) * kMaxAbsPartials * n_cuts) ||
        !rperm.reserve(sizeof(uint32_t) * 32 * (size_t)b_tiles))
        return done(M3D_ERR_DEVICE);
    w.qB_a = pB_s.p;
    w.dA_a = pA_s.p;
    w.qB_b = pB_d.p;
    w.dA_b = pA_d.p;
    w.an2 = ns2.as<float>();
    w.bn2 = nd2.as<float>();
    float* sc = scal.as<float>();   // [0] max |a row|^2, [1] max |b row|^2, [2], [3] the two overflow counters (u32)
    w.max_an2 = sc + 0;
    w.max_bn2 = sc + 1;
    w.overflow_count = scal.as<uint32_t>() + 2;
    w.overflow_count_r = scal.as<uint32_t>() + 3;
    w.premin = premin.as<float>();
    w.ring = ring.as<uint2>();
    w.ring_count = ring_count.as<uint32_t>();
    w.part_min = part_min.as<float>();
    w.evict_min = evict.as<float>();
    w.rev_premin = rev_premin.as<float>();
    w.rthr = rthr.as<float>();
    w.rcnt = rcnt.as<uint32_t>();
    w.rperm = rperm.as<uint32_t>();
    w.rcand = rcand.as<uint2>();
    w.rlist = rlist.as<uint2>();
    w.rlist_cnt = rlist_cnt.as<uint32_t>();
    w.overflow_list = over_list.as<uint32_t>();
    w.overflow_list_r = over_list_r.as<uint32_t>();
    w.nn_ab = nn_ab;
    w.nn_ba = nn_ba;

    bool ok = true;
    size_t next_event = 0;
    auto upload = [&](int side, const Cut& c) -> hipEvent_t {   // rows of a host matrix onto the link; the event that follows them (sliced)
        const MatchSide& sd = side == 0 ? side_a : side_b;
        if (sd.dev || !c.rows) return nullptr;
        double* dst = (side == 0 ? up_a : up_b) + (size_t)c.row0 * 33;
        ok = ok && hipMemcpyAsync(dst, sd.host + (size_t)c.row0 * 33, sizeof(double) * 33 * c.rows, hipMemcpyHostToDevice, cs) == hipSuccess;
        if (!sliced) return nullptr;
        hipEvent_t e = aux_event_of(ctx, next_event++);
        ok = ok && e && hipEventRecord(e, cs) == hipSuccess;
        in_flight = true;
        return e;
    };
    auto wait_for = [&](hipStream_t st, hipEvent_t e) {
        if (e) ok = ok && hipStreamWaitEvent(st, e, 0) == hipSuccess;
    };
    auto mark = [&](hipStream_t st) -> hipEvent_t {
        if (!sliced) return nullptr;
        hipEvent_t e = aux_event_of(ctx, next_event++);
        ok = ok && e && hipEventRecord(e, st) == hipSuccess;
        return e;
    };
    auto slice_max = [&](int side, const Cut& c, size_t cut_index, hipStream_t st) {   // max |v| of a slice -> its block of partial maxima
        launch_max_abs((side == 0 ? w.a : w.b) + (size_t)c.row0 * 33, (size_t)c.rows * 33, mabs.as<double>() + cut_index * kMaxAbsPartials, st);
    };
    auto finish_uploads = [&]() {   // every slice not yet sent; the main stream behind the link
        if (!sliced) return;
        for (size_t h = 1; h < qa.size(); ++h) wait_for(s, upload(0, qa[h]));
        for (size_t p = 1; p < pb.size(); ++p) wait_for(s, upload(1, pb[p]));
    };

    // ---- the first slices, and the scale --------------------------------------------------------------------------------------
    // The copy stream writes into blocks just taken from the lane's free list, which is ordered by the lane's MAIN stream only:
    // the copy (and second compute) stream start behind whatever the main stream still has queued (today every call
    // leaves its main stream idle before it parks a block; this makes the invariant explicit instead of assumed).
    if (sliced) {
        const hipEvent_t e0 = mark(s);
        wait_for(cs, e0);
        if (s2 && s2 != s) wait_for(s2, e0);
    }
    ok = ok && hipMemsetAsync(mabs.p, 0, sizeof(double) * kMaxAbsPartials * n_cuts, s) == hipSuccess;   // (cuts without rows, resident sides: no pass)
    wait_for(s, upload(0, qa[0]));
    wait_for(s, upload(1, pb[0]));
    MATCH_MARK("first slices on the link");
    const bool known_a = side_a.dev && side_a.max_abs >= 0.0, known_b = side_b.dev && side_b.max_abs >= 0.0;
    std::vector<double> hp(kMaxAbsPartials * n_cuts, 0.0);
    if (known_a) hp[0] = side_a.max_abs;   // (a resident side knows its max |v|, NaN included: no pass, no round trip)
    if (known_b) hp[kMaxAbsPartials * qa.size()] = side_b.max_abs;
    if (!known_a) slice_max(0, qa[0], 0, s);
    if (!known_b) slice_max(1, pb[0], qa.size(), s);
    // (the call's two waits go through page-locked memory and a polled word -- stream_wait_spin --: a copy into pageable memory is
    //  staged, and the runtime's wait wakes the caller 10-20 us after the stream has drained)
    const size_t hp_bytes = sizeof(double) * hp.size();
    ok = ok && ctx->h_reg.reserve(hp_bytes + 16);
    double* hp_pin = ok ? reinterpret_cast<double*>(ctx->h_reg.as<uint8_t>() + 16) : nullptr;
    if (ok && !(known_a && known_b)) {
        if (!known_a) ok = ok && hipMemcpyAsync(hp_pin, mabs.p, sizeof(double) * kMaxAbsPartials, hipMemcpyDeviceToHost, s) == hipSuccess;
        if (!known_b)
            ok = ok && hipMemcpyAsync(hp_pin + kMaxAbsPartials * qa.size(), mabs.as<double>() + kMaxAbsPartials * qa.size(),
                                      sizeof(double) * kMaxAbsPartials, hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipGetLastError() == hipSuccess && stream_wait_spin(ctx) == M3D_OK;
        if (ok && !known_a) std::memcpy(hp.data(), hp_pin, sizeof(double) * kMaxAbsPartials);
        if (ok && !known_b) std::memcpy(hp.data() + kMaxAbsPartials * qa.size(), hp_pin + kMaxAbsPartials * qa.size(), sizeof(double) * kMaxAbsPartials);
    }
    if (!ok) return done(M3D_ERR_DEVICE);
    MATCH_MARK("max |v| of the first slices known");
    auto fold = [](const std::vector<double>& v) {
        double mx = 0.0;
        for (double x : v) mx = (x > mx || x != x) ? x : mx;
        return mx;
    };
    const double mx0 = fold(hp);
    if (!(mx0 > 0.0 && std::isfinite(mx0) && mx0 < 1e300 && mx0 > 1e-300)) {   // zeros / NaN / inf: the fp32 screen's exact fallback handles them
        finish_uploads();
        return done(ok ? 1 : M3D_ERR_DEVICE);
    }
    // power-of-two scale that brings max |v| into [1024, 2048) (fp16 hi/lo split keeps 22 bits; norms / 2^15 fit) -- sliced: into
    // [512, 1024), so that slices still on their way may hold values up to twice the largest one seen
    int e2;
    (void)std::frexp(mx0, &e2);   // mx0 = f * 2^e2, f in [0.5, 1)
    const double scale = std::ldexp(1.0, (sliced ? 10 : 11) - e2);
    w.scale = scale;

    // ---- pack / warm up / scan, block by block --------------------------------------------------------------------------------
    ok = ok && hipMemsetAsync(sc, 0, 4 * sizeof(float), s) == hipSuccess;
    match_pack(w, 0, qa[0].row0, qa[0].rows, scale, s);
    std::vector<hipEvent_t> scanned;   // ends of the scan blocks queued on s2
    hipEvent_t prev = nullptr;         // the previous part's thresholds (its pack and norm maximum before them)
    for (size_t p = 0; p < pb.size(); ++p) {
        hipStream_t st = p % 2 ? s2 : s;
        if (p > 0) {
            wait_for(st, upload(1, pb[p]));
            wait_for(st, prev);
            if (pb[p].rows) slice_max(1, pb[p], qa.size() + p, st);
        }
        if (pb[p].rows) match_pack(w, 1, pb[p].row0, pb[p].rows, scale, st);
        if (p == 0) match_forward_warm(w, qa[0].row0, qa[0].rows, st);
        if (pb[p].rows) {
            match_reverse_thresholds(w, pb[p].row0, pb[p].rows, 0, st);
            match_order_part(w, pb[p].row0, pb[p].rows, st);   // (the part's database layout once more, in the thresholds' order)
        }
        prev = mark(st);
        match_scan(w, qa[0].row0, qa[0].rows, (uint32_t)p * spp, spp, w.plan.end((uint32_t)(p + 1) * spp - 1), 0, st);
        if (st != s) scanned.push_back(mark(st));
    }
    for (size_t h = 1; h < qa.size(); ++h) {
        wait_for(s, upload(0, qa[h]));
        wait_for(s, prev);
        for (hipEvent_t e : scanned) wait_for(s, e);
        scanned.clear();
        slice_max(0, qa[h], h, s);
        match_pack(w, 0, qa[h].row0, qa[h].rows, scale, s);
        match_forward_warm(w, qa[h].row0, qa[h].rows, s);
        for (const Cut& c : pb)
            if (c.rows) match_reverse_thresholds(w, c.row0, c.rows, (int)h, s);
        match_scan(w, qa[h].row0, qa[h].rows, 0, w.splits, b_tiles, (int)h, s);
    }
    for (hipEvent_t e : scanned) wait_for(s, e);
    in_flight = false;   // (everything queued elsewhere is now in front of s)
    MATCH_MARK("everything queued");
    for (const Cut& c : qa) match_verify_forward(w, c.row0, c.rows, s);
    for (const Cut& c : qa) match_reverse_bin(w, c.row0, c.rows, s);
    match_verify_reverse(w, s);
    if (pairs) {
        pairs->host[0] = 0;
        launch_mutual_pairs(nn_ab, nn_ba, na, nb, pairs->block_scratch, pairs->host, reinterpret_cast<uint2*>(pairs->host + 16), s);
    }
    uint32_t over[2] = {0, 0};
    ok = ok && hipMemcpyAsync(ctx->h_reg.p, w.overflow_count, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s) == hipSuccess;
    if (sliced) ok = ok && hipMemcpyAsync(hp_pin, mabs.p, hp_bytes, hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipGetLastError() == hipSuccess && stream_wait_spin(ctx) == M3D_OK;
    if (!ok) {
        in_flight = true;
        return done(M3D_ERR_DEVICE);
    }
    std::memcpy(over, ctx->h_reg.p, sizeof(over));
    if (sliced) std::memcpy(hp.data(), hp_pin, hp_bytes);
    MATCH_MARK("scans and verification done");
    if (sliced) {   // did every slice fit the scale chosen from the first ones?
        if (known_a) hp[0] = side_a.max_abs;
        if (known_b) hp[kMaxAbsPartials * qa.size()] = side_b.max_abs;
        const double mx = fold(hp);
        if (!(std::isfinite(mx) && mx * scale < 2048.0)) {   // no: once more, whole, on the matrices now resident
            done(0);
            g_match_path |= 4u;
            return match_mfma(ctx, MatchSide{nullptr, w.a, side_a.dev ? side_a.max_abs : -1.0}, na, MatchSide{nullptr, w.b, side_b.dev ? side_b.max_abs : -1.0},
                              nb, up_a, up_b, false, nn_ab, nn_ba, fallbacks, pairs);
        }
    }
    DevBuf exact_scratch;   // (a handful of queries as a rule; without the block the fall-back runs one workgroup per query)
    const uint32_t over_max = std::max(over[0], over[1]);
    const bool have_scratch = over_max && exact_scratch.reserve(nn_exact_scratch_bytes(over_max));
    const hipError_t fe = match_exact_fallbacks(w, over, have_scratch ? exact_scratch.p : nullptr, s);
    if (have_scratch) {
        (void)hipStreamSynchronize(s);   // (the block goes back to the lane's list below)
        exact_scratch.release();
    }
    if (fe != hipSuccess) return done(M3D_ERR_DEVICE);
    *fallbacks = (uint64_t)over[0] + over[1];
    if (pairs) pairs->done = over[0] == 0 && over[1] == 0;   // (nobody's nearest neighbour changed behind the cross-check)
    g_match_path |= 1u | (sliced ? 2u : 0u);
    return done(0);
}
}  // namespace

// ... on a lane the caller holds (m3d_match_mutual_nn, global_registration_on); arguments already checked
int m3d::match_mutual_nn_on(DeviceCtx* ctx, const MatchSide& side_src, size_t n_src, const MatchSide& side_dst, size_t n_dst,
                            int dim, size_t* out_src, size_t* out_dst, size_t* k_out) {
    MATCH_TIMELINE_BEGIN();
    HIPCHK(hipSetDevice(ctx->device));
    const double* feat_src = side_src.host;
    const double* feat_dst = side_dst.host;
    DevBuf fs, fd, bd, bi, nn01, nn10, fs32, fd32, ns2, nd2, ring, ring_count, evict, over_list, scal, mblk;
    auto done = [&](int r) {
        for (DevBuf* b : {&fs, &fd, &bd, &bi, &nn01, &nn10, &fs32, &fd32, &ns2, &nd2, &ring, &ring_count, &evict, &over_list, &scal, &mblk}) b->release();
        return r;
    };
    const uint32_t ns = (uint32_t)n_src, nd = (uint32_t)n_dst;
    // dim 33 (FPFH): screened exact search (m3d_match_kernels.hip): split-fp16 MFMA screen by default, the fp32
    // VALU screen with M3D_MATCH_SCREEN=fp32 or when the data does not fit fp16 scaling; M3D_MATCH_BRUTE=1 and
    // every other width: fp64 brute force.
    const bool screened = dim == 33 && !config().match_brute;
    const bool try_mfma = screened && !config().match_fp32_screen;
    // a side that is resident already (MatchSide::dev: a fragment's descriptors, m3d_global_registration.cpp) is not uploaded
    if ((!side_src.dev && !fs.reserve(sizeof(double) * (size_t)dim * ns)) || (!side_dst.dev && !fd.reserve(sizeof(double) * (size_t)dim * nd)) ||
        !nn01.reserve(sizeof(uint32_t) * ns) || !nn10.reserve(sizeof(uint32_t) * nd))
        return done(M3D_ERR_DEVICE);
    if (!mblk.reserve(sizeof(uint32_t) * mutual_blocks(ns)) || !ctx->h_match.reserve(64 + sizeof(uint32_t) * 2 * ((size_t)ns + 1)))
        return done(M3D_ERR_DEVICE);
    MATCH_MARK("blocks reserved");
    const double* fs_p = side_src.dev ? side_src.dev : fs.as<double>();
    const double* fd_p = side_dst.dev ? side_dst.dev : fd.as<double>();
    bool ok = true, resident = false, matched = false;
    SpecPairs spec_pairs;
    g_match_path = 0;
    if (try_mfma) {
        // the two std::threads of correspondence_matching.cpp:59-62 become ONE pass over the product tiles: the scan of the source
        // queries against the target rows also collects, per target row, the source rows that can be its nearest neighbour
        // (alone on the device: the kernels of other calls fill a call's upload gap anyway, and their lanes' streams are better left
        // with the hardware queues; match_pipeline = 2 slices whatever the size and the company: the tests' switch)
        const int mode = config().match_pipeline;
        const bool sliced = (!side_src.dev || !side_dst.dev) &&
                            (mode == 2 || (mode == 1 && ns >= kSliceMinRows && nd >= kSliceMinRows && lanes_held() <= 1));
        uint64_t fb = 0;
        spec_pairs.block_scratch = mblk.as<uint32_t>();
        spec_pairs.host = ctx->h_match.as<uint32_t>();
        const int r = match_mfma(ctx, side_src, ns, side_dst, nd, fs.as<double>(), fd.as<double>(), sliced, nn01.as<uint32_t>(),
                                 nn10.as<uint32_t>(), &fb, &spec_pairs);
        if (r < 0) return done(fail(M3D_ERR_DEVICE, "m3d_match_mutual_nn: HIP error"));
        matched = r == 0;
        resident = true;   // (either way both matrices are on the device)
        if (matched) g_match_fallbacks = fb;
    }
    if (!matched) {
        if (!resident)
            ok = (side_src.dev || hipMemcpyAsync(fs.p, feat_src, sizeof(double) * (size_t)dim * ns, hipMemcpyHostToDevice, ctx->stream) == hipSuccess) &&
                 (side_dst.dev || hipMemcpyAsync(fd.p, feat_dst, sizeof(double) * (size_t)dim * nd, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
        uint32_t s01 = splits_for(ns, nd, 512, 256), s10 = splits_for(nd, ns, 512, 256);
        const size_t part = std::max((size_t)s01 * ns, (size_t)s10 * nd);
        const uint32_t nmax = std::max(ns, nd);
        if (ok && (!bd.reserve(sizeof(double) * part) || !bi.reserve(sizeof(uint32_t) * part))) return done(M3D_ERR_DEVICE);
        if (ok && screened) {
            if (!ns2.reserve(sizeof(float) * ns) || !nd2.reserve(sizeof(float) * nd) || !ring.reserve(sizeof(uint2) * (size_t)kRing * part) ||
                !ring_count.reserve(sizeof(uint32_t) * part) || !evict.reserve(sizeof(float) * part) ||
                !over_list.reserve(sizeof(uint32_t) * nmax) || !scal.reserve(256) ||
                // fp32 rows carry one spare row (prefetch target of the last iteration)
                !fs32.reserve(sizeof(float) * (size_t)kScreenDimP * ((size_t)ns + 1)) ||
                !fd32.reserve(sizeof(float) * (size_t)kScreenDimP * ((size_t)nd + 1)))
                return done(M3D_ERR_DEVICE);
            float* sc = scal.as<float>();   // [0] max |src|^2, [1] max |dst|^2, [2] overflow counter (u32)
            uint32_t over01 = 0, over10 = 0;
            launch_to_f32_33(fs_p, ns, fs32.as<float>(), ns2.as<float>(), sc + 0, ctx->stream);
            launch_to_f32_33(fd_p, nd, fd32.as<float>(), nd2.as<float>(), sc + 1, ctx->stream);
            float h_max[2] = {0.0f, 0.0f};
            ok = hipMemcpyAsync(h_max, sc, sizeof(h_max), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                 hipStreamSynchronize(ctx->stream) == hipSuccess;
            ok = ok && launch_nn_screened33(fs_p, fs32.as<float>(), ns2.as<float>(), ns, fd_p, fd32.as<float>(), nd, h_max[1], s01,
                                            ring.as<uint2>(), ring_count.as<uint32_t>(), bd.as<float>(), evict.as<float>(),
                                            over_list.as<uint32_t>(), scal.as<uint32_t>() + 2, nn01.as<uint32_t>(), &over01,
                                            ctx->stream) == hipSuccess;
            ok = ok && launch_nn_screened33(fd_p, fd32.as<float>(), nd2.as<float>(), nd, fs_p, fs32.as<float>(), ns, h_max[0], s10,
                                            ring.as<uint2>(), ring_count.as<uint32_t>(), bd.as<float>(), evict.as<float>(),
                                            over_list.as<uint32_t>(), scal.as<uint32_t>() + 2, nn10.as<uint32_t>(), &over10,
                                            ctx->stream) == hipSuccess;
            g_match_fallbacks = (uint64_t)over01 + over10;
            g_match_path |= 8u;
        } else if (ok) {
            g_match_path |= 16u;
            launch_nn(fs_p, ns, fd_p, nd, dim, s01, bd.as<double>(), bi.as<uint32_t>(), nn01.as<uint32_t>(), ctx->stream);
            launch_nn(fd_p, nd, fs_p, ns, dim, s10, bd.as<double>(), bi.as<uint32_t>(), nn10.as<uint32_t>(), ctx->stream);
        }
    }
    MATCH_MARK("both searches queued or done");
    // cross-check, correspondence_matching.cpp:64-78, on the device: the mutual pairs in the order of the source index, written straight
    // into the lane's page-locked block (the two index arrays + a host loop with an unpredictable branch: 0.9 ms for 200 000 queries;
    // the loop without the branch: 0.3; this: 0.1)
    uint32_t* hm = ctx->h_match.as<uint32_t>();
    if (ok && !(matched && spec_pairs.done)) {
        hm[0] = 0;
        launch_mutual_pairs(nn01.as<uint32_t>(), nn10.as<uint32_t>(), ns, nd, mblk.as<uint32_t>(), hm, reinterpret_cast<uint2*>(hm + 16), ctx->stream);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!ok) return done(fail(M3D_ERR_DEVICE, "m3d_match_mutual_nn: HIP error"));
    MATCH_MARK("mutual pairs on the host");
    const size_t k = hm[0];
    for (size_t t = 0; t < k; ++t) {
        out_src[t] = hm[16 + 2 * t];
        out_dst[t] = hm[17 + 2 * t];
    }
    *k_out = k;
    MATCH_MARK("cross-check");
    const int rc = done(M3D_OK);
    MATCH_TIMELINE_END();
    return rc;
}

int m3d::device_max_abs(DeviceCtx* ctx, const double* dev, size_t n, double* out) {
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf part;
    RESERVE(part, sizeof(double) * kMaxAbsPartials);
    launch_max_abs(dev, n, part.as<double>(), ctx->stream);
    std::vector<double> hp(kMaxAbsPartials, 0.0);
    const bool ok = hipMemcpyAsync(hp.data(), part.p, sizeof(double) * kMaxAbsPartials, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    part.release();
    if (!ok) return fail(M3D_ERR_DEVICE, "device_max_abs: HIP error");
    double mx = 0.0;
    for (double v : hp) mx = (v > mx || v != v) ? v : mx;
    *out = mx;
    return M3D_OK;
}

