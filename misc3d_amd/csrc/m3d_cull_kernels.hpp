// m3d_cull_kernels.hpp -- launch interface of the spatially culled scoring path (m3d_cull_kernels.hip).
#pragma once
#include "m3d_bound_fp.hpp"   // kFrameStride, kBoundBins, kCumStride
#include "m3d_kernels.hpp"

namespace m3d {

constexpr int kTilePoints = 512;      // points per tile = one wave x 8 rows of 64 (kept in VGPRs)
constexpr int kBoxStride = 12;        // doubles per tile box record: centre xyz, half extents xyz, 2 pad, then SIX FLOATS (doubles 8..10):
                                      // the same box relative to SortedView::origin, half extents rounded outwards (cull_tiles32_k)
constexpr int kCull32Stride = 12;     // floats per fp32 box-test record of a hypothesis (minimal_fit_k -> cull_tiles32_k)
constexpr uint32_t kGroupsPerBlock = 8;  // 64-hypothesis groups per score_mask_k workgroup (<= 64)

// Hilbert-sorted copy of a resident cloud: SoA padded with NaN to a multiple of kTilePoints, plus one
// bounding box per tile (half extent < 0 marks an empty tile).
struct SortedView {
    const double* x;
    const double* y;
    const double* z;
    const double* boxes;  // n_tiles x kBoxStride
    uint32_t n_tiles;
    // optional, written by tile_boxes_k: every tile's points once more as fp32 offsets from the tile's box centre, in the
    // register layout of score_screen_k (kTileF32Floats floats per tile: [coordinate][row pair][lane] x (row 2j, row 2j + 1));
    // box slot 6 then says whether the tile can be screened (every offset finite).  A wave of score_screen_k loads 6 KB
    // ready to use instead of 12 KB of doubles it has to shift and convert.
    float* tile_f32 = nullptr;
    // optional, written by tile_frames_k (m3d_bound.hip): a robust local frame and a histogram per tile, from which plane_bound_k
    // takes an upper bound of a plane hypothesis' inlier count per (tile, hypothesis) pair.  They describe THESE tiles: a copy that
    // is re-partitioned drops them.
    const double* frames = nullptr;        // n_tiles x kFrameStride
    const uint16_t* frame_cum = nullptr;   // n_tiles x kCumStride
    bool has_dead = false;   // some points are tombstones (x = NaN in both copies: launch_poison_plane_inliers); planes only
    double max_abs = __builtin_inf();  // >= |coordinate| of every point of the cloud (the box tests' rounding margin); inf = unknown (nothing culled)
    // centre of the cloud's bounding box and the largest |coordinate - origin| (fp32 box tests work relative to it);
    // radius = inf: unknown, the fp64 box tests are used
    double origin[3] = {0.0, 0.0, 0.0};
    double radius = __builtin_inf();
};

constexpr int kTileF32Floats = 3 * kTilePoints;
void launch_tile_frames(const SortedView& s, double* frames, uint16_t* cum, hipStream_t st);
// ubsum[h] += upper bound of hypothesis h's inliers over the tiles it can touch, for the hypotheses of the list `surv` (written by
// the keep kernels of groups [group_begin, group_end): *surv_count ids; ubsum zero on entry); the keep bit of a hypothesis whose
// bound stays below best_count[0] is cleared, *surv_count ends at 0.  tickets: 1 + (hypotheses of the window / 64) words, zero
// before the first launch (the kernel leaves them zero).  cull32: the window's fp32 box-test records (null: the masks are
// read instead).  kind: planes (0) and cylinders (2: the shell over a tile as a slab, m3d_bound_fp.hpp); needs s.frames.
void launch_plane_bound(int kind, const SortedView& s, const double* score, const unsigned long long* masks, unsigned long long* keep,
                        uint32_t n_groups, uint32_t group_begin, uint32_t group_end, uint32_t* ubsum,
                        const uint32_t* best_count, uint32_t* surv_count, const uint32_t* surv, uint32_t* tickets,
                        const float* cull32, hipStream_t st,
                        bool always = false /* false: a list longer than half of the window is discarded unbounded (nothing worth pruning against) */,
                        const unsigned long long* touched = nullptr /* launch_cull_mask's per-hypothesis words for these groups (cylinders: read
                                                                       instead of repeating the box tests) */,
                        uint32_t touched_stride = 0);
void launch_tile_boxes(const SortedView& s, double* boxes, hipStream_t st);
// Tombstones (m3d_poison.hpp): the job that kills the inliers of the plane `model` (device) in place in the sorted copy
// `s`; *total (device, cleared by the owner) accumulates the number of points killed over all launches.
PoisonJob make_poison_job(const SortedView& s, const double* model, double thr, uint32_t* total);
void launch_poison_plane_inliers(const PoisonJob& job, hipStream_t st);


// masks: n_tiles x n_groups uint64, bit b of masks[t][g] = hypothesis 64 g + b may have inliers in tile t.
// ub (may be null; n_groups * 64 entries, zeroed here): number of tiles each hypothesis may touch.
void launch_cull_mask(int kind, const SortedView& s, const double* score, const uint8_t* valid, uint32_t h_count,
                      uint32_t n_groups, unsigned long long* masks, uint32_t* ub, hipStream_t st,
                      bool ub_is_zero = false, uint32_t group_begin = 0,
                      uint32_t group_end = 0xFFFFFFFFu /* only groups [group_begin, group_end) of the chunk (sharded fits) */,
                      const float* cull32 = nullptr /* minimal_fit_k's fp32 box-test records (kCull32Stride floats per
                                                       hypothesis): cull_tiles32_k instead of the fp64 tests (m3d_config.cull_fp32) */,
                      uint32_t* ubp = nullptr /* phased scoring (launch_score_phased): per hypothesis the touched tiles with index % 4 == 0
                                                 (low 16 bits) and == 1 (high 16 bits); zero on entry; fp32 box tests only */,
                      unsigned long long* touched = nullptr /* optional, ceil(n_tiles / 64) x touched_stride words: bit b of
                                                               touched[k][h] = hypothesis h of the chunk may have inliers in tile 64 k + b --
                                                               the masks once more, a word per HYPOTHESIS (plane_bound_k's view) */,
                      uint32_t touched_stride = 0, bool* touched_written = nullptr /* set when the launch wrote them (cull_hyp32_k only) */);
// keep[g]: hypotheses still worth scoring (ub[h] * 512 >= best_count[0]); ub == null -> all ones.
// zero_counts_rep != null: the same launch clears the kCountReplicas x rep_stride + kPairReplicas counter words.
void launch_keep_mask(const uint32_t* ub, const uint32_t* best_count, uint32_t n_groups, unsigned long long* keep,
                      hipStream_t st, uint32_t* zero_counts_rep = nullptr, uint32_t rep_stride = 0,
                      uint32_t group_offset = 0 /* the launch covers groups [group_offset, group_offset + n_groups) */,
                      uint32_t* surv_count = nullptr, uint32_t* surv = nullptr /* the kept hypotheses as a list as well (emit_survivors:
                                                                                  plane_bound_k's input; *surv_count zero before the launch) */);
// counts_rep[tile % kCountReplicas][h] += inliers of hypothesis h in the tiles whose (mask & keep) bit is set;
// counts_rep (kCountReplicas x rep_stride u32) zero on entry; launch_sum_replicas folds the replicas.
constexpr int kCountReplicas = 16;
// replicas of the launch's (tile, hypothesis) pair counter (m3d_stats.pairs_scored): one add per surviving wave,
// consecutive tiles hit different words so the adds never queue on one address
constexpr int kPairReplicas = 1024;
constexpr int kPairLead = kPairReplicas - 1;   // the last word: pairs the LEAD pass of the chunk evaluated (lead_fold_keep_k)
// ... of which the last kPairReplicas - kPairMain count the pairs the fp32 screen handed to the exact fp64 code
// (score_screen_k; m3d_stats.pairs_exact)
constexpr int kPairMain = 1008;
void launch_score_mask(int kind, const SortedView& s, const double* score, const unsigned long long* masks,
                       const unsigned long long* keep, uint32_t n_groups, uint32_t* counts_rep, uint32_t rep_stride,
                       uint32_t* pair_rep /* kPairReplicas u32, zero on entry: evaluated (tile, hypothesis) pairs */,
                       hipStream_t st, uint32_t group_begin = 0, uint32_t group_end = 0xFFFFFFFFu /* window of the chunk's groups */,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr /* both set: receive the launch's own start / stop
                       times (m3d_stats.ms_score_kernel) */,
                       bool thinned = false /* plane_bound_k has pruned the window: more groups per workgroup */);
// score_mfma_k (m3d_score_mfma.hip): the same counting with the screen on the matrix pipe -- planes, m3d_config.score_mfma.
// false: not applicable (another kind, tombstones in the copy, no fp32 tile offsets, switched off), nothing launched;
// launch_score_mask calls it first.
bool launch_score_mfma(int kind, const SortedView& s, const double* score, const unsigned long long* masks,
                       const unsigned long long* keep, uint32_t n_groups, uint32_t* counts_rep, uint32_t rep_stride,
                       uint32_t* pair_rep, hipStream_t st, uint32_t group_begin, uint32_t group_end, hipEvent_t ev_start,
                       hipEvent_t ev_stop);
// test probe of the MFMA screen: one tile (512 x 3 doubles, device), its box (6 doubles, device), n_h plane records ->
// out_q[n_h][512][2] = (u1, u2) / Sigma_h, out_h[n_h][3] = (h, Sigma_h, E_p), out_off[512][3] = the fp32 offsets
void launch_mfma_probe(const double* pts, const double* box, double max_abs, const double* recs, uint32_t n_h, double* out_q,
                       double* out_h, float* out_off, hipStream_t st);
// cull_lead_k: the box tests of groups [lead_groups, cull_end) and the counting of the leading groups [0, lead_groups) (which
// run their own box tests) in ONE launch -- the head of a fit's first chunk on one GPU.  keep[0 .. lead_groups) and the
// counter replicas must be prepared (minimal_fit_k's LeadPrep or keep_mask_k); ub is not written for the leading groups
// (nothing reads it).  Returns false, having launched nothing, when the fp32 box tests / the fp32 screen are off or the
// geometry does not fit: the caller then issues launch_cull_mask + launch_score_mask as before.
bool launch_cull_lead(int kind, const SortedView& s, const double* score, const float* cull32, unsigned long long* masks,
                      const unsigned long long* keep, uint32_t n_groups, uint32_t lead_groups, uint32_t* counts_rep,
                      uint32_t rep_stride, uint32_t* pair_rep, uint32_t* ub, uint32_t cull_end, hipStream_t st, uint32_t* ubp = nullptr);
int score_phases_for(int kind);   // 0 / 2 / 3: what m3d_config.score_phases means for this model kind
// the scoring of a window in phases with re-pruning in between (m3d_cull_kernels.hip, "PHASED scoring"); false: not applicable
bool launch_score_phased(int kind, const SortedView& s, const double* score, const unsigned long long* masks,
                         unsigned long long* keep, uint32_t n_groups, uint32_t* counts_rep, uint32_t rep_stride,
                         uint32_t* pair_rep, const uint32_t* ub, const uint32_t* ubp, const uint32_t* best_count, hipStream_t st,
                         uint32_t group_begin, uint32_t group_end, hipEvent_t ev_start, hipEvent_t ev_stop);
// The chunk's box tests INSIDE its scoring launch (every workgroup tests its tile against its own groups): for chunks that
// prune nothing (keep all ones).  false: preconditions not met (fp32 box tests / screen off), nothing launched.
bool launch_score_own_tests(int kind, const SortedView& s, const double* score, const float* cull32, unsigned long long* masks,
                            const unsigned long long* keep, uint32_t n_groups, uint32_t groups, uint32_t* counts_rep,
                            uint32_t rep_stride, uint32_t* pair_rep, hipStream_t st, hipEvent_t ev_start = nullptr,
                            hipEvent_t ev_stop = nullptr);
// (planes and spheres go through score_screen_k unless m3d_config.score_fp32_screen is 0)
// The device's prediction of the hypothesis the replay will end with (pick_best_k, m3d_cull_kernels.hip)
struct BestPick {
    unsigned long long index;   // absolute hypothesis index, ~0 when none
    uint32_t cnt, have;
    double params[8];           // its parameter record (kModelStride doubles): RefineModel's model -- NaN in slot 0 while `tie`
    uint32_t tie, pad;          // another hypothesis holds the same count: the replay's rmse rule decides, not the device
};
struct BestPickHost {           // mirror in pinned host memory, written by the same kernel
    unsigned long long index;
    uint32_t cnt, have;
    uint32_t seq, tie;          // PickFinal::seq, stored last: the host's completion word of the chunk; tie: see BestPick
};
// Optional tail of launch_sum_replicas (one-GPU fits): pick_best_k's decision taken by the workgroup of sum_replicas_k
// that finishes last -- no pick_best_k launch, and no event behind it: the host waits for pick_host->seq.
struct PickFinal {
    unsigned long long* key = nullptr;   // device: max over the chunk of (count << 32 | ~index); zero on entry and on exit
    unsigned long long* key2 = nullptr;  // device: max over the chunk of (count << 32 | index): a different index = a count tie
    uint32_t* ticket = nullptr;          // device: finished workgroups; zero on entry and on exit
    const double* params = nullptr;      // the chunk's parameter records
    unsigned long long index_base = 0;
    int first_chunk = 0;
    BestPick* pick = nullptr;
    BestPickHost* pick_host = nullptr;
    uint32_t seq = 0;
};
// Folds the replicas of hypotheses [h_begin, h_end): record = count | valid << 31 (valid != null, h < h_count) to
// `counts` (device-visible host memory, may be null) and `counts_dev` (may be null); pairs_out (may be null; TWO words) =
// sums of the pair counters (all pairs, pairs recounted in fp64); best_count != null: atomic running maximum of the valid hypotheses' counts.
void launch_sum_replicas(const uint32_t* counts_rep, uint32_t rep_stride, uint32_t h_end, uint32_t* counts,
                         const uint32_t* pair_rep, uint32_t* pairs_out, const uint8_t* valid, uint32_t h_count,
                         uint32_t* best_count, hipStream_t st, uint32_t h_begin = 0 /* hypotheses [h_begin, h_end) */,
                         uint32_t* counts_dev = nullptr /* the same records once more, in device memory */,
                         const PickFinal* pick = nullptr,
                         const unsigned long long* keep_final = nullptr /* phased scoring: hypotheses whose bit is clear report 0 */);
void launch_pick_best(const uint32_t* records, uint32_t count, unsigned long long index_base, const double* params,
                      bool first_chunk, BestPick* pick, BestPickHost* pick_host, hipStream_t st,
                      uint32_t* records_host = nullptr /* device-visible host copy of the records (sharded fits) */,
                      uint32_t* best_count = nullptr /* raised to the best valid count among the records */);
// lead pass folded + keep masks (and cleared counters) of groups [lead / 64, lead / 64 + n_groups_rest) in one launch;
// records[h] = count | valid << 31 for h < lead; best_count (not null) raised by the lead's best valid count.
void launch_lead_fold_keep(const uint32_t* counts_rep, uint32_t rep_stride, uint32_t lead, const uint8_t* valid,
                           uint32_t h_count, uint32_t* records, uint32_t* best_count, const uint32_t* ub,
                           unsigned long long* keep, uint32_t n_groups_rest, hipStream_t st,
                           uint32_t* records_dev = nullptr,
                           uint32_t group_begin = 0xFFFFFFFFu /* first group of the keep window; default lead / 64 */,
                           unsigned long long* pick_key = nullptr /* PickFinal::key: the lead's best goes in */,
                           unsigned long long* pick_key2 = nullptr /* PickFinal::key2 */,
                           uint32_t* surv_count = nullptr, uint32_t* surv = nullptr /* the kept hypotheses as a list as well */);
void launch_count_bits(const unsigned long long* masks, const unsigned long long* keep, uint32_t n_tiles,
                       uint32_t n_groups, unsigned long long* total, hipStream_t st);

}  // namespace m3d
