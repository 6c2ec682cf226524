// m3d_driver.hpp -- host side of the C ABI: device context, resident clouds, RANSAC driver.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/misc3d_amd.h"
#include "m3d_cull_kernels.hpp"
#include "m3d_kernels.hpp"

struct m3d_cloud;

namespace m3d {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

// grow-only device / pinned-host buffers
// release() hands the block to a free list (bounded) and reserve() looks there first: the one-call entry
// points (registration, matcher, ICP, normals, boundary detection) allocate a dozen or more buffers per call, and
// hipMalloc / hipFree cost more than many of those calls' kernels.  The lists are per (device, LANE): all device work of
// a lane runs on that lane's compute stream (DeviceCtx below), a block goes back to the list of the lane it was taken
// for and only that lane takes it again, so a recycled block is never touched out of order although several calls
// run on one device at a time.  dev_pool_trim(device) frees the lists of every lane of the device.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int dev = -1;
    int lane = 0;
    bool reserve(size_t bytes);
    void release();
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes);
    void release();
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

// One in-flight chunk of hypotheses (two slots: the next chunk is scored while the host replays
// the previous one).
struct ChunkSlot {
    DevBuf samples, score, params, valid, counts;
    DevBuf cull32;   // fp32 records of the box tests (cull_tiles32_k), pairwise interleaved
    DevBuf touched;     // cylinders with the histogram bound: the masks once more, a word per hypothesis and 64 tiles (launch_cull_mask)
    DevBuf masks, ub;   // the chunk's (tile, group) bit masks and touched-tile counters: per SLOT, so that the next chunk's box
                        // tests can run (DeviceCtx::pre_stream) while this chunk is being scored
    hipEvent_t pre_done = nullptr;   // MinimalFit + box tests of the chunk finished on pre_stream
    PinBuf h_samples, h_counts, h_valid;
    bool lead_fused = false;   // the chunk's lead pass ran inside cull_lead_k (no launch, no timing events of its own)
    hipEvent_t done = nullptr;
    hipEvent_t k0 = nullptr, k1 = nullptr;  // around the scoring kernel of the chunk (m3d_stats.ms_score_kernel)
    hipEvent_t k2 = nullptr, k3 = nullptr;  // around the scoring launch of the chunk's leading hypotheses (lead_groups > 0)
    uint32_t lead_groups = 0;
    bool host_has_records = true;   // sharded: the gathered records are (being) written to h_counts
    bool done_on_copy_stream = false;   // sharded RCCL windows: `done` follows the record copy on the copy stream
    bool scored = false;   // a scoring launch was issued for the chunk (k0 / k1 recorded)
    uint32_t poll_seq = 0;   // != 0: no `done` event; completion = BestPickHost::seq reaching this value (PickFinal)
    size_t begin = 0, end = 0;
    uint32_t h_pad = 0;
};

// One LANE of a device: a compute stream with everything a call needs beside its arguments (streams, events, scratch,
// pinned words).  A device has up to m3d_config.lanes of them (created on demand); a call holds ONE lane from entry to
// return (`mu`), so calls on different lanes -- other host threads -- run side by side on the device: one thread's upload
// and host-side replay under another's kernels (SURVEY.md 8(b) "Threading"; the caller this is for:
// /root/reference/src/pipeline.cpp:428-439, one std::thread per fragment pair).  Resident objects (m3d_cloud, m3d_reg)
// stay on the lane they were created on.
struct DeviceCtx {
    int device = -1;    // PHYSICAL HIP ordinal (hipSetDevice)
    int logical = -1;   // the ordinal the caller used: the same unless m3d_config.device_aliases maps several logical devices to one
                        // physical one -- each with lanes, streams, scratch, free lists and resident tables of its own
    int lane = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // RefineModel: the inlier list goes to the host while the GeneralFit sums run
    hipStream_t pre_stream = nullptr;    // MinimalFit + box tests of chunk k + 1 under the scoring launches of chunk k (fits of several chunks)
    hipEvent_t ev_compact = nullptr;
    std::mutex mu;  // one call at a time per lane (CtxLock / LaneLock)
    ChunkSlot slot[2];
    DevBuf partial, block_counts, total, idx, dist, sum_partial, sums, best_params, small;
    DevBuf masks, keep;        // culled scoring: (tile, 64-hypothesis group) bit masks, per-group keep masks
    DevBuf ub, best_count;     // bound-and-prune: surviving tiles per hypothesis, running best count
    DevBuf counts_rep;         // kCountReplicas copies of the per-hypothesis counters (short atomic chains)
    PinBuf h_small;
    PinBuf h_tie;              // approx_error_pair: two (count, error sum) results
    DevBuf tie_scratch;        // ... and the second error sum's partials / results
    void* seg_staging = nullptr;   // page-locked staging (m3d_host_alloc) for segmentation's index lists when the caller's array is pageable
    size_t seg_staging_cap = 0;    // ... in uint64 entries
    PinBuf h_inc;              // m3d_cloud_score_shard: the sampler's pruning incumbent on its way to / from the device
    uint32_t pick_seq = 0;     // last PickFinal::seq handed out (completion words of one-GPU speculative chunks)
    const double* last_best_dev = nullptr;   // device address of the last fit's best minimal model (a slot's params or best_params)
    // m3d_cloud_create's upload / sort scratch (a one-shot call -- upload, fit, destroy -- otherwise spends more time
    // in hipMalloc / hipFree than in the fit; the cloud's own buffers come back through DevBuf's free list)
    DevBuf cc_stage, cc_cell, cc_start, cc_fill, cc_sums, cc_total, cc_bbox;
    DevBuf pick;               // BestPick: the device's prediction of the winning hypothesis (probability-1 fits)
    PinBuf h_pick;             // BestPickHost mirror (+ at byte 64: inlier total of a compaction started on the prediction)
    bool spec_compaction = false;   // RefineModel's compaction has already been queued on pick->params
    // Deferred RefineModel (m3d_segment_plane_iterative on one GPU): a round whose early compaction ran on the hypothesis
    // the replay then named does not wait for it -- the inlier count is the scoring pass's, the index list goes to the
    // caller's pinned array by itself -- and its GeneralFit is finished from the pinned sums while the NEXT round's records
    // are awaited (two slots of h_best / h_moments / the total word: the next round's compaction is queued before that)
    // segment_impl, rounds on a large part of the cloud: the inlier list (megabytes) is written to device memory
    // (idx_out_override, the cluster's slice of a per-call buffer) and shipped by the copy engine under the rounds that
    // follow, instead of being stored over the host link by the compaction kernel itself (which then runs at the link's
    // rate: 370 us for 2.5 M indices); the copy stream is waited for once, at the end of the call
    uint64_t* idx_out_override = nullptr;
    bool defer_copy_sync = false;
    bool no_prune_hint = false;     // segment_impl: the next fit's chunk will prune nothing (no lead pass, no keep masks)
    bool defer_refine = false;
    int refine_slot = 0;
    struct DeferredRefine {
        bool pending = false;
        int slot = 0;
        uint32_t ni = 0;
        double* params_out = nullptr;
    } deferred;
    // a tombstone pass that waits for the next fit's minimal_fit_k launch to ride in (cloud_remove_issue -> issue_chunk)
    PoisonJob pending_poison;
    bool poison_pending = false;
    uint64_t* poison_expected_at = nullptr, poison_pending_count = 0;   // m3d_cloud::Work::poison_expected, credited at launch
    DevBuf poison_total;            // launch_poison_plane_inliers' running count
    bool ev_compact_early = false;  // ... and ev_compact was recorded right behind it (a removal has been queued after it)
    bool spec_hit = false;          // ... and the replay named the same hypothesis (cloud_fit_locked)
    PinBuf h_sums;             // GeneralFit: per-workgroup moment partials + coordinate sums, written by the kernels
    DevBuf moment_partial;     // fused RefineModel: per-workgroup raw moments of the compaction's counting pass
    PinBuf h_moments;          // ... folded (scan_blocks_k), device-visible: kFusedMomentDoubles doubles
    hipEvent_t ev_pre_gate = nullptr;         // recorded on `stream` where a fit starts; pre_stream waits for it (pre_stream_gate)
    uint32_t compact_epoch = 0;               // launch counter of the compaction scratch (compact_scratch)
    bool compaction_fused = false;            // the compaction in flight carried the moments
    uint64_t* compaction_idx_host = nullptr;  // ... and wrote the index list to this page-locked destination as well
    // segmentation: asked by refine() right before it queues RefineModel's compaction -- given the inlier count the scoring
    // pass reported, where should the partition of the NON-inliers go (null: no partition in this pass)?
    const std::function<const m3d::PartitionOut*(int64_t)>* partition_hook = nullptr;
    PinBuf h_best;             // best minimal model of a fit on its way to the host (read after RefineModel's wait)
    PinBuf h_sync;             // stream_wait_spin's completion word
    PinBuf h_reg;              // registration: a chunk's pass flags / counts and sums on their way to the host (polled, not waited for)
    uint32_t sync_seq = 0;
    DevBuf surv_list;          // plane_bound_k's input: the hypotheses the keep kernels kept (its length: best_count word 6)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    PinBuf h_match;            // the matcher's cross-checked pairs (launch_mutual_pairs): count at byte 0, (src, dst) u32 pairs from byte 64
    std::vector<hipEvent_t> aux_events;   // aux_event_of: the matcher's upload / part / scan events
};

constexpr int kMaxLanes = 8;
// entry points' bodies on a lane the caller already holds (m3d_registration.cpp); arguments checked by the callers
// one side of a match: the caller's array (uploaded by the call), or descriptors already resident on the device (dev != null;
// max_abs: their largest |value|, NaN if any, < 0 when unknown)
struct MatchSide {
    const double* host;
    const double* dev;
    double max_abs;
};
int match_mutual_nn_on(DeviceCtx* ctx, const MatchSide& feat_src, size_t n_src, const MatchSide& feat_dst, size_t n_dst, int dim,
                       size_t* out_src, size_t* out_dst, size_t* k_out);
// largest |value| of n doubles on the device (NaN if any is NaN), with a round trip; the lane's stream
int device_max_abs(DeviceCtx* ctx, const double* dev, size_t n, double* out);
int registration_ransac_on(DeviceCtx* ctx, m3d_cloud* csrc, m3d_cloud* cdst, const double* src, size_t n_src,
                           const double* dst, size_t n_dst, const size_t* corr_src, const size_t* corr_dst, size_t m,
                           double threshold, int max_iter, double edge_length_threshold, double confidence,
                           const uint64_t* seed, double* T_out, m3d_reg_stats* stats,
                           m3d_reg** session_out = nullptr /* the finished session (its target grid with original indices kept) for
                                                             information_matrix_on; the caller then calls reg_session_release */);
int information_matrix_on(DeviceCtx* ctx, m3d_cloud* csrc, m3d_cloud* cdst, const double* dst, size_t n_dst,
                          double max_correspondence_distance, const double* T, double* info, uint64_t* n_correspondences,
                          m3d_reg* session = nullptr);
void reg_session_release(m3d_reg* q);   // ... of a session handed out through session_out (the caller holds the lane)

void dev_pool_trim(int device);
int logical_device_count();           // what m3d_device_count returns: max(physical devices, m3d_config.device_aliases)
int physical_device(int logical);     // logical ordinal -> HIP ordinal (logical % physical devices under aliases); -1 + last error if invalid
DeviceCtx* get_ctx(int device);  // lane 0 of the device; nullptr + last error when the device is unusable
DeviceCtx* get_lane(int device, int lane);
hipStream_t copy_stream_of(DeviceCtx* ctx);   // the lane's copy / pre streams, created by the first call that needs them
hipStream_t pre_stream_of(DeviceCtx* ctx);
hipEvent_t aux_event_of(DeviceCtx* ctx, size_t k);   // the lane's k-th spare event (no timing), created on first use
int lanes_held();                              // calls holding a lane right now (all devices)
int lane_count();                // m3d_config.lanes, clamped to [1, kMaxLanes]

// Holds a lane for the calling thread: locks its mutex and makes it the lane DevBuf::reserve takes blocks for.
// (hipSetDevice stays with the callers: they report its failure.)
class CtxLock {
public:
    explicit CtxLock(DeviceCtx* c);
    ~CtxLock();
    CtxLock(const CtxLock&) = delete;
    CtxLock& operator=(const CtxLock&) = delete;

private:
    DeviceCtx* ctx_;
    int prev_lane_, prev_dev_;
};
// A lane of `device` for one call: the calling thread's own lane when it is free (threads are dealt lanes in the order
// they first arrive, so a single-threaded program lives on lane 0 and its scratch stays warm), else the lowest free one,
// else it waits for its own.  prefer >= 0: that lane instead of the thread's own (the batch entry points' workers).
// ctx == nullptr (+ last error) when the device is unusable.
class LaneLock {
public:
    explicit LaneLock(int device, int prefer = -1);
    ~LaneLock();
    LaneLock(const LaneLock&) = delete;
    LaneLock& operator=(const LaneLock&) = delete;
    DeviceCtx* ctx = nullptr;

private:
    int prev_lane_ = 0, prev_dev_ = -1;
};

}  // namespace m3d

struct m3d_cloud {
    m3d::DeviceCtx* ctx = nullptr;
    m3d::DevBuf x, y, z, nx, ny, nz;
    uint32_t n = 0, n_pad = 0;
    bool has_normals = false;
    // Z-order sorted copy for the culled scoring path (m3d_cull_kernels.hip)
    m3d::DevBuf sx, sy, sz, boxes;
    m3d::DevBuf tile_f32;   // SortedView::tile_f32 (n_tiles x kTileF32Floats floats)
    m3d::DevBuf frames, frame_cum;   // SortedView::frames / frame_cum (m3d_bound.hip), built by the first long plane fit (ensure_plane_frames)
    bool frames_ready = false;
    bool frames_failed = false;      // their allocation failed once: this cloud's plane fits go without the histogram bound
    bool one_shot = false;           // created for ONE fit (one_shot_fit): set-up that pays off over several fits is skipped
    uint32_t n_sorted = 0, n_tiles = 0;
    double max_abs = __builtin_inf();   // largest |coordinate| of the finite points (SortedView::max_abs)
    double origin[3] = {0.0, 0.0, 0.0};   // centre of the bounding box of the finite points (SortedView::origin)
    double radius = __builtin_inf();    // largest |coordinate - origin| (inf: unknown -> fp64 box tests)
    double bb[6] = {0, 0, 0, 0, 0, 0};  // bounding box of the points with three finite coordinates as created (lo, hi)
    bool bb_known = false;              // ... valid (there is such a point)
    // m3d_cloud_create's own clock (m3d_bench_cloud_setup_ms): total, and with m3d_config.kernel_timing the phases --
    // host-to-device copies + transposes, bounding box (incl. its round trip), Hilbert sort, tile boxes
    double setup_ms[5] = {0, 0, 0, 0, 0};
    // In-place shrinking (m3d_cloud_remove_inliers = SelectByIndex(inliers, invert), the tail of a
    // SegmentPlaneIterative round).  x/y/z above always hold the cloud AS CREATED (n0 points): index lists
    // and GeneralFit gathers refer to it through `orig`.  Once shrunk, n / n_pad / n_sorted / n_tiles and
    // view() / sorted() describe the working cloud in the ping-pong buffers below.
    struct Work {
        bool active = false;
        m3d::DevBuf bx[2], by[2], bz[2], bo[2], sbx[2], sby[2], sbz[2], sboxes, stile_f32;
        m3d::CloudView cur;
        m3d::SortedView scur;
        const uint32_t* cur_orig = nullptr;   // working index -> index in the cloud as created
        int pp = 0, spp = 0;                  // buffer sets that RECEIVE the next compaction
        bool cur_is_v0 = true;
        bool partition_done = false;   // the removal in flight found its creation-order partition already written (PartitionOut)
        int totals_slot = 0;           // which of the two pinned slots receives the totals of the removal in flight
        // a removal finished from a known count, its totals still to be checked (cloud_remove_check_pending)
        bool pending = false, pending_partition_done = false;
        int pending_slot = 0;
        uint32_t pending_new_n = 0, pending_new_sorted = 0;
        // tombstones (segmentation rounds in the clutter): the sorted copy's inliers are killed in place (x = NaN,
        // launch_poison_plane_inliers) instead of partitioned away; n_sorted stays the copy's PHYSICAL size, sorted_dead of
        // them are dead; a real compaction (mode 3 drops them) resets the count
        bool tombstones = false;        // the owner allows it (m3d_segment_plane_iterative's private cloud)
        uint32_t sorted_dead = 0;
        bool issue_poison = false;      // the removal in flight is a kill, not a partition (its total = points killed)
        bool pending_poison = false;
        uint32_t last_removed = 0;      // size of the previous removal (the speculative issue's only hint)
        uint64_t poison_expected = 0;   // points the kills of this cloud's lifetime should have removed
    } work;
    uint32_t n0 = 0, n_pad0 = 0, n_tiles0 = 0;
    m3d::CloudView view() const;        // working cloud
    m3d::CloudView base_view() const;   // cloud as created
    m3d::SortedView sorted() const;
    const uint32_t* orig() const { return work.active ? work.cur_orig : nullptr; }
};

// m3d_cloud_create with the Hilbert-sorted copy optional (m3d_device.cpp)
extern "C" m3d_cloud* m3d_cloud_create_impl(const double* xyz, const double* normals, size_t n, int device, int with_sorted_copy);
// ... on a lane the caller already holds (CtxLock / LaneLock): the registration session's two clouds share its lane
m3d_cloud* m3d_cloud_create_on(m3d::DeviceCtx* ctx, const double* xyz, const double* normals, size_t n, int with_sorted_copy);
void m3d_cloud_destroy_on(m3d_cloud* c);
