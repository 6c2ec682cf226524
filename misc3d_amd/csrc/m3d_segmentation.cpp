// m3d_segmentation.cpp -- segmentation::SegmentPlaneIterative (src/iterative_plane_segmentation.cpp:8-39) on a device-resident
// working cloud: rounds of fit_plane + removal of the inliers (pcd_copy->SelectByIndex(inliers, true), :33) as stable partitions
// of both copies or, in the clutter, tombstones in the Hilbert-sorted copy; deferred RefineModel; the one-GPU and sharded loops.
#include "m3d_driver_internal.hpp"

#pragma clang fp contract(off)

namespace m3d {


// pcd_copy = pcd_copy->SelectByIndex(inliers, true) (iterative_plane_segmentation.cpp:33) on the resident
// cloud: a stable partition keeps the non-inliers of `model_dev` (distance >= thr, or not comparable) in
// both copies -- original order (+ the map back to the cloud as created) and Hilbert-sorted (tile boxes
// recomputed).  The first call allocates the ping-pong buffers.
// Two halves: cloud_remove_issue enqueues the partitions and the copy of their totals (no host wait: the
// segmentation loop issues it behind RefineModel's kernels and lets RefineModel's own wait cover both),
// cloud_remove_finish -- after the stream has been waited for -- checks the totals and switches the cloud over.
// Without the finish nothing has changed for the caller (the partitions went into the spare buffer set).
constexpr size_t kRemoveTotalsOffset = 160;   // bytes into h_small (refine() uses 0..127 and 192..255)
// cloud_remove_prepare: buffers + the destination of the next partition of the cloud in creation order (what mode 2 of
// launch_compact, or a mode-0 compaction with a PartitionOut, writes)
static int cloud_remove_prepare(m3d_cloud* c, PartitionOut* out) {
    DeviceCtx* ctx = c->ctx;
    m3d_cloud::Work& w = c->work;
    if (c->has_normals) return fail(M3D_ERR_INVALID_ARG, "removing points from a cloud with normals is not supported");
    if (!w.active) {
        c->n0 = c->n;
        c->n_pad0 = c->n_pad;
        c->n_tiles0 = c->n_tiles;
        const size_t bytes = sizeof(double) * (size_t)c->n_pad;
        const uint32_t scap = c->n_tiles * kTilePoints;
        bool ok = true;
        for (int k = 0; k < 2 && ok; ++k)
            ok = w.bx[k].reserve(bytes) && w.by[k].reserve(bytes) && w.bz[k].reserve(bytes) &&
                 w.bo[k].reserve(sizeof(uint32_t) * (size_t)c->n_pad) && w.sbx[k].reserve(sizeof(double) * scap) &&
                 w.sby[k].reserve(sizeof(double) * scap) && w.sbz[k].reserve(sizeof(double) * scap);
        ok = ok && w.sboxes.reserve(sizeof(double) * kBoxStride * c->n_tiles) &&
             w.stile_f32.reserve(sizeof(float) * kTileF32Floats * (size_t)c->n_tiles);
        if (!ok) return M3D_ERR_DEVICE;
        launch_iota(w.bo[0].as<uint32_t>(), c->n, ctx->stream);
        w.cur = c->base_view();   // round 0 reads the uploaded cloud directly
        w.scur = c->sorted();
        w.cur_orig = w.bo[0].as<uint32_t>();
        w.pp = w.spp = 0;
        w.cur_is_v0 = true;
        w.active = true;
    }
    const int dst = w.cur_is_v0 ? 1 : w.pp;
    out->ox = w.bx[dst].as<double>();
    out->oy = w.by[dst].as<double>();
    out->oz = w.bz[dst].as<double>();
    out->oorig = w.bo[dst].as<uint32_t>();
    out->n_pad_cap = c->n_pad0;
    return M3D_OK;
}
// partition_done: the partition in creation order has been written by RefineModel's own compaction (PartitionOut)
// same_slot: a second issue of the SAME round (the first one, queued on the device's early pick, named another model): the
// totals go where the first one's went -- the other slot still belongs to the previous round's deferred check
// A removal of `removed` points may kill them in place in the sorted copy instead of partitioning it when it is a sliver
// (a sixteenth of the live points) and the dead stay below an eighth of the copy
static bool poison_fits(const m3d_cloud* c, uint64_t removed) {
    const m3d_cloud::Work& w = c->work;
    if (!w.tombstones || !w.active || c->has_normals || !config().sorted_tombstones) return false;
    const uint64_t alive = c->n_sorted - w.sorted_dead;
    return removed * 16 <= alive && ((uint64_t)w.sorted_dead + removed) * 8 <= c->n_sorted;
}
// expected_removed >= 0: the size of this removal is known (the scoring pass counted the inliers): planes may then be
// removed from the sorted copy by tombstones (poison_fits)
static int cloud_remove_issue(m3d_cloud* c, int kind, double thr, const double* model_dev, bool partition_done = false,
                              bool same_slot = false, int64_t expected_removed = -1) {
    DeviceCtx* ctx = c->ctx;
    m3d_cloud::Work& w = c->work;
    PartitionOut po;
    const int rp = cloud_remove_prepare(c, &po);
    if (rp != M3D_OK) return rp;
    const CloudView cur = w.cur;
    const uint32_t nb = (cur.n + kCompactTile - 1) / kCompactTile;
    const uint32_t snb = (c->n_sorted + kCompactTile - 1) / kCompactTile;
    const uint32_t scap = c->n_tiles0 * kTilePoints;
    CompactScratch scratch;
    if (const int rc = compact_scratch(ctx, std::max(nb, snb), &scratch); rc != M3D_OK) return rc;
    RESERVE(ctx->total, 16);
    RESERVE(ctx->h_small, 256);
    w.partition_done = partition_done;
    // the totals go to pinned host memory from the compaction kernels' own tails (two slots: a deferred check reads the
    // previous removal's totals after the next one has been queued) -- a copy command per round less
    if (!same_slot) w.totals_slot ^= 1;
    uint32_t* h_totals = reinterpret_cast<uint32_t*>(ctx->h_small.as<uint8_t>() + kRemoveTotalsOffset + 16 * w.totals_slot);
    if (!partition_done)
        launch_compact(kind, cur, model_dev, thr, 2, w.cur_orig, nullptr, nullptr, po.ox, po.oy, po.oz, po.oorig, po.n_pad_cap,
                       scratch, ctx->total.as<uint32_t>(), ctx->stream, nullptr, nullptr, nullptr,
                       nullptr, h_totals);
    w.issue_poison = kind == M3D_PLANE && expected_removed >= 0 && poison_fits(c, (uint64_t)expected_removed);
    if (w.issue_poison) {
        // (the kills add up in ctx->poison_total, cleared by the owner of the cloud -- segment_impl -- which compares the sum
        // with the inlier lists at the end; a real compaction in between checks the live count it leaves)
        RESERVE(ctx->poison_total, 16);
        // A round that does not wait for its RefineModel (DeviceCtx::deferred) goes straight on to the next fit: its kill
        // rides in that fit's minimal_fit_k launch.  The model is then read from the device's pick record, which stays put
        // until the next fit's records are folded (the winner's slot of the parameter array is rewritten by that launch).
        const bool ride = ctx->defer_refine && ctx->spec_hit && !ctx->poison_pending;
        const PoisonJob job = make_poison_job(w.scur, ride ? ctx->pick.as<BestPick>()->params : model_dev, thr,
                                              ctx->poison_total.as<uint32_t>());
        if (ride) {
            ctx->pending_poison = job;
            ctx->poison_pending = true;
            ctx->poison_expected_at = &w.poison_expected;
            ctx->poison_pending_count = (uint64_t)expected_removed;
        } else {
            launch_poison_plane_inliers(job, ctx->stream);
            w.poison_expected += (uint64_t)expected_removed;
        }
        HIPCHK(hipGetLastError());
        return M3D_OK;
    }
    // the same stable partition on the sorted copy (every inlier is a finite point; the copy's dead points go too), then
    // fresh tile boxes
    CloudView sview;
    sview.x = w.scur.x;
    sview.y = w.scur.y;
    sview.z = w.scur.z;
    sview.nx = sview.ny = sview.nz = nullptr;
    sview.n = c->n_sorted;
    sview.n_pad = w.scur.n_tiles * kTilePoints;
    if (const int rc = compact_scratch(ctx, std::max(nb, snb), &scratch); rc != M3D_OK) return rc;   // (a launch of its own: the next epoch)
    launch_compact(kind, sview, model_dev, thr, 3, nullptr, nullptr, nullptr, w.sbx[w.spp].as<double>(),
                   w.sby[w.spp].as<double>(), w.sbz[w.spp].as<double>(), nullptr, scap,
                   scratch, ctx->total.as<uint32_t>() + 1, ctx->stream, nullptr, nullptr, nullptr,
                   nullptr, h_totals + 1);
    HIPCHK(hipGetLastError());
    return M3D_OK;
}

// known_removed == null: the stream has been waited for, the totals are read and checked now.
// known_removed != null: the caller knows how many points the removal drops (the inlier count RefineModel reported) and has
// NOT waited for the removal's kernels: the cloud is switched over from that count, and the totals are checked by
// cloud_remove_check_pending once the stream is known to have passed them.
static int cloud_remove_finish(m3d_cloud* c, size_t* n_removed, const size_t* known_removed = nullptr) {
    DeviceCtx* ctx = c->ctx;
    m3d_cloud::Work& w = c->work;
    const CloudView cur = w.cur;
    const int dst = w.cur_is_v0 ? 1 : w.pp;
    uint32_t new_n, new_sorted;
    const uint32_t alive = c->n_sorted - w.sorted_dead;   // live points of the sorted copy
    if (known_removed) {
        if (*known_removed > cur.n || *known_removed > alive) return fail(M3D_ERR_INTERNAL, "more inliers than points");
        new_n = cur.n - (uint32_t)*known_removed;
        new_sorted = alive - (uint32_t)*known_removed;
        w.pending = true;
        w.pending_slot = w.totals_slot;
        w.pending_partition_done = w.partition_done;
        w.pending_new_n = new_n;
        w.pending_new_sorted = new_sorted;
        w.pending_poison = w.issue_poison;   // (a kill has no total of its own: cloud_remove_check_pending skips it)
    } else {
        if (w.issue_poison) return fail(M3D_ERR_INTERNAL, "a removal by tombstones needs its size");
        uint32_t h[2];
        std::memcpy(h, ctx->h_small.as<uint8_t>() + kRemoveTotalsOffset + 16 * w.totals_slot, sizeof(h));
        new_sorted = h[1];
        // (a partition written by RefineModel's compaction has no total of its own: the sorted copy's removal count stands
        // in, and the caller checks it against the length of the inlier list)
        new_n = w.partition_done ? (new_sorted <= alive && alive - new_sorted <= cur.n
                                        ? cur.n - (alive - new_sorted) : 0xFFFFFFFFu)
                                 : h[0];
    }
    if (new_n > cur.n || new_sorted > alive || cur.n - new_n != alive - new_sorted)
        return fail(M3D_ERR_INTERNAL, "the two copies of the cloud disagree on the removed points");
    if (n_removed) *n_removed = cur.n - new_n;
    w.last_removed = cur.n - new_n;
    if (w.issue_poison) {
        // killed in place: same arrays, same tiles, same (now slightly generous) boxes
        w.sorted_dead += cur.n - new_n;
        w.scur.has_dead = true;
    } else {
        w.scur.has_dead = false;
        w.scur.x = w.sbx[w.spp].as<double>();
        w.scur.y = w.sby[w.spp].as<double>();
        w.scur.z = w.sbz[w.spp].as<double>();
        w.scur.boxes = w.sboxes.as<double>();
        w.scur.tile_f32 = w.stile_f32.as<float>();
        w.scur.frames = nullptr;   // (new tiles: the frames of the copy as created do not describe them)
        w.scur.frame_cum = nullptr;
        w.scur.n_tiles = std::max<uint32_t>(1, (new_sorted + kTilePoints - 1) / kTilePoints);
        // compact_write_k pads to a multiple of 2048 (capped at the buffer size): whole tiles are NaN-clean
        launch_tile_boxes(w.scur, w.sboxes.as<double>(), ctx->stream);
        w.spp ^= 1;
        w.sorted_dead = 0;
        c->n_sorted = new_sorted;
        c->n_tiles = w.scur.n_tiles;
    }
    w.cur.x = w.bx[dst].as<double>();
    w.cur.y = w.by[dst].as<double>();
    w.cur.z = w.bz[dst].as<double>();
    w.cur.nx = w.cur.ny = w.cur.nz = nullptr;
    w.cur.n = new_n;
    w.cur.n_pad = std::max<uint32_t>(round_up(new_n, kScoreTile), kScoreTile);
    w.cur_orig = w.bo[dst].as<uint32_t>();
    if (w.cur_is_v0) {
        w.cur_is_v0 = false;
        w.pp = 0;  // bo[0] (iota) is free again: the next compaction goes to set 0
    } else {
        w.pp = dst ^ 1;
    }
    c->n = new_n;
    c->n_pad = w.cur.n_pad;
    return M3D_OK;
}

// the totals of a removal finished from a known count (cloud_remove_finish), once the stream has passed their copy
static int cloud_remove_check_pending(m3d_cloud* c) {
    m3d_cloud::Work& w = c->work;
    if (!w.pending) return M3D_OK;
    w.pending = false;
    uint32_t h[2];
    std::memcpy(h, c->ctx->h_small.as<uint8_t>() + kRemoveTotalsOffset + 16 * w.pending_slot, sizeof(h));
    if ((!w.pending_poison && h[1] != w.pending_new_sorted) || (!w.pending_partition_done && h[0] != w.pending_new_n))
        return fail(M3D_ERR_INTERNAL, "removed points and inlier list disagree");
    return M3D_OK;
}

static int cloud_remove_locked(m3d_cloud* c, int kind, double thr, const double* model_dev, size_t* n_removed) {
    const int rc = cloud_remove_issue(c, kind, thr, model_dev);
    if (rc != M3D_OK) return rc;
    HIPCHK(hipStreamSynchronize(c->ctx->stream));
    return cloud_remove_finish(c, n_removed);
}

}  // namespace m3d

using namespace m3d;

extern "C" {

int m3d_cloud_remove_inliers(m3d_cloud* c, int kind, double threshold, const double* model, size_t* n_removed) {
    if (!c || kind < 0 || kind > 2 || !model) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    DeviceCtx* ctx = c->ctx;
    CtxLock lock(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    RESERVE(ctx->small, 256);
    double tmp[kModelStride] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::memcpy(tmp, model, sizeof(double) * num_params(kind));
    HIPCHK(hipMemcpyAsync(ctx->small.p, tmp, sizeof(tmp), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return cloud_remove_locked(c, kind, threshold, ctx->small.as<double>(), n_removed);
}

}  // extern "C"
thread_local double m3d::g_seg_ms[6] = {0, 0, 0, 0, 0, 0};
extern "C" {


}  // extern "C"
namespace m3d {
// SegmentPlaneIterative, src/iterative_plane_segmentation.cpp:8-39
int segment_impl(const double* xyz, size_t n, double threshold, int max_iteration, double min_ratio,
                        const uint64_t* seed, int device, m3d_comm* comm, size_t max_clusters, double* planes,
                        size_t* cluster_offsets, size_t* cluster_indices, size_t* n_clusters,
                        double* cluster_points /* n x 3: the xyz of cluster_indices[i] at 3 i (may be null) */) {
    if (!planes || !cluster_offsets || !cluster_indices || !n_clusters || (!xyz && n))
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    *n_clusters = 0;
    cluster_offsets[0] = 0;
    if (n < 3) {  // :13-17: LogWarning + empty result
        set_error("Point cloud size has less than 3.");
        return M3D_FALSE;
    }
    const double t_call = now_ms();
    m3d_cloud* c0 = m3d_cloud_create(xyz, nullptr, n, device);
    if (!c0) return M3D_ERR_DEVICE;
    DeviceCtx* ctx = c0->ctx;
    c0->work.tombstones = true;   // (a private cloud, planes only: the sorted copy may carry dead points between rounds)
    bool poison_ready = false;
    int rc = M3D_OK;
    const double t_created = now_ms();
    double t_rounds = t_created, t_copied = t_created;
    size_t rounds_done = 0, big_rounds = 0;
    double t_big = 0;
    {
        CtxLock lock(ctx);
        // The cross-round state of a segmentation (a tombstone pass waiting to ride in the next fit, a RefineModel finished one
        // round late, lists leaving through the copy engine) lives on the device context and points into THIS call's cloud and
        // the caller's buffers: however the block is left, the context goes back to idle and keeps none of those pointers
        // (the explicit resets below stay where their order matters).
        struct StateGuard {
            DeviceCtx* ctx;
            ~StateGuard() {
                ctx->defer_refine = false;
                ctx->defer_copy_sync = false;
                ctx->idx_out_override = nullptr;
                ctx->poison_pending = false;
                ctx->poison_expected_at = nullptr;
                ctx->poison_pending_count = 0;
                ctx->deferred.pending = false;
                ctx->deferred.params_out = nullptr;
                ctx->spec_hit = false;
            }
        } state_guard{ctx};
        uint64_t seed0 = 0;
        rc = agree_seed(comm, seed, ctx->stream, &seed0);
        poison_ready = ctx->poison_total.reserve(16) && hipMemsetAsync(ctx->poison_total.p, 0, 16, ctx->stream) == hipSuccess;
        if (!poison_ready) c0->work.tombstones = false;
        ctx->deferred.pending = false;
        ctx->poison_pending = false;
        ctx->defer_refine = !comm && config().speculative_refine != 0;   // (one GPU: DeviceCtx::deferred)
        // A pageable destination is reached through staged copies, a blocking one per round, into pages that fault on first
        // touch (10 M points: 43 ms against 37): the rounds write into a page-locked staging array the device context keeps
        // -- the compaction kernels store the index lists straight into it -- and the lists are copied over at the end.
        size_t* idx_out = cluster_indices;
        constexpr size_t kStagingMax = (size_t)1 << 25;   // entries (256 MB); larger clouds keep the direct path
        if (n <= kStagingMax && !is_library_pinned(cluster_indices, sizeof(size_t) * n)) {
            if (ctx->seg_staging_cap < n) {
                if (ctx->seg_staging) m3d_host_free(ctx->seg_staging);
                ctx->seg_staging_cap = 0;
                ctx->seg_staging = m3d_host_alloc(sizeof(size_t) * (n + n / 4));
                if (ctx->seg_staging) ctx->seg_staging_cap = n + n / 4;
            }
            if (ctx->seg_staging) idx_out = static_cast<size_t*>(ctx->seg_staging);
        }
        // rounds on a large part of the cloud: lists of megabytes leave through the copy engine (DeviceCtx::idx_out_override)
        DevBuf seg_idx_dev;
        const bool lists_by_copy_engine = rc == M3D_OK && !comm && is_library_pinned(idx_out, sizeof(size_t) * n) &&
                                          n >= ((size_t)1 << 20) && seg_idx_dev.reserve(sizeof(uint64_t) * n);
        ctx->defer_copy_sync = lists_by_copy_engine;
        size_t count = 0, k = 0;
        size_t iterations_hint = 0;   // iterations the previous round took: sizes this round's second chunk up front
        const size_t target = (size_t)((1 - min_ratio) * (double)n);  // :28
        while (rc == M3D_OK && count < target && k < max_clusters) {
            if (c0->n < 3) {  // the reference's FitModel would throw here (ransac.h:510-513)
                rc = 2;
                break;
            }
            // ransac.FitModel(threshold, plane, inliers), :29-31; probability stays at the RANSAC default
            // (ransac.h:462); inlier indices refer to the cloud as created (c0->orig()).  The return value
            // (GeneralFit) is ignored by the reference.
            double* plane = planes + 4 * k;   // (written by the fit, or -- a deferred RefineModel -- while the next round runs)
            plane[0] = plane[1] = plane[2] = plane[3] = 0.0;
            size_t ni = 0;
            const size_t off = cluster_offsets[k];
            const double t_round0 = now_ms();
            const bool big_round = c0->n > (uint32_t)(n / 8);
            // The removal of the round's inliers (:33) is queued behind RefineModel's kernels, before RefineModel
            // waits for them: the pre-refinement model is already on the device and the inlier count is known
            // from the scoring pass, so the round costs one host wait less.  Not on the last round.
            bool removal_issued = false, partition_fused = false, spec_removal = false;
            PartitionOut part_out;
            // RefineModel's compaction evaluates the very flags the removal needs: it writes the partition of the cloud
            // in creation order as well (one count, one scan and one write launch less per round)
            const std::function<const PartitionOut*(int64_t)> partition_hook = [&](int64_t expected_ni) -> const PartitionOut* {
                // -2: asked before the inlier count is known (compaction queued on the device's own pick, run_ransac): the
                // partition goes to the spare buffers and is simply not used should this turn out to be the last round
                if (expected_ni == -3) {   // the speculative compaction has been queued: the sorted copy's removal behind it
                    // (a round that will kill its inliers in place cannot do so on a guess: the kill is queued once the
                    // replay has confirmed the pick -- still in front of the device, which is busy with the compaction)
                    if (partition_fused && !spec_removal && !poison_fits(c0, c0->work.last_removed) &&
                        cloud_remove_issue(c0, M3D_PLANE, threshold, ctx->pick.as<BestPick>()->params, true) == M3D_OK)
                        spec_removal = true;
                    return nullptr;
                }
                if (expected_ni == -2) {
                    if (k + 1 >= max_clusters) return nullptr;
                } else if (expected_ni <= 0 || count + (size_t)expected_ni >= target || k + 1 >= max_clusters) {
                    return nullptr;
                }
                if (cloud_remove_prepare(c0, &part_out) != M3D_OK) return nullptr;
                partition_fused = true;
                return &part_out;
            };
            const std::function<int(int64_t)> issue_removal = [&](int64_t expected_ni) -> int {
                if (expected_ni <= 0 || count + (size_t)expected_ni >= target || k + 1 >= max_clusters) return M3D_OK;
                removal_issued = true;
                if (spec_removal && ctx->spec_hit) return M3D_OK;   // (queued on the device's pick, which the replay confirmed)
                return cloud_remove_issue(c0, M3D_PLANE, threshold, ctx->last_best_dev, partition_fused, /*same_slot=*/spec_removal,
                                          expected_ni);
            };
            ctx->partition_hook = &partition_hook;
            ctx->idx_out_override = lists_by_copy_engine && big_round ? seg_idx_dev.as<uint64_t>() + off : nullptr;
            ctx->no_prune_hint = !comm && k > 0 && (uint64_t)c0->work.last_removed * 32 < c0->n;   // (the previous round's plane: < 3 % of the cloud)
            rc = cloud_fit_locked(c0, M3D_PLANE, threshold, (size_t)max_iteration, 0.9999, seed0 + k, plane,
                                  idx_out + off, &ni, nullptr, &issue_removal, &iterations_hint, comm);
            ctx->partition_hook = nullptr;
            ctx->idx_out_override = nullptr;
            ctx->no_prune_hint = false;
            if (rc < 0) break;
            rc = cloud_remove_check_pending(c0);   // (the previous round's removal: this round's wait lay behind it)
            if (rc != M3D_OK) break;
            if (ni == 0) {  // the reference would loop forever (:29,:35)
                rc = 2;
                break;
            }
            cluster_offsets[k + 1] = off + ni;
            count += ni;
            k++;
            rounds_done++;
            if (big_round) {
                big_rounds++;
                t_big += now_ms() - t_round0;
            }
            if (count >= target || k >= max_clusters) break;
            // pcd_copy = pcd_copy->SelectByIndex(inliers, true), :33 -- inliers of the PRE-refinement model,
            // still on the device (ctx->last_best_dev)
            size_t removed = 0;
            if (partition_fused && !removal_issued) {   // (the two hooks take the same decision from the same count)
                rc = fail(M3D_ERR_INTERNAL, "partition written without the removal being queued");
                break;
            }
            // (a removal queued by the hook is NOT waited for: its size is the inlier count; its totals are checked after
            // the next round's wait, which the stream reaches behind them)
            rc = removal_issued ? cloud_remove_finish(c0, &removed, &ni)
                                : cloud_remove_locked(c0, M3D_PLANE, threshold, ctx->last_best_dev, &removed);
            if (rc != M3D_OK) break;
            if (removed != ni) {
                rc = fail(M3D_ERR_INTERNAL, "removed points and inlier list disagree");
                break;
            }
        }
        *n_clusters = k;
        // the points killed in place in the sorted copy over the whole call against the inlier lists of those rounds
        uint32_t killed = 0;
        if (poison_ready && c0->work.poison_expected)
            (void)hipMemcpyAsync(&killed, ctx->poison_total.p, sizeof(killed), hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        if (lists_by_copy_engine) (void)hipStreamSynchronize(copy_stream_of(ctx));   // (the big rounds' lists)
        ctx->defer_copy_sync = false;
        seg_idx_dev.release();
        ctx->defer_refine = false;
        ctx->poison_pending = false;   // (the last round's kill has nobody left to serve)
        if (rc == M3D_OK || rc == 2) {   // the last round's RefineModel
            const int fr = finalize_deferred_refine(ctx);
            if (fr != M3D_OK) rc = fr;
        }
        ctx->deferred.pending = false;
        if (rc == M3D_OK) rc = cloud_remove_check_pending(c0);
        if ((rc == M3D_OK || rc == 2) && poison_ready && (uint64_t)killed != c0->work.poison_expected)
            rc = fail(M3D_ERR_INTERNAL, "the sorted copy's tombstones and the inlier lists disagree");
        t_rounds = now_ms();
        // the clusters' points (SelectByIndex, :32): gathered from the resident cloud as created, one copy back
        if ((rc == M3D_OK || rc == 2) && cluster_points && k && cluster_offsets[k]) {
            const size_t total = cluster_offsets[k];
            DevBuf d_idx, d_out;
            bool ok = d_idx.reserve(sizeof(uint64_t) * total) && d_out.reserve(sizeof(double) * 3 * total);
            ok = ok && hipMemcpyAsync(d_idx.p, idx_out, sizeof(uint64_t) * total, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
            if (ok) {
                launch_gather_points(c0->base_view(), d_idx.as<uint64_t>(), total, d_out.as<double>(), ctx->stream);
                ok = hipMemcpyAsync(cluster_points, d_out.p, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                     hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
            }
            d_idx.release();
            d_out.release();
            if (!ok) rc = fail(M3D_ERR_DEVICE, "gathering the clusters' points failed");
        }
        if (idx_out != cluster_indices && k) std::memcpy(cluster_indices, idx_out, sizeof(size_t) * cluster_offsets[k]);
        t_copied = now_ms();
    }
    m3d_cloud_destroy(c0);
    g_seg_ms[0] = now_ms() - t_call;
    g_seg_ms[1] = t_created - t_call;
    g_seg_ms[2] = t_rounds - t_created;
    g_seg_ms[3] = t_copied - t_rounds;
    g_seg_ms[4] = t_big;
    g_seg_ms[5] = (double)big_rounds + 1e-4 * (double)rounds_done;
    if (rc == 2) return 2;
    return rc == M3D_OK ? M3D_OK : rc;
}

}  // namespace m3d
extern "C" {

int m3d_segment_plane_iterative(const double* xyz, size_t n, double threshold, int max_iteration,
                                double min_ratio, const uint64_t* seed, int device, size_t max_clusters,
                                double* planes, size_t* cluster_offsets, size_t* cluster_indices,
                                size_t* n_clusters) {
    return segment_impl(xyz, n, threshold, max_iteration, min_ratio, seed, device, nullptr, max_clusters, planes,
                        cluster_offsets, cluster_indices, n_clusters);
}

int m3d_segment_plane_iterative_clouds(const double* xyz, size_t n, double threshold, int max_iteration,
                                       double min_ratio, const uint64_t* seed, int device, size_t max_clusters,
                                       double* planes, size_t* cluster_offsets, size_t* cluster_indices,
                                       double* cluster_points, size_t* n_clusters) {
    return segment_impl(xyz, n, threshold, max_iteration, min_ratio, seed, device, nullptr, max_clusters, planes,
                        cluster_offsets, cluster_indices, n_clusters, cluster_points);
}

int m3d_segment_plane_iterative_sharded(const double* xyz, size_t n, double threshold, int max_iteration,
                                        double min_ratio, const uint64_t* seed, int device, m3d_comm* comm,
                                        size_t max_clusters, double* planes, size_t* cluster_offsets,
                                        size_t* cluster_indices, size_t* n_clusters) {
    if (comm && comm->transport == m3d_comm::kRccl && comm->device != device)
        return fail(M3D_ERR_INVALID_ARG, "the RCCL communicator lives on another device");
    return segment_impl(xyz, n, threshold, max_iteration, min_ratio, seed, device, comm, max_clusters, planes,
                        cluster_offsets, cluster_indices, n_clusters);
}


}  // extern "C"
