"""Multi-GPU RANSAC: hypotheses sharded over ranks, one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Hypotheses are the independent unit of the reference loop (ransac.h:571-590 has no cross-iteration
dependency except the best-model reduction, ransac.h:592-613).  Every rank holds a replica of the
cloud in HBM (10 M points = 240 MB) and a sampler seeded identically, so all ranks see ONE sample
stream.  A window of the stream is cut into slices; slice j is scored by rank j % world through the
C ABI (m3d_cloud_score_shard: the host draws the other ranks' slices while the GPU scores its own).
One all-gather per window exchanges the per-hypothesis (valid, inlier count) records -- 4 bytes per
hypothesis, latency-bound on xGMI.  Every rank then replays the sequential best-update / adaptive-
stop rule in index order (m3d_replay_chunk), so the chosen hypothesis, the iteration count and the
inlier set are identical to the single-GPU run and independent of the number of GPUs.

The scorer is injected so the N>1 control path can be tested on CPU with gloo (the tests plug the
oracle in as the scorer; the product scorer is ``capi.Cloud``).  Scorer interface:
    make_sampler(kind, seed) -> object with .table(n_hyp) -> (n_hyp, m) uint32
    score_shard(sampler, thr, begin, end, slice, world, rank) -> (valid u8, counts u32) of this rank's share
    score_range(kind, thr, table, begin, end) -> (valid, models, counts)        (single hypotheses, rare)
    exact_error(kind, thr, model) -> (count, error);  refine(kind, thr, params, copy) -> (ret, params, inliers)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi



@dataclass
class ShardedFit:
    ret: int
    params: np.ndarray
    inliers: np.ndarray
    best_index: int
    count: int
    iterations: int
    fitness: float
    hypotheses_scored: int
    collectives: int
    best_model: np.ndarray = None     # the pre-refinement (minimal) model RefineModel's inliers belong to


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group), True
    return 1, 0, False


_AG_CACHE = {}


def _all_gather_records(rec, group, device):
    """all_gather of equal-length int32 record arrays -> (world, len) uint32 array.  On a GPU group the staging
    tensors (pinned host in / out, device in / out) are kept per (length, world, device): a fit exchanges a few
    tens of KB once or twice, so allocations and pageable copies would be most of the cost."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = len(rec)
    if device is None:
        t = torch.from_numpy(rec.view(np.int32))
        out = torch.empty(world * n, dtype=torch.int32)
        dist.all_gather_into_tensor(out, t, group=group)
        return out.numpy().view(np.uint32).reshape(world, n)
    key = (n, world, str(device))
    bufs = _AG_CACHE.get(key)
    if bufs is None:
        if len(_AG_CACHE) > 16:
            _AG_CACHE.clear()
        bufs = (torch.empty(n, dtype=torch.int32).pin_memory(), torch.empty(n, dtype=torch.int32, device=device),
                torch.empty(world * n, dtype=torch.int32, device=device),
                torch.empty(world * n, dtype=torch.int32).pin_memory())
        _AG_CACHE[key] = bufs
    src_pin, src_dev, out_dev, out_pin = bufs
    src_pin.numpy()[:] = rec.view(np.int32)
    src_dev.copy_(src_pin, non_blocking=True)
    dist.all_gather_into_tensor(out_dev, src_dev, group=group)
    out_pin.copy_(out_dev, non_blocking=True)
    torch.cuda.current_stream(device).synchronize()
    return out_pin.numpy().view(np.uint32).reshape(world, n)


def fit_sharded(scorer, n_points, kind, threshold=0.01, max_iteration=1000, probability=0.9999, seed=0,
                group=None, device=None, want_inliers=True, copy=True, slice_size=None) -> ShardedFit:
    """RANSAC::FitModel with the hypothesis loop sharded over the ranks of `group`."""
    world, rank, have_pg = _world(group)
    m = capi.MINIMAL_SAMPLE[kind]
    if probability <= 0 or probability > 1:
        raise capi.M3DError(capi.ERR_PROBABILITY, "Probability must be > 0 or <= 1.0")
    if n_points < m:
        raise capi.M3DError(capi.ERR_TOO_FEW_POINTS, "Can not fit model due to lack of points")
    H = int(max_iteration)
    sampler = scorer.make_sampler(kind, seed)

    st = capi.ReplayState()
    capi.lib().m3d_replay_init(C.byref(st))
    # window schedule: everything at once when the loop cannot stop early (probability == 1), else
    # geometric windows so that little is scored beyond the adaptive stop
    window = H if probability >= 1.0 else 128 * world
    begin = 0
    scored = 0
    collectives = 0
    best_model = None

    def model_of(i):
        i = int(i)
        if best_model is not None and i == best_model[0]:
            return best_model[1]
        if hasattr(scorer, "minimal_model"):          # host-side MinimalFit of that one sample (no launch)
            return scorer.minimal_model(kind, threshold, sampler.table(i + 1)[i])
        _, mod, _ = scorer.score_range(kind, threshold, sampler.table(i + 1), i, i + 1)   # one-row launch (rare)
        return mod[0]

    def rmse_of(_user, i):
        cnt, err = scorer.exact_error(kind, threshold, model_of(i))
        return 1e10 if cnt == 0 else err / float(np.sqrt(float(cnt)))

    cb = capi.RMSE_FN(rmse_of)
    # one slice per rank and window: the fewest launches per rank.  Rank r starts scoring once the host has walked the
    # stream to the end of its slice (~15 us per 10 000 hypotheses with the block sampler); interleaving two shorter
    # slices per rank, which hid the slower word-by-word sampler, costs more in extra launches than it hides
    # (tools/time_shard_rank.py: world 8, rank 7: 0.48 ms against 0.50 ms; world 2: 0.38 against 0.45)
    K = 1
    packed = hasattr(scorer, "score_shard_packed")
    while begin < H and not st.stopped:
        end = min(H, begin + window)
        n_win = end - begin
        sl = slice_size if slice_size else max(64, -(-n_win // (K * world)))
        n_slices = -(-n_win // sl)
        k_loc = -(-n_slices // world)             # slices per rank (padded)
        # local records laid out as k_loc slices of `sl` (only the globally last slice can be short,
        # and slices a rank does not own sit at the tail): zero padding lands beyond `end`
        rec = np.zeros(k_loc * sl, dtype=np.uint32)
        if packed:                                   # the C ABI ships and replays (valid << 31 | count) as is
            mine = scorer.score_shard_packed(sampler, threshold, begin, end, sl, world, rank)
            rec[: len(mine)] = mine
            scored += len(mine)
        else:
            v, c = scorer.score_shard(sampler, threshold, begin, end, sl, world, rank)
            scored += len(c)
            rec[: len(c)] = c | (v.astype(np.uint32) << np.uint32(31))      # counts < 2^31
        if have_pg:
            allrec = _all_gather_records(rec, group, device)             # (world, k_loc * sl)
            collectives += 1
        else:
            allrec = rec.reshape(1, -1)
        # slice j = (local slice j // world of rank j % world)  ->  global hypothesis order
        g = np.ascontiguousarray(allrec.reshape(world, k_loc, sl).transpose(1, 0, 2)).reshape(-1)[:n_win]
        prev_best = st.best_index
        if packed:
            g = np.ascontiguousarray(g)
            capi.lib().m3d_replay_chunk(C.byref(st), n_points, kind, H, probability, begin, end, None,
                                        g.ctypes.data_as(C.c_void_p), cb, None)
        else:
            gv = (g >> np.uint32(31)).astype(np.uint8)
            gc = g & np.uint32(0x7FFFFFFF)
            capi.lib().m3d_replay_chunk(C.byref(st), n_points, kind, H, probability, begin, end,
                                        gv.ctypes.data_as(C.c_void_p), gc.ctypes.data_as(C.c_void_p), cb, None)
        if st.best_index != prev_best:           # keep the best model across windows
            bi = int(st.best_index)
            best_model = None
            best_model = (bi, model_of(bi).copy())
        begin = end
        window = min(window * 2, 16384 * world)
        if probability < 1.0 and st.best_index >= 0 and st.current_iteration > st.count:
            # the adaptive bound only shrinks once a best model exists: one more window that covers what is
            # left of it (+6 % for invalid minimal fits) ends the loop with one more collective
            left = int(st.current_iteration) - int(st.count)
            window = min(-(-(left + left // 16 + 16) // 64) * 64, 16384 * world)

    best = best_model[1] if (st.best_index >= 0 and best_model is not None) else np.zeros(capi.NUM_PARAMS[kind])
    if st.best_index >= 0 and getattr(scorer, "refine_takes_expected", False):
        ret, params, inliers = scorer.refine(kind, threshold, best, copy=copy, expected=int(st.best_count))
    else:
        ret, params, inliers = scorer.refine(kind, threshold, best, copy=copy)
    if st.best_index >= 0 and len(inliers) != st.best_count:
        raise capi.M3DError(capi.ERR_INTERNAL, "refine pass and gathered counts disagree")
    if not want_inliers:
        inliers = inliers[:0]
    return ShardedFit(ret, params, inliers, int(st.best_index), int(st.count), int(st.iterations),
                      float(st.best_fitness), scored, collectives, np.asarray(best, dtype=np.float64).copy())


@dataclass
class ShardedSegmentation:
    ret: int                  # 1 = done, 2 = stopped early (a round found no inlier / fewer than 3 points left), 0 = N < 3
    planes: np.ndarray        # (k, 4)
    clusters: list            # k arrays of indices into the cloud as created, ascending
    collectives: int


def segment_plane_iterative_sharded(scorer, threshold, max_iteration=100, min_ratio=0.05, seed=0, group=None,
                                    device=None, max_clusters=None) -> ShardedSegmentation:
    """SegmentPlaneIterative (src/iterative_plane_segmentation.cpp:8-39) with every round's hypothesis
    loop sharded over the ranks: each rank holds a replica of the shrinking cloud,
    scores its share of the round's hypotheses, one all-gather per window, identical replay -> identical
    plane; then every rank removes the same inliers from its replica (m3d_cloud_remove_inliers:
    redundant compute, no point traffic).  Round k uses sampler seed `seed + k`, like the single-GPU
    m3d_segment_plane_iterative, so the result is independent of the number of GPUs."""
    n = int(scorer.n)
    planes, clusters = [], []
    collectives = 0
    if n < 3:                                   # :13-17 LogWarning + empty result
        return ShardedSegmentation(0, np.zeros((0, 4)), clusters, 0)
    target = int((1.0 - float(min_ratio)) * float(n))          # size_t((1 - min_ratio) * n), :28
    max_clusters = n if max_clusters is None else int(max_clusters)
    count, k, ret = 0, 0, 1
    while count < target and k < max_clusters:
        if scorer.n < 3:                        # the reference's FitModel would throw (ransac.h:510-513)
            ret = 2
            break
        r = fit_sharded(scorer, int(scorer.n), capi.PLANE, threshold, max_iteration, 0.9999, seed + k, group=group,
                        device=device, copy=True)
        collectives += r.collectives
        ni = len(r.inliers)
        if ni == 0:                             # the reference would loop forever (:29,:35)
            ret = 2
            break
        planes.append(np.asarray(r.params[:4], dtype=np.float64).copy())
        clusters.append(np.asarray(r.inliers).astype(np.int64))
        count += ni
        k += 1
        if count >= target or k >= max_clusters:
            break
        removed = scorer.remove_inliers(capi.PLANE, threshold, r.best_model)      # SelectByIndex(inliers, true), :33
        if removed != ni:
            raise capi.M3DError(capi.ERR_INTERNAL, "removed points and inlier list disagree")
    return ShardedSegmentation(ret, np.array(planes).reshape(-1, 4), clusters, collectives)



def reg_shard(ns, world, rank):
    """Survivor range [s0, s1) of `rank` and the padded shard length: whole groups of 64 hypotheses
    (the validation kernel's unit), ceil(groups / world) groups per rank, contiguous."""
    groups = -(-ns // 64)
    per = -(-groups // world) if groups else 0
    g0 = min(rank * per, groups)
    g1 = min(g0 + per, groups)
    if g0 == g1:
        return 0, 0, per * 64                  # nothing left for this rank
    return g0 * 64, min(g1 * 64, ns), per * 64


def _all_gather_f64(rec, group, device):
    """all_gather of equal-shape float64 arrays -> (world, *rec.shape); bit-exact transport."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(rec))
    if device is not None:
        t = t.to(device)
    out = torch.empty((world,) + tuple(rec.shape), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t.view(-1), group=group)
    return out.cpu().numpy()


def registration_ransac_sharded(session, group=None, device=None, gather=None, world=None, rank=None):
    """compute_transformation_ransac with the validation of every chunk's surviving hypotheses sharded over
    the ranks: source, target, grid and correspondences are replicated, the sessions are
    seeded identically (so every rank draws the same triples and keeps the same survivors), rank r validates
    a contiguous run of 64-hypothesis groups, ONE all-gather per chunk exchanges (count, sum d^2) as float64
    pairs (counts < 2^31 are exact in float64), every rank replays the whole chunk.  `session`:
    capi.RegSession (or anything with begin_chunk/validate/replay/finish).  `gather(rec) -> (world, ...)`
    replaces the torch.distributed all-gather (tests)."""
    if gather is None:
        world, rank, have_pg = _world(group)
        if have_pg:
            def gather(rec):
                return _all_gather_f64(rec, group, device)
        else:
            def gather(rec):
                return rec[None]
    collectives = 0
    while True:
        ns = session.begin_chunk()
        if ns is None:
            break
        s0, s1, shard = reg_shard(ns, world, rank)
        c, s = session.validate(s0, s1)
        rec = np.zeros((max(shard, 1), 2), dtype=np.float64)
        rec[: len(c), 0] = c
        rec[: len(c), 1] = s
        if shard:                     # ns == 0 on every rank alike: nothing to exchange
            allrec = np.asarray(gather(rec)).reshape(-1, 2)[:ns]
            collectives += 1
        else:
            allrec = np.zeros((0, 2))
        session.replay(allrec[:ns, 0].astype(np.uint32), np.ascontiguousarray(allrec[:ns, 1]))
    T, st = session.finish()
    st["collectives"] = collectives
    return T, st
