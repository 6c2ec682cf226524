"""Multi-GPU RANSAC: hypotheses sharded over ranks, one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Hypotheses are the independent unit of the reference loop (ransac.h:571-590 has no cross-iteration
dependency except the best-model reduction, ransac.h:592-613).  Every rank holds a replica of the
cloud in HBM (10 M points = 240 MB), draws the SAME sample table from the same std::mt19937 seed,
scores a CONTIGUOUS slice of each global chunk through the C ABI (m3d_cloud_score_range), and one
all-gather per chunk exchanges the per-hypothesis (valid, inlier count) records -- 5 bytes per
hypothesis, latency-bound.  Every rank then replays the sequential best-update / adaptive-stop rule
in index order (m3d_replay_chunk), so the chosen hypothesis, the iteration count and the inlier set
are identical to the single-GPU run and independent of the number of GPUs.

The scorer is injected so the N>1 control path can be tested on CPU with gloo (the tests plug the
oracle in as the scorer; the product scorer is ``capi.Cloud``).
"""
from __future__ import annotations

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


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _all_gather_records(valid, counts, per_rank, group, device):
    """all_gather of fixed-size (per_rank) uint8/uint32 records.  Returns (world*per_rank,) arrays."""
    import torch
    import torch.distributed as dist
    world, _ = _world(group)
    if world == 1:
        return valid, counts
    # one int32 tensor: counts, with the valid flag folded into bit 31 (counts < 2^31)
    rec = counts.astype(np.int64) | (valid.astype(np.int64) << 31)
    t = torch.from_numpy(rec.astype(np.int64).astype(np.uint32).view(np.int32).copy())
    if device is not None:
        t = t.to(device, non_blocking=False)
    out = torch.empty(world * per_rank, dtype=torch.int32, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    full = out.cpu().numpy().view(np.uint32)
    return ((full >> 31) & 1).astype(np.uint8), (full & 0x7FFFFFFF).astype(np.uint32)


def fit_sharded(scorer, n_points, kind, threshold=0.01, max_iteration=1000, probability=0.9999, seed=0,
                group=None, device=None, want_inliers=True) -> ShardedFit:
    """RANSAC::FitModel with the hypothesis loop sharded over the ranks of `group`.

    scorer: object with score_range(kind, thr, samples, begin, end) -> (valid, models, counts),
            exact_error(kind, thr, model) -> (count, error), refine(kind, thr, params) -> (ret, params, inliers)
            (``capi.Cloud`` has exactly this interface).
    """
    world, rank = _world(group)
    m = capi.MINIMAL_SAMPLE[kind]
    if probability <= 0 or probability > 1:
        raise capi.M3DError(capi.ERR_PROBABILITY, "Probability must be > 0 or <= 1.0")
    if n_points < m:
        raise capi.M3DError(capi.ERR_TOO_FEW_POINTS, "Can not fit model due to lack of points")
    H = int(max_iteration)
    table = capi.draw_samples(n_points, kind, H, seed) if H else np.zeros((0, m), dtype=np.uint32)

    import ctypes as C
    st = capi.ReplayState()
    capi.lib().m3d_replay_init(C.byref(st))
    my_models = {}          # global hypothesis index -> model (only this rank's slices)
    per_rank = 1024 if probability >= 1.0 else 128
    if probability >= 1.0:
        per_rank = max(64, -(-H // world))      # one round: early exit is impossible except fitness == 1
    cap = 16384
    begin = 0
    scored = 0
    collectives = 0

    def model_of(i):
        if i in my_models:
            return my_models[i]
        _, mod, _ = scorer.score_range(kind, threshold, table, i, i + 1)
        return mod[0]

    def rmse_of(i):
        cnt, err = scorer.exact_error(kind, threshold, model_of(int(i)))
        return 1e10 if cnt == 0 else err / np.sqrt(float(cnt))

    while begin < H and not st.stopped:
        end = min(H, begin + per_rank * world)
        lo = min(end, begin + rank * per_rank)
        hi = min(end, lo + per_rank)
        valid = np.zeros(per_rank, dtype=np.uint8)
        counts = np.zeros(per_rank, dtype=np.uint32)
        if hi > lo:
            v, mod, c = scorer.score_range(kind, threshold, table, lo, hi)
            valid[: hi - lo] = v
            counts[: hi - lo] = c
            for k in range(hi - lo):
                my_models[lo + k] = mod[k]
            scored += hi - lo
        gv, gc = _all_gather_records(valid, counts, per_rank, group, device)
        collectives += 1 if world > 1 else 0
        # records of rank r sit at [r*per_rank, (r+1)*per_rank): global order is contiguous
        n_chunk = end - begin
        gv = np.ascontiguousarray(gv[:n_chunk])
        gc = np.ascontiguousarray(gc[:n_chunk])
        cb = capi.RMSE_FN(lambda _u, i: float(rmse_of(i)))
        capi.lib().m3d_replay_chunk(C.byref(st), n_points, kind, H, probability, begin, end,
                                    gv.ctypes.data_as(C.c_void_p), gc.ctypes.data_as(C.c_void_p), cb, None)
        # keep only the best model of what has been replayed so far
        keep = my_models.get(st.best_index)
        my_models = {st.best_index: keep} if keep is not None else {}
        begin = end
        per_rank = min(per_rank * 2, cap)

    if st.best_index >= 0:
        best = model_of(int(st.best_index))
    else:
        best = np.zeros(capi.NUM_PARAMS[kind])
    ret, params, inliers = scorer.refine(kind, threshold, best)
    if st.best_index >= 0 and len(inliers) != st.best_count:
        raise capi.M3DError(capi.ERR_INTERNAL, "refine pass and gathered counts disagree")
    if not want_inliers:
        inliers = inliers[:0]
    return ShardedFit(ret, params, inliers, int(st.best_index), int(st.count), int(st.iterations),
                      float(st.best_fitness), scored, collectives)
