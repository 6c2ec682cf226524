"""Multi-GPU RANSAC: hypotheses sharded over ranks, one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Hypotheses are the independent unit of the reference loop (ransac.h:571-590 has no cross-iteration
dependency except the best-model reduction, ransac.h:592-613).  Every rank holds a replica of the
cloud in HBM (10 M points = 240 MB), draws the SAME sample table from the same std::mt19937 seed,
scores a CONTIGUOUS slice of each global chunk through the C ABI (m3d_cloud_score_range), and one
all-gather per chunk exchanges the per-hypothesis (valid, inlier count) records -- 4 bytes per
hypothesis, latency-bound.  Every rank then replays the sequential best-update / adaptive-stop rule
in index order (m3d_replay_chunk), so the chosen hypothesis, the iteration count and the inlier set
are identical to the single-GPU run and independent of the number of GPUs.

The scorer is injected so the N>1 control path can be tested on CPU with gloo (the tests plug the
oracle in as the scorer; the product scorer is ``capi.Cloud``).
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


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group), True
    return 1, 0, False


def _all_gather_records(valid, counts, per_rank, group, device, have_pg):
    """all_gather of fixed-size (per_rank) records; the valid flag rides in bit 31 of the count
    (counts < 2^31).  Returns (world*per_rank,) arrays in rank order = global hypothesis order."""
    if not have_pg:
        return valid, counts
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rec = (counts | (valid.astype(np.uint32) << np.uint32(31))).view(np.int32)
    t = torch.from_numpy(rec)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * per_rank, dtype=torch.int32, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    full = out.cpu().numpy().view(np.uint32)
    return (full >> np.uint32(31)).astype(np.uint8), full & np.uint32(0x7FFFFFFF)


def fit_sharded(scorer, n_points, kind, threshold=0.01, max_iteration=1000, probability=0.9999, seed=0,
                group=None, device=None, want_inliers=True) -> ShardedFit:
    """RANSAC::FitModel with the hypothesis loop sharded over the ranks of `group`.

    scorer: object with score_range(kind, thr, samples, begin, end) -> (valid, models, counts),
            exact_error(kind, thr, model) -> (count, error), refine(kind, thr, params) -> (ret, params, inliers)
            (``capi.Cloud`` has exactly this interface).
    """
    world, rank, have_pg = _world(group)
    m = capi.MINIMAL_SAMPLE[kind]
    if probability <= 0 or probability > 1:
        raise capi.M3DError(capi.ERR_PROBABILITY, "Probability must be > 0 or <= 1.0")
    if n_points < m:
        raise capi.M3DError(capi.ERR_TOO_FEW_POINTS, "Can not fit model due to lack of points")
    H = int(max_iteration)
    table = capi.draw_samples(n_points, kind, H, seed) if H else np.zeros((0, m), dtype=np.uint32)

    st = capi.ReplayState()
    capi.lib().m3d_replay_init(C.byref(st))
    if probability >= 1.0:
        per_rank = max(64, -(-H // world))      # one round: early exit is impossible except fitness == 1
    else:
        per_rank = 128
    cap = 16384
    begin = 0
    scored = 0
    collectives = 0
    best_model = None
    cur = {"lo": 0, "hi": 0, "models": None}     # this rank's slice of the chunk being replayed

    def model_of(i):
        i = int(i)
        if i == st.best_index and best_model is not None:
            return best_model
        if cur["models"] is not None and cur["lo"] <= i < cur["hi"]:
            return cur["models"][i - cur["lo"]]
        _, mod, _ = scorer.score_range(kind, threshold, table, i, i + 1)     # another rank's hypothesis: recompute
        return mod[0]

    def rmse_of(_user, i):
        cnt, err = scorer.exact_error(kind, threshold, model_of(i))
        return 1e10 if cnt == 0 else err / float(np.sqrt(float(cnt)))

    cb = capi.RMSE_FN(rmse_of)
    while begin < H and not st.stopped:
        end = min(H, begin + per_rank * world)
        lo = min(end, begin + rank * per_rank)
        hi = min(end, lo + per_rank)
        valid = np.zeros(per_rank, dtype=np.uint8)
        counts = np.zeros(per_rank, dtype=np.uint32)
        cur["lo"], cur["hi"], cur["models"] = lo, hi, None
        if hi > lo:
            v, mod, c = scorer.score_range(kind, threshold, table, lo, hi)
            valid[: hi - lo] = v
            counts[: hi - lo] = c
            cur["models"] = mod
            scored += hi - lo
        gv, gc = _all_gather_records(valid, counts, per_rank, group, device, have_pg)
        collectives += 1 if have_pg else 0
        # records of rank r sit at [r*per_rank, (r+1)*per_rank): global order is contiguous
        n_chunk = end - begin
        gv = np.ascontiguousarray(gv[:n_chunk])
        gc = np.ascontiguousarray(gc[:n_chunk])
        prev_best = st.best_index
        capi.lib().m3d_replay_chunk(C.byref(st), n_points, kind, H, probability, begin, end,
                                    gv.ctypes.data_as(C.c_void_p), gc.ctypes.data_as(C.c_void_p), cb, None)
        if st.best_index != prev_best:           # keep the best model across chunks
            bi = int(st.best_index)
            if cur["models"] is not None and lo <= bi < hi:
                best_model = cur["models"][bi - lo].copy()
            else:
                best_model = None
                best_model = model_of(bi).copy()
        begin = end
        per_rank = min(per_rank * 2, cap)

    if st.best_index >= 0:
        best = best_model if best_model is not None else model_of(int(st.best_index))
    else:
        best = np.zeros(capi.NUM_PARAMS[kind])
    ret, params, inliers = scorer.refine(kind, threshold, best)
    if st.best_index >= 0 and len(inliers) != st.best_count:
        raise capi.M3DError(capi.ERR_INTERNAL, "refine pass and gathered counts disagree")
    if not want_inliers:
        inliers = inliers[:0]
    return ShardedFit(ret, params, inliers, int(st.best_index), int(st.count), int(st.iterations),
                      float(st.best_fitness), scored, collectives)
