"""What an 8-GPU strong-scaling step is made of, measured on ONE GPU (no 8-GPU node is available to the builder): for C2 and
the two C3 fits, the time of rank k's share of the sharded scoring loop (m3d_cloud_score_shard with world = N: the whole
window's samples and minimal fits, the box tests and scoring of the rank's slice + the leading hypotheses, the records on
the host), for every rank k, against the same call with world = 1 and against the complete one-GPU fit.  The step at N GPUs
is then modelled as  max_k shard(k) + exchange + (fit(1) - shard(world 1)),  the last term being what every rank still does
alone (replay, RefineModel, the index list over PCIe).  `exchange` is the one number that cannot be measured here (an
ncclAllGather of 4 bytes per hypothesis over xGMI + the pick over the gathered records): 30 and 60 us are printed."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from misc3d_amd import capi, synth  # noqa: E402

CASES = {"c2": (0, 10_000, 11), "c3sph": (1, 50_000, 13), "c3cyl": (2, 50_000, 13)}
N = 1_000_000
REPS = 15


def best_of(fn, reps=REPS):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


for name, (kind, H, seed) in CASES.items():
    if kind == 0:
        pts, nrm = synth.plane_cloud_c2(N, 2), None
    elif kind == 1:
        pts, nrm = synth.sphere_cloud_c3(N, 4), None
    else:
        pts, nrm = synth.cylinder_cloud_c3(N, 3)
    with capi.Cloud(pts, nrm) as c:
        for _ in range(5):
            c.fit(kind, 0.01, H, 1.0, seed=seed, copy=False)
        t_fit = best_of(lambda: c.fit(kind, 0.01, H, 1.0, seed=seed, copy=False))

        def shard(world, rank):
            sl = -(-(-(-H // world)) // 64) * 64
            sm = c.make_sampler(kind, seed)
            try:
                t0 = time.perf_counter()
                c.score_shard_packed(sm, 0.01, 0, H, sl, world, rank)
                return time.perf_counter() - t0
            finally:
                sm.close()

        out = {"workload": name, "hypotheses": H, "ms_fit_1gpu": t_fit}
        for world in (1, 2, 4, 8):
            per_rank = []
            for rank in range(world):
                for _ in range(2):
                    shard(world, rank)
                per_rank.append(float(np.median([shard(world, rank) for _ in range(REPS)])) * 1e3)
            out[f"ms_shard_world{world}_max_rank"] = max(per_rank)
            out[f"ms_shard_world{world}_per_rank"] = [round(v, 4) for v in per_rank]
        alone = t_fit - out["ms_shard_world1_max_rank"]
        out["ms_alone_per_rank (replay, RefineModel, index list)"] = alone
        for world in (2, 4, 8):
            for ex in (0.03, 0.06):
                step = out[f"ms_shard_world{world}_max_rank"] + ex + max(alone, 0.0)
                out[f"modelled_speedup_world{world}_exchange_{int(ex * 1000)}us"] = t_fit / step
        print(json.dumps(out), flush=True)
