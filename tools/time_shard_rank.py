"""What one rank of an 8-GPU sharded C2 fit does on its own GPU (no process group needed): the sample stream of all
80 000 hypotheses is walked on the host, the rank's K slices are scored.  python tools/time_shard_rank.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth

N = 1_000_000
pts = synth.plane_cloud_c2(N, seed=2)
c = capi.Cloud(pts)
for world, K in ((1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (8, 1), (8, 2)):
    H = 10_000 * world
    sl = -(-H // (K * world))
    for rank in sorted({0, world // 2, world - 1}):
        ts = []
        for rep in range(12):
            t = time.perf_counter()
            s = c.make_sampler(0, 11)
            v, cnt = c.score_shard(s, 0.01, 0, H, sl, world, rank)
            ts.append((time.perf_counter() - t) * 1e3)
        print("world %d K %d rank %d: %d hypotheses scored, sampler + score_shard %.3f ms (min %.3f)" % (
            world, K, rank, len(cnt), float(np.median(ts[2:])), min(ts[2:])))
