"""BASELINE.json configs[4] (C5): iterative plane segmentation of a 10 M-point scene with every round's hypotheses
sharded over the ranks (one process per GPU, RCCL all-gather of 4 B per hypothesis per window):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \\
        tools/bench_c5_sharded.py [--points 10000000] [--max-iteration 1000]

Rank 0 prints one JSON line (rounds, planes found, wall time, collectives).  The result is identical for every N
(same planes, same clusters): rank 0 also checks it against the single-call m3d_segment_plane_iterative."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--max-iteration", type=int, default=1000)
    ap.add_argument("--min-ratio", type=float, default=0.05)
    ap.add_argument("--threshold", type=float, default=0.01)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from misc3d_amd import capi, distributed, synth
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    pts = synth.room_cloud_c5(a.points, 6)
    cloud = capi.Cloud(pts, device=local)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    r = distributed.segment_plane_iterative_sharded(cloud, a.threshold, a.max_iteration, a.min_ratio, seed=19, device=dev)
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        out = {"config": "C5 segment_plane_iterative, hypotheses sharded", "n_gpus": world, "points": a.points,
               "rounds": len(r.planes), "ret": r.ret, "ms": dt * 1e3, "collectives": r.collectives,
               "points_in_clusters": int(sum(len(c) for c in r.clusters))}
        if not a.no_check:
            rc, planes, clusters = capi.segment_plane_iterative(pts, a.threshold, max_iteration=a.max_iteration,
                                                                min_ratio=a.min_ratio, seed=19, device=local)
            out["equals_single_call"] = bool(len(planes) == len(r.planes) and np.array_equal(planes, r.planes) and all(
                np.array_equal(x.astype(np.int64), y) for x, y in zip(clusters, r.clusters)))
        print(json.dumps(out), flush=True)
    cloud.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
