"""Worker of tests/test_gpu_two_ranks.py: two (or more) processes, gloo process group, EVERY rank on GPU 0.

The product path (capi.Cloud / capi.RegSession -> C ABI -> HIP kernels) runs through the N > 1 drivers of
misc3d_amd/distributed.py exactly as under `bench.py --gpus N`, except that the records travel over gloo
instead of RCCL (one GPU cannot host two RCCL ranks).  Every rank compares its result with the one-call
single-GPU result; rank 0 prints TWO_RANK_OK when all ranks agree."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from misc3d_amd import capi, distributed, synth  # noqa: E402


def main():
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    fails = []

    def check(name, ok):
        if not ok:
            fails.append(name)

    # fits: all three models, with and without the adaptive stop
    clouds = {capi.PLANE: (synth.plane_cloud_c2(60000, seed=2), None), capi.SPHERE: (synth.sphere_cloud_c3(40000, 4), None),
              capi.CYLINDER: synth.cylinder_cloud_c3(40000, 3)}
    for kind, (pts, nrm) in clouds.items():
        for prob, H, seed in ((1.0, 3000, 11), (0.9999, 1000, 5), (0.99, 400, 9)):
            with capi.Cloud(pts, nrm) as c:
                one = c.fit(kind, 0.01, H, prob, seed=seed)
            with capi.Cloud(pts, nrm) as c:
                r = distributed.fit_sharded(c, len(pts), kind, 0.01, H, prob, seed)
            tag = f"fit kind {kind} prob {prob}"
            check(tag + " index", r.best_index == one.stats["best_index"] and r.iterations == one.stats["iterations"])
            check(tag + " inliers", np.array_equal(r.inliers, one.inliers))
            check(tag + " params", np.allclose(np.asarray(r.params), np.asarray(one.params), rtol=0, atol=1e-12))   # (other summation tree)
            check(tag + " collectives", r.collectives >= 1)
    # iterative segmentation
    room = synth.room_cloud_c5(120000, 6)
    rc1, planes1, clusters1 = capi.segment_plane_iterative(room, 0.01, max_iteration=200, min_ratio=0.05, seed=19)
    with capi.Cloud(room) as c:
        s = distributed.segment_plane_iterative_sharded(c, 0.01, 200, 0.05, seed=19)
    check("segmentation count", s.ret == rc1 and len(s.planes) == len(planes1) and len(planes1) >= 3)
    check("segmentation planes", np.allclose(s.planes, planes1, rtol=0, atol=1e-12))
    check("segmentation clusters", all(np.array_equal(a, b) for a, b in zip(s.clusters, clusters1)))
    # registration
    d = synth.registration_pair_c4(20000, seed=5, dim=33, true_fraction=0.5, sigma=0.001)
    a, b = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    for conf, H in ((1.0, 3000), (0.999, 100000)):
        T1, st1 = capi.registration_ransac(d["src"], d["dst"], a, b, threshold=0.03, max_iter=H, confidence=conf, seed=17)
        sess = capi.RegSession(d["src"], d["dst"], a, b, threshold=0.03, max_iter=H, confidence=conf, seed=17)
        T2, st2 = distributed.registration_ransac_sharded(sess)
        sess.close()
        check(f"registration conf {conf} T", np.array_equal(T1, T2))
        check(f"registration conf {conf} stats", all(st1[k] == st2[k] for k in ("best_index", "iterations", "validations", "fitness")))
    # ---- the C++ driver (m3d_cloud_fit_sharded & co., include/misc3d_amd.h) with the records over gloo through the
    # host transport of m3d_comm: same shard loop, same kernels as under RCCL, only the all-gather differs
    comm = capi.Comm.torch_host()
    check("comm geometry", comm.world == world and comm.rank == rank)
    for kind, (pts, nrm) in clouds.items():
        for prob, H, seed in ((1.0, 3000, 11), (1.0, 40000, 3), (0.9999, 1000, 5), (0.99, 400, 9), (1.0, 100, 2)):
            with capi.Cloud(pts, nrm) as c:
                one = c.fit(kind, 0.01, H, prob, seed=seed)
                before = comm.collectives
                r = c.fit_sharded(comm, kind, 0.01, H, prob, seed=seed)
            tag = f"C++ fit kind {kind} prob {prob} H {H}"
            check(tag + " index", r.stats["best_index"] == one.stats["best_index"] and r.stats["iterations"] == one.stats["iterations"]
                  and r.stats["count"] == one.stats["count"])
            check(tag + " inliers", np.array_equal(r.inliers, one.inliers))
            check(tag + " params", np.array_equal(np.asarray(r.params), np.asarray(one.params)))
            check(tag + " collectives", comm.collectives > before)
            if prob >= 1.0 and H > 128:     # every rank scores its slice only
                check(tag + " share", r.stats["hypotheses_scored"] < one.stats["hypotheses_scored"])
    # a fit without a seed: rank 0's random seed is shared, so the ranks still agree
    with capi.Cloud(*clouds[capi.PLANE]) as c:
        r = c.fit_sharded(comm, capi.PLANE, 0.01, 500, 0.9999, seed=None)
    t = torch.tensor([float(r.stats["best_index"]), float(len(r.inliers))], dtype=torch.float64)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    check("unseeded fit agrees across ranks", bool(torch.equal(t, tmax)))
    rc3, planes3, clusters3 = capi.segment_plane_iterative_sharded(room, comm, 0.01, max_iteration=200, min_ratio=0.05, seed=19)
    check("C++ segmentation count", rc3 == rc1 and len(planes3) == len(planes1))
    check("C++ segmentation planes", np.array_equal(planes3, planes1))
    check("C++ segmentation clusters", all(np.array_equal(a_, b_) for a_, b_ in zip(clusters3, clusters1)))
    for conf, H in ((1.0, 3000), (0.999, 100000)):
        T1, st1 = capi.registration_ransac(d["src"], d["dst"], a, b, threshold=0.03, max_iter=H, confidence=conf, seed=17)
        T3, st3 = capi.registration_ransac_sharded(d["src"], d["dst"], a, b, comm, threshold=0.03, max_iter=H, confidence=conf, seed=17)
        check(f"C++ registration conf {conf} T", np.array_equal(T1, T3))
        check(f"C++ registration conf {conf} stats", all(st1[k] == st3[k] for k in ("best_index", "iterations", "validations", "fitness")))
    comm.close()
    # every rank must have seen no failure
    t = torch.tensor([len(fails)], dtype=torch.int64)
    dist.all_reduce(t)
    if fails:
        print(f"rank {rank}: FAILED {fails}", flush=True)
    if rank == 0 and int(t.item()) == 0:
        print(f"TWO_RANK_OK world {world}", flush=True)
    dist.destroy_process_group()
    sys.exit(1 if int(t.item()) else 0)


if __name__ == "__main__":
    main()
