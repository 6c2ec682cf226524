"""Scratch check of the C++ sharded driver on one GPU: RCCL communicator of world 1 (+ host transport), identical
results and the cost of the sharded path against the one-call fit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if "--torch" in sys.argv:
    import torch
    print("torch imported first; cuda:", torch.cuda.is_available())
from misc3d_amd import capi, synth

pts = synth.plane_cloud_c2(1_000_000, seed=2)
cloud = capi.Cloud(pts)
one = cloud.fit(capi.PLANE, 0.01, 10000, 1.0, seed=11)
print("one-call", one.stats["best_index"], one.stats["n_inliers"])
for name, mk in (("rccl", lambda: capi.Comm.rccl(world=1, rank=0, device=0)),
                 ("host", lambda: capi.Comm.host(1, 0, lambda b: b))):
    t0 = time.perf_counter()
    comm = mk()
    print(name, "comm created in %.2f s" % (time.perf_counter() - t0))
    r = cloud.fit_sharded(comm, capi.PLANE, 0.01, 10000, 1.0, seed=11)
    ok = (r.stats["best_index"] == one.stats["best_index"] and np.array_equal(r.inliers, one.inliers)
          and np.array_equal(r.params, one.params))
    print(name, "identical:", ok, "collectives", comm.collectives)
    for prob, H, seed in ((0.9999, 1000, 5), (0.99, 400, 9)):
        a = cloud.fit(capi.PLANE, 0.01, H, prob, seed=seed)
        b = cloud.fit_sharded(comm, capi.PLANE, 0.01, H, prob, seed=seed)
        print(name, prob, "identical:", a.stats["best_index"] == b.stats["best_index"] and a.stats["iterations"] == b.stats["iterations"]
              and np.array_equal(a.inliers, b.inliers) and np.array_equal(a.params, b.params))
    for _ in range(30):
        cloud.fit(capi.PLANE, 0.01, 10000, 1.0, seed=11, copy=False)
        cloud.fit_sharded(comm, capi.PLANE, 0.01, 10000, 1.0, seed=11, copy=False)
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(300):
            cloud.fit(capi.PLANE, 0.01, 10000, 1.0, seed=11, copy=False)
        t1 = time.perf_counter()
        for _ in range(300):
            cloud.fit_sharded(comm, capi.PLANE, 0.01, 10000, 1.0, seed=11, copy=False)
        t2 = time.perf_counter()
        print(name, "one-call %.4f ms   sharded(world 1) %.4f ms   ratio %.3f" % ((t1 - t0) / 0.3, (t2 - t1) / 0.3, (t2 - t1) / (t1 - t0)))
    comm.close()
# segmentation
room = synth.room_cloud_c5(300000, 6)
rc1, planes1, clusters1 = capi.segment_plane_iterative(room, 0.01, max_iteration=200, min_ratio=0.05, seed=19)
comm = capi.Comm.rccl(world=1, rank=0, device=0)
rc2, planes2, clusters2 = capi.segment_plane_iterative_sharded(room, comm, 0.01, max_iteration=200, min_ratio=0.05, seed=19)
print("segmentation identical:", rc1 == rc2 and np.array_equal(planes1, planes2) and all(np.array_equal(a, b) for a, b in zip(clusters1, clusters2)), len(planes1), comm.collectives)
rc3, planes3, clusters3 = capi.segment_plane_iterative_multi(room, [0], 0.01, max_iteration=200, min_ratio=0.05, seed=19)
print("multi[0] identical:", rc1 == rc3 and np.array_equal(planes1, planes3))
d = synth.registration_pair_c4(20000, seed=5, dim=33, true_fraction=0.5, sigma=0.001)
a, b = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
T1, st1 = capi.registration_ransac(d["src"], d["dst"], a, b, threshold=0.03, max_iter=3000, confidence=1.0, seed=17)
T2, st2 = capi.registration_ransac_sharded(d["src"], d["dst"], a, b, comm, threshold=0.03, max_iter=3000, confidence=1.0, seed=17)
print("registration identical:", np.array_equal(T1, T2), st1["validations"], st2["validations"], comm.collectives)
comm.close()
print("TRY_SHARDED_DONE")
