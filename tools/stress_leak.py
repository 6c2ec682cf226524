"""Repeated calls of every entry point; device memory must return to its starting level (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from misc3d_amd import capi, synth, distributed

def free_mb():
    f, t = torch.cuda.mem_get_info(0)
    return f / 2**20

pts = synth.plane_cloud_c2(200_000, seed=2)
cp, cn = synth.cylinder_cloud_c3(100_000, 3)
d = synth.registration_pair_c4(20_000, seed=5)
def once():
    capi.fit(0, pts, None, 0.01, 500, 0.9999, seed=1)
    capi.fit(2, cp, cn, 0.01, 300, 0.9999, seed=1)
    with capi.Cloud(pts) as c:
        c.fit(0, 0.01, 2000, 1.0, seed=3)
        distributed.segment_plane_iterative_sharded(c, 0.02, 100, 0.3, seed=2)
    capi.segment_plane_iterative(pts, 0.02, 100, 0.2, seed=4)
    i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=3000, confidence=1.0, seed=17)
    with capi.RegSession(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=2000, confidence=1.0, seed=1) as s:
        distributed.registration_ransac_sharded(s)
    capi.kabsch(d["src"][:1000], d["dst"][:1000])
once()
torch.cuda.synchronize()
m0 = free_mb()
for k in range(15):
    once()
torch.cuda.synchronize()
m1 = free_mb()
print("free MiB before %.1f after %.1f  delta %.1f" % (m0, m1, m0 - m1))
assert m0 - m1 < 64, "device memory is leaking"
print("OK")
