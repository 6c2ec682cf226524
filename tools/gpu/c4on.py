import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(200000, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
it = int(os.environ.get("M3D_C4_ITERS", "100000"))
for _ in range(2):
    t0=time.perf_counter()
    T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=it, edge_length_threshold=0.9, confidence=1.0, seed=17)
    print((time.perf_counter()-t0)*1e3, st)
