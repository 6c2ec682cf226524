import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(200000, seed=5)
fs, fd = d["feat_src"].copy(), d["feat_dst"].copy()
# 300 source rows at one exact distance from one target row: more candidates than the row's slots -> exact fallback
rng = np.random.default_rng(1)
fs[1000:1300] = fd[77] + 1e-3 * np.sign(rng.normal(size=(300, 33)))
for rep in range(4):
    t0 = time.perf_counter()
    i0, i1 = capi.match_mutual_nn(fs, fd)
    print(f"{(time.perf_counter() - t0) * 1e3:.2f} ms  matches {len(i0)} fallbacks {capi.match_last_fallbacks()}", flush=True)
