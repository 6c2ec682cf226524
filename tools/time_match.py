"""C4 matcher (200 k x 200 k descriptors of width 33): wall clock of match_mutual_nn, four runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(int(os.environ.get("M3D_C4_POINTS", "200000")), seed=5)
for rep in range(4):
    t0 = time.perf_counter()
    i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    print(f"{(time.perf_counter() - t0) * 1e3:.2f} ms  matches {len(i0)} checksum {int(i0.sum() + 3 * i1.sum())} fallbacks {capi.match_last_fallbacks()}", flush=True)
