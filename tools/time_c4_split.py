"""Where does the C4 validation time go?  Same clouds, three correspondence sets: the matcher's (mixed), only TRUE
pairs, only FALSE pairs -- time per validated hypothesis for each, with and without the LDS-staged kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
n = 200000
d = synth.registration_pair_c4(n, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
inv = np.empty(n, dtype=np.int64); inv[d["perm"]] = np.arange(n)
good = inv[i0.astype(np.int64)] == i1.astype(np.int64)
sets = {"mixed": (i0, i1), "true": (i0[good], i1[good]), "false": (i0[~good], i1[~good])}
for name, (a, b) in sets.items():
    for lds in (0, 1):
        old = capi.set_config(reg_lds_staging=lds)
        H = 30000 if name != "false" else 300000
        t0 = time.perf_counter()
        T, st = capi.registration_ransac(d["src"], d["dst"], a, b, threshold=0.03, max_iter=H, edge_length_threshold=0.9, confidence=1.0, seed=17)
        dt = time.perf_counter() - t0
        capi.restore_config(old)
        print(f"{name:6s} lds={lds}: {dt*1e3:8.1f} ms  validations {st['validations']:6d}  {dt*1e6/max(st['validations'],1):7.2f} us/validation  fitness {st['fitness']:.3f}"
              f"  lds/global {st['lds_wave_hypotheses']}/{st['global_wave_hypotheses']}", flush=True)
