"""C4 registration (200 k <-> 200 k, 100 k hypotheses, confidence 1) with and without the LDS-staged validation kernel:
identical results, times, share of the work served from LDS."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
n = int(os.environ.get("M3D_C4_POINTS", "200000"))
d = synth.registration_pair_c4(n, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
res = {}
for lds in (1, 0, 1):
    old = capi.set_config(reg_lds_staging=lds)
    t0 = time.perf_counter()
    T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=100_000, edge_length_threshold=0.9, confidence=1.0, seed=17)
    dt = time.perf_counter() - t0
    capi.restore_config(old)
    print(f"lds={lds}: {dt*1e3:.1f} ms  validations {st['validations']} fitness {st['fitness']:.4f} best {st['best_index']} "
          f"lds/global wave-hyps {st['lds_wave_hypotheses']}/{st['global_wave_hypotheses']}  pose err {np.abs(T - d['T']).max():.2e}", flush=True)
    res.setdefault(lds, (T, {k: st[k] for k in ("best_index", "validations", "fitness", "est_k", "inlier_rmse")}))
    assert np.array_equal(res[lds][0], T)
assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1], (res[0][1], res[1][1])
print("identical with and without LDS staging")
