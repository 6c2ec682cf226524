import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from misc3d_amd import capi, synth
pts, nrm = synth.cylinder_cloud_c3(1_000_000, 3)
with capi.Cloud(pts, nrm) as c:
    for i in range(5):
        t0 = time.perf_counter(); g = c.fit(2, 0.01, 50_000, 1.0, seed=13); dt = (time.perf_counter() - t0) * 1e3
    print(f"{dt:.3f} ms", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in g.stats.items() if 'ms' in k or k in ('chunks', 'hypotheses_scored')})
