"""A/B on one box: C2 step time with and without the per-launch timing events (m3d_config.kernel_timing)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from misc3d_amd import capi, synth
pts = synth.plane_cloud_c2(1_000_000, seed=2)
c = capi.Cloud(pts)
for _ in range(50):
    c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
for rep in range(3):
    for kt in (1, 0):
        old = capi.set_config(kernel_timing=kt)
        t0 = time.perf_counter()
        for _ in range(300):
            g = c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
        dt = (time.perf_counter() - t0) / 300 * 1e3
        capi.restore_config(old)
        print(f"kernel_timing={kt}: {dt:.4f} ms/step  {({k: round(g.stats[k], 3) for k in ('ms_sample', 'ms_score', 'ms_refine', 'ms_total')})}")
