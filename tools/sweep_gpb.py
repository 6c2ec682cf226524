"""C2 step time against m3d_config.score_groups_per_block / score_min_workgroups (one box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from misc3d_amd import capi, synth
pts = synth.plane_cloud_c2(1_000_000, seed=2)
c = capi.Cloud(pts)
for _ in range(50):
    c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
def run(**kw):
    old = capi.set_config(**kw)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            g = c.fit(0, 0.01, 10000, 1.0, seed=11, copy=False)
        ts.append((time.perf_counter() - t0) / 200 * 1e3)
    capi.restore_config(old)
    print(kw, " ".join(f"{t:.4f}" for t in ts), "ms/step  pairs", g.stats["pairs_scored"], "exact", g.stats["pairs_exact"], flush=True)
for gpb in (2, 4, 8, 12, 16):
    run(score_groups_per_block=gpb)
run(score_fp32_screen=0)
for wgs in (4096, 8192, 32768):
    run(score_min_workgroups=wgs)
for lead in (64, 128, 256):
    run(lead_hypotheses=lead)
