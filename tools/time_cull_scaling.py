import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from misc3d_amd import capi, synth
for n in (250_000, 1_000_000, 4_000_000):
    pts = synth.plane_cloud_c2(n, seed=2)
    c = capi.Cloud(pts)
    for H in (640, 2560, 10000, 16384):
        s = capi.draw_samples(len(pts), 0, H, 11)
        c.time_score(0, 0.01, s, reps=3, mode=1)
        ms, _ = c.time_score(0, 0.01, s, reps=20, mode=1)
        tiles = -(-n // 512)
        print(f"points {n:8d} tiles {tiles:6d} hyp {H:6d}: cull {ms*1e3:7.1f} us  = {tiles*H/ms/1e6:8.1f} M box tests/ms -> {tiles*H/(ms*1e-3)/1e9:.0f} G tests/s")
    c.close()
