import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from misc3d_amd import capi, synth
n = int(os.environ.get("M3D_C5_POINTS", "10000000"))
pts = synth.room_cloud_c5(n, 6)
for rep in range(2):
    t0 = time.perf_counter(); c = capi.Cloud(pts); t1 = time.perf_counter()
    g = c.fit(0, 0.01, 1000, 0.9999, seed=19); t2 = time.perf_counter()
    print(f"cloud create {1e3*(t1-t0):.1f} ms; one fit (p=0.9999,1000 it) {1e3*(t2-t1):.2f} ms stats={ {k: round(v,3) if isinstance(v,float) else v for k,v in g.stats.items()} }")
    c.close()
    t0 = time.perf_counter(); rc, planes, clusters = capi.segment_plane_iterative(pts, 0.01, 1000, 0.05, seed=19); t1 = time.perf_counter()
    print(f"segment total {1e3*(t1-t0):.1f} ms, clusters {len(clusters)}")
