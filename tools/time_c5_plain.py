"""C5 (segment_plane_iterative on 10 M points): wall clock of six calls.

--devices N: what N GPUs are good for on this path -- N independent 10 M-point scenes in flight, one thread and one device each
(replicas: no collective; ctypes releases the GIL inside the library call).  Prints scenes per second for 1 and for N devices.
Hypothesis sharding of ONE scene's rounds (m3d_segment_plane_iterative_sharded) is exact and not faster: DESIGN.md 5."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
n_pts = int(os.environ.get("M3D_C5_POINTS", "10000000"))
reps = int(os.environ.get("M3D_C5_REPS", "6"))
n_dev = int(sys.argv[sys.argv.index("--devices") + 1]) if "--devices" in sys.argv else 1
pts = synth.room_cloud_c5(n_pts, 6)
ts = []
for rep in range(reps):
    t0 = time.perf_counter()
    rc, planes, clusters = capi.segment_plane_iterative(pts, 0.01, 1000, 0.05, seed=19)
    ts.append(1e3 * (time.perf_counter() - t0))
    br = capi.last_segment_ms()
print(" ".join(f"{t:.1f}" for t in ts), "ms; clusters", len(clusters), "points", int(sum(len(c) for c in clusters)))
print("last call inside the library:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in br.items()})
if n_dev > 1:
    n_dev = min(n_dev, capi.device_count())
    ref = [np.asarray(c) for c in clusters]
    ok = [True] * n_dev

    def worker(d):
        for _ in range(reps):
            _, _, cl = capi.segment_plane_iterative(pts, 0.01, 1000, 0.05, seed=19, device=d)
            ok[d] = ok[d] and len(cl) == len(ref) and all(np.array_equal(a, b) for a, b in zip(cl, ref))

    for d in range(n_dev):   # (first touch of every device outside the clock)
        capi.segment_plane_iterative(pts[:200000], 0.01, 100, 0.05, seed=1, device=d)
    th = [threading.Thread(target=worker, args=(d,)) for d in range(n_dev)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    one = 1e3 / (sum(ts[1:]) / max(len(ts) - 1, 1))
    print(f"{n_dev} devices, one scene stream each: {n_dev * reps / dt:.1f} scenes/s against {one:.1f} on one device "
          f"({n_dev * reps / dt / one:.2f} x); every result identical to the one-device result: {all(ok)}")
