"""The matcher's exact fall-back under load (200 k x 200 k x 33, C4's matrices): a target row with more candidates than its list holds, and
blocks of duplicated descriptors (flat regions give identical FPFH histograms) -- wall clock and fall-back count per case."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(200000, seed=5)
rng = np.random.default_rng(1)


def run(label, fs, fd):
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        i0, i1 = capi.match_mutual_nn(fs, fd)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{label:58s} {min(ts[1:]):7.2f} ms  matches {len(i0)}  fall-backs {capi.match_last_fallbacks()}", flush=True)


fs, fd = d["feat_src"].copy(), d["feat_dst"].copy()
run("C4 as it is", fs, fd)
fs[1000:1300] = fd[77] + 1e-3 * np.sign(rng.normal(size=(300, 33)))
run("300 source rows at one distance from target row 77", fs, fd)
for dup in (100, 1000, 10000):
    fs, fd = d["feat_src"].copy(), d["feat_dst"].copy()
    for m in (fs, fd):
        src = rng.integers(0, len(m), dup)
        for k in range(4):                        # every chosen row five times in the matrix
            m[rng.integers(0, len(m), dup)] = m[src]
    run(f"{dup} rows of each matrix duplicated four times over", fs, fd)
fs, fd = d["feat_src"].copy(), d["feat_dst"].copy()
fs[:20000] = fs[0]
fd[:20000] = fd[0]
run("20 000 identical rows in each matrix", fs, fd)
