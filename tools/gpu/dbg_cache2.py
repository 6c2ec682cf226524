import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from misc3d_amd import capi, synth
rng = np.random.default_rng(31)
g = np.arange(-12, 12) * 0.005
dst = np.stack(np.meshgrid(g, g, g[:6], indexing="ij"), -1).reshape(-1, 3)
partner = rng.choice(len(dst), 2500, replace=False)
src = dst[partner] + rng.choice([0.0, 0.0025], size=(2500, 3))
Tm = synth.rigid_transform(25.0, (0.2, -0.3, 1.0), (0.05, -0.02, 0.01))
src = src @ np.linalg.inv(Tm)[:3, :3].T + np.linalg.inv(Tm)[:3, 3]
cs = rng.integers(0, len(src), 400)
cd = partner[cs]
cd[::3] = rng.integers(0, len(dst), len(cd[::3]))
kw = dict(threshold=0.03, max_iter=2500, edge_length_threshold=0.5, confidence=1.0, seed=5)
old0 = capi.set_config(reg_prune=0)
with capi.RegSession(src, dst, cs, cd, **kw) as sess:
    ch = 0
    while (m := sess.begin_chunk()) is not None:
        recs = {}
        for cc in (2, 0):
            old = capi.set_config(reg_cache=cc, reg_prune=0)
            try:
                c, s = sess.validate(0, m)
            finally:
                capi.restore_config(old)
            recs[cc] = (c.copy(), s.copy())
        (c2, s2), (c0, s0) = recs[2], recs[0]
        bad = np.nonzero(~np.isclose(s2, s0, rtol=1e-12, atol=0))[0]
        print(f"chunk {ch}: m={m} counts_equal={np.array_equal(c2,c0)} mismatches={len(bad)}")
        for i in bad[:6]:
            print("   hyp", i, "count", c2[i], c0[i], "sum cache", repr(s2[i]), "walk", repr(s0[i]), "rel", (s2[i]-s0[i])/s0[i])
        sess.replay(c0, s0)
        ch += 1
    T, st = sess.finish()
print(st)
