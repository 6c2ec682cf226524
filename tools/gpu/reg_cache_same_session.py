import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from misc3d_amd import capi, synth
n = 12000
d = synth.registration_pair_c4(n, seed=23, dim=8, true_fraction=0.4, sigma=0.001)
src, dst = d["src"], d["dst"]
inv = np.empty(n, dtype=np.int64); inv[d["perm"]] = np.arange(n)
rng = np.random.default_rng(31)
cs = rng.integers(0, n, 500)
cd = np.where(rng.random(500) < 0.5, inv[cs], rng.integers(0, n, 500))
kw = dict(threshold=0.03, max_iter=2500, edge_length_threshold=0.5, confidence=1.0, seed=5)
with capi.RegSession(src, dst, cs, cd, **kw) as sess:
    ch = 0
    while (m := sess.begin_chunk()) is not None:
        recs = {}
        for cc in (2, 0, 2, 0):
            old = capi.set_config(reg_cache=cc)
            try:
                c, s = sess.validate(0, m)
            finally:
                capi.restore_config(old)
            recs.setdefault(cc, []).append((c.copy(), s.copy()))
        (c2, s2), (c2b, s2b) = recs[2]
        (c0, s0), (c0b, s0b) = recs[0]
        bad = np.nonzero(~np.isclose(s2, s0, rtol=1e-12, atol=0))[0]
        print(f"chunk {ch}: m={m} counts_equal={np.array_equal(c2,c0)} walk_repeatable={np.allclose(s0,s0b,rtol=1e-12,atol=0)} cache_repeatable={np.allclose(s2,s2b,rtol=1e-12,atol=0)} mismatches={len(bad)}")
        for i in bad[:8]:
            print("   hyp", i, "count", c2[i], c0[i], "sum cache", s2[i], "walk", s0[i], "diff", s2[i]-s0[i])
        sess.replay(c0, s0)
        ch += 1
    T, st = sess.finish()
print(st)
