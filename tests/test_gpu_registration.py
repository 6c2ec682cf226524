"""GPU parity tests of the registration half (through the C ABI) against the oracle:
3-point Kabsch + checkers + validation counts bit-exact hypothesis by hypothesis (same "K3x3"
specification on both sides), the RANSAC driver (best index, iteration/validation counts, est_k,
pose), n-point Kabsch within 1e-9, mutual-NN matcher bit-exact."""
import os

import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu


def _problem(n=3000, seed=3, m=800, true_fraction=0.4):
    d = synth.registration_pair_c4(n, seed=seed, dim=8, true_fraction=true_fraction, sigma=0.001)
    inv = np.empty(n, dtype=np.int64)
    inv[d["perm"]] = np.arange(n)
    rng = np.random.default_rng(seed + 1)
    cs = rng.integers(0, n, m)
    cd = np.where(rng.random(m) < true_fraction, inv[cs], rng.integers(0, n, m))
    return d, cs, cd


@pytest.mark.parametrize("conf,max_iter,seed", [(0.999, 3000, 17), (1.0, 700, 5), (0.9, 2000, 2)])
def test_registration_ransac_matches_oracle(capi, orc, conf, max_iter, seed):
    d, cs, cd = _problem()
    o = orc.registration_ransac(d["src"], d["dst"], cs, cd, thr=0.03, max_iter=max_iter, edge_thr=0.9,
                                confidence=conf, seed=seed)
    T, st = capi.registration_ransac(d["src"], d["dst"], cs, cd, threshold=0.03, max_iter=max_iter,
                                     edge_length_threshold=0.9, confidence=conf, seed=seed)
    assert st["best_index"] == o.best_index >= 0
    assert st["iterations"] == o.iterations and st["validations"] == o.validations and st["est_k"] == o.est_k
    assert st["fitness"] == o.fitness
    assert np.array_equal(T.view(np.uint64), o.T.view(np.uint64))     # same hypothesis, same arithmetic
    assert st["inlier_rmse"] == pytest.approx(o.inlier_rmse, rel=1e-12)
    assert np.allclose(T, d["T"], atol=0.02)


@pytest.mark.parametrize("frac,sigma,edge", [(0.4, 0.001, 0.9), (0.25, 0.004, 0.6)])
def test_registration_prune_is_exact(capi, orc, frac, sigma, edge):
    """Clouds large enough (>= 16 source tiles) for the two-phase validation: hypotheses that cannot
    reach the best inlier count of earlier chunks are dropped after every 8th tile.  The result must
    equal the oracle's and the unpruned run's (m3d_config.reg_prune = 0), partial overlap included."""
    n = 24_000
    d = synth.registration_pair_c4(n, seed=21, dim=8, true_fraction=frac, sigma=sigma)
    dst = d["dst"]
    src = d["src"].copy()
    src[src[:, 0] > np.quantile(src[:, 0], 0.65)] += 50.0     # a third of the source has no counterpart
    inv = np.empty(n, dtype=np.int64)
    inv[d["perm"]] = np.arange(n)
    rng = np.random.default_rng(3)
    cs = rng.integers(0, n, 1500)
    cd = np.where(rng.random(1500) < frac, inv[cs], rng.integers(0, n, 1500))
    kw = dict(threshold=0.03, max_iter=3000, edge_length_threshold=edge, confidence=1.0, seed=4)
    T, st = capi.registration_ransac(src, dst, cs, cd, **kw)
    # each optimisation switched off in turn (m3d_config)
    assert st["nn_fp32_screen"] == (1 if st["validations"] > 48 else 0)    # (the lists are built once a call validates in earnest)
    for env, val in (("reg_prune", 0), ("reg_neighbour_lists", 0), ("reg_sorted_lists", 0), ("reg_fp32_screen", 0)):
        old = capi.set_config(**{env: val})
        try:
            T0, st0 = capi.registration_ransac(src, dst, cs, cd, **kw)
        finally:
            capi.restore_config(old)
        assert np.array_equal(T, T0), env
        for k in ("best_index", "iterations", "validations", "est_k", "fitness", "inlier_rmse", "ties"):
            assert st[k] == st0[k], (env, k)
    assert 0.2 < st["fitness"] < 0.9
    o = orc.registration_ransac(src, dst, cs, cd, thr=0.03, max_iter=3000, edge_thr=edge, confidence=1.0, seed=4)
    assert st["best_index"] == o.best_index and st["validations"] == o.validations and st["fitness"] == o.fitness
    assert np.array_equal(T.view(np.uint64), o.T.view(np.uint64))


def test_registration_sorted_lists_are_exact(capi, orc):
    """The x-sorted neighbour lists with their early cut-off (m3d_config.reg_sorted_lists) and the plain lists are
    different ways to the same minimum: T, counts, tie decisions and the oracle's result are reproduced bit for bit."""
    n = 6000
    d = synth.registration_pair_c4(n, seed=9, dim=8, true_fraction=0.4, sigma=0.001)
    inv = np.empty(n, dtype=np.int64)
    inv[d["perm"]] = np.arange(n)
    rng = np.random.default_rng(3)
    cs = rng.integers(0, n, 1500)
    cd = np.where(rng.random(1500) < 0.4, inv[cs], rng.integers(0, n, 1500))
    kw = dict(threshold=0.03, max_iter=3000, edge_length_threshold=0.9, confidence=1.0, seed=4)
    res = {}
    for srt in (0, 1):
        old = capi.set_config(reg_sorted_lists=srt)
        try:
            res[srt] = capi.registration_ransac(d["src"], d["dst"], cs, cd, **kw)
        finally:
            capi.restore_config(old)
    T, st = res[1]
    T2, st2 = res[0]
    assert np.array_equal(T, T2)
    for k in ("best_index", "iterations", "validations", "est_k", "fitness", "inlier_rmse"):
        assert st[k] == st2[k], k
    o = orc.registration_ransac(d["src"], d["dst"], cs, cd, thr=0.03, max_iter=3000, edge_thr=0.9, confidence=1.0, seed=4)
    assert np.array_equal(T, o.T) and st["best_index"] == o.best_index and st["validations"] == o.validations


def _first_chunk_records(capi, src, dst, cs, cd, **kw):
    with capi.RegSession(src, dst, cs, cd, **kw) as sess:
        n = sess.begin_chunk()
        counts, sums = sess.validate(0, n)
        sess.replay(counts, sums)
        while (m := sess.begin_chunk()) is not None:
            c2, s2 = sess.validate(0, m)
            sess.replay(c2, s2)
        T, st = sess.finish()
    return counts.copy(), sums.copy(), T, st


@pytest.mark.parametrize("case", ["lattice", "dups", "near_ties", "far_offset", "tiny_scale", "one_cell", "dense"])
def test_registration_nn_screen_adversarial(capi, orc, case):
    """The fp32 screen of the validation's neighbour search (m3d_config.reg_fp32_screen: 16-byte list entries relative
    to the cell, winner evaluated in fp64, near ties handed to the fp64 walk) must leave every per-hypothesis inlier count
    and every sum of squared distances what the fp64 walk makes them -- each nearest distance is the same double -- on
    inputs built to confuse it: exact ties (lattice, duplicates), ties within 1e-9, scenes beyond
    what fp32 offsets can carry (the screen must stand down), everything in one cell, 100 points per cell."""
    rng = np.random.default_rng(12)
    thr, scale, shift, partner = 0.03, 1.0, np.zeros(3), None      # partner: src row i came from dst row partner[i] (default i)
    if case == "lattice":       # every query midway between lattice points: 2, 4 or 8 nearest neighbours at the SAME distance
        g = np.arange(-12, 12) * 0.005
        dst = np.stack(np.meshgrid(g, g, g[:6], indexing="ij"), -1).reshape(-1, 3)
        partner = rng.choice(len(dst), 2500, replace=False)
        src = dst[partner] + rng.choice([0.0, 0.0025], size=(2500, 3))
    elif case == "dups":
        base = rng.uniform(-0.2, 0.2, (1500, 3)) * [1, 1, 0.02]
        dst = np.concatenate([base, base, base[:500]])
        src = base[:1200] + rng.normal(0, 1e-3, (1200, 3))
    elif case == "near_ties":   # queries on the bisector of two target points, off it by 1e-9 .. 1e-13 of the spacing
        a = rng.uniform(-0.2, 0.2, (1500, 3)) * [1, 1, 0.05]
        d = rng.normal(size=(1500, 3))
        d *= 0.004 / np.linalg.norm(d, axis=1)[:, None]
        dst = np.concatenate([a - d, a + d])
        src = a + d * (10.0 ** rng.uniform(-13, -9, (1500, 1))) * rng.choice([-1, 1], (1500, 1))
    elif case == "far_offset":
        dst = rng.uniform(-0.2, 0.2, (3000, 3)) * [1, 1, 0.02]
        src = dst[:2000] + rng.normal(0, 1e-3, (2000, 3))
        shift = np.array([3.0e5, -2.0e5, 1.0e5])
    elif case == "tiny_scale":
        dst = rng.uniform(-0.2, 0.2, (3000, 3)) * [1, 1, 0.02]
        src = dst[:2000] + rng.normal(0, 1e-3, (2000, 3))
        scale = 1e-18
    elif case == "one_cell":    # the whole target inside one grid cell, the threshold a hundred times its extent
        dst = rng.uniform(0, 1e-4, (300, 3))
        src = dst[:200] + rng.normal(0, 1e-6, (200, 3))
    else:                       # dense: ~100 points per cell, lists of ~900 entries
        dst = rng.uniform(-0.05, 0.05, (60000, 3)) * [1, 1, 0.01]
        src = dst[:3000] + rng.normal(0, 3e-4, (3000, 3))
    Tm = synth.rigid_transform(25.0, (0.2, -0.3, 1.0), (0.05, -0.02, 0.01))
    src = src @ np.linalg.inv(Tm)[:3, :3].T + np.linalg.inv(Tm)[:3, 3]     # dst = Tm(src)
    src, dst, thr = (src + shift) * scale, (dst + shift) * scale, thr * scale
    ns = len(src)
    # correspondences: the partner, a third of them random
    cs = rng.integers(0, ns, 400)
    cd = cs.copy() if partner is None else partner[cs]
    cd[::3] = rng.integers(0, len(dst), len(cd[::3]))
    kw = dict(threshold=thr, max_iter=1500, edge_length_threshold=0.5, confidence=1.0, seed=5)
    res = {}
    for screen in (1, 0):
        old = capi.set_config(reg_fp32_screen=screen)
        try:
            res[screen] = _first_chunk_records(capi, src, dst, cs, cd, **kw)
        finally:
            capi.restore_config(old)
    c1, s1, T1, st1 = res[1]
    c0, s0, T0, st0 = res[0]
    assert len(c1) >= 3 and c1.max() > 0.5 * ns
    # (the sums are order-free: the counting sorts place the points of a cell in the order their atomics land, so the last
    # bits of a sum differ from run to run with or without the screen; inlier_rmse below is the serial-order sum)
    assert np.array_equal(c1, c0) and np.allclose(s1, s0, rtol=1e-12, atol=0.0)
    assert np.array_equal(T1, T0)
    for k in ("best_index", "iterations", "validations", "est_k", "fitness", "inlier_rmse"):
        assert st1[k] == st0[k], k
    assert st0["nn_fp32_screen"] == 0 and st0["nn_screen_fallbacks"] == 0
    if case in ("far_offset", "tiny_scale"):
        assert st1["nn_fp32_screen"] == 0          # offsets from a cell corner would not be good to fp32's last bit
    else:
        assert st1["nn_fp32_screen"] == 1
        if case in ("lattice", "dups"):
            assert st1["nn_screen_fallbacks"] > 100     # the ties were seen and handed over
        # (near_ties: the three-point poses are off by more than the 1e-9 the construction leaves -- few ties survive)
    if case not in ("tiny_scale", "dense"):
        o = orc.registration_ransac(src, dst, cs, cd, thr=thr, max_iter=1500, edge_thr=0.5, confidence=1.0, seed=5)
        assert st1["best_index"] == o.best_index and st1["validations"] == o.validations and st1["fitness"] == o.fitness
        assert np.array_equal(T1.view(np.uint64), o.T.view(np.uint64))


def _all_chunk_records(capi, src, dst, cs, cd, **kw):
    counts, sums = [], []
    with capi.RegSession(src, dst, cs, cd, **kw) as sess:
        while (m := sess.begin_chunk()) is not None:
            c2, s2 = sess.validate(0, m)
            counts.append(c2.copy())
            sums.append(s2.copy())
            sess.replay(c2, s2)
        T, st = sess.finish()
    return np.concatenate(counts), np.concatenate(sums), T, st


@pytest.mark.parametrize("case", ["overlap", "partial", "lattice", "dups", "near_ties", "far_offset", "tiny_scale", "dense", "sparse_target"])
def test_registration_candidate_cache_is_exact(capi, orc, case):
    """The validation's candidate cache (m3d_config.reg_cache, m3d_reg_cache.hip: per source point the 32 target points nearest
    to its position under the incumbent pose, held in registers, with a certificate that every other target point is farther)
    must leave every hypothesis' count and sum what the neighbour-list walk makes them: forced on from the first incumbent
    (reg_cache = 2) against off (0), every chunk's records, on ordinary data (where it must answer most pairs itself), partial
    overlap (source points with nothing near: the no-match certificate), exact ties and duplicates (the identity certificate
    must hand them over), coordinates beyond fp32's reach, scales the fp32 squares cannot carry (the cache must stand down),
    lists of hundreds of entries, and a target far sparser than the source."""
    rng = np.random.default_rng(31)
    thr, scale, shift, partner = 0.03, 1.0, np.zeros(3), None
    n_corr, edge = 500, 0.5
    if case in ("overlap", "partial", "sparse_target"):
        n = 12_000
        d = synth.registration_pair_c4(n, seed=23, dim=8, true_fraction=0.4, sigma=0.001)
        src, dst = d["src"].copy(), d["dst"]
        inv = np.empty(n, dtype=np.int64)
        inv[d["perm"]] = np.arange(n)
        if case == "partial":
            src[src[:, 0] > np.quantile(src[:, 0], 0.6)] += 50.0      # 40 % of the source has no counterpart at all
            src[::7] += 0.033                                          # ... and a seventh sits just outside the radius
        cs = rng.integers(0, n, n_corr)
        cd = np.where(rng.random(n_corr) < 0.5, inv[cs], rng.integers(0, n, n_corr))
        if case == "sparse_target":
            keep = np.sort(rng.choice(n, n // 20, replace=False))      # a twentieth of the target: fewer than 32 points within two cells
            remap = -np.ones(n, dtype=np.int64)
            remap[keep] = np.arange(len(keep))
            dst = dst[keep]
            ok = remap[cd] >= 0
            cs, cd = cs[ok], remap[cd[ok]]
            edge = 0.9
        Tm = None
    elif case == "lattice":
        g = np.arange(-12, 12) * 0.005
        dst = np.stack(np.meshgrid(g, g, g[:6], indexing="ij"), -1).reshape(-1, 3)
        partner = rng.choice(len(dst), 2500, replace=False)
        src = dst[partner] + rng.choice([0.0, 0.0025], size=(2500, 3))
    elif case == "dups":
        base = rng.uniform(-0.2, 0.2, (1500, 3)) * [1, 1, 0.02]
        dst = np.concatenate([base, base, base[:500]])
        src = base[:1200] + rng.normal(0, 1e-3, (1200, 3))
    elif case == "near_ties":
        a = rng.uniform(-0.2, 0.2, (1500, 3)) * [1, 1, 0.05]
        dd = rng.normal(size=(1500, 3))
        dd *= 0.004 / np.linalg.norm(dd, axis=1)[:, None]
        dst = np.concatenate([a - dd, a + dd])
        src = a + dd * (10.0 ** rng.uniform(-13, -9, (1500, 1))) * rng.choice([-1, 1], (1500, 1))
    elif case == "far_offset":
        dst = rng.uniform(-0.2, 0.2, (3000, 3)) * [1, 1, 0.02]
        src = dst[:2000] + rng.normal(0, 1e-3, (2000, 3))
        shift = np.array([3.0e5, -2.0e5, 1.0e5])
    elif case == "tiny_scale":
        dst = rng.uniform(-0.2, 0.2, (3000, 3)) * [1, 1, 0.02]
        src = dst[:2000] + rng.normal(0, 1e-3, (2000, 3))
        scale = 1e-18
    else:                       # dense: ~100 points per cell
        dst = rng.uniform(-0.05, 0.05, (60000, 3)) * [1, 1, 0.01]
        src = dst[:3000] + rng.normal(0, 3e-4, (3000, 3))
    if case not in ("overlap", "partial", "sparse_target"):
        Tm = synth.rigid_transform(25.0, (0.2, -0.3, 1.0), (0.05, -0.02, 0.01))
        src = src @ np.linalg.inv(Tm)[:3, :3].T + np.linalg.inv(Tm)[:3, 3]     # dst = Tm(src)
        src, dst, thr = (src + shift) * scale, (dst + shift) * scale, thr * scale
        cs = rng.integers(0, len(src), 400)
        cd = cs.copy() if partner is None else partner[cs]
        cd[::3] = rng.integers(0, len(dst), len(cd[::3]))
    kw = dict(threshold=thr, max_iter=2500, edge_length_threshold=edge, confidence=1.0, seed=5)
    # Every chunk's records with the pruning phases off -- a hypothesis the phases drop reports the sum over the tiles it saw, and
    # which points share a tile varies from session to session (the source's counting sort places the points of a cell in the
    # order their atomics land) -- and the call's results with them on.
    res = {}
    for prune in (0, 1):
        for cc in (2, 0):
            old = capi.set_config(reg_cache=cc, reg_prune=prune)
            try:
                res[prune, cc] = _all_chunk_records(capi, src, dst, cs, cd, **kw)
            finally:
                capi.restore_config(old)
    c1, s1, T1, st1 = res[0, 2]
    c0, s0, T0, st0 = res[0, 0]
    assert len(c1) >= 20 and c1.max() > {"partial": 0.3, "sparse_target": 0.1}.get(case, 0.5) * len(src)
    assert np.array_equal(c1, c0) and np.allclose(s1, s0, rtol=1e-12, atol=0.0)
    for prune in (0, 1):
        _, _, Ta, sta = res[prune, 2]
        _, _, Tb, stb = res[prune, 0]
        assert np.array_equal(Ta, Tb) and np.array_equal(Ta, T1)
        for k in ("best_index", "iterations", "validations", "est_k", "fitness", "inlier_rmse"):
            assert sta[k] == stb[k] == st1[k], (prune, k)
    assert st0["lds_wave_hypotheses"] == 0 and st0["global_wave_hypotheses"] == 0
    answered, walked = st1["lds_wave_hypotheses"], st1["global_wave_hypotheses"]
    if case == "tiny_scale":
        assert answered == 0 and walked == 0           # the cache stood down (cell edge 1e-20)
    else:
        assert answered + walked > 0
        if case in ("overlap", "partial", "far_offset", "sparse_target"):
            assert answered > 0.3 * (answered + walked), (answered, walked)
    if case not in ("tiny_scale", "dense"):
        o = orc.registration_ransac(src, dst, cs, cd, thr=thr, max_iter=2500, edge_thr=edge, confidence=1.0, seed=5)
        assert st1["best_index"] == o.best_index and st1["validations"] == o.validations and st1["fitness"] == o.fitness
        assert np.array_equal(T1.view(np.uint64), o.T.view(np.uint64))


def test_registration_nn_screen_fallback_rate(capi):
    """On ordinary data the screen decides practically every query itself."""
    d, cs, cd = _problem(n=20000, seed=4, m=1200, true_fraction=0.5)
    T, st = capi.registration_ransac(d["src"], d["dst"], cs, cd, threshold=0.03, max_iter=2000, edge_length_threshold=0.9,
                                     confidence=1.0, seed=3)
    assert st["nn_fp32_screen"] == 1 and st["validations"] > 100
    assert st["nn_screen_fallbacks"] < 2e-4 * st["validations"] * 20000


def test_registration_session_sharded_equals_single_call(capi):
    """m3d_reg session (begin_chunk / validate / replay) driven by distributed.registration_ransac_sharded:
    world 1, and two "ranks" simulated by two identically seeded sessions on this GPU in two threads with a
    barrier-based all-gather -- same pose, same statistics as the one-call m3d_registration_ransac."""
    import threading

    from misc3d_amd import distributed
    d, cs, cd = _problem(n=6000, seed=8, m=1500, true_fraction=0.5)
    kw = dict(threshold=0.03, max_iter=6000, edge_length_threshold=0.9, confidence=1.0, seed=23)
    T0, st0 = capi.registration_ransac(d["src"], d["dst"], cs, cd, **kw)
    assert st0["validations"] > 200       # several chunks with more than one 64-group of survivors
    keys = ("best_index", "iterations", "validations", "est_k", "fitness", "inlier_rmse", "ties")
    with capi.RegSession(d["src"], d["dst"], cs, cd, **kw) as sess:
        T1, st1 = distributed.registration_ransac_sharded(sess)
    assert np.array_equal(T0, T1) and all(st0[k] == st1[k] for k in keys)

    world = 2
    barrier = threading.Barrier(world)
    slots = [None] * world
    out = [None] * world
    err = []

    def run(rank):
        def gather(rec):
            slots[rank] = rec.copy()
            barrier.wait(timeout=120)
            allrec = np.stack(slots)
            barrier.wait(timeout=120)
            return allrec
        try:
            with capi.RegSession(d["src"], d["dst"], cs, cd, **kw) as sess:
                out[rank] = distributed.registration_ransac_sharded(sess, gather=gather, world=world, rank=rank)
        except Exception as e:      # pragma: no cover
            err.append(e)
            barrier.abort()

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not err, err
    for T2, st2 in out:
        assert np.array_equal(T0, T2) and all(st0[k] == st2[k] for k in keys)


def test_registration_source_overhangs_target_bbox(capi, orc):
    """Source points OUTSIDE the target's bounding box but within the threshold of a boundary point are
    correspondences too (a kd-tree radius search has no box): the grid must carry enough pad cells.  Target =
    a square patch; source = the same patch grown by up to one threshold on every side."""
    rng = np.random.default_rng(31)
    thr = 0.05
    dst = np.c_[rng.uniform(0, 1, 4000), rng.uniform(0, 1, 4000), rng.normal(0, 1e-3, 4000)]
    edge = np.c_[rng.uniform(-0.9 * thr, 1 + 0.9 * thr, 3000), rng.uniform(-0.9 * thr, 1 + 0.9 * thr, 3000),
                 rng.uniform(-0.5 * thr, 0.5 * thr, 3000)]
    src = np.concatenate([dst[:2000] + rng.normal(0, 1e-3, (2000, 3)), edge])
    cs = rng.integers(0, 2000, 600)
    cd = cs.copy()
    cd[::4] = rng.integers(0, 4000, len(cd[::4]))
    o = orc.registration_ransac(src, dst, cs, cd, thr=thr, max_iter=300, confidence=1.0, seed=2)
    T, st = capi.registration_ransac(src, dst, cs, cd, threshold=thr, max_iter=300, confidence=1.0, seed=2)
    assert st["best_index"] == o.best_index and st["validations"] == o.validations > 20
    assert st["fitness"] == o.fitness and np.array_equal(T.view(np.uint64), o.T.view(np.uint64))
    outside = ((src[:, :2] < 0) | (src[:, :2] > 1)).any(axis=1)
    cnt, _ = orc.reg_validate(src[outside], dst, np.eye(4).ravel(), thr)
    assert cnt > 100        # the case is real: many overhanging points do have a neighbour


def test_registration_grid_edge_cases(capi, orc):
    # target far from the origin, threshold comparable to the extent, points exactly on cell borders
    rng = np.random.default_rng(7)
    n = 1200
    src = rng.integers(-40, 40, size=(n, 3)) * 0.03003          # multiples of the cell size 1.001 * thr
    T = synth.rigid_transform(10.0, (0, 0, 1), (100.0, -50.0, 7.0))
    dst = src @ T[:3, :3].T + T[:3, 3]
    cs = rng.integers(0, n, 300)
    cd = cs.copy()
    cd[::3] = rng.integers(0, n, len(cd[::3]))
    for thr in (0.03, 0.5):
        o = orc.registration_ransac(src, dst, cs, cd, thr=thr, max_iter=400, confidence=1.0, seed=9)
        Tg, st = capi.registration_ransac(src, dst, cs, cd, threshold=thr, max_iter=400, confidence=1.0, seed=9)
        assert st["best_index"] == o.best_index and st["validations"] == o.validations
        assert st["fitness"] == o.fitness
        assert np.array_equal(Tg.view(np.uint64), o.T.view(np.uint64))


def test_registration_errors_and_identity(capi):
    rng = np.random.default_rng(0)
    p = rng.normal(size=(10, 3))
    with pytest.raises(capi.M3DError) as e:
        capi.registration_ransac(p[:2], p, [0, 1, 1], [0, 1, 2], seed=1)
    assert e.value.code == capi.ERR_TOO_FEW_POINTS and "less than 3" in str(e.value)
    T, st = capi.registration_ransac(p, p, [0, 1], [0, 1], seed=1)              # < 3 correspondences
    assert np.array_equal(T, np.eye(4)) and st["best_index"] == -1
    T, st = capi.registration_ransac(p, p, [0, 1, 2], [0, 1, 2], threshold=0.0, seed=1)
    assert np.array_equal(T, np.eye(4))
    with pytest.raises(capi.M3DError) as e:
        capi.registration_ransac(p, p, [0, 1, 99], [0, 1, 2], seed=1)
    assert e.value.code == capi.ERR_INVALID_ARG


@pytest.mark.parametrize("n", [3, 4, 1000, 100_000])
def test_kabsch_matches_oracle(capi, orc, n):
    rng = np.random.default_rng(n)
    T = synth.rigid_transform(25.0, (0.3, -1, 0.5), (1.0, 2.0, -0.5))
    src = rng.uniform(-2, 2, (n, 3))
    dst = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (n, 3)) * (n > 4)
    for scaling in (False, True):
        got = capi.kabsch(src, dst, scaling)
        ref = orc.umeyama(src, dst, scaling)
        assert np.allclose(got, ref, rtol=0, atol=1e-9)
    if n <= 4:
        assert np.allclose(capi.kabsch(src, dst), T, atol=1e-12)
    with pytest.raises(capi.M3DError) as e:
        capi.kabsch(src[:2], dst[:2])
    assert e.value.code == capi.ERR_TOO_FEW_POINTS


@pytest.mark.parametrize("dim,ns,nd", [(33, 3000, 2700), (33, 257, 5000), (8, 1500, 1500), (3, 900, 1000)])
def test_mutual_nn_matches_oracle(capi, orc, dim, ns, nd):
    rng = np.random.default_rng(dim + ns)
    fs = rng.uniform(0, 1, (ns, dim))
    fd = rng.uniform(0, 1, (nd, dim))
    k = min(ns, nd) // 3
    fd[:k] = np.abs(fs[ns - k:] + rng.normal(0, 0.01, (k, dim)))
    fd[k:k + 5] = fd[:5]                      # exact duplicates: ties go to the lowest index
    a, b = capi.match_mutual_nn(fs, fd)
    oa, ob = orc.match_mutual_nn(fs, fd)
    assert np.array_equal(a.astype(np.int64), oa) and np.array_equal(b.astype(np.int64), ob)
    assert len(a) > k // 2


def _brute(capi, fs, fd):
    """the fp64 brute-force kernel (m3d_config.match_brute): the unscreened GPU path"""
    old = capi.set_config(match_brute=1)
    try:
        return capi.match_mutual_nn(fs, fd)
    finally:
        capi.restore_config(old)


@pytest.mark.parametrize("screen", ["mfma", "fp32"])
@pytest.mark.parametrize("case", ["dups", "lattice", "huge", "tiny", "nan", "offset", "scales", "mixed"])
def test_mutual_nn_screen_adversarial(capi, orc, case, screen):
    """dim-33 inputs built to defeat the reduced-precision screens (split-fp16 MFMA, fp32 VALU): long runs of exact ties (ring eviction -> exact
    fallback), near-equal distances below fp32 resolution, values outside the fp32 range, NaNs,
    large common offsets (catastrophic cancellation in |a|^2+|b|^2-2ab).  The result must stay the
    exact fp64 answer: equal to the oracle and to the unscreened brute-force kernel."""
    rng = np.random.default_rng(len(case))
    ns, nd, dim = 700, 900, 33
    fs = rng.uniform(0, 1, (ns, dim))
    fd = rng.uniform(0, 1, (nd, dim))
    fd[:300] = fs[:300] + rng.normal(0, 1e-3, (300, dim))
    expect_fallback = False
    if case == "dups":            # 200 identical rows on both sides: every one of them ties
        fd[100:300] = fd[100]
        fs[400:600] = fs[400]
        fs[10] = fd[100]
        expect_fallback = True
    elif case == "lattice":       # distances differ only far below the fp32 error bound
        base = rng.uniform(0, 1, dim)
        fd[:] = base + 1e-9 * rng.integers(-3, 4, (nd, dim))
        fs[:] = base + 1e-9 * rng.integers(-3, 4, (ns, dim))
        expect_fallback = True
    elif case == "huge":          # squares overflow fp32 (the MFMA screen rescales by a power of two instead)
        fs *= 1e25
        fd *= 1e25
        expect_fallback = screen == "fp32"
    elif case == "tiny":          # squares underflow fp32
        fs *= 1e-25
        fd *= 1e-25
        expect_fallback = screen == "fp32"
    elif case == "nan":
        fs[5, 7] = np.nan
        fd[9, 0] = np.nan
        fd[11] = np.nan
    elif case == "offset":        # |a|^2 ~ 3e9 while neighbour distances are ~1: fp32 cancels completely
        fs += 1e4
        fd += 1e4
    elif case == "scales":        # one huge-norm database row inflates the bound of every query, and pushes every
        fd[500] *= 1e6            # other value below fp16 resolution after the common power-of-two scaling
    elif case == "mixed":         # six decades of dynamic range inside every row
        fs[:, ::2] *= 1e-6
        fd[:, ::2] *= 1e-6
        fs[:, 1::4] *= 1e-3
        fd[:, 1::4] *= 1e-3
    old = capi.set_config(match_fp32_screen=1 if screen == "fp32" else 0)
    try:
        a, b = capi.match_mutual_nn(fs, fd)
    finally:
        capi.restore_config(old)
    falls = capi.match_last_fallbacks()
    oa, ob = orc.match_mutual_nn(fs, fd)
    assert np.array_equal(a.astype(np.int64), oa) and np.array_equal(b.astype(np.int64), ob)
    ba, bb = _brute(capi, fs, fd)
    assert np.array_equal(a, ba) and np.array_equal(b, bb)
    if expect_fallback:
        assert falls > 0


@pytest.mark.parametrize("case", ["row_overflow", "hub_query", "ragged", "tiny_sides"])
def test_mutual_nn_one_pass_reverse_lists(capi, orc, case):
    """The MFMA screen finds BOTH directions in one scan: the target rows' candidates are collected in per-lane lists
    (8 entries per query and slice), binned by row (256 slots) and verified in fp64.  Built to overflow each of them:
    row_overflow -- 700 source rows at the same distance from one target row (more candidates than slots: the row takes
    the exact fallback); hub_query -- one source row that is the nearest of 400 target rows (its lane lists overflow:
    direct appends); ragged / tiny_sides -- sizes that are no multiple of the 32-row tiles, a side smaller than a tile."""
    rng = np.random.default_rng(7 + len(case))
    dim = 33
    if case == "row_overflow":
        ns, nd = 3000, 1200
        fs = rng.uniform(0, 1, (ns, dim))
        fd = rng.uniform(0, 1, (nd, dim))
        fs[1000:1700] = fd[77] + 1e-3 * np.sign(rng.normal(size=(700, dim)))      # 700 rows at one exact distance from target 77
    elif case == "hub_query":
        ns, nd = 2500, 3000
        fs = rng.uniform(0, 1, (ns, dim))
        fd = rng.uniform(0, 1, (nd, dim))
        fd[500:900] = fs[42] + rng.normal(0, 1e-4, (400, dim))                     # source 42 is the nearest of 400 targets
    elif case == "ragged":
        ns, nd = 1033, 2051
        fs = rng.uniform(0, 1, (ns, dim))
        fd = rng.uniform(0, 1, (nd, dim))
        fd[:500] = fs[:500] + rng.normal(0, 1e-3, (500, dim))
    else:
        ns, nd = 5, 1500
        fs = rng.uniform(0, 1, (ns, dim))
        fd = rng.uniform(0, 1, (nd, dim))
    for a_, b_ in ((fs, fd), (fd, fs)):                                            # each side once as the scan's queries
        a, b = capi.match_mutual_nn(a_, b_)
        falls = capi.match_last_fallbacks()
        oa, ob = orc.match_mutual_nn(a_, b_)
        assert np.array_equal(a.astype(np.int64), oa) and np.array_equal(b.astype(np.int64), ob)
        if case == "row_overflow" and a_ is fs:
            assert falls >= 1                                                      # target row 77 could not keep its candidates


def test_mutual_nn_screen_is_used(capi):
    """on ordinary descriptors the screen decides (almost) every query without the fallback"""
    d = synth.registration_pair_c4(20_000, seed=9)
    a, b = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    assert capi.match_last_fallbacks() < 40      # (of 40 000 searches: both directions)
    ba, bb = _brute(capi, d["feat_src"], d["feat_dst"])
    assert np.array_equal(a, ba) and np.array_equal(b, bb)


def test_c4_pipeline_properties(capi):
    """BASELINE config C4 shape at reduced N (20k): matcher -> RANSAC recovers the generating pose;
    determinism for a fixed seed."""
    d = synth.registration_pair_c4(20_000, seed=5)
    i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    assert len(i0) > 0.2 * 20_000
    inv = np.empty(20_000, dtype=np.int64)
    inv[d["perm"]] = np.arange(20_000)
    true_frac = np.mean(inv[i0.astype(np.int64)] == i1.astype(np.int64))
    assert true_frac > 0.5
    T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=20_000,
                                     confidence=1.0, seed=17)
    assert st["iterations"] == 20_000 and st["fitness"] > 0.95
    assert np.allclose(T, d["T"], atol=5e-3)
    T2, st2 = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=20_000,
                                       confidence=1.0, seed=17)
    assert np.array_equal(T, T2) and st2["best_index"] == st["best_index"]


@pytest.mark.parametrize("case", ["refine", "identity_far", "no_convergence", "duplicates", "nonfinite"])
def test_registration_icp_matches_oracle(capi, orc, case):
    """Point-to-point ICP (what the reference's examples chain on the RANSAC pose): same
    correspondence set (index work: bit-exact, lowest target index on exact ties), same iteration count, same
    fitness; pose within 1e-9 (the n-point Kabsch sums are order-free on the GPU, serial in the oracle)."""
    d = synth.registration_pair_c4(6000, seed=11, dim=8, sigma=0.001)
    src, dst = d["src"].copy(), d["dst"].copy()
    init = d["T"].copy()
    init[:3, 3] += np.array([0.012, -0.009, 0.006])
    kw = dict(max_iteration=30, relative_fitness=1e-6, relative_rmse=1e-6)
    max_dist = 0.02
    if case == "identity_far":       # poor overlap: the iteration is chaotic (1e-16 differences in an update grow), so
        init = None                  # only the evaluation of the initial pose and ONE update are compared
        max_dist = 0.05
        kw["max_iteration"] = 1
    elif case == "no_convergence":
        kw["max_iteration"] = 2
    elif case == "duplicates":
        dst = np.concatenate([dst, dst[:2000], dst[100:1100]])        # exact ties: the lowest index must win
    elif case == "nonfinite":
        src[[3, 500]] = np.nan
        dst[[7, 900]] = np.inf
        dst[11, 2] = np.nan
    T, st, corr = capi.registration_icp(src, dst, max_dist, init, want_correspondences=True, **kw)
    oT, ofit, orm, oit, ocorr = orc.registration_icp(src, dst, max_dist, init, max_iter=kw["max_iteration"],
                                                     rel_fitness=1e-6, rel_rmse=1e-6)
    assert st["iterations"] == oit
    assert np.array_equal(corr, ocorr)
    assert st["correspondences"] == int((ocorr >= 0).sum()) and st["fitness"] == ofit
    assert st["inlier_rmse"] == pytest.approx(orm, rel=1e-9, abs=1e-15)
    assert np.allclose(T, oT, rtol=0, atol=1e-9)
    if case == "refine":
        assert st["converged"] == 1 and st["fitness"] > 0.99
        assert np.abs(T - d["T"]).max() < 1e-3
    if case == "duplicates":
        assert (corr[corr >= 0] < len(d["dst"])).all()
    with pytest.raises(capi.M3DError):
        capi.registration_icp(src, dst, 0.0, init)


def test_information_matrix_and_global_registration(capi, orc):
    """GetInformationMatrixFromPointClouds (the acceptance test of GlobalRegistration): the
    correspondence count info(5,5) is exact, the moment sums agree with the serial-order oracle to 1e-9
    relative; the composed global_registration accepts the true pose and rejects a wrong one."""
    d = synth.registration_pair_c4(5000, seed=13, dim=33, sigma=0.001)
    for T in (d["T"], np.eye(4)):
        info, nc = capi.information_matrix(d["src"], d["dst"], 0.03, T)
        ref = orc.information_matrix(d["src"], d["dst"], 0.03, T)
        assert nc == int(ref[5, 5]) == int(info[5, 5]) and info[3, 3] == info[4, 4] == nc
        assert np.allclose(info, ref, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(ref).max()))
        assert np.array_equal(info, info.T)
    ok, pose, info = capi.global_registration(d["src"], d["dst"], d["feat_src"], d["feat_dst"], voxel_size=0.03 / 1.4,
                                              max_iter=3000, seed=3)
    assert ok and np.abs(pose - d["T"]).max() < 0.02 and info[5, 5] > 0.9 * 5000
    far = d["dst"] + 100.0                                     # nothing overlaps: RANSAC finds no support
    ok2, pose2, info2 = capi.global_registration(d["src"], far, d["feat_src"], np.random.default_rng(0).uniform(
        0, 1, d["feat_dst"].shape), voxel_size=0.03 / 1.4, max_iter=500, seed=3)
    assert (not ok2 and np.array_equal(info2, np.eye(6))) or np.allclose(pose2, np.eye(4))
