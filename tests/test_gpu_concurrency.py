"""The boundary's threading clause (SURVEY.md 8(b) "Threading": re-entrant C ABI) and GlobalRegistration (8(f) N2) on the GPU,
against the oracle:

* m3d_global_registration = ReconstructionPipeline::GlobalRegistration (src/pipeline.cpp:790-828) vs the oracle's restatement;
* m3d_global_registration_batch = BuildPoseGraphForScene's one std::thread per fragment pair (src/pipeline.cpp:428-439) vs
  the same pairs one call at a time;
* >= 8 host threads firing mixed m3d_match_mutual_nn / m3d_registration_ransac / m3d_cloud_fit / m3d_fit_* /
  m3d_segment_plane_iterative / m3d_global_registration calls at ONE device (lanes, m3d_driver.hpp), every result compared
  with the oracle's, computed beforehand on one thread."""
import threading

import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu


def _pair(n, seed, dim=33, true_fraction=0.3, sigma=0.001):
    d = synth.registration_pair_c4(n, seed=seed, dim=dim, true_fraction=true_fraction, sigma=sigma)
    return d["src"], d["dst"], d["feat_src"], d["feat_dst"], d["T"]


def _same_result(g, o, n_min):
    ok, T, info = g[:3]
    ook, oT, oinfo = o[:3]
    assert ok == ook
    assert np.array_equal(T.view(np.uint64), oT.view(np.uint64))          # same hypothesis, same arithmetic
    assert info[5, 5] == oinfo[5, 5]                                      # the correspondence count is exact
    assert np.array_equal(info == 0.0, oinfo == 0.0)                      # the structural zeros are zeros
    assert np.allclose(info, oinfo, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(oinfo).max()))
    if ok and not np.array_equal(oinfo, np.eye(6)):
        assert info[5, 5] / n_min >= 0.3


@pytest.mark.parametrize("n,seed,max_iter,conf", [(4000, 13, 3000, 0.999), (6000, 2, 1500, 1.0), (2500, 7, 800, 0.999)])
def test_global_registration_matches_oracle(capi, orc, n, seed, max_iter, conf):
    src, dst, fs, fd, T_true = _pair(n, seed)
    vox = 0.03 / 1.4
    o = orc.global_registration(src, dst, fs, fd, vox, max_iter=max_iter, confidence=conf, seed=seed + 1)
    g = capi.global_registration(src, dst, fs, fd, vox, max_iter=max_iter, confidence=conf, seed=seed + 1, want_stats=True)
    _same_result(g, o, n)
    assert g[3]["n_matches"] == o[3] > 100
    assert g[0] and np.abs(g[1] - T_true).max() < 0.02 and g[3]["identity_shortcut"] == 0
    assert g[3]["n_info_correspondences"] == g[2][5, 5] > 0.9 * n


def test_global_registration_rejects_and_shortcuts(capi, orc):
    """(false, pose, I6) when the pose explains too little of the clouds (pipeline.cpp:821-824); (true, I4, I6) when the
    solver returns the identity (:814-816) -- here: fewer than 3 mutual matches, Open3D's default RegistrationResult;
    the solver's LogError (fewer than 3 points) is an error code."""
    src, dst, fs, fd, _ = _pair(3000, 5)
    vox = 0.03 / 1.4
    # a third of the source overlaps the target: the pose is found, the information matrix says no
    cut = src.copy()
    cut[800:] = np.random.default_rng(1).uniform(40.0, 60.0, size=(len(src) - 800, 3))
    o = orc.global_registration(cut, dst, fs, fd, vox, max_iter=2000, seed=9)
    g = capi.global_registration(cut, dst, fs, fd, vox, max_iter=2000, seed=9, want_stats=True)
    _same_result(g, o, 3000)
    assert not g[0] and np.array_equal(g[2], np.eye(6)) and not np.array_equal(g[1], np.eye(4))
    assert 0 < g[3]["n_info_correspondences"] < 0.3 * 3000
    # descriptors without mutual matches -> identity -> accepted with I6 (the reference's behaviour, not a judgement)
    fd0 = np.zeros_like(fd)
    fd0[:, 0] = np.arange(len(fd)) + 10.0
    fs0 = np.zeros_like(fs)
    fs0[:, 1] = np.arange(len(fs)) + 10.0
    o = orc.global_registration(src, dst, fs0, fd0, vox, max_iter=500, seed=1)
    g = capi.global_registration(src, dst, fs0, fd0, vox, max_iter=500, seed=1, want_stats=True)
    assert o[3] == g[3]["n_matches"] <= 2
    assert g[0] and o[0] and np.array_equal(g[1], np.eye(4)) and np.array_equal(g[2], np.eye(6))
    assert g[3]["identity_shortcut"] == 1
    with pytest.raises(capi.M3DError, match="less than 3"):
        capi.global_registration(src[:2], dst, fs[:2], fd, vox)
    with pytest.raises(ValueError):
        orc.global_registration(src[:2], dst, fs[:2], fd, vox)


def test_global_registration_batch_equals_single_calls(capi, orc):
    """Eight pairs of different sizes through m3d_global_registration_batch (devices = [0], four in flight) == the same pairs
    one call at a time == the oracle (two of them); more than one lane was used; a bad pair fails alone."""
    vox = 0.03 / 1.4
    sizes = [3000, 5000, 2000, 4000, 2500, 3500, 6000, 1500]
    pairs = [_pair(n, 40 + k)[:4] for k, n in enumerate(sizes)]
    seeds = [100 + k for k in range(len(pairs))]
    single = [capi.global_registration(*p, vox, max_iter=1500, seed=s) for p, s in zip(pairs, seeds)]
    batch = capi.global_registration_batch(pairs, vox, max_iter=1500, seeds=seeds, devices=(0,), inflight=4, want_stats=True)
    for k, (b, s) in enumerate(zip(batch, single)):
        assert b[0] == s[0] and np.array_equal(b[1], s[1]) and np.array_equal(b[2], s[2]), k
    for k in (1, 6):
        o = orc.global_registration(*pairs[k], vox, max_iter=1500, seed=seeds[k])
        _same_result(batch[k], o, sizes[k])
    assert len({b[3]["lane"] for b in batch}) >= 2 and all(b[3]["device"] == 0 for b in batch)
    one = capi.global_registration_batch(pairs[:3], vox, max_iter=1500, seeds=seeds[:3], devices=(0,), inflight=1)
    for b, s in zip(one, single[:3]):
        assert np.array_equal(b[1], s[1]) and np.array_equal(b[2], s[2])
    assert capi.global_registration_batch([], vox) == []
    bad = list(pairs[:3])
    bad[1] = (bad[1][0][:2], bad[1][1], bad[1][2][:2], bad[1][3])
    with pytest.raises(capi.M3DError, match="pair 1.*less than 3"):
        capi.global_registration_batch(bad, vox, max_iter=500, seeds=seeds[:3])


def test_register_fragment_pairs_resident_equals_pairwise_calls(capi, orc):
    """m3d_register_fragment_pairs (fragments resident for the call, pairs by index: src/pipeline.cpp:415-440) == the same pairs
    through m3d_global_registration from host arrays, bit for bit; one pair against the oracle; dim != 33 (brute-force matcher,
    no cached maximum) as well; the shipped python API on top of it."""
    vox = 0.03 / 1.4
    for dim in (33, 8):
        base = synth.registration_pair_c4(3000, seed=70 + dim, dim=dim, sigma=0.001)
        # five "fragments": the source and four differently transformed / permuted / noised copies of it
        rng = np.random.default_rng(dim)
        frags, feats = [base["src"]], [base["feat_src"]]
        for k in range(4):
            Tk = synth.rigid_transform(20.0 + 15.0 * k, (1, k + 1, 2), (0.1 * k, -0.2, 0.05 * k))
            perm = rng.permutation(3000)
            frags.append(np.ascontiguousarray((base["src"] @ Tk[:3, :3].T + Tk[:3, 3] + rng.normal(0, 0.001, (3000, 3)))[perm]))
            feats.append(np.ascontiguousarray(np.abs(base["feat_src"] + rng.normal(0, 0.01, base["feat_src"].shape))[perm]))
        pairs = [(s, t) for s in range(5) for t in range(s + 1, 5)]
        seeds = [300 + k for k in range(len(pairs))]
        res = capi.register_fragment_pairs(frags, feats, pairs, vox, max_iter=1500, seeds=seeds, inflight=4, want_stats=True)
        for (s, t), sd, r in zip(pairs, seeds, res):
            g = capi.global_registration(frags[s], frags[t], feats[s], feats[t], vox, max_iter=1500, seed=sd)
            assert r[0] == g[0] and np.array_equal(r[1], g[1]) and np.array_equal(r[2], g[2]), (dim, s, t)
            assert r[0] and r[3]["n_matches"] > 2500
        o = orc.global_registration(frags[1], frags[3], feats[1], feats[3], vox, max_iter=1500, seed=seeds[pairs.index((1, 3))])
        _same_result(res[pairs.index((1, 3))], o, 3000)
        if dim == 33:
            import misc3d_amd as m3d
            api = m3d.reconstruction.register_fragment_pairs(frags, [f.T for f in feats], pairs=pairs, voxel_size=vox, max_iter=1500,
                                                             seeds=seeds)      # (dim, N) matrices, as open3d Feature.data
            assert [(a[0], a[1]) for a in api] == pairs
            # the default: loop closures only -- adjacent fragments are the odometry's (src/pipeline.cpp:752-764)
            dflt = m3d.reconstruction.register_fragment_pairs(frags, [f.T for f in feats], voxel_size=vox, max_iter=300)
            assert [(a[0], a[1]) for a in dflt] == [(s, t) for (s, t) in pairs if t > s + 1]
            for a, r in zip(api, res):
                assert a[2] == r[0] and np.array_equal(a[3], r[1]) and np.array_equal(a[4], r[2])
            ok, T, info = m3d.reconstruction.global_registration(frags[0], frags[2], feats[0], feats[2], vox, 1500, seed=seeds[1])
            assert ok == res[1][0] and np.array_equal(T, res[1][1]) and np.array_equal(info, res[1][2])
    with pytest.raises(capi.M3DError, match="out of range"):
        capi.register_fragment_pairs(frags, feats, [(0, 7)], vox)
    assert capi.register_fragment_pairs(frags, feats, [], vox) == []


def test_concurrent_mixed_calls_match_the_oracle(capi, orc):
    """The pipeline.cpp:428-439 pattern: ten host threads, each running its own mix of entry points against device 0 at the
    same time, several rounds; every result is compared with the oracle's."""
    rng = np.random.default_rng(0)
    vox = 0.03 / 1.4
    jobs = []          # (name, run() -> result, check(result))

    def add_fit(kind, pts, nrm, seed, resident):
        o = orc.fit(kind, pts, nrm, thr=0.01, max_iter=300, prob=0.9999, seed=seed)

        def check(g):
            assert g.ret == o.ret and g.stats["best_index"] == o.best_index
            assert np.array_equal(g.inliers, o.inliers)
            assert np.allclose(g.params, o.params, rtol=0, atol=1e-9)
        if resident:
            cloud = capi.Cloud(pts, nrm)
            jobs.append((f"cloud_fit{kind}", lambda: cloud.fit(kind, 0.01, 300, 0.9999, seed=seed), check))
        else:
            jobs.append((f"fit{kind}", lambda: capi.fit(kind, pts, nrm, 0.01, 300, 0.9999, seed=seed), check))

    add_fit(0, synth.plane_cloud_c1(30000, 1), None, 7, True)
    add_fit(0, synth.plane_cloud_c1(20000, 2), None, 8, False)
    add_fit(1, synth.sphere_cloud_c3(25000, 4), None, 9, True)
    cp, cn = synth.cylinder_cloud_c3(20000, 3)
    add_fit(2, cp, cn, 10, False)

    room = synth.room_cloud_c5(60000, 6)
    oseg = orc.segment_plane_iterative(room, 0.02, 100, 0.05, seed=19)

    def check_seg(g):
        _rc, planes, clusters = g
        assert len(clusters) == len(oseg[2]) >= 4
        for a, b in zip(clusters, oseg[2]):
            assert np.array_equal(a, b)
        assert np.allclose(planes, oseg[1], rtol=0, atol=1e-9)
    jobs.append(("segment", lambda: capi.segment_plane_iterative(room, 0.02, 100, 0.05, seed=19), check_seg))

    src, dst, fs, fd, _ = _pair(4000, 21)
    om = orc.match_mutual_nn(fs, fd)

    def check_match(g):
        assert np.array_equal(g[0], om[0]) and np.array_equal(g[1], om[1])
    jobs.append(("match", lambda: capi.match_mutual_nn(fs, fd), check_match))

    cs, cd = om
    oreg = orc.registration_ransac(src, dst, cs, cd, thr=0.03, max_iter=2000, edge_thr=0.9, confidence=0.999, seed=5)

    def check_reg(g):
        T, st = g
        assert st["best_index"] == oreg.best_index and st["validations"] == oreg.validations
        assert np.array_equal(T.view(np.uint64), oreg.T.view(np.uint64))
    jobs.append(("ransac", lambda: capi.registration_ransac(src, dst, cs, cd, threshold=0.03, max_iter=2000,
                                                            edge_length_threshold=0.9, confidence=0.999, seed=5), check_reg))

    ogr = orc.global_registration(src, dst, fs, fd, vox, max_iter=1500, seed=3)
    jobs.append(("global", lambda: capi.global_registration(src, dst, fs, fd, vox, max_iter=1500, seed=3),
                 lambda g: _same_result(g, ogr, 4000)))

    oinfo = orc.information_matrix(src, dst, 0.03, oreg.T)

    def check_info(g):
        assert g[1] == int(oinfo[5, 5]) and np.allclose(g[0], oinfo, rtol=1e-9, atol=1e-9 * np.abs(oinfo).max())
    jobs.append(("info", lambda: capi.information_matrix(src, dst, 0.03, oreg.T), check_info))

    import os
    n_threads, rounds = 10, int(os.environ.get("M3D_CONCURRENCY_ROUNDS", "6"))   # (a soak: M3D_CONCURRENCY_ROUNDS=100)
    start = threading.Barrier(n_threads)
    errors, done = [], [0] * n_threads

    def run(t):
        try:
            order = np.random.default_rng(100 + t).permutation(len(jobs) * rounds) % len(jobs)
            start.wait(timeout=120)
            for j in order:
                name, call, check = jobs[j]
                try:
                    check(call())
                except Exception as e:       # noqa: BLE001
                    raise AssertionError(f"thread {t}, job {name}: {e!r}") from e
                done[t] += 1
        except Exception as e:               # noqa: BLE001
            errors.append(e)
            start.abort()

    ths = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=600)
    assert not errors, errors[:3]
    assert all(d == len(jobs) * rounds for d in done), done
    del rng


def test_lanes_one_is_the_serial_library(capi, orc):
    """m3d_config.lanes = 1: one call at a time per device (rounds 1-4's behaviour) -- same results, one lane."""
    vox = 0.03 / 1.4
    pairs = [_pair(2500, 60 + k)[:4] for k in range(4)]
    seeds = [7, 8, 9, 10]
    ref = capi.global_registration_batch(pairs, vox, max_iter=800, seeds=seeds, inflight=4, want_stats=True)
    old = capi.set_config(lanes=1)
    try:
        one = capi.global_registration_batch(pairs, vox, max_iter=800, seeds=seeds, inflight=4, want_stats=True)
    finally:
        capi.restore_config(old)
    assert {b[3]["lane"] for b in one} == {0}
    for a, b in zip(ref, one):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_set_config_while_other_threads_compute(capi, orc):
    """m3d_set_config from one thread while four others fit: a call works with the settings as they were when it
    took its lane -- a snapshot per call -- so flipping the scoring switches (histogram bound, phases, pre-stream, fp32 screen, box
    tests) under running fits changes which kernels the NEXT calls use and never a result.  Every fit against the oracle's."""
    import threading
    clouds = {0: (synth.plane_cloud_c1(150_000, 3), None), 2: synth.cylinder_cloud_c3(120_000, 5), 1: (synth.sphere_cloud_c3(100_000, 4), None)}
    H = {0: 12_000, 1: 6_000, 2: 9_000}
    ref = {k: orc.fit(k, p, n, thr=0.01, max_iter=H[k], prob=1.0, seed=31) for k, (p, n) in clouds.items()}
    resident = {k: capi.Cloud(p, n) for k, (p, n) in clouds.items()}
    stop = threading.Event()
    errors, done = [], [0]
    old = capi.get_config()

    def flipper():
        rng = np.random.default_rng(1)
        while not stop.is_set():
            capi.set_config(plane_bound=int(rng.integers(0, 3)), score_phases=int(rng.choice([-1, 0, 2, 3])),
                            prestream=int(rng.integers(0, 2)), score_fp32_screen=int(rng.integers(0, 2)),
                            cull_fp32=int(rng.integers(0, 2)), first_chunk=int(rng.choice([0, 2048, 4096])),
                            speculative_refine=int(rng.integers(0, 2)))

    def fitter(tid):
        try:
            for it in range(12):
                k = (tid + it) % 3
                g = resident[k].fit(k, 0.01, H[k], 1.0, seed=31)
                o = ref[k]
                if not (g.ret == o.ret and g.stats["best_index"] == o.best_index and np.array_equal(g.inliers, o.inliers)
                        and np.allclose(g.params, o.params, rtol=0, atol=1e-9)):
                    errors.append((tid, it, k, g.stats["best_index"], o.best_index))
                done[0] += 1
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    fl = threading.Thread(target=flipper)
    th = [threading.Thread(target=fitter, args=(t,)) for t in range(4)]
    fl.start()
    for t in th:
        t.start()
    for t in th:
        t.join()
    stop.set()
    fl.join()
    capi.restore_config(old)
    for c in resident.values():
        c.close()
    assert not errors, errors[:5]
    assert done[0] == 48


def test_fit_batch_equals_single_fits(capi, orc):
    """m3d_cloud_fit_batch (python/py_common.cpp:11-78's callers' loops as ONE call): clouds placed on lanes of the caller's
    choice (m3d_cloud_create_lane), 36 jobs of mixed kinds -- every result equal to m3d_cloud_fit on the same cloud, three of them
    to the oracle's; jobs of one cloud run in the order given (a removal between two of them would be seen); a bad job fails with
    its index and the others keep their results; probability errors come back per job."""
    data = [(0, synth.plane_cloud_c1(60_000, 1), None), (1, synth.sphere_cloud_c3(50_000, 4), None),
            (2,) + synth.cylinder_cloud_c3(50_000, 3), (0, synth.plane_cloud_c1(200_000, 7), None)]
    clouds = [capi.Cloud(p, n, lane=k % capi.get_config().lanes) for k, (_, p, n) in enumerate(data)]
    jobs = []
    for rep in range(9):
        for k, (kind, _, _) in enumerate(data):
            jobs.append((clouds[k], kind, 0.01, 800 + 100 * rep, 0.9999 if rep % 2 else 1.0, 40 + rep))
    res = capi.fit_batch(jobs, inflight=0)
    assert len(res) == len(jobs)
    for (c, kind, thr, it, prob, seed), r in zip(jobs, res):
        one = c.fit(kind, thr, it, prob, seed=seed)
        assert r.ret == one.ret and r.stats["best_index"] == one.stats["best_index"] and r.stats["iterations"] == one.stats["iterations"]
        assert np.array_equal(r.inliers, one.inliers) and np.array_equal(r.params, one.params)
    for k in (0, 1, 2):
        kind, p, n = data[k]
        o = orc.fit(kind, p, n, thr=0.01, max_iter=800, prob=1.0, seed=40)
        assert res[k].stats["best_index"] == o.best_index and np.array_equal(res[k].inliers, o.inliers)
    # one lane at a time gives the same
    res1 = capi.fit_batch(jobs[:8], inflight=1)
    for a, b in zip(res[:8], res1):
        assert np.array_equal(a.inliers, b.inliers) and np.array_equal(a.params, b.params)
    assert capi.fit_batch([]) == []
    bad = list(jobs[:4])
    bad[2] = (clouds[2], 2, 0.01, 100, 1.5, 1)          # SetProbability's range check, job 2
    with pytest.raises(capi.M3DError, match="job 2"):
        capi.fit_batch(bad)
    with pytest.raises(capi.M3DError):
        capi.Cloud(data[0][1], lane=99)
    for c in clouds:
        c.close()
