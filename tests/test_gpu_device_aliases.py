"""The multi-DEVICE code paths executed on a one-GPU box: m3d_config.device_aliases makes ordinals 0..N-1
logical devices on the physical one -- each with lanes, streams, scratch, free lists and resident tables of its own, as a
second GPU would have -- so that run_on_devices (m3d_fit_multi, m3d_segment_plane_iterative_multi: one thread, one replica and
one in-process communicator per device), m3d_global_registration_batch and m3d_register_fragment_pairs (pairs dealt round-robin,
one resident-fragment table per device) run with n_dev = 2, 3 and 8.  What this proves: the dealing, the per-device state, the
in-process exchange and its abort path.  What it cannot: peer traffic and RCCL between devices (nothing here crosses a link)."""
import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def aliases(capi):
    old = capi.set_config(device_aliases=8)
    assert capi.device_count() >= 8
    yield capi
    capi.restore_config(old)


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_fit_multi_over_logical_devices(aliases, orc, n_dev):
    capi = aliases
    devs = list(range(n_dev))
    for kind, (pts, nrm), H in ((0, (synth.plane_cloud_c1(120_000, 3), None), 4000),
                                (1, (synth.sphere_cloud_c3(60_000, 4), None), 3000),
                                (2, synth.cylinder_cloud_c3(60_000, 5), 3000)):
        one = capi.fit_multi(kind, pts, [0], nrm, 0.01, H, 1.0, seed=21)
        many = capi.fit_multi(kind, pts, devs, nrm, 0.01, H, 1.0, seed=21)
        assert many.ret == one.ret and many.stats["best_index"] == one.stats["best_index"]
        assert np.array_equal(many.inliers, one.inliers) and np.array_equal(many.params, one.params)
        if kind == 0:
            o = orc.fit(kind, pts, nrm, thr=0.01, max_iter=H, prob=1.0, seed=21)
            assert many.stats["best_index"] == o.best_index and np.array_equal(many.inliers, o.inliers)
            assert np.allclose(many.params, o.params, rtol=0, atol=1e-9)
    # the adaptive stop as well (every rank must leave the loop at the same window)
    pts = synth.plane_cloud_c1(80_000, 9)
    a = capi.fit_multi(0, pts, [0], None, 0.01, 5000, 0.9999, seed=2)
    b = capi.fit_multi(0, pts, devs, None, 0.01, 5000, 0.9999, seed=2)
    assert a.stats["iterations"] == b.stats["iterations"] and np.array_equal(a.inliers, b.inliers)


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_segmentation_multi_over_logical_devices(aliases, orc, n_dev):
    capi = aliases
    room = synth.room_cloud_c5(120_000, 6)
    rc1, planes1, clusters1 = capi.segment_plane_iterative(room, 0.01, max_iteration=200, min_ratio=0.05, seed=19)
    rc2, planes2, clusters2 = capi.segment_plane_iterative_multi(room, list(range(n_dev)), 0.01, 200, 0.05, seed=19)
    assert rc2 == rc1 and len(planes1) >= 3 and np.array_equal(planes1, planes2)
    assert len(clusters1) == len(clusters2) and all(np.array_equal(a, b) for a, b in zip(clusters1, clusters2))
    if n_dev == 3:
        ro, po, co = orc.segment_plane_iterative(room, 0.01, max_iteration=200, min_ratio=0.05, seed=19, lookahead=128)
        assert ro == 0 and rc2 == 1 and len(co) == len(clusters2) and all(np.array_equal(a, b) for a, b in zip(co, clusters2))


def _pair(n, seed, dim=33):
    d = synth.registration_pair_c4(n, seed=seed, dim=dim, sigma=0.001)
    return d["src"], d["dst"], d["feat_src"], d["feat_dst"]


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_global_registration_batch_over_logical_devices(aliases, orc, n_dev):
    capi = aliases
    vox = 0.03 / 1.4
    sizes = [3000, 5000, 2000, 4000, 2500, 3500, 6000, 1500, 2200, 2800, 3300]
    pairs = [_pair(n, 140 + k) for k, n in enumerate(sizes)]
    seeds = [500 + k for k in range(len(pairs))]
    one = capi.global_registration_batch(pairs, vox, max_iter=1200, seeds=seeds, devices=(0,), inflight=2, want_stats=True)
    many = capi.global_registration_batch(pairs, vox, max_iter=1200, seeds=seeds, devices=tuple(range(n_dev)), inflight=2, want_stats=True)
    for k, (a, b) in enumerate(zip(one, many)):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), k
        assert b[3]["device"] == k % n_dev          # pair k is dealt to devices[k % n_dev]
    assert len({b[3]["device"] for b in many}) == min(n_dev, len(pairs))
    o = orc.global_registration(*pairs[4], vox, max_iter=1200, seed=seeds[4])
    assert many[4][0] == o[0] and np.allclose(many[4][1], o[1], atol=1e-9)
    # a bad pair fails alone, on whatever device it was dealt to
    bad = list(pairs[:5])
    bad[3] = (bad[3][0][:2], bad[3][1], bad[3][2][:2], bad[3][3])
    with pytest.raises(capi.M3DError, match="pair 3.*less than 3"):
        capi.global_registration_batch(bad, vox, max_iter=300, seeds=seeds[:5], devices=tuple(range(n_dev)))


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_register_fragment_pairs_over_logical_devices(aliases, n_dev):
    capi = aliases
    vox = 0.03 / 1.4
    base = synth.registration_pair_c4(3000, seed=77, dim=33, sigma=0.001)
    rng = np.random.default_rng(n_dev)
    frags, feats = [base["src"]], [base["feat_src"]]
    for k in range(5):
        Tk = synth.rigid_transform(20.0 + 15.0 * k, (1, k + 1, 2), (0.1 * k, -0.2, 0.05 * k))
        perm = rng.permutation(3000)
        frags.append(np.ascontiguousarray((base["src"] @ Tk[:3, :3].T + Tk[:3, 3] + rng.normal(0, 0.001, (3000, 3)))[perm]))
        feats.append(np.ascontiguousarray(np.abs(base["feat_src"] + rng.normal(0, 0.01, base["feat_src"].shape))[perm]))
    pairs = [(s, t) for s in range(6) for t in range(s + 1, 6)]
    seeds = [900 + k for k in range(len(pairs))]
    one = capi.register_fragment_pairs(frags, feats, pairs, vox, max_iter=1200, seeds=seeds, devices=(0,), inflight=2, want_stats=True)
    many = capi.register_fragment_pairs(frags, feats, pairs, vox, max_iter=1200, seeds=seeds, devices=tuple(range(n_dev)), inflight=2,
                                        want_stats=True)
    for k, (a, b) in enumerate(zip(one, many)):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), k
        assert b[3]["device"] == k % n_dev
    assert all(r[0] for r in many)
    with pytest.raises(capi.M3DError, match="out of range"):
        capi.register_fragment_pairs(frags, feats, [(0, 1), (0, 9)], vox, devices=tuple(range(n_dev)))


def test_a_failing_rank_aborts_the_local_group(aliases):
    """world = 8, one rank's device does not exist: that rank gives up before its first exchange, the in-process communicator
    is aborted (m3d_multi.cpp run_on_devices), nobody waits for it, the call reports ITS error, and the library works on."""
    capi = aliases
    pts = synth.plane_cloud_c1(50_000, 1)
    with pytest.raises(capi.M3DError, match="device 12"):
        capi.fit_multi(0, pts, [0, 1, 2, 12, 4, 5, 6, 7], None, 0.01, 2000, 1.0, seed=3)
    with pytest.raises(capi.M3DError):
        capi.segment_plane_iterative_multi(synth.room_cloud_c5(40_000, 2), [0, 1, 2, 3, 4, 5, 6, 12], 0.01, 100, 0.05, seed=3)
    with pytest.raises(capi.M3DError):
        capi.fit_multi(0, pts, [0, 0], None, 0.01, 100, 1.0, seed=3)      # logical ordinals must still be distinct
    ok = capi.fit_multi(0, pts, list(range(8)), None, 0.01, 2000, 1.0, seed=3)
    one = capi.fit(0, pts, None, 0.01, 2000, 1.0, seed=3)
    assert ok.stats["best_index"] == one.stats["best_index"] and np.array_equal(ok.inliers, one.inliers)


def test_logical_devices_keep_their_own_state(aliases):
    """Resident clouds on two logical devices of one GPU: own lanes (the same lane number on either), own contexts; a cloud
    stays on the device it was created on; fits on both from two threads at once agree with a fit on device 0."""
    import threading
    capi = aliases
    pts = synth.plane_cloud_c1(100_000, 5)
    ref = capi.fit(0, pts, None, 0.01, 3000, 1.0, seed=8)
    clouds = [capi.Cloud(pts, device=d) for d in (0, 5, 7)]
    res = {}
    def work(k, c):
        for _ in range(5):
            res[k] = c.fit(0, 0.01, 3000, 1.0, seed=8)
    th = [threading.Thread(target=work, args=(k, c)) for k, c in enumerate(clouds)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(3):
        assert res[k].stats["best_index"] == ref.stats["best_index"] and np.array_equal(res[k].inliers, ref.inliers)
    for c in clouds:
        c.close()
    with pytest.raises(capi.M3DError):
        capi.fit(0, pts, None, 0.01, 100, 1.0, seed=1, device=40)       # beyond the aliases: still an invalid ordinal
