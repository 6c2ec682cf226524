"""The C++ sharded driver (m3d_cloud_fit_sharded, m3d_segment_plane_iterative_sharded / _multi,
m3d_registration_ransac_sharded) on ONE GPU: a real RCCL communicator of world size 1 (librccl bound at run time,
ncclAllGather on the library's stream), the host transport, and the one-process `devices[]` form.  Results must be
bit-identical to the one-call entry points.  World sizes 2 and 3 run in tests/test_gpu_two_ranks.py (all ranks on
GPU 0, records over gloo through m3d_comm's host transport); RCCL itself at N > 1 needs N GPUs (bench.py --gpus N)."""
import numpy as np
import pytest

from misc3d_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl1():
    c = capi.Comm.rccl(world=1, rank=0, device=0)
    yield c
    c.close()


def _same(a, b):
    return (a.ret == b.ret and a.stats["best_index"] == b.stats["best_index"] and a.stats["iterations"] == b.stats["iterations"]
            and a.stats["count"] == b.stats["count"] and np.array_equal(a.inliers, b.inliers) and np.array_equal(a.params, b.params))


@pytest.mark.parametrize("kind", [capi.PLANE, capi.SPHERE, capi.CYLINDER])
def test_fit_sharded_rccl_world1(rccl1, kind):
    pts, nrm = {capi.PLANE: (synth.plane_cloud_c2(80000, seed=2), None), capi.SPHERE: (synth.sphere_cloud_c3(50000, 4), None),
                capi.CYLINDER: synth.cylinder_cloud_c3(50000, 3)}[kind]
    with capi.Cloud(pts, nrm) as c:
        for prob, H, seed in ((1.0, 5000, 11), (1.0, 20000, 4), (0.9999, 1000, 5), (0.99, 300, 9), (1.0, 64, 1), (1.0, 1, 1)):
            before = rccl1.collectives
            assert _same(c.fit(kind, 0.01, H, prob, seed=seed), c.fit_sharded(rccl1, kind, 0.01, H, prob, seed=seed)), (kind, prob, H)
            assert rccl1.collectives > before            # the exchange really ran
        # comm == NULL is the one-call fit
        assert _same(c.fit(kind, 0.01, 500, 0.9999, seed=3), c.fit_sharded(None, kind, 0.01, 500, 0.9999, seed=3))


def test_fit_sharded_phased_scoring(rccl1):
    """The phased scoring of cylinders (the default from 131 k points on) inside the sharded loop: same incumbent on every rank,
    same result as the one-call fit."""
    pts, nrm = synth.cylinder_cloud_c3(150_000, 3)
    with capi.Cloud(pts, nrm) as c:
        a = c.fit(capi.CYLINDER, 0.01, 6000, 1.0, seed=13)
        b = c.fit_sharded(rccl1, capi.CYLINDER, 0.01, 6000, 1.0, seed=13)
        assert _same(a, b)
        old = capi.set_config(score_phases=0)
        try:
            assert _same(a, c.fit_sharded(rccl1, capi.CYLINDER, 0.01, 6000, 1.0, seed=13))
        finally:
            capi.restore_config(old)


def test_fit_sharded_host_transport_and_errors(rccl1):
    pts = synth.plane_cloud_c2(60000, seed=2)
    calls = []
    comm = capi.Comm.host(1, 0, lambda b: (calls.append(len(b)), b)[1])
    with capi.Cloud(pts) as c:
        assert _same(c.fit(0, 0.01, 3000, 1.0, seed=11), c.fit_sharded(comm, 0, 0.01, 3000, 1.0, seed=11))
        assert calls and all(n % 256 == 0 for n in calls)     # whole 64-hypothesis groups of 4-byte records
        with pytest.raises(capi.M3DError) as e:
            c.fit_sharded(comm, 0, 0.01, 100, 1.5, seed=1)
        assert e.value.code == capi.ERR_PROBABILITY
        old = capi.set_config(dense_scoring=1)
        try:
            with pytest.raises(capi.M3DError):
                c.fit_sharded(comm, 0, 0.01, 100, 1.0, seed=1)    # the sharded path is the culled one
        finally:
            capi.restore_config(old)
    # a transport that fails is reported, not swallowed
    bad = capi.Comm.host(1, 0, lambda b: b[:-1])
    with capi.Cloud(pts) as c:
        with pytest.raises(capi.M3DError):
            c.fit_sharded(bad, 0, 0.01, 300, 1.0, seed=1)
    comm.close()
    bad.close()


def test_segmentation_sharded_and_multi(rccl1):
    room = synth.room_cloud_c5(150000, 6)
    rc1, planes1, clusters1 = capi.segment_plane_iterative(room, 0.01, max_iteration=200, min_ratio=0.05, seed=19)
    assert len(planes1) >= 3
    for rc2, planes2, clusters2 in (capi.segment_plane_iterative_sharded(room, rccl1, 0.01, 200, 0.05, seed=19),
                                    capi.segment_plane_iterative_multi(room, [0], 0.01, 200, 0.05, seed=19)):
        assert rc2 == rc1 and np.array_equal(planes1, planes2)
        assert len(clusters1) == len(clusters2) and all(np.array_equal(a, b) for a, b in zip(clusters1, clusters2))
    with pytest.raises(capi.M3DError):
        capi.segment_plane_iterative_multi(room, [0, 0], 0.01, 200, 0.05, seed=19)     # distinct devices only
    f1 = capi.fit(0, room, None, 0.01, 2000, 1.0, seed=5)
    f2 = capi.fit_multi(0, room, [0], None, 0.01, 2000, 1.0, seed=5)
    assert f1.stats["best_index"] == f2.stats["best_index"] and np.array_equal(f1.inliers, f2.inliers)


def test_registration_sharded_rccl_world1(rccl1):
    d = synth.registration_pair_c4(20000, seed=5, dim=33, true_fraction=0.5, sigma=0.001)
    a, b = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    for conf, H in ((1.0, 3000), (0.999, 100000)):
        T1, st1 = capi.registration_ransac(d["src"], d["dst"], a, b, threshold=0.03, max_iter=H, confidence=conf, seed=17)
        T2, st2 = capi.registration_ransac_sharded(d["src"], d["dst"], a, b, rccl1, threshold=0.03, max_iter=H, confidence=conf, seed=17)
        assert np.array_equal(T1, T2)
        assert all(st1[k] == st2[k] for k in ("best_index", "iterations", "validations", "fitness", "est_k"))
