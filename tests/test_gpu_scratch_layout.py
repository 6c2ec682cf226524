"""Scratch-layout invariants of a lane: a lane's scratch -- survivor lists and their tickets, mask and counter
slots, the compaction's slots, pinned words -- is laid out for the LAST call's sizes and re-used by the next one.  Round 4's
one real bug (8aeb631: plane_bound_k's tickets sat behind a variable-length list) was of this class and was found by a
random soak; these tests walk the class on purpose: window lengths that shrink and grow across fits on ONE cloud and one
lane, for every kind, with the scoring paths whose scratch depends on the window (histogram bound at every size, phased
scoring, one / several chunks, the sharded loop), clouds of different sizes alternating on one lane, and every result
compared with the oracle."""
import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu

# hypotheses per fit: below / above the lead pass (128), one and several chunks (cap 24 576, first chunk 2 048), odd sizes
SEQ = [10000, 300, 24576, 2048, 40000, 64, 8192, 129, 30000, 4097]
N = 6000
_ORACLE = {}


def _cloud(kind):
    if kind == 0:
        return synth.plane_cloud_c1(N, 21), None
    if kind == 1:
        return synth.sphere_cloud_c3(N, 22), None
    return synth.cylinder_cloud_c3(N, 23)


def _oracle(orc, kind, H):
    if (kind, H) not in _ORACLE:
        pts, nrm = _cloud(kind)
        _ORACLE[(kind, H)] = orc.fit(kind, pts, nrm, thr=0.01, max_iter=H, prob=1.0, seed=1000 + H)
    return _ORACLE[(kind, H)]


def _check(g, o, what):
    assert g.ret == o.ret and g.stats["best_index"] == o.best_index, what
    assert np.array_equal(g.inliers, o.inliers), what
    assert np.allclose(g.params, o.params, rtol=0, atol=1e-9), what


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("cfg", [{}, {"plane_bound": 2}, {"score_phases": 3, "plane_bound": 0}, {"chunk_cap": 4096, "first_chunk": 1024},
                                 {"prestream": 0, "plane_bound": 2}], ids=["default", "bound_always", "three_phases", "small_chunks", "no_prestream"])
def test_window_lengths_shrink_and_grow_on_one_cloud(capi, orc, kind, cfg):
    pts, nrm = _cloud(kind)
    old = capi.set_config(**cfg)
    try:
        with capi.Cloud(pts, nrm) as c:
            for H in SEQ + SEQ[::-1]:
                _check(c.fit(kind, 0.01, H, 1.0, seed=1000 + H), _oracle(orc, kind, H), (kind, cfg, H))
    finally:
        capi.restore_config(old)


@pytest.mark.parametrize("kind", [0, 2])
def test_window_lengths_shrink_and_grow_in_the_sharded_loop(capi, orc, kind):
    """m3d_cloud_fit_sharded over a world-1 RCCL communicator: the window of a rank, its gathered records, the short first piece."""
    pts, nrm = _cloud(kind)
    comm = capi.Comm.rccl(world=1, rank=0, device=0)
    old = capi.set_config(plane_bound=2)
    try:
        with capi.Cloud(pts, nrm) as c:
            for H in SEQ:
                _check(c.fit_sharded(comm, kind, 0.01, H, 1.0, seed=1000 + H), _oracle(orc, kind, H), (kind, "sharded", H))
                _check(c.fit(kind, 0.01, H, 1.0, seed=1000 + H), _oracle(orc, kind, H), (kind, "one call after sharded", H))
    finally:
        capi.restore_config(old)
        comm.close()


def test_clouds_of_different_sizes_alternate_on_one_lane(capi, orc):
    """Tile counts that shrink and grow under a lane's mask / frame / compaction scratch: a 40 000-point and a 3 000-point cloud
    fitted in turn (plane, histogram bound always), then segmented; lanes = 1 keeps everything on one lane."""
    big = synth.plane_cloud_c1(40000, 31)
    small = synth.plane_cloud_c1(3000, 32)
    room = synth.room_cloud_c5(30000, 33)
    ob = {H: orc.fit(0, big, None, thr=0.01, max_iter=H, prob=1.0, seed=H) for H in (9000, 200)}
    os_ = {H: orc.fit(0, small, None, thr=0.01, max_iter=H, prob=1.0, seed=H) for H in (9000, 200)}
    oseg = orc.segment_plane_iterative(room, 0.02, 100, 0.05, seed=5)
    old = capi.set_config(lanes=1, plane_bound=2)
    try:
        with capi.Cloud(big) as cb, capi.Cloud(small) as cs:
            for H in (9000, 200, 9000):
                _check(cb.fit(0, 0.01, H, 1.0, seed=H), ob[H], ("big", H))
                _check(cs.fit(0, 0.01, H, 1.0, seed=H), os_[H], ("small", H))
                _rc, planes, clusters = capi.segment_plane_iterative(room, 0.02, 100, 0.05, seed=5)
                assert len(clusters) == len(oseg[2]) and all(np.array_equal(a, b) for a, b in zip(clusters, oseg[2]))
                assert np.allclose(planes, oseg[1], rtol=0, atol=1e-9)
                _check(capi.fit(0, small, None, 0.01, H, 1.0, seed=H), os_[H], ("one-shot small", H))
    finally:
        capi.restore_config(old)
