"""m3d_match_mutual_nn with its matrices going up in slices under the scan (m3d_config.match_pipeline).

A sliced call cuts the queries in two and the database in four parts of whole splits, chooses the fp16 scale from the first slices,
lets the norm bounds grow as slices are packed, keeps one set of reverse thresholds per query slice, and runs the scan's blocks on
two streams.  None of that may show in the result: every case here is compared with the CPU oracle (ANNMatcher::Match restated,
oracle/misc3d_oracle_reg.c) and, where the oracle is too slow, with the unsliced call -- which the suite compares with the oracle at
full size elsewhere (tests/test_gpu_full_size_vs_oracle.py).  match_pipeline = 2 slices whatever the size, so that ragged sizes,
empty parts and sides smaller than a slice are reached with matrices the oracle handles in milliseconds."""
import threading

import numpy as np
import pytest

from misc3d_amd import synth

pytestmark = pytest.mark.gpu
DIM = 33
MFMA, SLICED, REDONE, FP32 = 1, 2, 4, 8


def _pair(rng, ns, nd, mutual=None):
    fs = rng.uniform(0, 1, (ns, DIM))
    fd = rng.uniform(0, 1, (nd, DIM))
    k = min(ns, nd) // 3 if mutual is None else mutual
    fd[:k] = np.abs(fs[ns - k:] + rng.normal(0, 0.01, (k, DIM)))
    return fs, fd


def _sliced(capi, fs, fd, mode=2):
    old = capi.set_config(match_pipeline=mode)
    try:
        a, b = capi.match_mutual_nn(fs, fd)
        return a.astype(np.int64), b.astype(np.int64), capi.match_last_path(), capi.match_last_fallbacks()
    finally:
        capi.restore_config(old)


@pytest.mark.parametrize("ns,nd", [(5000, 7000), (1537, 1025), (2048, 4096), (9000, 3000), (513, 20000), (20000, 513), (40, 2500),
                                   (2500, 40), (12345, 12345)])
def test_sliced_matches_oracle_ragged_sizes(capi, orc, ns, nd):
    """sizes that are no multiple of a tile, a slice or a part; parts without rows; a side smaller than the first slice"""
    rng = np.random.default_rng(ns + 7 * nd)
    fs, fd = _pair(rng, ns, nd)
    fd[-3:] = fd[:3]                      # exact duplicates in the last part: ties go to the lowest index, found in the first
    a, b, path, _ = _sliced(capi, fs, fd)
    oa, ob = orc.match_mutual_nn(fs, fd)
    assert np.array_equal(a, oa) and np.array_equal(b, ob)
    assert path & MFMA and path & SLICED and not path & REDONE


@pytest.mark.parametrize("where", ["second_query_slice", "last_database_part", "both"])
@pytest.mark.parametrize("factor", [1.9, 3.9, 4.0, 40.0])
def test_later_slices_larger_than_the_first(capi, orc, where, factor):
    """the scale is chosen from the first slices with one bit of headroom -- their largest |v| lands in [512, 1024), the screen's
    bounds hold up to 2048: a later value up to twice (here, with the first maximum at the bottom of its binade, four times) the
    largest one seen stays sliced, anything beyond is found at the end and the search redone whole on the resident matrices.
    Same answer either way."""
    rng = np.random.default_rng(int(factor * 10) + len(where))
    ns, nd = 6000, 5000
    fs, fd = _pair(rng, ns, nd)
    np.minimum(fs, 0.99, out=fs)
    np.minimum(fd, 0.99, out=fd)
    fs[0, 0] = fd[0, 0] = 1.0            # the first slices' maximum, exactly: scaled to 512
    if where in ("second_query_slice", "both"):
        fs[ns - 700:, :] *= min(factor, 1.9)     # larger norms throughout the slice (the bounds' side of it) ...
        fs[ns - 1, 5] = factor                   # ... and the value that decides
    if where in ("last_database_part", "both"):
        fd[nd - 300:, :] *= min(factor, 1.9)
        fd[nd - 1, 7] = factor
    a, b, path, _ = _sliced(capi, fs, fd)
    oa, ob = orc.match_mutual_nn(fs, fd)
    assert np.array_equal(a, oa) and np.array_equal(b, ob)
    assert path & MFMA
    assert bool(path & REDONE) == (factor >= 4.0)


@pytest.mark.parametrize("case", ["nan_late", "inf_late", "nan_first", "zeros", "huge", "tiny"])
def test_values_the_screen_cannot_hold(capi, orc, case):
    """NaN / inf in a slice that arrives after the scale was chosen, in the first slice, all-zero matrices, magnitudes outside fp32"""
    rng = np.random.default_rng(len(case))
    ns, nd = 4000, 4500
    fs, fd = _pair(rng, ns, nd)
    if case == "nan_late":
        fs[ns - 5, 3] = np.nan
        fd[nd - 9] = np.nan
    elif case == "inf_late":
        fd[nd - 2, 1] = np.inf
    elif case == "nan_first":
        fs[1, 1] = np.nan
    elif case == "zeros":
        fs[:] = 0.0
        fd[:] = 0.0
    elif case == "huge":
        fs *= 1e30
        fd *= 1e30
    elif case == "tiny":
        fs *= 1e-30
        fd *= 1e-30
    a, b, path, _ = _sliced(capi, fs, fd)
    oa, ob = orc.match_mutual_nn(fs, fd)
    assert np.array_equal(a, oa) and np.array_equal(b, ob)
    if case in ("nan_late", "inf_late", "nan_first", "zeros"):
        assert path & FP32 and not path & MFMA
    else:
        assert path & MFMA and path & SLICED


@pytest.mark.parametrize("case", ["dups", "lattice", "offset", "scales_late", "mixed", "row_overflow_late", "hub_query_late"])
def test_sliced_screen_adversarial(capi, orc, case):
    """tests/test_gpu_registration.py's inputs against the screen, with the hard rows in the LATER slices: runs of exact ties, distances below the
    bound, a common offset, one huge-norm row that arrives after the first windows were set, six decades inside every row, a target
    row with more candidates than slots and a query that is the nearest of 400 rows -- both in the second query slice"""
    rng = np.random.default_rng(3 + len(case))
    ns, nd = 5200, 4800
    fs, fd = _pair(rng, ns, nd)
    if case == "dups":
        fd[nd - 400:nd - 200] = fd[nd - 400]
        fs[ns - 600:ns - 400] = fs[ns - 600]
        fs[ns - 10] = fd[nd - 400]
    elif case == "lattice":
        base = rng.uniform(0, 1, DIM)
        fd[:] = base + 1e-9 * rng.integers(-3, 4, (nd, DIM))
        fs[:] = base + 1e-9 * rng.integers(-3, 4, (ns, DIM))
    elif case == "offset":
        fs += 1e4
        fd += 1e4
    elif case == "scales_late":
        fd[nd - 100] *= 1.9              # inside the headroom: the bound of every window set before it arrived was smaller
        fs[ns - 50] *= 1.9
    elif case == "mixed":
        fs[:, ::2] *= 1e-6
        fd[:, ::2] *= 1e-6
        fs[:, 1::4] *= 1e-3
        fd[:, 1::4] *= 1e-3
    elif case == "row_overflow_late":
        fs[ns - 900:ns - 200] = fd[nd - 77] + 1e-3 * np.sign(rng.normal(size=(700, DIM)))
    elif case == "hub_query_late":
        fd[nd - 900:nd - 500] = fs[ns - 42] + rng.normal(0, 1e-4, (400, DIM))
    for a_, b_ in ((fs, fd), (fd, fs)):
        a, b, path, falls = _sliced(capi, a_, b_)
        oa, ob = orc.match_mutual_nn(a_, b_)
        assert np.array_equal(a, oa) and np.array_equal(b, ob)
        assert path & MFMA and path & SLICED
        if case in ("dups", "lattice") or (case == "row_overflow_late" and a_ is fs):
            assert falls > 0


def test_sliced_equals_whole_at_full_size(capi):
    """C4's matrices (200 000 x 200 000 x 33): the default configuration slices them (the call is alone on the device), and the pairs
    are those of the unsliced call -- which test_c4_full_size_vs_oracle compares with the oracle"""
    d = synth.registration_pair_c4(200000, seed=5)
    a1, b1, path1, f1 = _sliced(capi, d["feat_src"], d["feat_dst"], mode=1)
    a0, b0, path0, f0 = _sliced(capi, d["feat_src"], d["feat_dst"], mode=0)
    assert path1 & SLICED and not path1 & REDONE and path0 & MFMA and not path0 & SLICED
    assert np.array_equal(a1, a0) and np.array_equal(b1, b0) and len(a1) > 50000
    assert f1 < 40 and f0 < 40


def test_sliced_calls_from_several_threads(capi, orc):
    """sliced calls on several lanes at once (each with its own copy / second compute stream and events), every result against the oracle"""
    cases = []
    for t in range(6):
        rng = np.random.default_rng(100 + t)
        fs, fd = _pair(rng, 3000 + 517 * t, 5000 - 311 * t)
        cases.append((fs, fd, orc.match_mutual_nn(fs, fd)))
    errors = []
    old = capi.set_config(match_pipeline=2)

    def work(t):
        try:
            fs, fd, (oa, ob) = cases[t]
            for _ in range(4):
                a, b = capi.match_mutual_nn(fs, fd)
                if not (np.array_equal(a.astype(np.int64), oa) and np.array_equal(b.astype(np.int64), ob)):
                    errors.append(("mismatch", t))
                if not capi.match_last_path() & SLICED:
                    errors.append(("not sliced", t))
        except Exception as e:   # noqa: BLE001
            errors.append((repr(e), t))

    try:
        threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    finally:
        capi.restore_config(old)
    assert not errors, errors
