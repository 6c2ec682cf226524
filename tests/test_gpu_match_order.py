"""The matcher scans the database in an order of its own (round 6: rows sorted by their reverse-search threshold within chunks of
1024 rows -- rev_order_k, m3d_match_kernels.hip -- so that ONE test per side decides whether a tile holds a candidate).  Rings and
candidate lists then carry positions, nn64_verify_k / rev_bin_k translate them back, and ties must still go to the lowest ROW.
None of that may show: every case is compared with the CPU oracle (ANNMatcher::Match restated, oracle/misc3d_oracle_reg.c), sliced
and unsliced."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DIM = 33
MFMA = 1


def _both(capi, fs, fd):
    out = []
    for mode in (0, 2):
        old = capi.set_config(match_pipeline=mode)
        try:
            a, b = capi.match_mutual_nn(fs, fd)
            out.append((a.astype(np.int64), b.astype(np.int64), capi.match_last_path()))
        finally:
            capi.restore_config(old)
    return out


def _clusters(rng, n, spreads):
    """descriptors in clusters of very different density: the reverse thresholds of neighbouring rows differ by orders of magnitude"""
    k = len(spreads)
    centres = rng.uniform(0.2, 0.8, (k, DIM))
    which = rng.integers(0, k, n)
    return np.abs(centres[which] + rng.normal(0, 1, (n, DIM)) * np.asarray(spreads)[which, None])


@pytest.mark.parametrize("ns,nd", [(700, 700), (3000, 1023), (3000, 1024), (3000, 1025), (2500, 2049), (300, 9000), (9000, 300), (6000, 6000)])
def test_chunks_of_every_shape(capi, orc, ns, nd):
    """databases of less than a chunk, exactly one, one row more, several with a ragged last one"""
    rng = np.random.default_rng(ns * 31 + nd)
    fs = _clusters(rng, ns, [0.002, 0.02, 0.2])
    fd = _clusters(rng, nd, [0.002, 0.02, 0.2])
    k = min(ns, nd) // 4
    fd[rng.permutation(nd)[:k]] = np.abs(fs[rng.permutation(ns)[:k]] + rng.normal(0, 1e-3, (k, DIM)))
    oa, ob = orc.match_mutual_nn(fs, fd)
    for a, b, path in _both(capi, fs, fd):
        assert path & MFMA
        assert np.array_equal(a, oa) and np.array_equal(b, ob)


def test_equal_rows_keep_the_lowest_index(capi, orc):
    """whole runs of identical descriptors (equal thresholds, equal distances): the order inside a chunk is by row, the nearest
    neighbour among equals is the lowest ROW on both sides -- whatever position the scan met it at"""
    rng = np.random.default_rng(11)
    ns, nd = 4000, 5000
    fs = rng.uniform(0, 1, (ns, DIM))
    fd = rng.uniform(0, 1, (nd, DIM))
    for j0 in (100, 1020, 2047, 4990):            # (runs across chunk boundaries too)
        fd[j0:j0 + 8] = fd[j0]
    fd[3000:3005] = fs[17]                        # five exact copies of a query, and the query five times
    fs[2000:2005] = fs[17]
    fs[3990:3995] = fd[1020]
    oa, ob = orc.match_mutual_nn(fs, fd)
    for a, b, _ in _both(capi, fs, fd):
        assert np.array_equal(a, oa) and np.array_equal(b, ob)
    hit = {int(i): int(j) for i, j in zip(oa, ob)}
    assert hit.get(17) == 3000                    # the lowest row of the five copies, matched by the lowest of the equal queries


def test_rows_without_a_usable_threshold(capi, orc):
    """rows far from every query of the sample the thresholds come from (the first eighth of the queries): their thresholds are the
    largest of their chunks; rows that are nobody's nearest neighbour; a database ordered by density to begin with"""
    rng = np.random.default_rng(5)
    ns, nd = 8000, 6000
    fs = np.abs(rng.normal(0.5, 0.05, (ns, DIM)))
    fs[: ns // 8] = np.abs(rng.normal(0.2, 0.01, (ns // 8, DIM)))      # the sample sits in one corner
    fd = np.abs(rng.normal(0.5, 0.05, (nd, DIM)))
    fd[::7] = np.abs(rng.normal(0.9, 0.3, (len(fd[::7]), DIM)))        # every seventh row far from everything
    order = np.argsort(np.linalg.norm(fd - 0.5, axis=1))               # (dense rows first)
    fd = np.ascontiguousarray(fd[order])
    oa, ob = orc.match_mutual_nn(fs, fd)
    for a, b, _ in _both(capi, fs, fd):
        assert np.array_equal(a, oa) and np.array_equal(b, ob)
