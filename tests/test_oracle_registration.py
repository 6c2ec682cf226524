"""Known-answer tests pinning the registration half of the oracle (oracle/misc3d_oracle_reg.c):
Kabsch/umeyama against numpy.linalg.svd, the RANSAC driver on exactly solvable cases, mutual nearest
neighbour against scipy's cKDTree.  (Open3D / Eigen are absent: PARITY UNPINNED vs the real
reference, see the file header.)"""
import numpy as np
import pytest

from misc3d_amd import synth


def _kabsch_numpy(src, dst, scaling=False):
    ms, md = src.mean(0), dst.mean(0)
    s, d = src - ms, dst - md
    sigma = d.T @ s / len(src)
    U, S, Vt = np.linalg.svd(sigma)
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    c = 1.0
    if scaling:
        c = (S * np.diag(D)).sum() / ((s ** 2).sum() / len(src))
    T = np.eye(4)
    T[:3, :3] = c * R
    T[:3, 3] = md - c * R @ ms
    return T


def test_k3x3_matches_numpy_svd(orc):
    rng = np.random.default_rng(0)
    for trial in range(200):
        A = rng.normal(size=(3, 3))
        if trial % 4 == 1:
            A[:, 2] = A[:, 0] * 0.3 - A[:, 1]        # rank 2 (what 3 correspondences always give)
        if trial % 4 == 2:
            A = -A @ A.T                             # symmetric negative definite
        R, sv = orc.k3x3_rotation(A)
        U, S, Vt = np.linalg.svd(A)
        D = np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
        assert np.allclose(R, U @ D @ Vt, atol=1e-9), trial
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.linalg.det(R) > 0.999999
        assert np.allclose(np.abs(sv), S, atol=1e-12 * max(1, S[0]))


def test_umeyama_recovers_exact_transform(orc):
    rng = np.random.default_rng(1)
    T = synth.rigid_transform(40.0, (1, 1, 1), (0.3, -0.1, 0.2))
    for n in (3, 4, 10, 1000):
        src = rng.uniform(-1, 1, (n, 3))
        dst = src @ T[:3, :3].T + T[:3, 3]
        got = orc.umeyama(src, dst)
        assert np.allclose(got, T, atol=1e-12)
        assert np.allclose(got, _kabsch_numpy(src, dst), atol=1e-12)
    # with scaling
    src = rng.uniform(-1, 1, (50, 3))
    dst = 2.5 * (src @ T[:3, :3].T) + T[:3, 3]
    got = orc.umeyama(src, dst, True)
    assert np.allclose(got[:3, :3], 2.5 * T[:3, :3], atol=1e-12) and np.allclose(got[:3, 3], T[:3, 3], atol=1e-12)
    # reflection: the best ROTATION is returned (det = +1), like Eigen::umeyama
    dst = src * np.array([1, 1, -1.0])
    got = orc.umeyama(src, dst)
    assert np.linalg.det(got[:3, :3]) > 0.999 and np.allclose(got, _kabsch_numpy(src, dst), atol=1e-10)
    # noisy
    dst = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (50, 3))
    assert np.allclose(orc.umeyama(src, dst), _kabsch_numpy(src, dst), atol=1e-10)


def test_validate_counts_strict_radius(orc):
    dst = np.array([[0, 0, 0], [1, 0, 0], [5, 5, 5.0]])
    src = np.array([[0.03, 0, 0], [1, 0.05, 0], [9, 9, 9.0], [0, 0.05, 0]])
    cnt, e2 = orc.reg_validate(src, dst, np.eye(4), 0.05)
    # |d|^2 < r*r is strict (std::lower_bound on radius*radius): 0.05 away is NOT a correspondence
    assert cnt == 1 and e2 == pytest.approx(0.03 ** 2, rel=1e-15)


def _small_problem(n=1500, seed=3, true_fraction=0.4):
    d = synth.registration_pair_c4(n, seed=seed, dim=8, true_fraction=true_fraction, sigma=0.001)
    inv = np.empty(n, dtype=np.int64)
    inv[d["perm"]] = np.arange(n)
    rng = np.random.default_rng(seed + 1)
    m = 600
    cs = rng.integers(0, n, m)
    cd = np.where(rng.random(m) < true_fraction, inv[cs], rng.integers(0, n, m))
    return d, cs, cd


def test_registration_ransac_recovers_pose(orc):
    d, cs, cd = _small_problem()
    r = orc.registration_ransac(d["src"], d["dst"], cs, cd, thr=0.03, max_iter=2000, edge_thr=0.9,
                                confidence=0.999, seed=17)
    assert r.ret == 0 and r.best_index >= 0 and r.fitness > 0.9
    assert np.allclose(r.T, d["T"], atol=0.02)
    assert r.iterations <= 2000 and r.est_k <= 2000
    # confidence 1.0 never shortens the loop
    r1 = orc.registration_ransac(d["src"], d["dst"], cs, cd, thr=0.03, max_iter=300, confidence=1.0, seed=17)
    assert r1.iterations == 300 and r1.est_k == 300
    # degenerate inputs
    assert orc.registration_ransac(d["src"][:2], d["dst"], cs, cd, seed=1).ret == -1
    r0 = orc.registration_ransac(d["src"], d["dst"], cs[:2], cd[:2], seed=1)
    assert r0.ret == 0 and np.array_equal(r0.T, np.eye(4))
    r0 = orc.registration_ransac(d["src"], d["dst"], cs, cd, thr=0.0, seed=1)
    assert np.array_equal(r0.T, np.eye(4))


def test_mutual_nn_vs_ckdtree(orc):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    fs = rng.uniform(0, 1, (700, 33))
    fd = rng.uniform(0, 1, (650, 33))
    fd[:200] = fs[100:300] + rng.normal(0, 0.01, (200, 33))
    a, b = orc.match_mutual_nn(fs, fd)
    n01 = cKDTree(fd).query(fs)[1]
    n10 = cKDTree(fs).query(fd)[1]
    keep = [i for i in range(len(fs)) if n10[n01[i]] == i]
    assert np.array_equal(a, keep) and np.array_equal(b, n01[keep])
    assert len(a) >= 190 and np.all(np.diff(a) > 0)


def test_oracle_icp_against_scipy(orc):
    """Independent cross-check of the ICP restatement: the same loop written with scipy's cKDTree and numpy's SVD
    Kabsch reaches the same pose (1e-9), fitness and iteration count."""
    from scipy.spatial import cKDTree
    d = synth.registration_pair_c4(3000, seed=4, dim=8, sigma=0.001)
    src, dst = d["src"], d["dst"]
    init = d["T"].copy()
    init[:3, 3] += np.array([0.01, 0.008, -0.007])
    T, fit, rm, it, corr = orc.registration_icp(src, dst, 0.02, init)
    tree = cKDTree(dst)
    Tn = init.copy()
    pcd = src @ Tn[:3, :3].T + Tn[:3, 3]

    def result(p):
        dd, jj = tree.query(p, k=1)
        ok = dd ** 2 < 0.02 ** 2
        return ok, jj, (ok.sum() / len(p)), (np.sqrt((dd[ok] ** 2).sum() / ok.sum()) if ok.any() else 0.0)

    ok, jj, f0, r0 = result(pcd)
    n_it = 0
    for n_it in range(1, 31):
        s, t = pcd[ok], dst[jj[ok]]
        ms, mt = s.mean(0), t.mean(0)
        U, S, Vt = np.linalg.svd((t - mt).T @ (s - ms) / len(s))
        D = np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
        R = U @ D @ Vt
        upd = np.eye(4)
        upd[:3, :3], upd[:3, 3] = R, mt - R @ ms
        Tn = upd @ Tn
        pcd = pcd @ R.T + upd[:3, 3]
        ok, jj, f1, r1 = result(pcd)
        done = abs(f0 - f1) < 1e-6 and abs(r0 - r1) < 1e-6
        f0, r0 = f1, r1
        if done:
            break
    assert it == n_it and fit == pytest.approx(f0, abs=1e-12) and rm == pytest.approx(r0, rel=1e-9)
    assert np.allclose(T, Tn, rtol=0, atol=1e-9)
    assert np.array_equal(corr >= 0, ok) and np.array_equal(corr[ok], jj[ok])


def test_is_identity_follows_eigen(orc):
    """Eigen::MatrixBase::isIdentity(1e-8) (src/pipeline.cpp:814): diagonal |x - 1| <= min(|x|, 1) prec, off-diagonal
    |x| <= prec -- boundary cases in exact binary fractions around prec."""
    import ctypes as C
    f = orc.lib().orc_is_identity4
    f.restype = C.c_int

    def ident(T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        return bool(f(T.ctypes.data_as(C.c_void_p), C.c_double(1e-8)))

    assert ident(np.eye(4))
    for r, c in ((0, 3), (2, 1), (3, 0)):
        T = np.eye(4); T[r, c] = 1e-8
        assert ident(T)                        # <= : the bound itself passes
        T[r, c] = -1.0000001e-8
        assert not ident(T)
    T = np.eye(4); T[1, 1] = 1.0 + 9e-9
    assert ident(T)
    T[1, 1] = 1.0 + 2e-8
    assert not ident(T)
    T = np.eye(4); T[2, 2] = 1.0 - 9e-9       # min(|x|, 1) = |x| < 1: the bound scales with x
    assert ident(T)
    T[2, 2] = 0.0
    assert not ident(T)
    T = np.eye(4); T[0, 0] = np.nan
    assert not ident(T)


def test_global_registration_is_the_composition(orc):
    """orc_global_registration (src/pipeline.cpp:790-828) against its parts called one by one: match -> RANSAC at 1.4 voxel ->
    identity shortcut -> information matrix -> info(5,5) / min(N) < 0.3; accept, reject and shortcut cases."""
    d = synth.registration_pair_c4(1500, seed=3, dim=33, sigma=0.001)
    vox = 0.03 / 1.4

    def by_parts(src, dst, fs, fd, **kw):
        cs, cd = orc.match_mutual_nn(fs, fd)
        r = orc.registration_ransac(src, dst, cs, cd, thr=vox * 1.4, **kw)
        T = r.T
        if np.allclose(T, np.eye(4), rtol=0, atol=1e-8):
            return True, T, np.eye(6), len(cs)
        info = orc.information_matrix(src, dst, vox * 1.4, T)
        if info[5, 5] / min(len(src), len(dst)) < 0.3:
            return False, T, np.eye(6), len(cs)
        return True, T, info, len(cs)

    kw = dict(max_iter=600, edge_thr=0.9, confidence=0.999, seed=4)
    cut = d["src"].copy()
    cut[400:] = np.random.default_rng(1).uniform(40.0, 60.0, size=(1100, 3))
    fs0 = np.zeros_like(d["feat_src"]); fs0[:, 1] = np.arange(1500) + 10.0
    fd0 = np.zeros_like(d["feat_dst"]); fd0[:, 0] = np.arange(1500) + 10.0
    cases = {"accept": (d["src"], d["dst"], d["feat_src"], d["feat_dst"]), "reject": (cut, d["dst"], d["feat_src"], d["feat_dst"]),
             "no matches": (d["src"], d["dst"], fs0, fd0)}
    seen = set()
    for name, args in cases.items():
        g = orc.global_registration(*args, vox, **kw)
        p = by_parts(*args, **kw)
        assert g[0] == p[0] and g[3] == p[3], name
        assert np.array_equal(g[1], p[1]) and np.array_equal(g[2], p[2]), name
        seen.add((g[0], bool(np.array_equal(g[2], np.eye(6))), bool(np.array_equal(g[1], np.eye(4)))))
    assert seen == {(True, False, False), (False, True, False), (True, True, True)}
    with pytest.raises(ValueError):
        orc.global_registration(d["src"][:2], d["dst"], d["feat_src"][:2], d["feat_dst"], vox)
