"""CPU-side check of the drop-in boundary: the C-ABI shared object loads without a GPU and exports
every function declared in include/misc3d_amd.h; compute entry points fail loudly (no fallback)."""
import ctypes
import re

import numpy as np
import pytest


def _declared_functions(header_text):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    return sorted(set(re.findall(r"\b(m3d_[a-z0-9_]+)\s*\(", text)) - {"m3d_rmse_fn", "m3d_allgather_fn"})


def test_header_symbols_exported(capi):
    import os
    names = _declared_functions(open(capi.HEADER_PATH).read())
    assert len(names) >= 40
    for need in ("m3d_cloud_fit_sharded", "m3d_comm_create_rccl", "m3d_segment_plane_iterative_multi", "m3d_segment_plane_iterative_clouds",
                 "m3d_registration_ransac_sharded", "m3d_set_config", "m3d_global_registration", "m3d_global_registration_batch",
                 "m3d_register_fragment_pairs", "m3d_cloud_fit_batch", "m3d_cloud_create_lane"):
        assert need in names
    # measurement hooks live in their own header: none of them in the product header
    assert not [n for n in names if n.startswith("m3d_bench_") or "time_score" in n]
    bench = _declared_functions(open(os.path.join(os.path.dirname(capi.HEADER_PATH), "misc3d_amd_bench.h")).read())
    assert bench and all(n.startswith("m3d_bench_") for n in bench)
    L = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in names + bench if not hasattr(L, n)]
    assert not missing, missing


def test_config_roundtrip_and_env_defaults(capi):
    c = capi.get_config()
    assert c.lead_hypotheses % 64 == 0 and c.lead_hypotheses >= 64 and 1 <= c.score_groups_per_block <= 64
    old = capi.set_config(dense_scoring=1, lead_hypotheses=100, score_groups_per_block=999)   # bad values fall back
    try:
        n = capi.get_config()
        assert n.dense_scoring == 1 and n.lead_hypotheses == 128 and n.score_groups_per_block == 8
    finally:
        capi.restore_config(old)
    assert capi.get_config().dense_scoring == old.dense_scoring


def test_host_communicator_needs_no_gpu(capi):
    seen = []
    comm = capi.Comm.host(3, 1, lambda b: (seen.append(b), b * 3)[1])
    assert comm.world == 3 and comm.rank == 1 and comm.collectives == 0
    comm.close()
    with pytest.raises(capi.M3DError):
        capi.Comm.host(2, 5, lambda b: b)        # rank out of range


def test_version_and_error_string(capi):
    assert b"gfx950" in capi.lib().m3d_version()
    assert isinstance(capi.last_error(), str)


def test_argument_validation_needs_no_gpu(capi):
    pts = np.zeros((2, 3))
    with pytest.raises(capi.M3DError) as e:
        capi.fit(capi.PLANE, pts, seed=1)                      # ransac.h:510-513
    assert e.value.code == capi.ERR_TOO_FEW_POINTS and "lack of points" in str(e.value)
    with pytest.raises(capi.M3DError) as e:
        capi.fit(capi.PLANE, np.zeros((10, 3)), probability=1.5, seed=1)   # ransac.h:483-485
    assert e.value.code == capi.ERR_PROBABILITY
    with pytest.raises(capi.M3DError) as e:
        capi.fit(capi.CYLINDER, np.zeros((10, 3)), normals=None, seed=1)   # py_common.cpp:50-52
    assert e.value.code == capi.ERR_NO_NORMALS and "requires normals" in str(e.value)


def test_global_registration_argument_validation_needs_no_gpu(capi):
    """m3d_global_registration / _batch / m3d_register_fragment_pairs check their arguments in front of any device work:
    the solver's LogError (fewer than 3 points, src/transform_estimation.cpp:130-133), distinct devices, index ranges."""
    rng = np.random.default_rng(0)
    pts, feat = rng.normal(size=(50, 3)), rng.uniform(size=(50, 33))
    with pytest.raises(capi.M3DError) as e:
        capi.global_registration(pts[:2], pts, feat[:2], feat, 0.01)
    assert e.value.code == capi.ERR_TOO_FEW_POINTS and "less than 3" in str(e.value)
    with pytest.raises(capi.M3DError) as e:
        capi.global_registration_batch([(pts, pts, feat, feat)], 0.01, devices=(0, 0))
    assert e.value.code == capi.ERR_INVALID_ARG and "distinct" in str(e.value)
    with pytest.raises(capi.M3DError) as e:
        capi.register_fragment_pairs([pts, pts], [feat, feat], [(0, 2)], 0.01)
    assert e.value.code == capi.ERR_INVALID_ARG and "out of range" in str(e.value)
    assert capi.global_registration_batch([], 0.01) == [] and capi.register_fragment_pairs([pts], [feat], [], 0.01) == []
    with pytest.raises(ValueError):
        capi.global_registration(pts, pts, feat[:10], feat, 0.01)       # one descriptor per point


def test_round5_config_fields(capi):
    """lanes, wait_spin_us, prestream, chunk_cap, first_chunk, reg_cells_per_radius, match_pipeline: defaults and sanitising (appended fields)."""
    c = capi.get_config()
    assert (c.lanes, c.wait_spin_us, c.prestream, c.chunk_cap, c.first_chunk, c.reg_cells_per_radius, c.match_pipeline) == (4, 500, 1, 24576, 2048, 4, 1)
    old = capi.set_config(lanes=99, wait_spin_us=-5, chunk_cap=100, first_chunk=-1, reg_cells_per_radius=0, prestream=7, match_pipeline=9)
    try:
        n = capi.get_config()
        assert (n.lanes, n.wait_spin_us, n.prestream, n.chunk_cap, n.first_chunk, n.reg_cells_per_radius, n.match_pipeline) == (4, 500, 1, 1024, 0, 4, 1)
        if not capi.experimental():      # the product build REJECTS the refuted variants' switches (m3d_kernels.hpp)
            for k in ("score_mfma", "score_waves4", "compact_one_pass"):
                with pytest.raises(capi.M3DError, match="EXPERIMENTAL"):
                    capi.set_config(**{k: 1})
            n = capi.get_config()
            assert (n.score_mfma, n.score_waves4, n.compact_one_pass) == (0, 0, 0)
    finally:
        capi.restore_config(old)


def test_no_cpu_fallback(capi):
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.M3DError) as e:
        capi.fit(capi.PLANE, np.random.default_rng(0).normal(size=(100, 3)), seed=1)
    assert e.value.code == capi.ERR_DEVICE


def test_draw_samples_matches_oracle_sampler(capi, orc):
    # host-only entry point: std::mt19937 + utils.h:81-97 rejection sampler
    for kind, n in ((capi.PLANE, 1000), (capi.SPHERE, 37), (capi.CYLINDER, 5)):
        a = capi.draw_samples(n, kind, 500, 1234)
        b = orc.draw_samples(n, capi.MINIMAL_SAMPLE[kind], 500, 1234)
        assert np.array_equal(a.astype(np.uint64), b)


def test_block_sampler_stream_is_the_scalar_stream(capi, orc):
    # the library draws 624 words at a time with a multiply-shift remainder (m3d_mt19937.hpp); the oracle draws word by
    # word with `%`.  Tiny clouds force redraws (duplicates) in nearly every sample, sizes around powers of two and
    # 2^31 exercise the remainder constants, 3000 hypotheses cross many block boundaries at every alignment.
    sizes = {capi.PLANE: (3, 4, 7, 255, 256, 257, 65535, 65536, 65537, 10**6, 10**7, 2**31 - 1),
             capi.SPHERE: (4, 5, 9, 1023, 1024, 1025, 999983, 2**30, 2**30 + 1),
             capi.CYLINDER: (2, 3, 6, 127, 128, 129, 50000, 2**31 - 1)}
    for kind, ns in sizes.items():
        for i, n in enumerate(ns):
            seed = 7919 * i + kind
            a = capi.draw_samples(n, kind, 3000, seed)
            b = orc.draw_samples(n, capi.MINIMAL_SAMPLE[kind], 3000, seed)
            assert np.array_equal(a.astype(np.uint64), b), (kind, n)
    # seeds are taken mod 2^32 (std::mt19937::seed)
    a = capi.draw_samples(1000, capi.PLANE, 100, 5)
    b = capi.draw_samples(1000, capi.PLANE, 100, 5 + 2**32)
    assert np.array_equal(a, b)


def test_replay_matches_oracle_driver(capi, orc):
    # pure host logic: replay of ransac.h:592-613 over the oracle's per-hypothesis trace
    from misc3d_amd import synth
    pts = synth.plane_cloud_c1(3000, seed=3)
    for prob, max_iter in ((0.9999, 200), (1.0, 64), (0.5, 100)):
        r = orc.fit(orc.PLANE, pts, thr=0.01, max_iter=max_iter, prob=prob, seed=5, trace=True)
        tr = r.trace
        errs = tr["errors"]
        cnts = tr["counts"]

        def rmse(i):
            return errs[i] / np.sqrt(float(cnts[i])) if cnts[i] else 1e10

        st = capi.replay(len(pts), capi.PLANE, max_iter, prob, tr["valid"].astype(np.uint8),
                         cnts.astype(np.uint32), rmse)
        assert st.best_index == r.best_index
        assert st.count == r.count
        assert st.iterations == r.iterations
        assert st.best_fitness == r.fitness


def test_replay_accepts_packed_records(capi):
    """valid == NULL: counts carry MinimalFit's return in bit 31 (the record the ranks exchange)"""
    import ctypes as C
    rng = np.random.default_rng(3)
    n, H = 5000, 700
    valid = (rng.random(H) < 0.9).astype(np.uint8)
    counts = rng.integers(0, n // 2, H).astype(np.uint32)
    counts[rng.integers(0, H, 40)] = counts.max()            # fitness ties -> the rmse callback is exercised
    rmse = lambda i: float((i * 2654435761) % 1000) / 1000.0
    for prob, mi in ((0.99, H), (1.0, H), (0.9999, 300)):
        a = capi.replay(n, capi.PLANE, mi, prob, valid[:mi], counts[:mi], rmse)
        st = capi.ReplayState()
        capi.lib().m3d_replay_init(C.byref(st))
        packed = np.ascontiguousarray(counts[:mi] | (valid[:mi].astype(np.uint32) << np.uint32(31)))
        cb = capi.RMSE_FN(lambda _u, i: rmse(int(i)))
        capi.lib().m3d_replay_chunk(C.byref(st), n, capi.PLANE, mi, prob, 0, mi, None,
                                    packed.ctypes.data_as(C.c_void_p), cb, None)
        for f, _ in capi.ReplayState._fields_:
            assert getattr(a, f) == getattr(st, f), (prob, f)
