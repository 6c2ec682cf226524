"""Generate the committed golden vectors of tests/golden/*.npz.

The reference has no tests or golden vectors of its own and cannot be built or
imported in this image, so these fixtures are produced by the ORACLE (oracle/, the CPU restatement --
"parity unpinned", see its header) in this container and committed as data: inputs and expected outputs.
They (1) freeze the oracle against silent changes and platform differences (CPU test), and (2) let the
HIP path be checked against data without building the oracle (GPU test).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from misc3d_amd import capi, synth  # noqa: E402

THR = 0.01


def fits():
    out = {}
    clouds = {0: (synth.plane_cloud_c1(1500, 1), None), 1: (synth.sphere_cloud_c3(1500, 4), None),
              2: synth.cylinder_cloud_c3(1500, 3)}
    for kind, (pts, nrm) in clouds.items():
        pts = np.ascontiguousarray(pts)
        m = capi.MINIMAL_SAMPLE[kind]
        samples = capi.draw_samples(len(pts), kind, 200, 21 + kind)           # host-only sampler of the C ABI
        osamp = oracle.draw_samples(len(pts), m, 200, 21 + kind)
        assert np.array_equal(samples.astype(np.uint64), np.asarray(osamp, dtype=np.uint64).reshape(samples.shape))
        v, models, counts, errors = oracle.score_samples(kind, pts, nrm, THR, samples.astype(np.uint64))
        pre = f"k{kind}_"
        out[pre + "points"] = pts
        if nrm is not None:
            out[pre + "normals"] = np.ascontiguousarray(nrm)
        out[pre + "samples"] = samples
        out[pre + "valid"] = np.asarray(v, dtype=np.uint8)
        out[pre + "models"] = np.asarray(models)
        out[pre + "counts"] = np.asarray(counts, dtype=np.uint64)
        for tag, (mi, prob, seed) in {"a": (300, 0.9999, 7), "b": (150, 1.0, 11)}.items():
            r = oracle.fit(kind, pts, nrm, thr=THR, max_iter=mi, prob=prob, seed=seed)
            out[pre + tag + "_args"] = np.array([mi, prob, seed], dtype=np.float64)
            out[pre + tag + "_ret_best_count_iter"] = np.array([r.ret, r.best_index, r.count, r.iterations], dtype=np.int64)
            out[pre + tag + "_fitness"] = np.array([r.fitness])
            out[pre + tag + "_params"] = r.params
            out[pre + tag + "_inliers"] = r.inliers.astype(np.uint64)
    np.savez_compressed(os.path.join(HERE, "fits.npz"), **out)


def segmentation():
    pts = np.ascontiguousarray(synth.room_cloud_c5(3000, 6))
    rc, planes, clusters = oracle.segment_plane_iterative(pts, 0.02, 120, 0.15, seed=19)
    out = {"points": pts, "args": np.array([0.02, 120, 0.15, 19]), "rc": np.array([rc]), "planes": planes,
           "offsets": np.cumsum([0] + [len(c) for c in clusters]).astype(np.uint64),
           "indices": np.concatenate(clusters).astype(np.uint64)}
    np.savez_compressed(os.path.join(HERE, "segmentation.npz"), **out)


def registration():
    d = synth.registration_pair_c4(1200, seed=5, dim=33, true_fraction=0.5, sigma=0.001)
    a, b = oracle.match_mutual_nn(d["feat_src"], d["feat_dst"])
    o = oracle.registration_ransac(d["src"], d["dst"], a, b, thr=0.03, max_iter=800, edge_thr=0.9, confidence=1.0, seed=17)
    T = oracle.umeyama(d["src"][a[:200]], d["dst"][b[:200]], with_scaling=False)
    Ts = oracle.umeyama(d["src"][a[:200]], 1.5 * d["dst"][b[:200]], with_scaling=True)
    out = {"src": d["src"], "dst": d["dst"], "feat_src": d["feat_src"], "feat_dst": d["feat_dst"],
           "match_src": np.asarray(a, dtype=np.uint64), "match_dst": np.asarray(b, dtype=np.uint64),
           "ransac_args": np.array([0.03, 800, 0.9, 1.0, 17]), "ransac_T": o.T,
           "ransac_stats": np.array([o.best_index, o.iterations, o.validations, o.est_k], dtype=np.int64),
           "ransac_fitness_rmse": np.array([o.fitness, o.inlier_rmse]), "kabsch_T": T, "kabsch_T_scaled": Ts}
    np.savez_compressed(os.path.join(HERE, "registration.npz"), **out)


def global_registration():
    """ReconstructionPipeline::GlobalRegistration (src/pipeline.cpp:790-828) on registration.npz's pair: accepted, and rejected
    when three quarters of the source are replaced by points that overlap nothing.  Inputs: registration.npz (+ the replacement)."""
    d = synth.registration_pair_c4(1200, seed=5, dim=33, true_fraction=0.5, sigma=0.001)
    args = dict(max_iter=1500, edge_thr=0.9, confidence=0.999, seed=23)
    vox = 0.03 / 1.4
    ok, T, info, nm = oracle.global_registration(d["src"], d["dst"], d["feat_src"], d["feat_dst"], vox, **args)
    far = np.random.default_rng(2).uniform(40.0, 60.0, size=(900, 3))
    cut = d["src"].copy()
    cut[300:] = far
    ok2, T2, info2, nm2 = oracle.global_registration(cut, d["dst"], d["feat_src"], d["feat_dst"], vox, **args)
    assert ok and not ok2
    np.savez_compressed(os.path.join(HERE, "global_registration.npz"), args=np.array([vox, 1500, 0.9, 0.999, 23]),
                        ok=np.array([ok, ok2]), T=np.stack([T, T2]), info=np.stack([info, info2]), matches=np.array([nm, nm2]),
                        far=far)


def next_rows():
    """SURVEY.md 8(f) rows that are in: normals from an organised map (N3), point-to-point ICP (N1)."""
    rng = np.random.default_rng(7)
    w, h, k = 48, 36, 2
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    z = 1.0 + 0.002 * u - 0.001 * v + 0.03 * np.sin(u / 7.0) + rng.normal(0, 1e-4, (h, w))
    xyz = np.stack([(u - w / 2) / 120.0 * z, (v - h / 2) / 120.0 * z, z], -1).reshape(-1, 3)
    xyz[rng.random(w * h) < 0.04] = np.nan
    vp = np.array([0.05, -0.02, -0.3])
    nrm = oracle.normals_from_map(xyz, w, h, k, vp)
    d = synth.registration_pair_c4(1500, seed=9, dim=8, sigma=0.001)
    init = d["T"].copy()
    init[:3, 3] += np.array([0.01, -0.006, 0.004])
    T, fit, rm, it, corr = oracle.registration_icp(d["src"], d["dst"], 0.02, init)
    info = oracle.information_matrix(d["src"], d["dst"], 0.03, T)
    uv = rng.uniform(0, 1, (1800, 2))
    uv = uv[np.hypot(uv[:, 0] - 0.5, uv[:, 1] - 0.5) > 0.2]
    bpts = np.ascontiguousarray(np.c_[uv[:, 0], uv[:, 1], 0.25 * uv[:, 0] - 0.1 * uv[:, 1] + rng.normal(0, 1e-4, len(uv))])
    bidx = oracle.detect_boundary_points(bpts, None, 2, 0.07, 30, 90.0)
    np.savez_compressed(os.path.join(HERE, "next_rows.npz"), info=info, boundary_points=bpts, boundary_index=bidx,
                        boundary_args=np.array([2, 0.07, 30, 90.0]), map_xyz=xyz, map_shape_k=np.array([w, h, k]),
                        map_view_point=vp, map_normals=nrm, icp_src=d["src"], icp_dst=d["dst"], icp_init=init,
                        icp_T=T, icp_fit_rmse=np.array([fit, rm]), icp_iterations=np.array([it]), icp_corr=corr)


def example_cloud():
    """The reference's example cloud, examples/data/segmentation/test.ply (40 458 points, binary little-endian
    double xyz written by Open3D) -- a DATA file, committed verbatim as tests/golden/segmentation_test.ply -- run
    through the calls the reference's examples make on it: SegmentPlaneIterative(pcd, 0.01, 100, 0.1)
    (examples/cpp/segment_plane_iterative.cpp:12-18) and, on the largest plane, DetectBoundaryPoints with
    KDTreeSearchParamHybrid(0.02, 30) (examples/cpp/ransac_and_boundary.cpp:44-47).  The examples seed their
    sampler from random_device; the fixture fixes seed 0.  /root/reference exists only in the build container, so
    the copy step is skipped when the data file is already in place."""
    import shutil
    from misc3d_amd import io
    dst = os.path.join(HERE, "segmentation_test.ply")
    src = "/root/reference/examples/data/segmentation/test.ply"
    if os.path.exists(src):
        shutil.copyfile(src, dst)
    pts = np.ascontiguousarray(io.read_ply(dst)["points"])
    rc, planes, clusters = oracle.segment_plane_iterative(pts, 0.01, 100, 0.1, seed=0)
    big = max(range(len(clusters)), key=lambda i: len(clusters[i]))
    plane_pts = np.ascontiguousarray(pts[np.asarray(clusters[big], dtype=np.int64)])
    bidx = oracle.detect_boundary_points(plane_pts, None, 2, 0.02, 30, 90.0)
    fit = oracle.fit(0, pts, None, thr=0.01, max_iter=1000, prob=0.9999, seed=0)
    np.savez_compressed(os.path.join(HERE, "example_cloud.npz"), args=np.array([0.01, 100, 0.1, 0]), rc=np.array([rc]),
                        planes=planes, offsets=np.cumsum([0] + [len(c) for c in clusters]).astype(np.uint64),
                        indices=np.concatenate(clusters).astype(np.uint64), largest=np.array([big]),
                        boundary_index=np.asarray(bidx, dtype=np.uint64),
                        fit_ret_best_count_iter=np.array([fit.ret, fit.best_index, fit.count, fit.iterations], dtype=np.int64),
                        fit_params=fit.params, fit_inliers=fit.inliers.astype(np.uint64))


if __name__ == "__main__":
    fits()
    example_cloud()
    segmentation()
    registration()
    global_registration()
    next_rows()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
