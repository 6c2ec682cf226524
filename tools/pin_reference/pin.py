"""Pin the oracle to the real reference: export the golden inputs, (elsewhere) run tools/pin_reference/pin_reference on
them, compare the reference's records with the oracle built for each floating-point association.

    python tools/pin_reference/pin.py export  <dir>      # writes <dir>/k{0,1,2}.in from tests/golden/fits.npz (+ a larger seeded set)
    <dir>/pin_reference <dir>                            # on a box with Eigen + Open3D 0.15.1 (tools/pin_reference/run.sh)
    python tools/pin_reference/pin.py compare <dir>      # which M3D_FP_ORDER / ORC_FP_ORDER reproduces the reference bit for bit
    python tools/pin_reference/pin.py selftest <dir>     # no reference needed: writes .ref files FROM THE ORACLE and compares
                                                         # (checks the file formats and this script end to end)

Legs: (1) per-hypothesis records of the reference's estimators on a fixed sample table (k<kind>.in/.ref);
(2) the reference's OWN RANSAC<...>::FitModel on a fixed seed (pin_seed.h replaces the sampler's random_device):
return value, parameters, the "run {} iterations" count and the full inlier list, with the adaptive stop (probability
0.9999) and without (1.0) -- d<kind>_<case>.in/.ref; (3) misc3d::segmentation::SegmentPlaneIterative itself on the
reference's example cloud (examples/data/segmentation/test.ply = tests/golden/segmentation_test.ply, its example's
arguments) and on a synthetic room -- seg_<case>.in/.ref; (4) optional: Open3D's RegistrationRANSACBasedOnCorrespondence
as RANSACSolver::Solve calls it (reg.in/.ref; informational: the oracle restates Open3D 0.15.1 from memory).

`compare` also writes <dir>/reference_fits.npz -- the reference's own vectors in the layout of tests/golden/fits.npz
(inputs + valid / models / counts): committed under tests/golden/ they turn DESIGN.md's "parity unpinned" into pinned.
"""
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
NPAR = {0: 4, 1: 4, 2: 7}
M = {0: 3, 1: 4, 2: 2}
THR = 0.01


def load_inputs():
    d = np.load(os.path.join(ROOT, "tests", "golden", "fits.npz"))
    out = {}
    for kind in (0, 1, 2):
        pts = d[f"k{kind}_points"]
        nrm = d[f"k{kind}_normals"] if f"k{kind}_normals" in d.files else None
        out[kind] = (np.ascontiguousarray(pts), None if nrm is None else np.ascontiguousarray(nrm),
                     np.ascontiguousarray(d[f"k{kind}_samples"]).astype(np.uint64))
    return out


def export(dirname):
    os.makedirs(dirname, exist_ok=True)
    for kind, (pts, nrm, samples) in load_inputs().items():
        with open(os.path.join(dirname, f"k{kind}.in"), "wb") as f:
            f.write(struct.pack("<4Qd", len(pts), 0 if nrm is None else 1, len(samples), M[kind], THR))
            f.write(pts.astype("<f8").tobytes())
            if nrm is not None:
                f.write(nrm.astype("<f8").tobytes())
            f.write(samples.astype("<u8").tobytes())
    export_driver(dirname)
    print("wrote", dirname, "/k{0,1,2}.in, d{0,1,2}_{adaptive,exhaustive}.in, seg_{example,room}.in, reg.in")


DRIVER_CASES = {"adaptive": (300, 0.9999, 5), "exhaustive": (300, 1.0, 6)}      # tag -> (max_iteration, probability, seed)


def seg_inputs():
    """tag -> (points, threshold, max_iteration, min_ratio, seed)"""
    from misc3d_amd import io as m3d_io, synth
    ply = m3d_io.read_ply(os.path.join(ROOT, "tests", "golden", "segmentation_test.ply"))
    pts = np.ascontiguousarray(ply[0] if isinstance(ply, tuple) else (ply["points"] if isinstance(ply, dict) else ply), dtype=np.float64)
    return {"example": (pts, 0.01, 100, 0.1, 0),                                # examples/cpp/segment_plane_iterative.cpp:12-18
            "room": (synth.room_cloud_c5(30000, 6), 0.01, 60, 0.05, 3)}


def reg_inputs():
    from misc3d_amd import synth
    d = synth.registration_pair_c4(3000, seed=5)
    import oracle
    i0, i1 = oracle.match_mutual_nn(d["feat_src"], d["feat_dst"])
    return d["src"], d["dst"], np.asarray(i0, dtype=np.uint64), np.asarray(i1, dtype=np.uint64), 0.03, 2000, 0.9, 0.999, 17


def export_driver(dirname):
    for kind, (pts, nrm, _samples) in load_inputs().items():
        for tag, (max_iter, prob, seed) in DRIVER_CASES.items():
            with open(os.path.join(dirname, f"d{kind}_{tag}.in"), "wb") as f:
                f.write(struct.pack("<4Q2d", len(pts), 0 if nrm is None else 1, max_iter, seed, THR, prob))
                f.write(pts.astype("<f8").tobytes())
                if nrm is not None:
                    f.write(nrm.astype("<f8").tobytes())
    for tag, (pts, thr, max_iter, min_ratio, seed) in seg_inputs().items():
        with open(os.path.join(dirname, f"seg_{tag}.in"), "wb") as f:
            f.write(struct.pack("<3Q2d", len(pts), max_iter, seed, thr, min_ratio))
            f.write(pts.astype("<f8").tobytes())
    src, dst, i0, i1, thr, max_iter, edge, conf, seed = reg_inputs()
    with open(os.path.join(dirname, "reg.in"), "wb") as f:
        f.write(struct.pack("<5Q3d", len(src), len(dst), len(i0), max_iter, seed, thr, edge, conf))
        f.write(np.ascontiguousarray(src, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(dst, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(np.stack([i0, i1], axis=1), dtype="<u8").tobytes())


def read_driver_ref(path):
    with open(path, "rb") as f:
        ret, npar = struct.unpack("<qQ", f.read(16))
        params = np.frombuffer(f.read(8 * npar), dtype="<f8")
        count, ni = struct.unpack("<qQ", f.read(16))
        inliers = np.frombuffer(f.read(8 * ni), dtype="<u8")
    return dict(ret=ret, params=params, count=count, inliers=inliers)


def read_seg_ref(path):
    with open(path, "rb") as f:
        (k,) = struct.unpack("<Q", f.read(8))
        planes, sizes = [], []
        for _ in range(k):
            planes.append(np.frombuffer(f.read(32), dtype="<f8"))
            sizes.append(struct.unpack("<Q", f.read(8))[0])
        clusters = [np.frombuffer(f.read(24 * sz), dtype="<f8").reshape(sz, 3) for sz in sizes]
    return np.array(planes).reshape(k, 4), clusters


def oracle_driver(order, dirname):
    """the oracle's fits / segmentations of the exported cases under association `order` (subprocess) -> npz"""
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import oracle, pin\n"
            "out = {}\n"
            "for kind, (pts, nrm, _s) in pin.load_inputs().items():\n"
            "    for tag, (mi, prob, seed) in pin.DRIVER_CASES.items():\n"
            "        r = oracle.fit(kind, pts, nrm, thr=pin.THR, max_iter=mi, prob=prob, seed=seed)\n"
            "        pre = f'd{kind}_{tag}_'\n"
            "        out[pre + 'ret'] = np.int64(r.ret); out[pre + 'params'] = r.params; out[pre + 'count'] = np.int64(r.count); out[pre + 'inliers'] = r.inliers\n"
            "for tag, (pts, thr, mi, mr, seed) in pin.seg_inputs().items():\n"
            "    rc, planes, clusters = oracle.segment_plane_iterative(pts, thr, max_iteration=mi, min_ratio=mr, seed=seed)\n"
            "    out[f'seg_{tag}_planes'] = planes; out[f'seg_{tag}_sizes'] = np.array([len(c) for c in clusters], dtype=np.int64)\n"
            "    out[f'seg_{tag}_points'] = np.concatenate([pts[c.astype(np.int64)] for c in clusters]) if clusters else np.zeros((0, 3))\n"
            "src, dst, i0, i1, thr, mi, edge, conf, seed = pin.reg_inputs()\n"
            "rr = oracle.registration_ransac(src, dst, i0.astype(np.int64), i1.astype(np.int64), thr=thr, max_iter=mi, edge_thr=edge, confidence=conf, seed=seed)\n"
            "out['reg_T'] = rr.T; out['reg_fitness'] = np.float64(rr.fitness); out['reg_rmse'] = np.float64(rr.inlier_rmse)\n"
            "np.savez(sys.argv[1], **out)\n") % (ROOT, HERE)
    tmp = os.path.join(dirname, f"oracle_driver_{order}.npz")
    subprocess.run([sys.executable, "-c", code, tmp], check=True, env=dict(os.environ, M3D_FP_ORDER=str(order)))
    return np.load(tmp)


def compare_driver(dirname, orders):
    """legs 2-4 against the oracle built for each association in `orders`; -> {order: all legs that exist agree}"""
    verdict = {}
    for order in orders:
        o = oracle_driver(order, dirname)
        ok_all, seen = True, 0
        for kind in (0, 1, 2):
            for tag in DRIVER_CASES:
                path = os.path.join(dirname, f"d{kind}_{tag}.ref")
                if not os.path.exists(path):
                    continue
                seen += 1
                r = read_driver_ref(path)
                pre = f"d{kind}_{tag}_"
                same_list = np.array_equal(r["inliers"], o[pre + "inliers"])
                dpar = float(np.max(np.abs(r["params"] - o[pre + "params"]))) if len(r["params"]) == len(o[pre + "params"]) else np.inf
                same_count = r["count"] < 0 or r["count"] == int(o[pre + "count"])
                same_ret = r["ret"] == int(o[pre + "ret"])
                # (GeneralFit: the reference's plane covariance is an OpenMP reduction and its sphere fit an SVD: 1e-9, not bits)
                ok = same_list and same_count and same_ret and dpar <= 1e-9
                print(f"order {order} driver kind {kind} {tag}: ret {'==' if same_ret else '!='}  iterations "
                      f"{'==' if same_count else '!='} ({r['count']} vs {int(o[pre + 'count'])})  inlier list "
                      f"{'identical' if same_list else 'DIFFERS'} ({len(r['inliers'])})  |params diff| {dpar:.2e}")
                ok_all = ok_all and ok
        for tag in ("example", "room"):
            path = os.path.join(dirname, f"seg_{tag}.ref")
            if not os.path.exists(path):
                continue
            seen += 1
            planes, clusters = read_seg_ref(path)
            sizes = o[f"seg_{tag}_sizes"]
            same_sizes = len(clusters) == len(sizes) and all(len(c) == int(s) for c, s in zip(clusters, sizes))
            same_pts = same_sizes and np.array_equal(bits(np.concatenate(clusters) if clusters else np.zeros((0, 3))),
                                                     bits(o[f"seg_{tag}_points"]))
            dpl = float(np.max(np.abs(planes - o[f"seg_{tag}_planes"]))) if same_sizes and len(planes) else (0.0 if same_sizes else np.inf)
            print(f"order {order} segmentation {tag}: {len(clusters)} clusters, sizes {'==' if same_sizes else '!='}  "
                  f"cluster points {'bit-equal' if same_pts else 'DIFFER'}  |plane diff| {dpl:.2e}")
            ok_all = ok_all and same_pts and dpl <= 1e-9
        path = os.path.join(dirname, "reg.ref")
        if os.path.exists(path):      # informational: Open3D's sampler and SVD are restated from memory
            with open(path, "rb") as f:
                T = np.frombuffer(f.read(128), dtype="<f8").reshape(4, 4)
                fit, rmse = struct.unpack("<2d", f.read(16))
            print(f"order {order} registration (informational): |T diff| {float(np.max(np.abs(T - o['reg_T']))):.2e}  fitness "
                  f"{fit:.6f} vs {float(o['reg_fitness']):.6f}  rmse {rmse:.6g} vs {float(o['reg_rmse']):.6g}")
        verdict[order] = ok_all if seen else None
    return verdict


def read_ref(dirname, kind):
    with open(os.path.join(dirname, f"k{kind}.ref"), "rb") as f:
        H, npar = struct.unpack("<2Q", f.read(16))
        valid = np.frombuffer(f.read(H), dtype=np.uint8)
        models = np.frombuffer(f.read(8 * H * npar), dtype="<f8").reshape(H, npar)
        counts = np.frombuffer(f.read(8 * H), dtype="<u8")
        errors = np.frombuffer(f.read(8 * H), dtype="<f8")
        best, gf_ok = struct.unpack("<qQ", f.read(16))
        refined = np.frombuffer(f.read(8 * npar), dtype="<f8")
    return dict(valid=valid, models=models, counts=counts, errors=errors, best=best, gf_ok=gf_ok, refined=refined)


def oracle_records(order):
    """(valid, models, counts, errors) of the oracle built for association `order`, in a subprocess (the oracle module
    picks its library from M3D_FP_ORDER at import)."""
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import oracle, pin\n"
            "out = {}\n"
            "for kind, (pts, nrm, samples) in pin.load_inputs().items():\n"
            "    v, m, c, e = oracle.score_samples(kind, pts, nrm, %r, samples)\n"
            "    out[f'v{kind}'] = np.asarray(v, dtype=np.uint8); out[f'm{kind}'] = np.asarray(m); out[f'c{kind}'] = np.asarray(c, dtype=np.uint64); out[f'e{kind}'] = np.asarray(e)\n"
            "np.savez(sys.argv[1], **out)\n") % (ROOT, HERE, THR)
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"m3d_pin_oracle_{order}.npz")
    subprocess.run([sys.executable, "-c", code, tmp], check=True, env=dict(os.environ, M3D_FP_ORDER=str(order)))
    return np.load(tmp)


def bits(a):
    return np.nan_to_num(np.ascontiguousarray(a, dtype=np.float64), nan=-7.0).view(np.uint64)


def compare(dirname):
    refs = {k: read_ref(dirname, k) for k in (0, 1, 2)}
    names = {0: "Eigen >= 3.3 (default)", 1: "Eigen 3.2 three-sums", 2: "Eigen 3.4 determinant"}
    verdict = {}
    for order in (0, 1, 2):
        o = oracle_records(order)
        ok_all = True
        for kind in (0, 1, 2):
            r = refs[kind]
            v = r["valid"].astype(bool)
            same_valid = np.array_equal(r["valid"], o[f"v{kind}"])
            same_models = same_valid and np.array_equal(bits(r["models"][v]), bits(o[f"m{kind}"][v]))
            same_counts = same_valid and np.array_equal(r["counts"], o[f"c{kind}"])
            same_errors = same_valid and np.array_equal(bits(r["errors"]), bits(o[f"e{kind}"]))
            n_model_diff = int((bits(r["models"][v]) != bits(o[f"m{kind}"][v])).any(axis=1).sum()) if same_valid else -1
            print(f"order {order} ({names[order]}), kind {kind}: valid {'==' if same_valid else '!='}  models "
                  f"{'bit-equal' if same_models else f'{n_model_diff} differ'}  counts {'==' if same_counts else '!='}  "
                  f"serial error sums {'bit-equal' if same_errors else 'differ'}")
            ok_all = ok_all and same_models and same_counts and same_errors
        verdict[order] = ok_all
    drv = compare_driver(dirname, (0, 1, 2))
    for k, v in drv.items():
        if v is not None:
            verdict[k] = verdict[k] and v
    if all(v is None for v in drv.values()):
        print("(no driver / segmentation .ref files in", dirname, "-- only the per-hypothesis leg was compared)")
    good = [k for k, v in verdict.items() if v]
    print("associations that reproduce the reference bit for bit:", good or "NONE -- see the per-kind lines above")
    if good:
        print(f"-> build with M3D_FP_ORDER={good[0]} / ORC_FP_ORDER={good[0]} as the default (misc3d_amd/csrc/m3d_fp.hpp, "
              f"oracle/misc3d_oracle.c) and commit {dirname}/reference_fits.npz under tests/golden/")
    out = {}
    for kind, (pts, nrm, samples) in load_inputs().items():
        pre = f"k{kind}_"
        out[pre + "points"] = pts
        if nrm is not None:
            out[pre + "normals"] = nrm
        out[pre + "samples"] = samples.astype(np.uint32)
        out[pre + "valid"] = refs[kind]["valid"]
        out[pre + "models"] = refs[kind]["models"]
        out[pre + "counts"] = refs[kind]["counts"]
        out[pre + "errors"] = refs[kind]["errors"]
        out[pre + "refined"] = refs[kind]["refined"]
    np.savez_compressed(os.path.join(dirname, "reference_fits.npz"), **out)
    return 0 if good else 1


def selftest(dirname):
    """Writes .ref files from the ORACLE (association 0) in pin_reference's format and runs `compare` on them."""
    export(dirname)
    o = oracle_records(0)
    for kind in (0, 1, 2):
        H = len(o[f"v{kind}"])
        with open(os.path.join(dirname, f"k{kind}.ref"), "wb") as f:
            f.write(struct.pack("<2Q", H, NPAR[kind]))
            f.write(o[f"v{kind}"].tobytes())
            f.write(np.ascontiguousarray(o[f"m{kind}"], dtype="<f8").tobytes())
            f.write(o[f"c{kind}"].astype("<u8").tobytes())
            f.write(np.ascontiguousarray(o[f"e{kind}"], dtype="<f8").tobytes())
            f.write(struct.pack("<qQ", int(np.argmax(o[f"c{kind}"])), 1))
            f.write(np.zeros(NPAR[kind], dtype="<f8").tobytes())
    # the driver, segmentation and registration legs in pin_reference.cpp's formats, from the oracle (association 0)
    od = oracle_driver(0, dirname)
    for kind in (0, 1, 2):
        for tag in DRIVER_CASES:
            pre = f"d{kind}_{tag}_"
            with open(os.path.join(dirname, f"d{kind}_{tag}.ref"), "wb") as f:
                f.write(struct.pack("<qQ", int(od[pre + "ret"]), NPAR[kind]))
                f.write(np.ascontiguousarray(od[pre + "params"], dtype="<f8").tobytes())
                f.write(struct.pack("<qQ", int(od[pre + "count"]), len(od[pre + "inliers"])))
                f.write(od[pre + "inliers"].astype("<u8").tobytes())
    for tag in ("example", "room"):
        sizes = od[f"seg_{tag}_sizes"]
        with open(os.path.join(dirname, f"seg_{tag}.ref"), "wb") as f:
            f.write(struct.pack("<Q", len(sizes)))
            for k, sz in enumerate(sizes):
                f.write(np.ascontiguousarray(od[f"seg_{tag}_planes"][k], dtype="<f8").tobytes())
                f.write(struct.pack("<Q", int(sz)))
            f.write(np.ascontiguousarray(od[f"seg_{tag}_points"], dtype="<f8").tobytes())
    with open(os.path.join(dirname, "reg.ref"), "wb") as f:
        f.write(np.ascontiguousarray(od["reg_T"], dtype="<f8").tobytes())
        f.write(struct.pack("<2dQ", float(od["reg_fitness"]), float(od["reg_rmse"]), 0))
    # pin_seed.h's token replacement against this image's libstdc++ (no reference needed): a sampler-shaped class seeded
    # through `std::random_device` must produce std::mt19937(seed)'s stream, and seed + 1 for the second object
    chk = os.path.join(dirname, "pin_seed_check.cpp")
    with open(chk, "w") as f:
        f.write('#include "pin_seed.h"\n#include <cstdio>\nextern "C" { uint64_t m3d_pin_seed = 0, m3d_pin_calls = 0; }\n'
                'struct S { std::mt19937 g; S() { std::random_device rd; g = std::mt19937(rd()); } };\n'
                'int main() { m3d_pin_seed = 41; S a, b; std::mt19937 r0(41), r1(42); int bad = 0;\n'
                '  for (int i = 0; i < 1000; ++i) bad += (a.g() != r0()) + (b.g() != r1());\n'
                '  std::printf("%d\\n", bad); return bad != 0; }\n')
    exe = os.path.join(dirname, "pin_seed_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", HERE, chk, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "0", "pin_seed.h does not replace std::random_device here"
    print("pin_seed.h: std::random_device replaced, streams equal std::mt19937(seed), (seed + 1)")
    return compare(dirname)


if __name__ == "__main__":
    cmd, dirname = sys.argv[1], sys.argv[2]
    sys.exit({"export": lambda d: export(d) or 0, "compare": compare, "selftest": selftest}[cmd](dirname))
