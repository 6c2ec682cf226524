"""Pin the oracle to the real reference: export the golden inputs, (elsewhere) run tools/pin_reference/pin_reference on
them, compare the reference's records with the oracle built for each floating-point association.

    python tools/pin_reference/pin.py export  <dir>      # writes <dir>/k{0,1,2}.in from tests/golden/fits.npz (+ a larger seeded set)
    <dir>/pin_reference <dir>                            # on a box with Eigen + Open3D 0.15.1 (tools/pin_reference/run.sh)
    python tools/pin_reference/pin.py compare <dir>      # which M3D_FP_ORDER / ORC_FP_ORDER reproduces the reference bit for bit
    python tools/pin_reference/pin.py selftest <dir>     # no reference needed: writes .ref files FROM THE ORACLE and compares
                                                         # (checks the file formats and this script end to end)

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
    print("wrote", dirname, "/k{0,1,2}.in")


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
    return compare(dirname)


if __name__ == "__main__":
    cmd, dirname = sys.argv[1], sys.argv[2]
    sys.exit({"export": lambda d: export(d) or 0, "compare": compare, "selftest": selftest}[cmd](dirname))
