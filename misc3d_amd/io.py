"""Open3D-free reader / writer for the PLY point clouds the reference ships under examples/data
(e.g. examples/data/segmentation/test.ply, binary little-endian, double x y z).

Only what those files need: one `vertex` element with scalar properties; ascii or binary_little_endian;
x y z (+ nx ny nz, + red green blue) are picked out, other properties are skipped; other elements
(faces) are ignored.  Returns float64 arrays ready for the m3d.* functions."""
from __future__ import annotations

import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def read_ply(path):
    """-> dict(points (N,3) float64, normals (N,3) float64 or None, colors (N,3) float64 in [0,1] or None)"""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt = None
        elements = []            # (name, count, [(prop, dtype)])
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append((tok[-1], None))
                else:
                    if tok[1] not in _TYPES:
                        raise ValueError(f"unsupported PLY property type {tok[1]}")
                    elements[-1][2].append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"unsupported PLY format {fmt}")
        if not elements or elements[0][0] != "vertex":
            raise ValueError("the first PLY element must be `vertex`")
        _, n, props = elements[0]
        if any(t is None for _, t in props):
            raise ValueError("list properties on the vertex element are not supported")
        dt = np.dtype([(name, "<" + t) for name, t in props])
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            data = {name: rows[:, k] for k, (name, _) in enumerate(props)}
        else:
            raw = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
            data = {name: raw[name] for name, _ in props}

    def pick(names, scale=1.0):
        if not all(k in data for k in names):
            return None
        return np.ascontiguousarray(np.stack([np.asarray(data[k], dtype=np.float64) for k in names], axis=1)) * scale

    color_scale = 1.0
    if "red" in data and np.asarray(data["red"]).dtype.kind in "ui":
        color_scale = 1.0 / 255.0
    return {"points": pick(("x", "y", "z")), "normals": pick(("nx", "ny", "nz")),
            "colors": pick(("red", "green", "blue"), color_scale)}


def write_ply(path, points, normals=None, binary=True):
    """double-precision x y z (+ nx ny nz), the layout of the reference's example clouds"""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    cols = [pts]
    names = ["x", "y", "z"]
    if normals is not None:
        cols.append(np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3))
        names += ["nx", "ny", "nz"]
    arr = np.concatenate(cols, axis=1)
    header = "ply\nformat %s 1.0\nelement vertex %d\n%send_header\n" % (
        "binary_little_endian" if binary else "ascii", len(pts), "".join(f"property double {k}\n" for k in names))
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        if binary:
            f.write(arr.astype("<f8").tobytes())
        else:
            np.savetxt(f, arr, fmt="%.17g")
