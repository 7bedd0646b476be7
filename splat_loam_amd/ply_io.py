"""On-disk model format of the reference: binary little-endian
PLY, one `vertex` element with float32 properties
`x y z opacity scale_0 scale_1 rot_0..rot_3 f_dc_0..f_dc_2` holding the RAW
(pre-activation) parameters (scene/gaussian_model.py:123-168; read back by
`load_ply`, :170-221, which looks properties up by name).  Written with NumPy
only (the reference uses `plyfile`, which produces the same byte layout for a
single all-float32 element), so models saved here load in the reference's
`mesh` / `eval_*` commands and vice versa.
"""
from __future__ import annotations

import os

import numpy as np

PROPS = ["x", "y", "z", "opacity", "scale_0", "scale_1", "rot_0", "rot_1", "rot_2", "rot_3",
         "f_dc_0", "f_dc_1", "f_dc_2"]


def save_ply(path, xyz, opacity_raw, scaling_raw, rotation_raw) -> None:
    def host(a):      # NumPy, or tensors on any device (the reference saves `.detach().cpu().numpy()`, gaussian_model.py:133-141)
        return np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=np.float32)
    xyz, opacity_raw, scaling_raw, rotation_raw = (host(a) for a in (xyz, opacity_raw, scaling_raw, rotation_raw))
    n = xyz.shape[0]
    data = np.concatenate([xyz.reshape(n, 3), opacity_raw.reshape(n, 1), scaling_raw.reshape(n, 2),
                           rotation_raw.reshape(n, 4), np.zeros((n, 3), np.float32)], axis=1).astype("<f4")
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {p}\n" for p in PROPS) + "end_header\n"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())


def load_ply(path) -> dict:
    """dict(xyz (N,3), opacity (N,1), scaling (N,S), rotation (N,4)) of raw parameters; properties are
    looked up by name like the reference does, so extra/reordered float properties are tolerated."""
    with open(path, "rb") as f:
        blob = f.read()
    end = blob.index(b"end_header\n") + len(b"end_header\n")
    lines = blob[:end].decode("ascii").splitlines()
    if lines[0] != "ply" or "binary_little_endian" not in lines[1]:
        raise ValueError("only binary little-endian PLY is supported")
    n, names = 0, []
    kinds = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
             "int": "<i4", "int32": "<i4", "uint": "<u4", "short": "<i2", "ushort": "<u2", "char": "i1"}
    dtype = []
    in_vertex = False
    for ln in lines:
        tok = ln.split()
        if tok[:1] == ["element"]:
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                n = int(tok[2])
        elif tok[:1] == ["property"] and in_vertex:
            dtype.append((tok[2], kinds[tok[1]]))
            names.append(tok[2])
    rec = np.frombuffer(blob, dtype=np.dtype(dtype), count=n, offset=end)
    col = lambda name: np.asarray(rec[name], dtype=np.float32)
    scale_names = sorted([p for p in names if p.startswith("scale_")], key=lambda s: int(s.split("_")[-1]))
    rot_names = sorted([p for p in names if p.startswith("rot")], key=lambda s: int(s.split("_")[-1]))
    return dict(xyz=np.stack([col("x"), col("y"), col("z")], 1), opacity=col("opacity")[:, None],
                scaling=np.stack([col(s) for s in scale_names], 1), rotation=np.stack([col(s) for s in rot_names], 1))
