"""Model PLY schema of the reference (scene/gaussian_model.py:123-221)."""
import numpy as np

from splat_loam_amd import ply_io


def test_roundtrip_and_header(tmp_path):
    rng = np.random.default_rng(0)
    n = 257
    xyz, op, sc, rot = rng.normal(size=(n, 3)), rng.normal(size=(n, 1)), rng.normal(size=(n, 2)), rng.normal(size=(n, 4))
    p = tmp_path / "models" / "0000.ply"
    ply_io.save_ply(p, xyz, op, sc, rot)
    head = open(p, "rb").read(400).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 257\nproperty float x\n")
    order = [ln.split()[2] for ln in head.splitlines() if ln.startswith("property")]
    assert order == ["x", "y", "z", "opacity", "scale_0", "scale_1", "rot_0", "rot_1", "rot_2", "rot_3",
                     "f_dc_0", "f_dc_1", "f_dc_2"]
    assert p.stat().st_size == len(head[:head.index("end_header\n") + 11]) + n * 13 * 4
    m = ply_io.load_ply(p)
    for k, ref in (("xyz", xyz), ("opacity", op), ("scaling", sc), ("rotation", rot)):
        assert np.array_equal(m[k], ref.astype(np.float32))


def _g8():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_formats.npz"))


def test_g8_table_equals_what_the_reference_hands_to_plyfile(tmp_path):
    """G8 (tools/make_golden.py:g8): the structured array the reference's GaussianModel.save_ply
    (scene/gaussian_model.py:137-168) builds for `PlyElement.describe(elements, "vertex")`, recorded with a stand-in
    for plyfile: this repo's writer must put exactly that table on disk — names, order, dtypes, values."""
    import hashlib
    g = _g8()
    p = tmp_path / "0000.ply"
    ply_io.save_ply(p, g["in_xyz"], g["in_opacity"], g["in_scaling"], g["in_rotation"])
    blob = open(p, "rb").read()
    end = blob.index(b"end_header\n") + 11
    head = blob[:end].decode("ascii").split("\n")
    assert head[:3] == ["ply", "format binary_little_endian 1.0", f"element {g['ply_element_name']} {len(g['in_xyz'])}"]
    props = [ln.split() for ln in head if ln.startswith("property")]
    assert [pr[2] for pr in props] == list(g["ply_names"])
    # plyfile writes numpy '<f4' as the PLY type `float`
    assert all(dt == "<f4" for dt in g["ply_dtypes"]) and all(pr[1] == "float" for pr in props)
    table = np.frombuffer(blob, dtype="<f4", offset=end).reshape(-1, len(props))
    assert np.array_equal(table, g["ply_table"])
    assert len(blob) == end + g["ply_table"].nbytes
    # the file the reference's load_ply was run on when the fixture was made is the file written today
    assert hashlib.sha256(blob).hexdigest() == str(g["ours_sha256"])
    for k in ("xyz", "opacity", "scaling", "rotation"):
        assert np.array_equal(g["ref_loaded_" + k], g["in_" + k])


def test_g8_loader_reads_the_reference_table(tmp_path):
    """The other direction: a file holding the reference's table (header as plyfile writes a single float32
    element, then the table's bytes) read by this repo's load_ply."""
    g = _g8()
    names = list(g["ply_names"])
    head = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(g["ply_table"])
    head += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    p = tmp_path / "ref.ply"
    with open(p, "wb") as f:
        f.write(head.encode("ascii"))
        f.write(np.ascontiguousarray(g["ply_table"], dtype="<f4").tobytes())
    m = ply_io.load_ply(p)
    for k in ("xyz", "opacity", "scaling", "rotation"):
        assert np.array_equal(m[k], g["in_" + k])
