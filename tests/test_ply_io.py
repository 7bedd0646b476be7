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
