"""The C-ABI library loads without a GPU, exports every symbol include/sls_abi.h
declares, and its host-side entry points behave (no device compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from splat_loam_amd import _abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sls_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sls_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = C.CDLL(_abi.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in sls_abi.h but not exported"
        assert n in _abi.EXPORTS, f"{n} has no ctypes prototype in _abi.py"


def test_constants_match_spec_header():
    lib = _abi.lib()
    spec = open(os.path.join(ROOT, "include", "sls_spec.h")).read()
    assert lib.sls_rec_stride() == int(re.search(r"#define SLS_REC_STRIDE (\d+)", spec).group(1)) == 20
    assert lib.sls_grec_stride() == int(re.search(r"#define SLS_GREC_STRIDE (\d+)", spec).group(1)) == 16
    tw, th = _abi.tile_size()
    assert (tw * th) % 64 == 0 and tw % 8 == 0 and th % 8 == 0
    assert lib.sls_timing_slots() >= 12 and lib.sls_timing_name(7) == b"render_fwd"


def test_camera_from_matrices_and_ray_tables(oracle32):
    lib = _abi.lib()
    H, W = 64, 1024
    K = synth.spherical_K(H, W)
    view, proj = synth.camera_matrices(K, synth.keyframe_poses(4)[3])
    cam = _abi.SlsCamera()
    _abi.check(lib.sls_camera_from_matrices(view.ctypes.data, proj.ctypes.data, H, W, 1.0, C.byref(cam)), "camera")
    ocam = oracle32.camera(H, W, view, proj, tile=_abi.tile_size())
    assert cam.wrap == ocam.wrap == 1
    assert np.array_equal(np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32), ocam.fcam[:4])
    assert np.array_equal(np.array(list(cam.Rvw) + list(cam.tvw), np.float32), ocam.fcam[7:19])
    assert (cam.near_cut, cam.far_cut) == (np.float32(0.2), np.float32(100.0))
    col = np.empty((W, 2), np.float32); row = np.empty((H, 2), np.float32)
    _abi.check(lib.sls_ray_tables(C.byref(cam), col.ctypes.data, row.ctypes.data), "tables")
    ocol, orow = oracle32.ray_tables(ocam)
    assert np.array_equal(col, ocol) and np.array_equal(row, orow)          # bit-exact
    # narrow horizontal field of view: no wrap
    view, proj = synth.camera_matrices(synth.spherical_K(H, W, hfov_deg=120.0))
    _abi.check(lib.sls_camera_from_matrices(view.ctypes.data, proj.ctypes.data, H, W, 1.0, C.byref(cam)), "camera")
    assert cam.wrap == 0


def test_error_reporting_never_throws():
    lib = _abi.lib()
    cam = _abi.SlsCamera()
    rc = lib.sls_camera_from_matrices(None, None, 4, 4, 1.0, C.byref(cam))
    assert rc == -1 and b"null pointer" in lib.sls_last_error()
    proj = np.eye(4, dtype=np.float32); proj[1, 0] = 0.3                    # skewed K
    rc = lib.sls_camera_from_matrices(np.eye(4, dtype=np.float32).ctypes.data, proj.ctypes.data, 4, 4, 1.0, C.byref(cam))
    assert rc == -1 and b"skew" in lib.sls_last_error()
    with pytest.raises(RuntimeError, match="skew"):
        _abi.check(rc, "sls_camera_from_matrices")
    grp = (_abi.SlsAdamGroup * 1)()
    assert lib.sls_adam_step(grp, 0, 0.9, 0.999, 1e-15, 1, None) == -1
    assert lib.sls_adam_step(grp, 1, 0.9, 0.999, 1e-15, 0, None) == -1       # step is 1-based
    assert lib.sls_knn_scratch_bytes(0) == 0 and lib.sls_knn_scratch_bytes(1000) > 1000 * 32
    assert lib.sls_sort_scratch_bytes(0) >= 1024 and lib.sls_stage1_scratch_bytes(1000) >= 16


def test_product_never_touches_the_checker():
    """No module of the product imports or opens anything under oracle/."""
    pkg = os.path.join(ROOT, "splat_loam_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"(from|import)\s+oracle|oracle/|liboracle", txt):
                    bad.append(os.path.join(dirpath, f))
    for shim in ("diff_surfel_spherical_rasterization/__init__.py", "simple_knn/_C.py", "gsaligner/__init__.py"):
        if re.search(r"oracle", open(os.path.join(ROOT, shim)).read()):
            bad.append(shim)
    assert not bad, bad


def test_engine_bucket_surgery_on_cpu():
    """Host logic of MappingEngine.remap (prune / densify between keyframes): the flat optimiser buckets keep the
    survivors' rows group by group and append zero rows."""
    import torch
    from splat_loam_amd.engine import carry_bucket
    n_old, n_new = 7, 8
    groups = [torch.arange(n_old * w, dtype=torch.float32).view(n_old, w) + 100.0 * k for k, w in enumerate((3, 1, 2, 4))]
    buf = torch.cat([g.reshape(-1) for g in groups])
    keep = torch.tensor([1, 0, 1, 1, 0, 0, 1], dtype=torch.bool)
    out = carry_bucket(buf, keep, n_new)
    assert out.numel() == 10 * n_new
    off = 0
    for g, w in zip(groups, (3, 1, 2, 4)):
        got = out[off:off + w * n_new].view(n_new, w)
        assert torch.equal(got[:4], g[keep]) and float(got[4:].abs().max()) == 0.0
        off += w * n_new
    import pytest
    with pytest.raises(ValueError):
        carry_bucket(buf, keep, 3)


def test_frame_from_precomp_inverts_the_reference_construction():
    """`cov3D_precomp` in the layout the reference's model builds (scene/gaussian_model.py:20-36: a (N,4,4) transform
    with [:3,:3] = (R diag(s_u, s_v, 1))^T and the centre in row 3; utils/general_utils.py:13-48 for R(q)) converts back
    to the (scales, w-x-y-z quaternion) the rasterizer takes — up to the quaternion's sign."""
    import torch
    from splat_loam_amd.rasterizer import frame_from_precomp
    g = torch.Generator().manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(2000, 4, generator=g), dim=1)
    q[:4] = torch.tensor([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]])   # the four pure branches
    s = torch.rand(2000, 2, generator=g) * 0.2 + 1e-3
    r, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    L = R @ torch.diag_embed(torch.cat([s, torch.ones(2000, 1)], 1))
    trans = torch.zeros(2000, 4, 4)
    trans[:, :3, :3] = L.permute(0, 2, 1)
    trans[:, 3, :3] = torch.randn(2000, 3, generator=g)
    trans[:, 3, 3] = 1
    for t in (trans, trans[:, :3, :3].contiguous()):
        s2, q2 = frame_from_precomp(t)
        sign = torch.sign((q * q2).sum(1, keepdim=True))
        assert float((s2 - s).abs().max()) <= 1e-6
        assert float((q2 * sign - q).abs().max()) <= 2e-6
        assert float((q2.norm(dim=1) - 1).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        frame_from_precomp(torch.zeros(5, 6))


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """The Python side mirrors the header's structs by hand (splat_loam_amd/_abi.py): a field added on one side only moves
    every later field silently.  A C program compiled against include/sls_abi.h prints size and field offsets; the ctypes
    classes must agree field by field (names included)."""
    import ctypes as C
    import os
    import shutil
    import subprocess
    from splat_loam_amd import _abi
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    classes = ("SlsCamera", "SlsMappingConfig", "SlsMappingStatus", "SlsAlignerParams", "SlsAlignerResult", "SlsAdamGroup")
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "sls_abi.h"', "int main(void) {"]
    for name in classes:
        cls = getattr(_abi, name)
        lines.append(f'  printf("{name} size %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.strip()}
    for name in classes:
        cls = getattr(_abi, name)
        assert got[(name, "size")] == C.sizeof(cls), f"{name}: {got[(name, 'size')]} bytes in the header, {C.sizeof(cls)} in ctypes"
        for field, _ in cls._fields_:
            assert got[(name, field)] == getattr(cls, field).offset, f"{name}.{field}"
