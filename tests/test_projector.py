"""BASELINE config 1 (CPU plumbing): synthetic HDL-64-like scan -> 64x1024 spherical
range/normal/valid images -> back-projection with the reference-pinned
depth_to_points convention returns the scan."""
import math

import numpy as np
import torch

from splat_loam_amd import projector, renderer
from splat_loam_amd.scene import Camera


def _scan(H=64, n_az=2000, seed=0):
    rng = np.random.default_rng(seed)
    el = np.radians(np.linspace(2.0, -24.8, H))
    az = np.linspace(-math.pi, math.pi, n_az, endpoint=False) + 1e-3
    A, E = np.meshgrid(az, el)
    R = 8.0 + 4.0 * np.sin(3 * A) + 2.0 * np.cos(5 * E) + rng.uniform(0, 0.05, A.shape)
    pts = np.stack([R * np.cos(A) * np.cos(E), R * np.sin(A) * np.cos(E), R * np.sin(E)], -1).reshape(-1, 3)
    return pts.astype(np.float32)


def test_scan_to_images_contract_and_roundtrip():
    H, W = 64, 1024
    cloud = _scan(H)
    out = projector.scan_to_images(cloud, H, W, depth_min=0.5, depth_max=100.0)
    assert out["range_image"].shape == (H, W) and out["normals_image"].shape == (H, W, 3)
    assert out["valid"].dtype == bool and out["valid"].mean() > 0.95
    assert abs(out["hfov"] - 2 * math.pi) < 1e-9 and out["K"][0, 0] < 0 and out["K"][1, 1] < 0
    lut, valid = out["lut"], out["valid"]
    assert np.array_equal(lut == -1, ~valid)
    assert np.allclose(out["range_image"][valid], np.linalg.norm(cloud[lut[valid]], axis=1), rtol=1e-6)
    n = out["normals_image"][valid]
    assert np.allclose(n, -cloud[lut[valid]] / np.linalg.norm(cloud[lut[valid]], axis=1, keepdims=True), atol=1e-6)
    # back-projection through the reference-pinned convention (c-0.5, r-0.5): the direction of every
    # valid pixel is within half a bin (diagonal) of the point that landed in it
    cam = Camera(out["K"], out["range_image"][None], out["normals_image"].transpose(2, 0, 1),
                 valid[None].astype(np.uint8), None, data_device="cpu")
    pts = renderer.depth_to_points(cam, torch.from_numpy(out["range_image"][None]), False).numpy().transpose(1, 2, 0)
    src = cloud[lut[valid]]
    cosang = (pts[valid] * src).sum(1) / (np.linalg.norm(pts[valid], axis=1) * np.linalg.norm(src, axis=1))
    ang = np.arccos(np.clip(cosang, -1, 1))
    half_diag = 0.5 * math.hypot(out["hfov"] / W, out["vfov"] / H)
    assert ang.max() <= half_diag * 1.01
    assert np.allclose(np.linalg.norm(pts[valid], axis=1), np.linalg.norm(src, axis=1), rtol=1e-5)


def test_points_at_pixel_centres_roundtrip_exactly_and_nearest_wins():
    H, W = 16, 64
    el_max, el_min = math.radians(10), math.radians(-20)
    K = np.array([[-W / (2 * math.pi), 0, W / 2 - 1], [0, -H / (el_max - el_min), H * el_max / (el_max - el_min) - 1],
                  [0, 0, 1]], np.float64)
    Kinv = np.linalg.inv(K)
    pts, want = [], {}
    for (c, r, rho) in ((0, 0, 5.0), (63, 15, 7.0), (31, 8, 9.0), (31, 8, 4.0), (12, 3, 150.0), (40, 9, 0.1)):
        a, e, _ = Kinv @ np.array([c - 0.5, r - 0.5, 1.0])
        pts.append(rho * np.array([math.cos(a) * math.cos(e), math.sin(a) * math.cos(e), math.sin(e)]))
    pts = np.array(pts, np.float32)
    lut = projector.project(pts, K, H, W, depth_min=0.5, depth_max=100.0)
    assert lut[0, 0] == 0 and lut[15, 63] == 1
    assert lut[8, 31] == 3                       # the nearer of the two returns in that pixel
    assert lut[3, 12] == -1 and lut[9, 40] == -1   # outside the depth window
    assert (lut >= 0).sum() == 3


# ---- the HIP projector (csrc/sls_projector.hip) against the NumPy one ----------------------------------
import pytest


def _jittered_scan(H, W, seed, frac=0.35, dup=False):
    """Points inside their pixel, at most `frac` px from the centre (no float32-vs-float64 boundary cases)."""
    rng = np.random.default_rng(seed)
    el_max, el_min = math.radians(3.0), math.radians(-25.0)
    K = np.array([[-W / (2 * math.pi), 0, W / 2 - 1], [0, -H / (el_max - el_min), H * el_max / (el_max - el_min) - 1],
                  [0, 0, 1]], np.float64)
    cc, rr = np.meshgrid(np.arange(W), np.arange(H))
    keep = rng.uniform(size=cc.shape) < 0.9
    cc, rr = cc[keep], rr[keep]
    if dup:      # several returns per pixel: the nearest must win
        cc, rr = np.concatenate([cc, cc[::2], cc[::3]]), np.concatenate([rr, rr[::2], rr[::3]])
    u = cc - 0.5 + rng.uniform(-frac, frac, cc.shape)
    v = rr - 0.5 + rng.uniform(-frac, frac, cc.shape)
    az, el = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
    rho = rng.uniform(0.2, 120.0, cc.shape)          # some outside the (0.5, 100] window
    pts = np.stack([rho * np.cos(az) * np.cos(el), rho * np.sin(az) * np.cos(el), rho * np.sin(el)], -1)
    perm = rng.permutation(len(pts))
    return pts[perm].astype(np.float32), K.astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,dup", [(64, 1024, False), (64, 2048, True), (16, 64, True)])
def test_hip_projector_matches_numpy_exactly(device, H, W, dup):
    cloud, K = _jittered_scan(H, W, seed=H + W, dup=dup)
    want = projector.project(cloud, K, H, W, 0.5, 100.0)
    proj = projector.DeviceProjector(H, W, 0.5, 100.0, device=device)
    cd, Kd = torch.tensor(cloud, device=device), torch.tensor(K.reshape(-1), device=device)
    for _ in range(2):                               # second pass: the scratch was left ready by the first
        lut, rng_img, nrm, valid = proj.project(cd, Kd)
        lut = lut.cpu().numpy()
        assert np.array_equal(lut, want)             # integer image: bit-exact
        ok = want >= 0
        assert np.array_equal(valid.cpu().numpy(), ok)
        ranges = np.linalg.norm(cloud, axis=1)       # float32, as scene/preprocessing.py:55
        exp_r = np.where(ok, ranges[want], 0.0).astype(np.float32)
        assert np.array_equal(rng_img.cpu().numpy(), exp_r)
        exp_n = np.where(ok[..., None], -cloud[want] / ranges[want, None], 0.0).astype(np.float32)
        assert np.array_equal(nrm.cpu().numpy(), exp_n)


@pytest.mark.gpu
def test_hip_projector_scan_to_images_like_numpy(device):
    H, W = 64, 1024
    cloud = _scan(H)
    want = projector.scan_to_images(cloud, H, W, depth_min=0.5, depth_max=100.0)
    proj = projector.DeviceProjector(H, W, 0.5, 100.0, device=device)
    got = proj.scan_to_images(torch.tensor(cloud, device=device))
    K = got["K"].cpu().numpy()
    assert np.allclose(K, want["K"], rtol=2e-6, atol=2e-5)
    assert abs(float(got["hfov"]) - want["hfov"]) < 1e-6 and abs(float(got["vfov"]) - want["vfov"]) < 1e-6
    lut = got["lut"].cpu().numpy()
    # float32 angles on the device, float64 on the host: a return within 1e-3 px of a pixel border may
    # land next door; everything else is identical
    diff = lut != want["lut"]
    assert diff.mean() < 2e-3
    same = ~diff & (lut >= 0)
    assert np.array_equal(got["range_image"].cpu().numpy()[same], want["range_image"][same])
    assert np.allclose(got["normals_image"].cpu().numpy()[same], want["normals_image"][same], atol=1e-6)
    # partial azimuth coverage: padded field of view instead of the full circle
    part = cloud[np.abs(np.arctan2(cloud[:, 1], cloud[:, 0])) < 1.0]
    Kp, vfov, hfov, _, _ = projector.calculate_spherical_intrinsics(part, H, W)
    intr = proj.intrinsics(torch.tensor(part, device=device)).cpu().numpy()
    assert np.allclose(intr[:9].reshape(3, 3), Kp, rtol=2e-6, atol=2e-5) and abs(intr[10] - hfov) < 1e-6


@pytest.mark.gpu
def test_hip_projector_edge_cases(device):
    H, W = 16, 64
    proj = projector.DeviceProjector(H, W, 0.5, 100.0, device=device)
    K = torch.tensor([-W / (2 * math.pi), 0, W / 2 - 1, 0, -H / 0.5, H * 0.25 / 0.5 - 1, 0, 0, 1],
                     dtype=torch.float32, device=device)
    empty = torch.zeros((0, 3), dtype=torch.float32, device=device)
    lut, rng_img, nrm, valid = proj.project(empty, K)
    assert int((lut != -1).sum()) == 0 and not bool(valid.any()) and float(rng_img.abs().sum()) == 0.0
    assert float(proj.intrinsics(empty)[11]) == 0.0
    # origin and NaN points are ignored; equal ranges: the later point wins (NumPy's stable ordering)
    pts = torch.tensor([[0, 0, 0], [float("nan"), 1, 1], [5, 0, 0], [5, 0, 0]], dtype=torch.float32, device=device)
    lut, rng_img, nrm, valid = proj.project(pts, K)
    want = projector.project(pts.cpu().numpy()[2:], K.cpu().numpy().reshape(3, 3), H, W, 0.5, 100.0)
    got = lut.cpu().numpy()
    assert np.array_equal(got >= 0, want >= 0) and int((got >= 0).sum()) == 1
    assert got.max() == want.max() + 2 == 3
    with pytest.raises(RuntimeError):
        proj.project(pts.cpu(), K)


@pytest.mark.gpu
def test_device_images_make_a_keyframe_without_leaving_the_gpu(device):
    """DeviceProjector output -> Camera (tensors taken as they are) == the NumPy path's Camera."""
    H, W = 64, 1024
    cloud, _ = _jittered_scan(H, W, seed=5)
    proj = projector.DeviceProjector(H, W, 0.5, 100.0, device=device)
    cd = torch.tensor(cloud, device=device)
    got = proj.scan_to_images(cd)
    cam = Camera(got["K"], got["range_image"][None], got["normals_image"].permute(2, 0, 1), got["valid"][None], None,
                 data_device=str(device))
    assert cam.image_depth.data_ptr() == got["range_image"].data_ptr()      # no copy
    lut_np = projector.project(cloud, got["K"].cpu().numpy(), H, W, 0.5, 100.0)
    assert (got["lut"].cpu().numpy() != lut_np).mean() < 2e-3       # (pixel-border cases, float32 vs float64 angles)
    ref = Camera(got["K"].cpu().numpy(), got["range_image"].cpu().numpy()[None],
                 got["normals_image"].cpu().numpy().transpose(2, 0, 1), got["valid"].cpu().numpy()[None], None,
                 data_device=str(device))
    for k in ("image_depth", "image_normal", "projection_matrix", "world_view_transform"):
        assert torch.equal(getattr(cam, k), getattr(ref, k)), k
    assert torch.equal(cam.image_valid.bool(), ref.image_valid.bool())
