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
