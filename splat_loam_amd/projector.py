"""Spherical projection of a LiDAR scan to range / normal / valid images — the
CPU plumbing of scene/preprocessing.py:24-83 (BASELINE config 1).

The reference does this with `pyprojections` 0.0.3 (C++/pybind, PyPI sdist, not
installed and not vendored — "unpinned").  What the tree pins is the output
contract (`range_image (H,W) f32`, `normals_image (H,W,3)`, `valid (H,W) bool`,
`lut (H,W)` with -1 for empty pixels, the depth window `(depth_min, depth_max]`,
normals = -unit(point) by default, scene/preprocessing.py:42-64,112) and the
back-projection it must agree with: pixel (c, r) <-> (az, el) = K^-1 [c-0.5, r-0.5, 1]
(utils/graphic_utils.py:41-59).  This module defines K so that
`depth_to_points(project(cloud)) == cloud` for points at bin centres:

    fx = -W/hfov, cx = W*az_max/hfov - 1     (az decreases with the column index)
    fy = -H/vfov, cy = H*el_max/vfov - 1     (row 0 is the top beam)
    pixel index of image coordinate u:  floor(u + 1)   (centre of pixel c is u = c - 0.5)

NumPy only; nearest return wins (z-buffer on range).
"""
from __future__ import annotations

import math

import numpy as np


def calculate_spherical_intrinsics(cloud: np.ndarray, H: int, W: int, full_azimuth_threshold_deg: float = 300.0):
    """(K, vfov, hfov, az_max, el_max) for a (N,3) cloud.  A scan spanning more than
    `full_azimuth_threshold_deg` of azimuth is treated as a 360-degree sensor."""
    pts = np.asarray(cloud, dtype=np.float64)
    rng = np.linalg.norm(pts, axis=1)
    ok = rng > 0
    az = np.arctan2(pts[ok, 1], pts[ok, 0])
    el = np.arcsin(np.clip(pts[ok, 2] / rng[ok], -1.0, 1.0))
    el_min, el_max = float(el.min()), float(el.max())
    span = float(az.max() - az.min())
    if math.degrees(span) >= full_azimuth_threshold_deg:
        az_max, hfov = math.pi, 2.0 * math.pi
    else:
        pad = span / max(W - 1, 1) * 0.5
        az_max, hfov = float(az.max()) + pad, span + 2 * pad
    pad = (el_max - el_min) / max(H - 1, 1) * 0.5
    el_max, vfov = el_max + pad, (el_max - el_min) + 2 * pad
    fx, fy = -W / hfov, -H / vfov
    K = np.array([[fx, 0.0, W * az_max / hfov - 1.0], [0.0, fy, H * el_max / vfov - 1.0], [0.0, 0.0, 1.0]],
                 dtype=np.float32)
    return K, vfov, hfov, az_max, el_max


def project(cloud: np.ndarray, K: np.ndarray, H: int, W: int, depth_min: float, depth_max: float):
    """lut (H,W) int64: index of the nearest point falling into each pixel, -1 if none."""
    pts = np.asarray(cloud, dtype=np.float64)
    rng = np.linalg.norm(pts, axis=1)
    keep = (rng > depth_min) & (rng <= depth_max)
    az = np.arctan2(pts[:, 1], pts[:, 0])
    el = np.arcsin(np.clip(pts[:, 2] / np.maximum(rng, 1e-30), -1.0, 1.0))
    K = np.asarray(K, dtype=np.float64)
    u = K[0, 0] * az + K[0, 2]
    v = K[1, 1] * el + K[1, 2]
    c = np.floor(u + 1.0).astype(np.int64)
    r = np.floor(v + 1.0).astype(np.int64)
    hfov_px = abs(K[0, 0]) * 2.0 * math.pi
    if abs(hfov_px - W) <= 1.0:
        c = np.mod(c, W)                      # 360-degree image: az = +-pi are the same column
    keep &= (c >= 0) & (c < W) & (r >= 0) & (r < H)
    lut = np.full((H, W), -1, dtype=np.int64)
    idx = np.nonzero(keep)[0]
    order = idx[np.argsort(-rng[idx], kind="stable")]      # far first, so the nearest write lands last
    lut[r[order], c[order]] = order
    return lut


def scan_to_images(cloud: np.ndarray, H: int, W: int, depth_min: float = 0.5, depth_max: float = 100.0):
    """The image triple the reference's Preprocessor builds (scene/preprocessing.py:42-64):
    returns dict(K, range_image (H,W), normals_image (H,W,3), valid (H,W) bool, lut)."""
    cloud = np.asarray(cloud, dtype=np.float32)
    K, vfov, hfov, _, _ = calculate_spherical_intrinsics(cloud, H, W)
    lut = project(cloud, K, H, W, depth_min, depth_max)
    invalid = lut == -1
    ranges = np.linalg.norm(cloud, axis=1).astype(np.float32)
    range_image = ranges[lut]
    range_image[invalid] = 0.0
    normals = np.zeros_like(cloud, dtype=np.float32)
    ok = (ranges > depth_min) & (ranges <= depth_max)
    normals[ok] = -cloud[ok] / ranges[ok, None]                          # scene/preprocessing.py:112
    normals_image = normals[lut]
    normals_image[invalid] = 0.0
    return dict(K=K, vfov=vfov, hfov=hfov, range_image=range_image, normals_image=normals_image,
                valid=~invalid, lut=lut)
