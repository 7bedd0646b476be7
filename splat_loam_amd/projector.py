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


class DeviceProjector:
    """The same projection on the MI355X (csrc/sls_projector.hip through the C ABI): the scan never
    leaves the device, K is computed and consumed there, no host synchronisation.  One instance per
    image size; images are freshly allocated per scan (they become the keyframe's targets).

        proj = DeviceProjector(H, W, depth_min, depth_max)
        out = proj.scan_to_images(cloud_cuda)     # dict like scan_to_images(), device tensors; K (3,3) on device
    """

    def __init__(self, H: int, W: int, depth_min: float = 0.5, depth_max: float = 100.0,
                 full_azimuth_threshold_deg: float = 300.0, device="cuda:0"):
        import torch
        from . import _abi
        self.H, self.W = int(H), int(W)
        self.depth_min, self.depth_max = float(depth_min), float(depth_max)
        self.thr = float(full_azimuth_threshold_deg)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("DeviceProjector runs on the GPU only (no CPU fallback); use scan_to_images() on the host")
        lib = _abi.lib()
        self._bytes = int(lib.sls_projector_scratch_bytes(self.H, self.W))
        self._scratch = torch.empty((self._bytes + 7) // 8, dtype=torch.int64, device=self.dev)
        _abi.check(lib.sls_projector_prepare(self.H, self.W, self._scratch.data_ptr(), self._bytes, self._stream()),
                   "sls_projector_prepare")

    def _stream(self):
        import torch
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _cloud(self, cloud):
        import torch
        if not (isinstance(cloud, torch.Tensor) and cloud.is_cuda):
            raise RuntimeError("cloud must be a CUDA tensor (no CPU fallback)")
        if cloud.dtype != torch.float32 or cloud.dim() != 2 or cloud.shape[1] != 3:
            raise RuntimeError("cloud must be (N,3) float32")
        return cloud.contiguous()

    def intrinsics(self, cloud):
        """12 device floats: K row-major, vfov, hfov, any-point flag (calculate_spherical_intrinsics)."""
        import torch
        from . import _abi
        cloud = self._cloud(cloud)
        out = torch.empty(12, dtype=torch.float32, device=self.dev)
        _abi.check(_abi.lib().sls_projector_intrinsics(cloud.shape[0], cloud.data_ptr(), self.H, self.W, self.thr,
                                                       out.data_ptr(), self._scratch.data_ptr(), self._bytes,
                                                       self._stream()), "sls_projector_intrinsics")
        return out

    def project(self, cloud, K):
        """(lut int32 (H,W), range_image (H,W), normals_image (H,W,3), valid bool (H,W)) for K: 9+ device floats."""
        import torch
        from . import _abi
        cloud = self._cloud(cloud)
        if not (isinstance(K, torch.Tensor) and K.is_cuda and K.dtype == torch.float32 and K.numel() >= 9):
            raise RuntimeError("K must be a CUDA float32 tensor with the 9 row-major entries first")
        K = K.contiguous()
        H, W = self.H, self.W
        lut = torch.empty((H, W), dtype=torch.int32, device=self.dev)
        rng = torch.empty((H, W), dtype=torch.float32, device=self.dev)
        nrm = torch.empty((H, W, 3), dtype=torch.float32, device=self.dev)
        valid = torch.empty((H, W), dtype=torch.uint8, device=self.dev)
        _abi.check(_abi.lib().sls_projector_project(cloud.shape[0], cloud.data_ptr(), K.data_ptr(), H, W, self.depth_min,
                                                    self.depth_max, lut.data_ptr(), rng.data_ptr(), nrm.data_ptr(),
                                                    valid.data_ptr(), self._scratch.data_ptr(), self._bytes,
                                                    self._stream()), "sls_projector_project")
        return lut, rng, nrm, valid.view(torch.bool)

    def scan_to_images(self, cloud):
        intr = self.intrinsics(cloud)
        lut, rng, nrm, valid = self.project(cloud, intr)
        return dict(K=intr[:9].view(3, 3), vfov=intr[9], hfov=intr[10], range_image=rng, normals_image=nrm,
                    valid=valid, lut=lut)
