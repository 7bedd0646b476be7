"""Seeded synthetic scenes: a 360-degree spherical LiDAR
camera with a KITTI-like vertical field of view and N random surfels facing
the sensor.  NumPy only, so the CPU checker and the HIP path see identical
float32 inputs.  No dataset exists in the build container; every test and
bench.py use this generator and say so (`"data": "synthetic"`).
"""
from __future__ import annotations

import math

import numpy as np

EL_MIN_DEG = -24.8
EL_MAX_DEG = 2.0


def spherical_K(H: int, W: int, el_min_deg: float = EL_MIN_DEG, el_max_deg: float = EL_MAX_DEG,
                hfov_deg: float = 360.0) -> np.ndarray:
    """K with u = fx*az + cx, v = fy*el + cy (the intrinsics convention
    utils/graphic_utils.py:41-55 inverts).  Row 0 is the top beam."""
    el_min, el_max = math.radians(el_min_deg), math.radians(el_max_deg)
    hfov = math.radians(hfov_deg)
    fx = -W / hfov
    cx = W / 2.0
    fy = -H / (el_max - el_min)
    cy = -fy * el_max
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)


def camera_matrices(K: np.ndarray, world_T_lidar: np.ndarray | None = None):
    """(viewmatrix, projmatrix) exactly as scene/cameras.py:43-50 builds them."""
    if world_T_lidar is None:
        world_T_lidar = np.eye(4)
    view = np.ascontiguousarray(np.linalg.inv(np.asarray(world_T_lidar, dtype=np.float64)).T, dtype=np.float32)
    proj = np.eye(4, dtype=np.float32)
    proj[:3, :3] = np.asarray(K, dtype=np.float32).T
    return view, proj


def keyframe_poses(n: int) -> list[np.ndarray]:
    """n poses 0.5 m apart along +x with <= 2 deg yaw."""
    poses = []
    for k in range(n):
        yaw = math.radians(2.0) * math.sin(1.7 * k)
        T = np.eye(4)
        T[:3, :3] = [[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]]
        T[0, 3] = 0.5 * k
        poses.append(T)
    return poses


def _quat_from_R(R: np.ndarray) -> np.ndarray:
    """Batched rotation matrix -> unit quaternion (w,x,y,z), w >= 0."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q = np.empty((R.shape[0], 4))
    tr = m00 + m11 + m22
    # robust branch selection
    c0 = tr > 0
    c1 = ~c0 & (m00 >= m11) & (m00 >= m22)
    c2 = ~c0 & ~c1 & (m11 >= m22)
    c3 = ~c0 & ~c1 & ~c2
    s = np.sqrt(np.maximum(tr[c0] + 1.0, 1e-30)) * 2
    q[c0] = np.stack([0.25 * s, (R[c0, 2, 1] - R[c0, 1, 2]) / s, (R[c0, 0, 2] - R[c0, 2, 0]) / s,
                      (R[c0, 1, 0] - R[c0, 0, 1]) / s], 1)
    s = np.sqrt(np.maximum(1.0 + m00[c1] - m11[c1] - m22[c1], 1e-30)) * 2
    q[c1] = np.stack([(R[c1, 2, 1] - R[c1, 1, 2]) / s, 0.25 * s, (R[c1, 0, 1] + R[c1, 1, 0]) / s,
                      (R[c1, 0, 2] + R[c1, 2, 0]) / s], 1)
    s = np.sqrt(np.maximum(1.0 + m11[c2] - m00[c2] - m22[c2], 1e-30)) * 2
    q[c2] = np.stack([(R[c2, 0, 2] - R[c2, 2, 0]) / s, (R[c2, 0, 1] + R[c2, 1, 0]) / s, 0.25 * s,
                      (R[c2, 1, 2] + R[c2, 2, 1]) / s], 1)
    s = np.sqrt(np.maximum(1.0 + m22[c3] - m00[c3] - m11[c3], 1e-30)) * 2
    q[c3] = np.stack([(R[c3, 1, 0] - R[c3, 0, 1]) / s, (R[c3, 0, 2] + R[c3, 2, 0]) / s,
                      (R[c3, 1, 2] + R[c3, 2, 1]) / s, 0.25 * s], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[q[:, 0] < 0] *= -1
    return q


def make_scene(N: int, H: int, W: int, seed: int = 0, range_lo: float = 3.0, range_hi: float = 60.0,
               scale_lo: float = 0.02, scale_hi: float = 0.10, opac_lo: float = 0.1, opac_hi: float = 0.99,
               max_tilt_deg: float = 30.0):
    """Returns dict(K, means (N,3), scales (N,2), rots (N,4) wxyz unit, opac (N,1)), all float32."""
    rng = np.random.default_rng(seed)
    el_min, el_max = math.radians(EL_MIN_DEG), math.radians(EL_MAX_DEG)
    az = rng.uniform(-math.pi, math.pi, N)
    el = rng.uniform(el_min, el_max, N)
    rng_m = np.exp(rng.uniform(math.log(range_lo), math.log(range_hi), N))
    ray = np.stack([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)], 1)
    means = ray * rng_m[:, None]
    # normal = -ray tilted by <= max_tilt about a random tangent axis
    helper = np.where(np.abs(ray[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    e1 = np.cross(ray, helper)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(ray, e1)
    tilt = rng.uniform(0, math.radians(max_tilt_deg), N)
    phi = rng.uniform(0, 2 * math.pi, N)
    tn = -ray * np.cos(tilt)[:, None] + (np.cos(phi)[:, None] * e1 + np.sin(phi)[:, None] * e2) * np.sin(tilt)[:, None]
    tn /= np.linalg.norm(tn, axis=1, keepdims=True)
    h2 = np.where(np.abs(tn[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    t0 = np.cross(tn, h2)
    t0 /= np.linalg.norm(t0, axis=1, keepdims=True)
    t1 = np.cross(tn, t0)
    psi = rng.uniform(0, 2 * math.pi, N)
    tu = np.cos(psi)[:, None] * t0 + np.sin(psi)[:, None] * t1
    tv = np.cross(tn, tu)
    R = np.stack([tu, tv, tn], 2)  # columns
    rots = _quat_from_R(R)
    scales = np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), (N, 2)))
    opac = rng.uniform(opac_lo, opac_hi, (N, 1))
    return dict(K=spherical_K(H, W), means=means.astype(np.float32), scales=scales.astype(np.float32),
                rots=rots.astype(np.float32), opac=opac.astype(np.float32))


def make_targets(H: int, W: int, scene: dict, seed: int = 0):
    """Cheap measurement images for the mapper-style loss: per-pixel range =
    0.97 x nearest surfel centre range splatted coarsely, valid = ones."""
    K = scene["K"].astype(np.float64)
    m = scene["means"].astype(np.float64)
    rho = np.linalg.norm(m, axis=1)
    az = np.arctan2(m[:, 1], m[:, 0])
    el = np.arcsin(m[:, 2] / rho)
    c = np.clip(np.rint(K[0, 0] * az + K[0, 2]).astype(np.int64), 0, W - 1)
    r = np.clip(np.rint(K[1, 1] * el + K[1, 2]).astype(np.int64), 0, H - 1)
    depth = np.full((H, W), np.inf)
    np.minimum.at(depth, (r, c), rho)
    med = float(np.median(rho))
    depth[~np.isfinite(depth)] = med
    depth = (0.97 * depth).astype(np.float32)[None]
    valid = np.ones((1, H, W), dtype=np.uint8)
    return depth, valid
