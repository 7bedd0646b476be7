"""CPU restatement (NumPy, float64) of the frame-to-keyframe registration of
splat_loam_amd/csrc/sls_aligner.hip — TEST INFRASTRUCTURE ONLY (tests/, never
imported by the product).

"Parity unpinned": the reference's `gsaligner` submodule is not vendored
(/root/reference/.gitmodules, pixi.toml:42), only its call interface is visible
(slam/tracker.py:141-197).  The algorithm restated here is this repository's own
specification; the functional tests (a known motion between two
synthetic scans is recovered) anchor it, the GPU tests compare the HIP kernels with it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class Params:
    num_iterations: int = 15
    min_inliers: int = 64
    max_distance: float = 1.0
    min_cos_angle: float = math.cos(math.radians(80.0))
    huber_delta: float = 0.10
    range_weight: float = 0.25
    range_huber: float = 0.30
    depth_min: float = 0.5
    depth_max: float = 100.0
    damping: float = 1e-6


def cam_of(K, H, W):
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    wrap = abs(abs(fx) * 2.0 * math.pi - W) <= 1.0
    return dict(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, wrap=wrap)


def normals(cam, depth, points, depth_min):
    """depth (H,W), points (H,W,3) -> normals (H,W,3); follows aligner_normals_kernel."""
    H, W = cam["H"], cam["W"]
    d = np.asarray(depth, np.float64).reshape(H, W)
    p = np.asarray(points, np.float64).reshape(H, W, 3)
    n = np.zeros((H, W, 3))
    ok = d > depth_min
    up, dn = np.roll(p, -1, 0), np.roll(p, 1, 0)
    okv = ok & np.roll(ok, -1, 0) & np.roll(ok, 1, 0)
    okv[0] = okv[-1] = False
    rt, lf = np.roll(p, -1, 1), np.roll(p, 1, 1)
    okh = np.roll(ok, -1, 1) & np.roll(ok, 1, 1)
    if not cam["wrap"]:
        okh[:, 0] = okh[:, -1] = False
    c = np.cross(up - dn, rt - lf)
    ln = np.linalg.norm(c, axis=2)
    good = okv & okh & (ln > 1e-12)
    c = c / np.maximum(ln, 1e-300)[..., None]
    s = np.where((c * p).sum(2) > 0.0, -1.0, 1.0)
    n[good] = (c * s[..., None])[good]
    return n


def _huber(e, delta):
    a = np.abs(e)
    return np.where(a <= delta, 1.0, delta / np.maximum(a, 1e-300))


def linearize(cam, prm: Params, ref_depth, ref_points, ref_normals, q_depth, q_points, T):
    """Returns sys (30,): H upper triangle (21) | b (6) | chi2 | inliers | valid query pixels."""
    H, W = cam["H"], cam["W"]
    rd = np.asarray(ref_depth, np.float64).reshape(-1)
    rp = np.asarray(ref_points, np.float64).reshape(-1, 3)
    rn = np.asarray(ref_normals, np.float64).reshape(-1, 3)
    qd = np.asarray(q_depth, np.float64).reshape(-1)
    qp = np.asarray(q_points, np.float64).reshape(-1, 3)
    T = np.asarray(T, np.float64)
    valid = (qd > prm.depth_min) & (qd <= prm.depth_max)
    p = qp @ T[:3, :3].T + T[:3, 3]
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    rxy2 = x * x + y * y
    rho2 = rxy2 + z * z
    rxy, rho = np.sqrt(rxy2), np.sqrt(rho2)
    ok = valid & (rho > prm.depth_min) & (rxy > 1e-6)
    az, el = np.arctan2(y, x), np.arctan2(z, np.maximum(rxy, 1e-300))
    u, v = cam["fx"] * az + cam["cx"], cam["fy"] * el + cam["cy"]
    c = np.floor(u + 1.0).astype(np.int64)
    r = np.floor(v + 1.0).astype(np.int64)
    if cam["wrap"]:
        c = np.mod(c, W)
    ok &= (c >= 0) & (c < W) & (r >= 0) & (r < H)
    j = np.where(ok, r * W + c, 0)
    dr = rd[j]
    n = rn[j]
    ok &= (dr > prm.depth_min) & (dr <= prm.depth_max) & (np.abs(n).sum(1) > 0)
    diff = p - rp[j]
    cosang = -(n * p).sum(1) / np.maximum(rho, 1e-300)
    ok &= ((diff * diff).sum(1) <= prm.max_distance ** 2) & (cosang >= prm.min_cos_angle)
    sys = np.zeros(30)
    sys[29] = valid.sum()
    sys[28] = ok.sum()
    if not ok.any():
        return sys

    def add(J, e, w):
        Hm = (J * w[:, None]).T @ J
        sys[:21] += Hm[np.triu_indices(6)]
        sys[21:27] += (J * (w * e)[:, None]).sum(0)
        sys[27] += (w * e * e).sum()

    p_, n_, diff_ = p[ok], n[ok], diff[ok]
    e = (n_ * diff_).sum(1)
    J = np.concatenate([n_, np.cross(p_, n_)], 1)
    add(J, e, _huber(e, prm.huber_delta))
    if prm.range_weight > 0.0:
        rdi = rd.reshape(H, W)
        rr, cc = r[ok], c[ok]
        cl, cr = cc - 1, cc + 1
        if cam["wrap"]:
            cl, cr = np.mod(cl, W), np.mod(cr, W)
        inb = (cl >= 0) & (cr < W)
        a, b = rdi[rr, np.clip(cl, 0, W - 1)], rdi[rr, np.clip(cr, 0, W - 1)]
        gu = np.where(inb & (a > prm.depth_min) & (b > prm.depth_min), 0.5 * (b - a), 0.0)
        inr = (rr > 0) & (rr < H - 1)
        a, b = rdi[np.clip(rr - 1, 0, H - 1), cc], rdi[np.clip(rr + 1, 0, H - 1), cc]
        gv = np.where(inr & (a > prm.depth_min) & (b > prm.depth_min), 0.5 * (b - a), 0.0)
        xo, yo, zo = p_[:, 0], p_[:, 1], p_[:, 2]
        rxy2o, rho2o = rxy2[ok], rho2[ok]
        rxyo, rhoo = np.sqrt(rxy2o), np.sqrt(rho2o)
        er = rhoo - dr[ok]
        iu, iv = cam["fx"] / rxy2o, cam["fy"] / (rxyo * rho2o)
        g = np.stack([xo / rhoo - gu * (-yo * iu) - gv * (-xo * zo * iv),
                      yo / rhoo - gu * (xo * iu) - gv * (-yo * zo * iv),
                      zo / rhoo - gv * (rxy2o * iv)], 1)
        Jr = np.concatenate([g, np.cross(p_, g)], 1)
        add(Jr, er, prm.range_weight * _huber(er, prm.range_huber))
    return sys


def se3_exp(xi):
    v, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-6:
        A, B, C = 1 - th * th / 6, 0.5 - th * th / 24, 1 / 6 - th * th / 120
    else:
        A, B = math.sin(th) / th, (1 - math.cos(th)) / (th * th)
        C = (1 - A) / (th * th)
    R = np.eye(3) + A * K + B * K @ K
    V = np.eye(3) + B * K + C * K @ K
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, V @ v
    return T


def solve_update(sys, T, prm: Params):
    Hm = np.zeros((6, 6))
    Hm[np.triu_indices(6)] = sys[:21]
    Hm = Hm + Hm.T - np.diag(np.diag(Hm))
    if sys[28] < prm.min_inliers:
        return T, 0.0
    xi = np.linalg.solve(Hm + prm.damping * np.eye(6), -sys[21:27])
    return se3_exp(xi) @ T, float(np.linalg.norm(xi))


def align(cam, prm: Params, ref_depth, ref_points, ref_normals, q_depth, q_points, T0):
    T = np.array(T0, np.float64)
    step = 0.0
    for _ in range(prm.num_iterations):
        sys = linearize(cam, prm, ref_depth, ref_points, ref_normals, q_depth, q_points, T)
        T, step = solve_update(sys, T, prm)
    sys = linearize(cam, prm, ref_depth, ref_points, ref_normals, q_depth, q_points, T)
    fitness = sys[28] / sys[29] if sys[29] > 0 else 0.0
    return T, fitness, dict(chi2=sys[27], inliers=int(sys[28]), valid_query=int(sys[29]), last_step=step)
