"""Dense float64 autograd restatement of the rasterizer forward (TEST
INFRASTRUCTURE; see sls_oracle.c header — "parity unpinned").

Independent of sls_oracle.c in two ways: (1) it uses the textbook ray-plane
formulation  x = t d,  u = Tu.(x - p)/su,  v = Tv.(x - p)/sv  instead of the
cancellation-free Hu/Hv form the kernels use, and (2) gradients come from
torch autograd, not from hand-written derivatives.  It is O(N * H * W) and
meant for N <= ~100 surfels on images of a few hundred pixels.

Integer decisions that are not differentiable (tile rectangles, the global
(depth, index) order) are taken from the C checker's preprocess so both
evaluate the same function.
"""
from __future__ import annotations

import numpy as np
import torch

from .oracle import FAR, NEAR, Camera

# the C spec constants are float literals (include/sls_spec.h); use their exact values
ALPHA_MAX = float(np.float32(0.99))
ALPHA_MIN = float(np.float32(1.0) / np.float32(255.0))
T_MIN = float(np.float32(1.0e-4))


def build_rotation(q: torch.Tensor) -> torch.Tensor:
    """utils/general_utils.py:13-37 without the normalisation."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def dense_forward(cam: Camera, tables, pre: dict, means, scales, rots, opac):
    """means (N,3), scales (N,2), rots (N,4), opac (N,1): float64 tensors
    (requires_grad as desired).  pre: output of Oracle(float64).preprocess.
    Returns allmap (7,H,W)."""
    dt = torch.float64
    H, W = cam.H, cam.W
    TW, TH = cam.tile
    fc = torch.tensor(cam.fcam, dtype=dt)
    fx, fy, cx, cy, mod = fc[0], fc[1], fc[2], fc[3], fc[4]
    Rvw = fc[7:16].reshape(3, 3)
    tvw = fc[16:19]
    col = torch.tensor(tables[0], dtype=dt)
    row = torch.tensor(tables[1], dtype=dt)
    # pixel rays (H,W,3)
    d = torch.stack([col[None, :, 0] * row[:, None, 0], col[None, :, 1] * row[:, None, 0],
                     row[:, None, 1].expand(H, W)], dim=-1)
    pc = torch.arange(W, dtype=dt)[None, :].expand(H, W)
    pr = torch.arange(H, dtype=dt)[:, None].expand(H, W)
    tile_x = (torch.arange(W) // TW)[None, :].expand(H, W)
    tile_y = (torch.arange(H) // TH)[:, None].expand(H, W)

    p = means @ Rvw.T + tvw
    rho_c = p.norm(dim=1)
    rxy = p[:, :2].norm(dim=1)
    az = torch.atan2(p[:, 1], p[:, 0])
    el = torch.atan2(p[:, 2], rxy)
    cpx = fx * az + cx
    cpy = fy * el + cy
    # use the checker's polynomial-atan2 values, keep the analytic gradient
    rec = torch.tensor(pre["rec"], dtype=dt)
    cpx = cpx + (rec[:, 16] - cpx).detach()
    cpy = cpy + (rec[:, 17] - cpy).detach()
    Rq = build_rotation(rots)
    Tu = Rq[:, :, 0] @ Rvw.T
    Tv = Rq[:, :, 1] @ Rvw.T
    Tn = Rq[:, :, 2] @ Rvw.T
    sig = torch.where((Tn * p).sum(1) > 0, -1.0, 1.0).to(dt)
    n = sig[:, None] * Tn
    su, sv = scales[:, 0] * mod, scales[:, 1] * mod

    radii = torch.tensor(pre["radii"])
    rect = torch.tensor(pre["rect"])
    depth_key = torch.tensor(np.asarray(pre["depth"], dtype=np.float32).astype(np.float64))
    order = sorted([i for i in range(means.shape[0]) if radii[i] > 0],
                   key=lambda i: (float(depth_key[i]), i))

    T = torch.ones(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    D = torch.zeros(H, W, dtype=dt)
    Nn = torch.zeros(H, W, 3, dtype=dt)
    M1 = torch.zeros(H, W, dtype=dt)
    M2 = torch.zeros(H, W, dtype=dt)
    dist = torch.zeros(H, W, dtype=dt)
    med = torch.zeros(H, W, dtype=dt)
    mscale = FAR / (FAR - NEAR)
    GX = cam.GX
    for i in order:
        txlo, ncols, tylo, nrows = [int(v) for v in rect[i]]
        in_x = ((tile_x - txlo) % GX) < ncols
        in_y = (tile_y >= tylo) & (tile_y < tylo + nrows)
        member = in_x & in_y & ~done
        if not bool(member.any()):
            continue
        nd = (d * n[i]).sum(-1)
        valid3d = nd < 0
        nd_safe = torch.where(valid3d, nd, torch.full_like(nd, -1.0))
        t = (n[i] * p[i]).sum() / nd_safe
        x = t[..., None] * d - p[i]
        u = (x * Tu[i]).sum(-1) / su[i]
        v = (x * Tv[i]).sum(-1) / sv[i]
        rho3 = u * u + v * v
        dx = pc - cpx[i]
        if cam.wrap:
            dx = torch.where(dx > 0.5 * W, dx - W, torch.where(dx < -0.5 * W, dx + W, dx))
        dy = pr - cpy[i]
        rho2 = 2.0 * (dx * dx + dy * dy)
        use3d = valid3d & (rho3 <= rho2)
        rho = torch.where(use3d, rho3, rho2)
        depth = torch.where(use3d, t, rho_c[i].expand(H, W))
        G = torch.exp(-0.5 * rho)
        alpha = torch.clamp(opac[i, 0] * G, max=ALPHA_MAX)
        active = member & (depth >= NEAR) & (alpha >= ALPHA_MIN)
        testT = T * (1 - alpha)
        newly_done = active & (testT < T_MIN)
        done = done | newly_done
        blend = active & ~newly_done
        w = torch.where(blend, alpha * T, torch.zeros_like(T))
        A = 1 - T
        depth_s = torch.where(blend, depth, torch.ones_like(depth))
        m = mscale * (1 - NEAR / depth_s)
        dist = dist + (m * m * A + M2 - 2 * m * M1) * w
        D = D + depth_s * w
        M1 = M1 + m * w
        M2 = M2 + m * m * w
        med = torch.where(blend & (T > 0.5), depth, med)
        Nn = Nn + w[..., None] * n[i]
        T = torch.where(blend, testT, T)
    return torch.cat([D[None], (1 - T)[None], Nn.permute(2, 0, 1), med[None], dist[None]], dim=0)
