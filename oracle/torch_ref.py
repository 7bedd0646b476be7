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


def dense_forward_tiled(cam: Camera, tables, pre: dict, means, scales, rots, opac):
    """dense_forward() with the tiles as a batch: the same float64 autograd formulation, but every surfel is evaluated
    only on the 16x16 tiles of its rectangle (the pixels outside never see it in dense_forward() either: `member`) and
    step k handles the k-th list entry of EVERY tile at once — O(longest list) torch steps on (tiles, 256) tensors
    instead of O(N) steps on (H, W) ones: 2 000 surfels on 64x512 in seconds and ~2 GB of autograd state instead of 20.
    Same arguments, same result as dense_forward (tests/test_oracle.py holds the two against each other)."""
    dt = torch.float64
    H, W = cam.H, cam.W
    TW, TH = cam.tile
    GX, GY = cam.GX, (H + TH - 1) // TH
    NT, PX = GX * GY, TW * TH
    fc = torch.tensor(cam.fcam, dtype=dt)
    fx, fy, cx, cy, mod = fc[0], fc[1], fc[2], fc[3], fc[4]
    Rvw = fc[7:16].reshape(3, 3)
    tvw = fc[16:19]
    col = torch.tensor(tables[0], dtype=dt)
    row = torch.tensor(tables[1], dtype=dt)
    p = means @ Rvw.T + tvw
    rho_c = p.norm(dim=1)
    az = torch.atan2(p[:, 1], p[:, 0])
    el = torch.atan2(p[:, 2], p[:, :2].norm(dim=1))
    rec = torch.tensor(pre["rec"], dtype=dt)
    cpx = fx * az + cx
    cpy = fy * el + cy
    cpx = cpx + (rec[:, 16] - cpx).detach()
    cpy = cpy + (rec[:, 17] - cpy).detach()
    Rq = build_rotation(rots)
    Tu = Rq[:, :, 0] @ Rvw.T
    Tv = Rq[:, :, 1] @ Rvw.T
    Tn = Rq[:, :, 2] @ Rvw.T
    sig = torch.where((Tn * p).sum(1) > 0, -1.0, 1.0).to(dt)
    n = sig[:, None] * Tn
    npv = (n * p).sum(1)
    su, sv = scales[:, 0] * mod, scales[:, 1] * mod
    op = opac[:, 0]
    radii = np.asarray(pre["radii"])
    rect = np.asarray(pre["rect"])
    depth_key = np.asarray(pre["depth"], dtype=np.float32).astype(np.float64)
    lists = [[] for _ in range(NT)]
    for i in sorted((i for i in range(means.shape[0]) if radii[i] > 0), key=lambda i: (float(depth_key[i]), i)):
        txlo, ncols, tylo, nrows = (int(v) for v in rect[i])
        for ty in range(tylo, tylo + nrows):
            for k in range(ncols):
                lists[ty * GX + (txlo + k) % GX].append(i)
    kmax = max((len(l) for l in lists), default=0)
    L = torch.full((NT, max(kmax, 1)), -1, dtype=torch.long)
    for t, l in enumerate(lists):
        if l:
            L[t, :len(l)] = torch.tensor(l, dtype=torch.long)
    # pixel grids per tile: (NT, PX); pixels beyond a ragged image edge are finished from the start
    ty = torch.arange(NT) // GX
    tx = torch.arange(NT) % GX
    rr = (ty[:, None] * TH + (torch.arange(PX) // TW)[None, :])
    cc = (tx[:, None] * TW + (torch.arange(PX) % TW)[None, :])
    inside = (rr < H) & (cc < W)
    rr_c, cc_c = rr.clamp(max=H - 1), cc.clamp(max=W - 1)
    d = torch.stack([col[cc_c, 0] * row[rr_c, 0], col[cc_c, 1] * row[rr_c, 0], row[rr_c, 1]], dim=-1)      # (NT, PX, 3)
    pc, pr = cc.to(dt), rr.to(dt)
    T = torch.ones(NT, PX, dtype=dt)
    done = ~inside
    D = torch.zeros(NT, PX, dtype=dt)
    Nn = torch.zeros(NT, PX, 3, dtype=dt)
    M1 = torch.zeros(NT, PX, dtype=dt)
    M2 = torch.zeros(NT, PX, dtype=dt)
    dist = torch.zeros(NT, PX, dtype=dt)
    med = torch.zeros(NT, PX, dtype=dt)
    mscale = FAR / (FAR - NEAR)
    for k in range(kmax):
        idx = L[:, k]
        has = (idx >= 0)[:, None]
        i = idx.clamp(min=0)
        if not bool((has & ~done).any()):
            continue
        ni, pi = n[i], p[i]                                    # (NT, 3)
        nd = (d * ni[:, None, :]).sum(-1)
        valid3d = nd < 0
        t = npv[i][:, None] / torch.where(valid3d, nd, torch.full_like(nd, -1.0))
        x = t[..., None] * d - pi[:, None, :]
        u = (x * Tu[i][:, None, :]).sum(-1) / su[i][:, None]
        v = (x * Tv[i][:, None, :]).sum(-1) / sv[i][:, None]
        rho3 = u * u + v * v
        dx = pc - cpx[i][:, None]
        if cam.wrap:
            dx = torch.where(dx > 0.5 * W, dx - W, torch.where(dx < -0.5 * W, dx + W, dx))
        dy = pr - cpy[i][:, None]
        rho2 = 2.0 * (dx * dx + dy * dy)
        use3d = valid3d & (rho3 <= rho2)
        rho = torch.where(use3d, rho3, rho2)
        depth = torch.where(use3d, t, rho_c[i][:, None].expand(NT, PX))
        alpha = torch.clamp(op[i][:, None] * torch.exp(-0.5 * rho), max=ALPHA_MAX)
        active = has & ~done & (depth >= NEAR) & (alpha >= ALPHA_MIN)
        testT = T * (1 - alpha)
        newly_done = active & (testT < T_MIN)
        done = done | newly_done
        blend = active & ~newly_done
        wgt = torch.where(blend, alpha * T, torch.zeros_like(T))
        depth_s = torch.where(blend, depth, torch.ones_like(depth))
        m = mscale * (1 - NEAR / depth_s)
        dist = dist + (m * m * (1 - T) + M2 - 2 * m * M1) * wgt
        D = D + depth_s * wgt
        M1 = M1 + m * wgt
        M2 = M2 + m * m * wgt
        med = torch.where(blend & (T > 0.5), depth, med)
        Nn = Nn + wgt[..., None] * ni[:, None, :]
        T = torch.where(blend, testT, T)
    planes = torch.cat([D[:, None], (1 - T)[:, None], Nn.permute(0, 2, 1), med[:, None], dist[:, None]], dim=1)   # (NT, 7, PX)
    img = planes.reshape(GY, GX, 7, TH, TW).permute(2, 0, 3, 1, 4).reshape(7, GY * TH, GX * TW)
    return img[:, :H, :W]
