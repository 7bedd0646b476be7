"""Pure-PyTorch, tile-batched CPU rasterizer with autograd — the CPU baseline BASELINE.json names ("a pure-PyTorch
CPU rasterizer timed on the host cores").  TEST / BENCH INFRASTRUCTURE (see sls_oracle.c's header): imported by
tests/ and by bench.py's `cpu_baseline` leg only, never by the product.

Same function as the checker and the HIP kernels: preprocess (projection, surfel frame, the
cancellation-free Hu/Hv record), binning (3-sigma extent -> tile rectangle -> instances sorted by (tile, depth
bits, index)), per-tile front-to-back blend.  Everything is vectorised torch: the per-tile blend evaluates the
(entries x 256 pixels) alpha matrix at once, transmittance by `cumprod`, termination by the first index whose
inclusive product drops below 1e-4; gradients come from autograd.  Lists are cut at the deepest entry any pixel of
the tile consumes (found in a no-grad pre-pass over 256-entry slabs), which is what makes 500k surfels feasible.

The integer decisions (tile rectangles) use torch's own atan2/asin, so a rectangle may differ from the checker's
by a tile in rare rounding cases; the rectangles are conservative, so the image does not change
(tests/test_oracle.py compares the image with the checker's).
"""
from __future__ import annotations

import math

import numpy as np
import torch

NEAR, FAR = float(np.float32(0.2)), 100.0
ALPHA_MAX = float(np.float32(0.99))
ALPHA_MIN = float(np.float32(1.0) / np.float32(255.0))
T_MIN = float(np.float32(1.0e-4))
CUTOFF, RMIN_PX, TILE = 3.0, 2.1213203435596424, 16


def camera_dict(H, W, viewmatrix, projmatrix, scale_modifier=1.0):
    V = np.asarray(viewmatrix, np.float64)
    K = np.asarray(projmatrix, np.float64)[:3, :3].T
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    wrap = abs(abs(fx) * 2.0 * math.pi - W) <= 1.0 and W % TILE == 0
    return dict(H=int(H), W=int(W), fx=fx, fy=fy, cx=cx, cy=cy, mod=float(scale_modifier), wrap=bool(wrap),
                Rvw=torch.tensor(V[:3, :3].T, dtype=torch.float32), tvw=torch.tensor(V[3, :3], dtype=torch.float32))


def _rotation_columns(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    tu = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)], 1)
    tv = torch.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)], 1)
    tn = torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], 1)
    return tu, tv, tn


def preprocess(cam, means, scales, rots, opac):
    """Per-surfel record (autograd-tracked) + the tile rectangle and sort key (no grad)."""
    p = means @ cam["Rvw"].T + cam["tvw"]
    rho = p.norm(dim=1)
    rxy = p[:, :2].norm(dim=1)
    tu, tv, tn = _rotation_columns(rots)
    Tu, Tv, Tn = tu @ cam["Rvw"].T, tv @ cam["Rvw"].T, tn @ cam["Rvw"].T
    c = (Tn * p).sum(1)
    sig = torch.where(c > 0, -torch.ones_like(c), torch.ones_like(c))
    n = sig[:, None] * Tn
    su, sv = scales[:, 0] * cam["mod"], scales[:, 1] * cam["mod"]
    A = sig[:, None] * Tv / su[:, None]
    B = -sig[:, None] * Tu / sv[:, None]
    rec = dict(Hu=torch.linalg.cross(A, p), Hv=torch.linalg.cross(B, p), n=n, npv=sig * c, rhoc=rho,
               dc=p / rho[:, None], o=opac.reshape(-1),
               cpx=cam["fx"] * torch.atan2(p[:, 1], p[:, 0]) + cam["cx"],
               cpy=cam["fy"] * torch.atan2(p[:, 2], rxy) + cam["cy"])
    with torch.no_grad():
        H, W = cam["H"], cam["W"]
        GX, GY = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        rad = CUTOFF * torch.maximum(su, sv)
        inside = rad < rho
        theta = torch.where(inside, torch.asin(torch.clamp(rad / rho, max=1.0)), torch.full_like(rho, math.pi))
        daz = torch.where(inside & (rad < rxy), torch.asin(torch.clamp(rad / rxy.clamp_min(1e-30), max=1.0)),
                          torch.full_like(rho, math.pi))
        rx = torch.clamp(abs(cam["fx"]) * daz, min=RMIN_PX)
        ry = torch.clamp(abs(cam["fy"]) * theta, min=RMIN_PX)
        lim = 1.0e9
        xlo = torch.floor(rec["cpx"] - rx + 0.5).clamp(-lim, lim).long()
        xhi = torch.floor(rec["cpx"] + rx + 0.5).clamp(-lim, lim).long()
        ylo = torch.floor(rec["cpy"] - ry + 0.5).clamp(-lim, lim).long().clamp_min(0)
        yhi = torch.floor(rec["cpy"] + ry + 0.5).clamp(-lim, lim).long().clamp_max(H - 1)
        vis = (rho >= NEAR) & (rho < 1e18) & (ylo <= yhi)
        if cam["wrap"]:
            full = (xhi - xlo + 1) >= W
            a = torch.div(xlo, TILE, rounding_mode="floor")
            b = torch.div(xhi, TILE, rounding_mode="floor")
            ncols = torch.where(full, torch.full_like(a, GX), torch.clamp(b - a + 1, max=GX))
            txlo = torch.where(full, torch.zeros_like(a), torch.remainder(a, GX))
        else:
            xlo, xhi = xlo.clamp_min(0), xhi.clamp_max(W - 1)
            vis = vis & (xlo <= xhi)
            txlo = torch.div(xlo, TILE, rounding_mode="floor")
            ncols = torch.div(xhi, TILE, rounding_mode="floor") - txlo + 1
        tylo = torch.div(ylo, TILE, rounding_mode="floor")
        nrows = torch.div(yhi, TILE, rounding_mode="floor") - tylo + 1
        ncols, nrows = torch.where(vis, ncols, torch.zeros_like(ncols)), torch.where(vis, nrows, torch.zeros_like(nrows))
        radii = torch.where(vis, torch.ceil(torch.maximum(rx, ry)).long(), torch.zeros_like(xlo)).to(torch.int32)
    return rec, dict(txlo=txlo, ncols=ncols, tylo=tylo, nrows=nrows, radii=radii, GX=GX, GY=GY, depth=rho.detach())


def bin_sort(b):
    """Instances sorted by (tile, depth bits, surfel index) -> (vals, ranges)."""
    cnt = b["ncols"] * b["nrows"]
    N = cnt.numel()
    idx = torch.repeat_interleave(torch.arange(N), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(idx.numel()) - start[idx]
    nc = b["ncols"][idx].clamp_min(1)
    ty = b["tylo"][idx] + torch.div(k, nc, rounding_mode="floor")
    tx = torch.remainder(b["txlo"][idx] + torch.remainder(k, nc), b["GX"])
    tile = ty * b["GX"] + tx
    bits = b["depth"].to(torch.float32).view(torch.int32).long()           # positive floats order like their bits
    key = (tile << 32) | bits[idx]
    order = torch.sort(key, stable=True).indices                            # ties keep the surfel-index order
    vals, tile = idx[order], tile[order]
    T = b["GX"] * b["GY"]
    counts = torch.bincount(tile, minlength=T)
    ends = torch.cumsum(counts, 0)
    return vals, torch.stack([ends - counts, ends], 1)


def _tile_alpha(q, d, pc, pr, wrapW):
    """(n, P) alpha (0 where skipped), depth, use3 for the entries `q` (dict of (n, ...)) and the tile's pixels."""
    dl = d[None] - q["dc"][:, None]
    nd = (q["n"][:, None] * d[None]).sum(-1)
    valid3 = nd < 0
    rinv = 1.0 / torch.where(valid3, nd, -torch.ones_like(nd))
    u = (q["Hu"][:, None] * dl).sum(-1) * rinv
    v = (q["Hv"][:, None] * dl).sum(-1) * rinv
    t = q["npv"][:, None] * rinv
    rho3 = u * u + v * v
    dx = pc[None] - q["cpx"][:, None]
    if wrapW:
        dx = dx - wrapW * torch.round(dx / wrapW)
    dy = pr[None] - q["cpy"][:, None]
    rho2 = 2.0 * (dx * dx + dy * dy)
    use3 = valid3 & (rho3 <= rho2)
    rho = torch.where(use3, rho3, rho2)
    depth = torch.where(use3, t, q["rhoc"][:, None].expand_as(t))
    alpha = torch.clamp(q["o"][:, None] * torch.exp(-0.5 * rho), max=ALPHA_MAX)
    live = (depth >= NEAR) & (alpha >= ALPHA_MIN)
    return torch.where(live, alpha, torch.zeros_like(alpha)), depth


def _blend(alpha, depth, nrm):
    """Front-to-back blend of an (n, P) alpha matrix -> the 7 channels (P,) each + per-pixel consumed count."""
    n = alpha.shape[0]
    Tinc = torch.cumprod(1.0 - alpha, dim=0)
    Texc = torch.cat([torch.ones_like(Tinc[:1]), Tinc[:-1]], 0)
    with torch.no_grad():
        term = (alpha > 0) & (Tinc < T_MIN)
        first = torch.where(term.any(0), term.float().argmax(0), torch.full((alpha.shape[1],), n))
        upd = (alpha > 0) & (torch.arange(n)[:, None] < first[None])
    w = torch.where(upd, alpha * Texc, torch.zeros_like(alpha))
    dep = torch.where(upd, depth, torch.ones_like(depth))
    m = (FAR / (FAR - NEAR)) * (1.0 - NEAR / dep)
    A, D = w.sum(0), (w * dep).sum(0)
    Nn = w.t() @ nrm
    M1, M2 = (w * m).sum(0), (w * m * m).sum(0)
    with torch.no_grad():
        med_ok = upd & (Texc > 0.5)
        last = (med_ok * torch.arange(1, n + 1)[:, None]).max(0).values
    med = torch.where(last > 0, depth.gather(0, (last - 1).clamp_min(0)[None])[0], torch.zeros_like(A))
    return D, A, Nn, med, A * M2 - M1 * M1, first


def rasterize(cam, means, scales, rots, opac, tiles=None, stats=None):
    """allmap (7,H,W) float32 with autograd.  `tiles`: render only these tile ids (the others stay zero) — used by
    bench.py to time a stated subset of the 500k-surfel scene."""
    H, W = cam["H"], cam["W"]
    rec, b = preprocess(cam, means, scales, rots, opac)
    with torch.no_grad():
        vals, ranges = bin_sort(b)
    col = (torch.arange(W, dtype=torch.float64) - cam["cx"]) / cam["fx"]
    row = (torch.arange(H, dtype=torch.float64) - cam["cy"]) / cam["fy"]
    ccol, scol, crow, srow = (torch.cos(col).float(), torch.sin(col).float(), torch.cos(row).float(), torch.sin(row).float())
    wrapW = float(W) if cam["wrap"] else 0.0
    allmap = torch.zeros((7, H, W), dtype=torch.float32)
    planes = [[] for _ in range(7)]
    where = []
    consumed = 0
    for t in (range(b["GX"] * b["GY"]) if tiles is None else tiles):
        a, e = int(ranges[t, 0]), int(ranges[t, 1])
        if e <= a:
            continue
        ty, tx = divmod(int(t), b["GX"])
        ys = torch.arange(ty * TILE, min(ty * TILE + TILE, H))
        xs = torch.arange(tx * TILE, min(tx * TILE + TILE, W))
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        py, px = py.reshape(-1), px.reshape(-1)
        d = torch.stack([ccol[px] * crow[py], scol[px] * crow[py], srow[py]], 1)
        pc, pr = px.float(), py.float()
        # no-grad pre-pass over slabs of 256 entries: how deep does any pixel of this tile go?
        with torch.no_grad():
            Trun = torch.ones(px.numel())
            done = torch.zeros(px.numel(), dtype=torch.bool)
            need = e - a
            for s0 in range(a, e, 256):
                ids = vals[s0:min(s0 + 256, e)]
                al, _ = _tile_alpha({k: v[ids] for k, v in rec.items()}, d, pc, pr, wrapW)
                Tinc = Trun[None] * torch.cumprod(1.0 - al, dim=0)
                done = done | ((al > 0) & (Tinc < T_MIN)).any(0)
                Trun = Tinc[-1]
                if bool(done.all()):
                    need = min(s0 + 256, e) - a
                    break
        ids = vals[a:a + need]
        q = {k: v[ids] for k, v in rec.items()}
        alpha, depth = _tile_alpha(q, d, pc, pr, wrapW)
        D, A, Nn, med, dist, first = _blend(alpha, depth, q["n"])
        consumed += int(first.max())
        flat = py * W + px
        where.append(flat)
        for c, v in enumerate((D, A, Nn[:, 0], Nn[:, 1], Nn[:, 2], med, dist)):
            planes[c].append(v)
    if where:
        flat = torch.cat(where)
        allmap = torch.stack([torch.zeros(H * W).index_put((flat,), torch.cat(planes[c])) for c in range(7)]).view(7, H, W)
    if stats is not None:
        stats.update(R=int(vals.numel()), R_eff=consumed)
    return b["radii"], allmap
