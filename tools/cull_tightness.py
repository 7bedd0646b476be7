#!/usr/bin/env python3
"""Diagnostic: how tight are the support half-extents (ex, ey) that preprocess stores for the
wave-level cull, compared with the true footprint (pixels with alpha >= 1/255) of a surfel?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from splat_loam_amd import synth
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_forward
N, H, W = 500000, 64, 2048
dev = torch.device("cuda:0")
sc = synth.make_scene(N, H, W, seed=0)
view, proj = synth.camera_matrices(sc["K"])
s = GaussianRasterizationSettings(H, W, 1.0, torch.tensor(view, device=dev), torch.tensor(proj, device=dev))
t = {k: torch.tensor(sc[k], device=dev) for k in ("means", "scales", "rots", "opac")}
st = rasterize_forward(s, t["means"], t["opac"], t["scales"], t["rots"])
rec = st.rec.cpu().numpy().reshape(N, 20).astype(np.float64)
radii = st.radii.cpu().numpy()
from splat_loam_amd.rasterizer import get_camera
ce = get_camera(s, dev)
col = ce.col_cs.cpu().numpy().astype(np.float64).reshape(W, 2); row = ce.row_cs.cpu().numpy().astype(np.float64).reshape(H, 2)
vis = np.nonzero(radii > 0)[0]
sel = np.random.default_rng(0).choice(vis, 3000, replace=False)
out = []
for g in sel:
    r = rec[g]
    Hu, npv, Hv, rhoc, n, op, dc, cpx, cpy, ex, ey = r[0:3], r[3], r[4:7], r[7], r[8:11], r[11], r[12:15], r[16], r[17], r[18], r[19]
    if ex < 0:
        continue
    x0 = int(np.floor(cpx)) - 64; xs = np.arange(x0, x0 + 129); ys = np.arange(0, H)
    X, Y = np.meshgrid(xs, ys); Xw = np.mod(X, W)
    d = np.stack([col[Xw, 0] * row[Y, 0], col[Xw, 1] * row[Y, 0], row[Y, 1]], -1)
    nd = d @ n
    with np.errstate(all="ignore"):
        u = ((d - dc) @ Hu) / nd; v = ((d - dc) @ Hv) / nd; tt = npv / nd
    rho3 = u * u + v * v
    rho2 = 2.0 * ((X - cpx) ** 2 + (Y - cpy) ** 2)
    use3 = (nd < 0) & (rho3 <= rho2)
    rho = np.where(use3, rho3, rho2)
    dep = np.where(use3, tt, rhoc)
    hit = (np.minimum(0.99, op * np.exp(-0.5 * rho)) >= 1 / 255) & (dep >= 0.2)
    if not hit.any():
        out.append((ex, ey, 0, 0, 0)); continue
    yy, xx = np.nonzero(hit)
    tx = np.abs(xs[xx] - cpx).max(); ty = np.abs(yy - cpy).max()
    out.append((ex, ey, tx, ty, hit.sum()))
o = np.array(out)
print("samples", len(o), "no-hit", int((o[:, 4] == 0).sum()))
print("stored ex median %.2f mean %.2f | true x half-extent median %.2f mean %.2f" % (np.median(o[:, 0]), o[:, 0].mean(), np.median(o[:, 2]), o[:, 2].mean()))
print("stored ey median %.2f mean %.2f | true y half-extent median %.2f mean %.2f" % (np.median(o[:, 1]), o[:, 1].mean(), np.median(o[:, 3]), o[:, 3].mean()))
print("box area ratio stored/true (mean of (2ex+1)(2ey+1) / (2tx+1)(2ty+1)): %.2f" % (((2 * o[:, 0] + 1) * (2 * o[:, 1] + 1)).sum() / ((2 * o[:, 2] + 1) * (2 * o[:, 3] + 1)).sum()))
assert np.all(o[:, 0] + 1e-3 >= o[:, 2]) and np.all(o[:, 1] + 1e-3 >= o[:, 3]), "support extents must be conservative"
