#!/usr/bin/env python3
"""Offline model of the tile kernels' lane utilisation (CPU only, NumPy): for a sample of tiles of the bench scene,
evaluate every (pixel, list entry) pair, find where every pixel terminates, and count — for several candidate
wave layouts — the steps a wave would issue and the lanes that do useful work.  Decides which restructuring of
sls_render_block.hip is worth building before any GPU time is spent.

    python tools/sim_lane_use.py [--n 500000] [--tiles 48]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ALPHA_MIN, ALPHA_MAX, T_MIN = 1.0 / 255.0, 0.99, 1e-4


def tile_pairs(rec, idx, x0, y0, col, row, W, wrap, near):
    """alpha[entry, pixel] (0 where skipped), box-pass[entry, pixel] for one 16x16 tile."""
    q = rec[idx].astype(np.float64)                               # (n, 20)
    px = x0 + np.arange(16)[None, :].repeat(16, 0).reshape(-1)   # (256,)
    py = y0 + np.arange(16)[:, None].repeat(16, 1).reshape(-1)
    d = np.stack([col[px, 0] * row[py, 0], col[px, 1] * row[py, 0], row[py, 1]], 1).astype(np.float64)   # (256,3)
    dl = d[None] - q[:, None, 12:15]
    nd = (q[:, None, 8:11] * d[None]).sum(-1)
    rinv = 1.0 / np.where(nd == 0, 1e-30, nd)
    u = (q[:, None, 0:3] * dl).sum(-1) * rinv
    v = (q[:, None, 4:7] * dl).sum(-1) * rinv
    t = q[:, None, 3] * rinv
    rho3 = u * u + v * v
    dx = px[None].astype(np.float64) - q[:, None, 16]
    if wrap:
        dx = dx - W * np.rint(dx / W)
    dy = py[None].astype(np.float64) - q[:, None, 17]
    rho2 = 2.0 * (dx * dx + dy * dy)
    use3 = (nd < 0) & (rho3 <= rho2)
    rho = np.where(use3, rho3, rho2)
    depth = np.where(use3, t, q[:, None, 7])
    alpha = np.minimum(ALPHA_MAX, q[:, None, 11] * np.exp(-0.5 * rho))
    alpha = np.where((depth < near) | (alpha < ALPHA_MIN), 0.0, alpha)
    box = (np.abs(dx) <= q[:, None, 18]) & (np.abs(dy) <= q[:, None, 19])
    return alpha, box, dx, dy, q[:, 18], q[:, 19]


def support_extents(sc, fx, fy):
    """(ex, ey) of sls_preprocess.hip:212-241 for the identity pose, float64."""
    p = sc["means"].astype(np.float64)
    s = sc["scales"].astype(np.float64)
    q = sc["rots"].astype(np.float64)
    o = sc["opac"].astype(np.float64).reshape(-1)
    r, x, y, z = q.T
    Tu = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)], 1)
    Tv = np.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)], 1)
    rxy2 = p[:, 0] ** 2 + p[:, 1] ** 2
    rho2 = rxy2 + p[:, 2] ** 2
    rho, rxy = np.sqrt(rho2), np.sqrt(rxy2)
    smax = s.max(1)
    lo = 255.0 * o
    rho_max = 2.0 * np.log(np.maximum(lo, 1.0 + 1e-9)) * 1.001 + 1e-3
    kk = np.sqrt(rho_max)
    rad = kk * smax
    th = np.where(rad < rho, np.arcsin(np.minimum(rad / rho, 1.0)), np.pi)
    daz = np.where((rad < rho) & (rad < rxy), np.arcsin(np.minimum(rad / rxy, 1.0)), np.pi)
    r2 = np.sqrt(0.5 * rho_max)
    ok = (rxy > 2 * rad) & (rxy > 0.5 * rho)
    au = (p[:, 0] * Tu[:, 1] - p[:, 1] * Tu[:, 0]) / rxy2
    av = (p[:, 0] * Tv[:, 1] - p[:, 1] * Tv[:, 0]) / rxy2
    rat_xy = rad / np.maximum(rxy - rad, 1e-9)
    az_ell = kk * np.sqrt((s[:, 0] * au) ** 2 + (s[:, 1] * av) ** 2) + 0.75 * rat_xy ** 2
    zr = p[:, 2] / rxy / rho2
    eu = -zr * (p[:, 0] * Tu[:, 0] + p[:, 1] * Tu[:, 1]) + rxy / rho2 * Tu[:, 2]
    ev = -zr * (p[:, 0] * Tv[:, 0] + p[:, 1] * Tv[:, 1]) + rxy / rho2 * Tv[:, 2]
    rat = rad / np.maximum(rho - rad, 1e-9)
    el_ell = kk * np.sqrt((s[:, 0] * eu) ** 2 + (s[:, 1] * ev) ** 2) + 1.5 * rat ** 2
    daz = np.where(ok, np.minimum(daz, az_ell * 1.02), daz)
    th = np.where(ok, np.minimum(th, el_ell * 1.02), th)
    ex = np.maximum(abs(fx) * daz, r2) * 1.001 + 0.05
    ey = np.maximum(abs(fy) * th, r2) * 1.001 + 0.05
    ex[lo <= 1.0] = -1e30
    ey[lo <= 1.0] = -1e30
    return ex, ey


def terminate(alpha):
    """per pixel: number of entries processed (index of the terminating entry, or n)."""
    n = alpha.shape[0]
    Tinc = np.cumprod(1.0 - alpha, axis=0)
    term = (Tinc < T_MIN) & (alpha > 0)
    first = np.where(term.any(0), term.argmax(0), n)
    return first                                                   # entries [0, first) are blended


def count_scheme(alpha, box, dxdy, exy, need, groups, slots, share_staging_px=16):
    """groups: list of pixel-index arrays that own a survivor list each and advance together inside ONE wave
    (a wave = all `groups` passed in one call are the sub-groups of one wave; lanes = sum(len(g)) * slots / ...).
    Returns (steps, useful_lanes, evaluated_lanes, staged_rounds)."""
    n = alpha.shape[0]
    dx, dy = dxdy
    ex, ey = exy
    wave_px = np.concatenate(groups)
    wave_need = need[wave_px].max()
    steps = useful = evaluated = rounds = 0
    for r0 in range(0, wave_need, 64):
        r1 = min(r0 + 64, n)
        rounds += 1
        gsteps = 0
        for g in groups:
            act = g[need[g] > r0]                                  # pixels of the group still running
            if act.size == 0:
                continue
            # support box of the entry vs the bounding box of the group's active pixels (what cull_pass does)
            ddx, ddy = dx[r0:r1][:, act], dy[r0:r1][:, act]
            # exact box-vs-box overlap in x and y
            px_lo, px_hi = ddx.min(1), ddx.max(1)
            py_lo, py_hi = ddy.min(1), ddy.max(1)
            passes = (px_lo <= ex[r0:r1]) & (px_hi >= -ex[r0:r1]) & (py_lo <= ey[r0:r1]) & (py_hi >= -ey[r0:r1])
            npass = int(passes.sum())
            gsteps = max(gsteps, -(-npass // slots))
            a = alpha[r0:r1][passes][:, g]
            live = (np.arange(r0, r1)[passes][:, None] < need[g][None, :]) & (a > 0)
            useful += int(live.sum())
        steps += gsteps
    evaluated = steps * 64
    return steps, useful, evaluated, rounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500_000)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--tiles", type=int, default=48)
    args = ap.parse_args()
    from oracle.oracle import Oracle
    from splat_loam_amd import synth
    N, H, W = args.n, args.height, args.width
    sc = synth.make_scene(N, H, W, seed=0)
    view, proj = synth.camera_matrices(sc["K"])
    o = Oracle(np.float32)
    o.set_threads(o.max_threads())
    cam = o.camera(H, W, view, proj)
    col, row = o.ray_tables(cam)
    pre = o.preprocess(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    binned = o.bin_sort(cam, pre)
    rec, ranges, vals = pre["rec"].copy(), binned["ranges"], binned["vals"]
    rec[:, 18], rec[:, 19] = support_extents(sc, cam.fx, cam.fy)        # (the checker does not fill the cull extents)
    rng = np.random.default_rng(0)
    tiles = rng.choice(cam.T, size=min(args.tiles, cam.T), replace=False)

    pix = np.arange(256).reshape(16, 16)
    def blocks(bw, bh):
        return [pix[y:y + bh, x:x + bw].reshape(-1) for y in range(0, 16, bh) for x in range(0, 16, bw)]
    schemes = {
        "8x2 x4 slots (current)": [([b], 4) for b in blocks(8, 2)],
        "4x4 x4 slots": [([b], 4) for b in blocks(4, 4)],
        "8x2 wave = 2 halves 4x2, own lists": [([b[[0, 1, 2, 3, 8, 9, 10, 11]], b[[4, 5, 6, 7, 12, 13, 14, 15]]], 4) for b in blocks(8, 2)],
        "8x2 wave = 4 quarters 2x2, own lists": [([b[[0, 1, 8, 9]], b[[2, 3, 10, 11]], b[[4, 5, 12, 13]], b[[6, 7, 14, 15]]], 4) for b in blocks(8, 2)],
        "8x2 wave = 4 quarters 4x1, own lists": [([b[0:4], b[4:8], b[8:12], b[12:16]], 4) for b in blocks(8, 2)],
        "4x4 wave = 4 quarters 2x2, own lists": [([b[[0, 1, 4, 5]], b[[2, 3, 6, 7]], b[[8, 9, 12, 13]], b[[10, 11, 14, 15]]], 4) for b in blocks(4, 4)],
        "4x2 x8 slots": [([b], 8) for b in blocks(4, 2)],
        "2x2 x16 slots": [([b], 16) for b in blocks(2, 2)],
    }
    tot = {k: np.zeros(4, np.int64) for k in schemes}
    pairs_useful = 0
    for t in tiles:
        a, b = ranges[t]
        if b <= a:
            continue
        ty, tx = divmod(int(t), cam.GX)
        alpha, box, dx, dy, ex, ey = tile_pairs(rec, vals[a:b], tx * 16, ty * 16, col, row, W, cam.wrap, 0.2)
        need = terminate(alpha)
        pairs_useful += int(((np.arange(alpha.shape[0])[:, None] < need[None]) & (alpha > 0)).sum())
        for name, waves in schemes.items():
            for groups, slots in waves:
                tot[name] += np.array(count_scheme(alpha, box, (dx, dy), (ex, ey), need, groups, slots))
    print(f"{len(tiles)} tiles, useful (pixel, entry) pairs: {pairs_useful}")
    base = tot["8x2 x4 slots (current)"][0]
    for name, v in tot.items():
        steps, useful, evaluated, rounds = v
        print(f"{name:42s} steps {steps:8d} ({steps / base:5.2f}x)  lane use {useful / max(evaluated, 1):5.1%}  staged rounds {rounds}")


if __name__ == "__main__":
    main()
