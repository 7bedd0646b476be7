"""The forward tile kernel's block-level footprint test (csrc/sls_tile.hpp: make_block_cone / cone_outside /
disc_reaches) restated in NumPy float32 and checked for what makes it legal: it may only drop a (block, surfel)
pair if NO pixel of the block can receive alpha >= 1/255 from that surfel.  The GPU parity tests check the same
thing end to end (n_contrib is bit-exact against the checker, which has no cull at all); this test pins the
inequality itself on the CPU, on the bench scene's own records, and records how tight it is."""
import numpy as np

from oracle.oracle import Oracle
from splat_loam_amd import synth

F = np.float32


def block_cone(cam, pcx, pcy, hx, hy):
    az, el = F((pcx - cam.cx) / cam.fx), F((pcy - cam.cy) / cam.fy)
    sa, ca, se, ce = np.sin(az, dtype=F), np.cos(az, dtype=F), np.sin(el, dtype=F), np.cos(el, dtype=F)
    kx, ky = F(hx / cam.fx), F(hy / cam.fy)
    d0 = np.array([ca * ce, sa * ce, se], F)
    Dx = np.array([-kx * sa * ce, kx * ca * ce, 0], F)
    Dy = np.array([-ky * ca * se, -ky * sa * se, ky * ce], F)
    span = abs(kx) + abs(ky)
    return d0, Dx, Dy, F(0.5 * span * span * 1.01 + 4e-6)


def cone_outside(cone, q):
    d0, Dx, Dy, eps = cone
    l = d0[None] - q[:, 12:15]
    a = (q[:, 0:3] * l).sum(1, dtype=F)
    b = (q[:, 4:7] * l).sum(1, dtype=F)
    e = (q[:, 8:11] * d0[None]).sum(1, dtype=F)
    n2 = a * a + b * b
    kn = q[:, 15] * np.sqrt(n2)
    g = a[:, None] * q[:, 0:3] + b[:, None] * q[:, 4:7] + kn[:, None] * q[:, 8:11]
    reach = np.abs(g @ Dx) + np.abs(g @ Dy) + eps * np.abs(g).sum(1)
    return n2 + kn * e - reach > F(2e-4) * (n2 + kn * np.abs(e))


def test_block_footprint_test_never_drops_a_contributing_pair():
    N, H, W = 60000, 64, 1024
    sc = synth.make_scene(N, H, W, seed=3, range_lo=1.0, range_hi=40.0, scale_lo=0.01, scale_hi=0.4, max_tilt_deg=75.0)
    view, proj = synth.camera_matrices(sc["K"], synth.keyframe_poses(3)[2])
    o = Oracle(np.float32)
    cam = o.camera(H, W, view, proj)
    col, row = o.ray_tables(cam)
    pre = o.preprocess(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    binned = o.bin_sort(cam, pre)
    rec = pre["rec"].copy()
    kc = np.sqrt(2.0 * np.log(np.maximum(255.0 * sc["opac"].reshape(-1).astype(np.float64), 1.0)) * 1.001 + 1e-3) * 1.0001
    rec[:, 15] = kc.astype(F)                         # (the checker leaves the culling aids empty)
    rng = np.random.default_rng(0)
    dropped = kept = contributing = 0
    for t in rng.choice(cam.T, size=40, replace=False):
        a, b = binned["ranges"][t]
        if b <= a:
            continue
        q = rec[binned["vals"][a:b]]
        ty, tx = divmod(int(t), cam.GX)
        for by in range(0, 16, 2):
            for bx in range(0, 16, 8):
                x0, y0 = tx * 16 + bx, ty * 16 + by
                px = (x0 + np.arange(8))[None, :].repeat(2, 0).reshape(-1)
                py = (y0 + np.arange(2))[:, None].repeat(8, 1).reshape(-1)
                d = np.stack([col[px, 0] * row[py, 0], col[px, 1] * row[py, 0], row[py, 1]], 1).astype(np.float64)
                qq = q.astype(np.float64)
                nd = qq[:, 8:11] @ d.T
                hu, hv = qq[:, 0:3] @ d.T - (qq[:, 0:3] * qq[:, 12:15]).sum(1)[:, None], qq[:, 4:7] @ d.T - (qq[:, 4:7] * qq[:, 12:15]).sum(1)[:, None]
                rho3 = (hu * hu + hv * hv) / np.where(nd == 0, 1e-300, nd * nd)
                cut = 2.0 * np.log(np.maximum(255.0 * qq[:, 11], 1.0))
                in3d = ((nd < 0) & (rho3 <= cut[:, None])).any(1)          # some pixel of the block is in the 3D footprint
                out = cone_outside(block_cone(cam, x0 + 3.5, y0 + 0.5, 3.5, 0.5), q)
                assert not (out & in3d).any(), "the cone test dropped a pair whose 3D footprint reaches the block"
                dropped += int(out.sum()); kept += int((~out).sum()); contributing += int(in3d.sum())
    assert dropped > 0.5 * (dropped + kept), "on whole tile lists most (block, surfel) pairs are separable"
    assert kept <= 1.25 * contributing + 200, (kept, contributing)    # and the test is tight


def test_taylor_remainder_bound_of_the_block_rays():
    """|d(x, y) - d0 - x Dx - y Dy| <= eps for every pixel of a block, also on coarse images (large angular pitch)."""
    for H, W, hfov in ((64, 2048, 360.0), (16, 64, 360.0), (32, 128, 120.0), (128, 1024, 360.0)):
        K = synth.spherical_K(H, W, hfov_deg=hfov)
        view, proj = synth.camera_matrices(K)
        o = Oracle(np.float32)
        cam = o.camera(H, W, view, proj)
        col, row = o.ray_tables(cam)
        for (bw, bh) in ((8, 2), (4, 4)):
            for y0 in range(0, H, bh):
                for x0 in range(0, W, max(bw, W // 16)):
                    hx, hy = 0.5 * (bw - 1), 0.5 * (bh - 1)
                    d0, Dx, Dy, eps = block_cone(cam, x0 + hx, y0 + hy, hx, hy)
                    for yy in range(bh):
                        for xx in range(bw):
                            px, py = min(x0 + xx, W - 1), min(y0 + yy, H - 1)
                            d = np.array([col[px, 0] * row[py, 0], col[px, 1] * row[py, 0], row[py, 1]], np.float64)
                            x = (px - (x0 + hx)) / hx if hx else 0.0
                            y = (py - (y0 + hy)) / hy if hy else 0.0
                            r = d - d0.astype(np.float64) - x * Dx.astype(np.float64) - y * Dy.astype(np.float64)
                            assert np.linalg.norm(r) <= eps, (H, W, x0, y0, r, eps)      # |g . r| <= |g|_1 |r|_2
