"""D10 — the tile-level footprint test of the binning (include/sls_det_math.h: sls_tile_outside, used by the
preprocess kernel and by the checker alike).  It decides which instances of a surfel's tile rectangle exist at
all, so it must never drop a (tile, surfel) pair that contributes to a pixel:
  * switching it off must not change the rendered image, the transmittance or the gradients by a single bit
    (an instance the test drops would have been skipped at every pixel of its tile);
  * an independent float64 evaluation of alpha over every pixel of the dropped tiles finds nothing >= 1/255;
  * and it is worth having: >= 9 % fewer instances on the bench scene (C3), ~15 % in the consumed list prefixes."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from splat_loam_amd import synth

CASES = [
    ("bench_like", 60000, 64, 1024, 360.0, None, {}),
    ("aniso_tilted", 8000, 64, 1024, 360.0, None, dict(range_lo=1.5, range_hi=12.0, scale_lo=0.01, scale_hi=0.6, max_tilt_deg=80.0)),
    ("dense_near_pose", 5000, 64, 512, 360.0, 2, dict(range_lo=1.0, range_hi=8.0, scale_lo=0.05, scale_hi=0.5)),
    ("ragged_nowrap", 4000, 40, 200, 120.0, None, {}),
    ("half_pixel_offset", 6000, 32, 512, 360.0, 1, dict(range_lo=2.0, range_hi=20.0)),
]


def _scene(N, H, W, hfov, pose_idx, kw, seed=5):
    sc = synth.make_scene(N, H, W, seed=seed, **kw)
    if hfov != 360.0:
        sc["K"] = synth.spherical_K(H, W, hfov_deg=hfov)
    pose = None if pose_idx is None else synth.keyframe_poses(3)[pose_idx]
    view, proj = synth.camera_matrices(sc["K"], pose)
    return sc, view, proj


@pytest.mark.parametrize("name,N,H,W,hfov,pose_idx,kw", CASES, ids=[c[0] for c in CASES])
def test_tile_cull_changes_no_pixel_and_no_gradient(oracle32, name, N, H, W, hfov, pose_idx, kw):
    sc, view, proj = _scene(N, H, W, hfov, pose_idx, kw)
    off = (-0.5, -0.5) if name == "half_pixel_offset" else (0.0, 0.0)
    args = (sc["means"], sc["scales"], sc["rots"], sc["opac"])
    on = oracle32.forward(oracle32.camera(H, W, view, proj, pix_offset=off, tile_cull=3), *args)
    no = oracle32.forward(oracle32.camera(H, W, view, proj, pix_offset=off, tile_cull=False), *args)
    R_on, R_no = on["binned"]["R"], no["binned"]["R"]
    assert R_on <= R_no and np.array_equal(on["pre"]["rect"], no["pre"]["rect"])
    assert np.array_equal(on["allmap"].view(np.uint32), no["allmap"].view(np.uint32)), f"{name}: the image changed"
    assert np.array_equal(on["fwd"]["pixT"].view(np.uint32), no["fwd"]["pixT"].view(np.uint32))
    dL = np.random.default_rng(1).normal(size=(7, H, W)).astype(np.float32)
    g_on, g_no = oracle32.backward(on, dL), oracle32.backward(no, dL)
    for k in ("dmeans", "dscales", "drots", "dopac"):
        assert np.array_equal(g_on[k].view(np.uint32), g_no[k].view(np.uint32)), f"{name}: {k} changed"
    # masks are consistent with the counts
    t, m = on["pre"]["tiles"], on["pre"]["tmask"]
    nrect = on["pre"]["rect"][:, 1] * on["pre"]["rect"][:, 3]
    pc = np.array([bin(int(x)).count("1") for x in m], dtype=np.uint32)
    small = nrect <= 64
    assert np.array_equal(t[small], pc[small]) and np.array_equal(t[~small], nrect[~small].astype(np.uint32))
    assert np.array_equal(no["pre"]["tiles"], nrect.astype(np.uint32))
    print(f"\n[{name}] instances {R_no} -> {R_on} (-{100 * (1 - R_on / max(R_no, 1)):.1f} %)")


def test_dropped_tiles_hold_no_pixel_above_the_alpha_threshold(oracle32):
    """Independent of the shared header: float64 alpha of every pixel of every DROPPED (surfel, tile) pair, from the
    spec and the raw inputs, never reaches 1/255 (a sample of the bench-like scene's drops)."""
    N, H, W = 60000, 64, 1024
    sc, view, proj = _scene(N, H, W, 360.0, None, {}, seed=9)
    cam = oracle32.camera(H, W, view, proj, tile_cull=3)
    pre = oracle32.preprocess(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    rect, mask = pre["rect"], pre["tmask"]
    nrect = rect[:, 1] * rect[:, 3]
    rng = np.random.default_rng(0)
    cand = np.nonzero((nrect >= 3) & (nrect <= 64) & (pre["tiles"] < nrect))[0]
    assert len(cand) > 500
    V = np.asarray(view, np.float64)
    Rvw, tvw = V[:3, :3].T, V[3, :3]
    checked = 0
    for i in rng.choice(cand, size=400, replace=False):
        p = Rvw @ sc["means"][i].astype(np.float64) + tvw
        w, x, y, z = sc["rots"][i].astype(np.float64)
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                       [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        Tu, Tv, Tn = Rvw @ Rq[:, 0], Rvw @ Rq[:, 1], Rvw @ Rq[:, 2]
        su, sv = sc["scales"][i].astype(np.float64)
        o = float(np.asarray(sc["opac"][i]).reshape(-1)[0])
        rho = np.linalg.norm(p)
        cpx = cam.fx * np.arctan2(p[1], p[0]) + cam.cx
        cpy = cam.fy * np.arctan2(p[2], np.hypot(p[0], p[1])) + cam.cy
        txlo, ncols, tylo, nrows = (int(v) for v in rect[i])
        for idx in range(int(nrect[i])):
            if (int(mask[i]) >> idx) & 1:
                continue
            ky, kx = divmod(idx, ncols)
            tx, ty = (txlo + kx) % cam.GX, tylo + ky
            cols = np.arange(tx * 16, min(tx * 16 + 16, W), dtype=np.float64)
            rows = np.arange(ty * 16, min(ty * 16 + 16, H), dtype=np.float64)
            az, el = (cols[None, :] - cam.cx) / cam.fx, (rows[:, None] - cam.cy) / cam.fy
            d = np.stack([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el) * np.ones_like(az)], -1)
            nd = d @ Tn
            t = (Tn @ p) / np.where(nd == 0, 1e-300, nd)
            hit = t[..., None] * d - p
            u, v = (hit @ Tu) / su, (hit @ Tv) / sv
            rho3 = u * u + v * v
            dx = cols[None, :] - cpx
            dx = dx - W * np.rint(dx / W) if cam.wrap else dx
            rho2 = 2.0 * (dx * dx + (rows[:, None] - cpy) ** 2)
            valid3 = (np.sign(Tn @ p) * nd > 0)          # the sensor-facing side: n.d < 0 with n = -sign(Tn.p) Tn
            rho_eff = np.where(valid3 & (rho3 <= rho2), rho3, rho2)
            alpha = np.minimum(0.99, o * np.exp(-0.5 * rho_eff))
            assert alpha.max() < 1.0 / 255.0, (i, idx, alpha.max())
            checked += 1
    assert checked >= 400


@pytest.mark.parametrize("kmin", [3, 6])
def test_tile_cull_on_the_bench_scene(kmin):
    """C3 (500k surfels, 64x2048), identical image; rectangles of >= 3 tiles tested: 9.9 % fewer instances, 17.9 %
    shorter consumed prefixes (the default); >= 6 tiles (a third of the tests): 6 % / 13.7 %."""
    o = Oracle(np.float32)
    N, H, W = 500_000, 64, 2048
    sc = synth.make_scene(N, H, W, seed=0)
    view, proj = synth.camera_matrices(sc["K"], None)
    args = (sc["means"], sc["scales"], sc["rots"], sc["opac"])
    on = o.forward(o.camera(H, W, view, proj, tile_cull=kmin), *args, frag_tol=0.0)
    no = o.forward(o.camera(H, W, view, proj, tile_cull=False), *args, frag_tol=0.0)
    assert np.array_equal(on["allmap"].view(np.uint32), no["allmap"].view(np.uint32))
    R_on, R_no = on["binned"]["R"], no["binned"]["R"]
    c_on, c_no = int(on["fwd"]["tile_consumed"].sum()), int(no["fwd"]["tile_consumed"].sum())
    print(f"\n[C3, rectangles of >= {kmin} tiles] instances {R_no} -> {R_on} (-{100 * (1 - R_on / R_no):.1f} %), consumed prefixes {c_no} -> {c_on} "
          f"(-{100 * (1 - c_on / c_no):.1f} %)")
    assert (R_on <= 0.905 * R_no and c_on <= 0.83 * c_no) if kmin == 3 else (R_on <= 0.95 * R_no and c_on <= 0.875 * c_no)
