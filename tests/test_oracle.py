"""The CPU checker itself: known answers, analytic-vs-autograd gradients in
float64, finite differences, integer invariants and edge cases.  "parity
unpinned" (oracle/sls_oracle.c header): there are no reference tests or golden
vectors for the rasterizer, so the checker is anchored on (i) an independent
float64 autograd formulation (oracle/torch_ref.py), (ii) finite differences,
(iii) consistency with the reference's back-projection (tests/test_golden.py).
"""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from splat_loam_amd import synth


def _scene64(N, H, W, seed, **kw):
    sc = synth.make_scene(N, H, W, seed=seed, **kw)
    sc["rots"] = sc["rots"].astype(np.float64)
    sc["rots"] /= np.linalg.norm(sc["rots"], axis=1, keepdims=True)
    return sc


def test_det_atan2_accuracy(oracle32, oracle64):
    rng = np.random.default_rng(0)
    y, x = rng.normal(size=200000), rng.normal(size=200000)
    y[:8] = [0, 0, 1, -1, 0.0, 1e-30, -1e-30, 5]
    x[:8] = [1, -1, 0, 0, 0.0, -1, -1, 5]
    for o, tol in ((oracle32, 4e-7), (oracle64, 1.2e-7)):   # f32: polynomial 1.1e-7 + ulp(pi) 2.4e-7
        got = o.atan2(y, x).astype(np.float64)
        ref = np.arctan2(y.astype(o.dtype).astype(np.float64), x.astype(o.dtype).astype(np.float64))
        assert np.abs(got - ref).max() <= tol
    assert oracle32.atan2(np.zeros(1), np.zeros(1))[0] == 0.0


def test_forward_matches_independent_autograd_formulation(oracle64):
    N, H, W = 40, 16, 64
    sc = _scene64(N, H, W, 1, range_lo=2.0, range_hi=6.0, scale_lo=0.05, scale_hi=0.4)
    view = np.linalg.inv(synth.keyframe_poses(2)[1]).T          # float64, exactly orthonormal
    _, proj = synth.camera_matrices(sc["K"])
    cam = oracle64.camera(H, W, view, proj)
    assert cam.wrap == 1
    st = oracle64.forward(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    t = [torch.tensor(np.asarray(sc[k], np.float64), requires_grad=True) for k in ("means", "scales", "rots", "opac")]
    am = torch_ref.dense_forward(cam, st["tables"], st["pre"], *t)
    assert np.abs(am.detach().numpy() - st["allmap"]).max() < 1e-12
    Wt = np.random.default_rng(5).normal(size=(7, H, W))
    (am * torch.tensor(Wt)).sum().backward()
    bw = oracle64.backward(st, Wt)
    assert np.abs(t[0].grad.numpy() - bw["dmeans"]).max() < 1e-10
    assert np.abs(t[1].grad.numpy() - bw["dscales"]).max() < 1e-10
    assert np.abs(t[3].grad.numpy() - bw["dopac"]).max() < 1e-10
    # rotations: equal up to the component along q, which F.normalize's backward
    # annihilates (the Hu/Hv form assumes an orthonormal R(q))
    q = sc["rots"]
    proj_t = lambda g: g - (g * q).sum(1, keepdims=True) * q
    assert np.abs(proj_t(t[2].grad.numpy()) - proj_t(bw["drots"])).max() < 1e-10


def test_backward_matches_finite_differences(oracle64):
    N, H, W = 12, 16, 32
    sc = _scene64(N, H, W, 3, range_lo=2.0, range_hi=5.0, scale_lo=0.1, scale_hi=0.5)
    view, proj = synth.camera_matrices(sc["K"])
    cam = oracle64.camera(H, W, view.astype(np.float64), proj)
    Wt = np.random.default_rng(1).normal(size=(7, H, W))
    Wt[5] = 0  # the median channel is piecewise constant in the surfel order

    def L(m, s, r, o):
        return (oracle64.forward(cam, m, s, r, o)["allmap"] * Wt).sum()
    args = [np.asarray(sc[k], np.float64) for k in ("means", "scales", "rots", "opac")]
    st = oracle64.forward(cam, *args)
    bw = oracle64.backward(st, Wt)
    eps = 1e-6
    for which, an in ((0, bw["dmeans"]), (1, bw["dscales"]), (3, bw["dopac"])):
        for i in range(0, N, 3):
            for k in range(args[which].shape[1]):
                a = [x.copy() for x in args]; b = [x.copy() for x in args]
                a[which][i, k] += eps; b[which][i, k] -= eps
                fd = (L(*a) - L(*b)) / (2 * eps)
                assert abs(fd - an[i, k]) <= 2e-5 * max(1.0, abs(fd)), (which, i, k, fd, an[i, k])


def test_single_surfel_known_answer(oracle32):
    H, W = 32, 128
    K = synth.spherical_K(H, W).astype(np.float64)
    view, proj = synth.camera_matrices(K.astype(np.float32))
    c, r, rng_m, o = 40, 12, 7.5, 0.8
    az, el = (c - K[0, 2]) / K[0, 0], (r - K[1, 2]) / K[1, 1]
    ray = np.array([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)])
    tn = -ray
    tu = np.cross(tn, [0.0, 0.0, 1.0]); tu /= np.linalg.norm(tu)
    tv = np.cross(tn, tu)
    q = synth._quat_from_R(np.stack([tu, tv, tn], 1)[None])[0]
    cam = oracle32.camera(H, W, view, proj)
    st = oracle32.forward(cam, (ray * rng_m)[None], np.array([[0.3, 0.3]]), q[None], np.array([[o]]))
    am = st["allmap"]
    assert abs(am[1, r, c] - o) < 2e-5 and abs(am[0, r, c] - rng_m * o) < 2e-4
    assert np.abs(am[2:5, r, c] - (-ray) * o).max() < 2e-5      # normal faces the sensor
    assert abs(am[5, r, c] - rng_m) < 1e-4 and st["radii"][0] > 0
    # the same surfel seen from behind: the normal is flipped to face the sensor
    q_back = synth._quat_from_R(np.stack([tu, -tv, -tn], 1)[None])[0]
    st2 = oracle32.forward(cam, (ray * rng_m)[None], np.array([[0.3, 0.3]]), q_back[None], np.array([[o]]))
    assert np.abs(st2["allmap"][2:5, r, c] - (-ray) * o).max() < 2e-5


def test_backprojected_surfel_lands_on_its_pixel(oracle32):
    """A surfel placed at ray(c, r) * range renders that range at pixel (c, r) (D1)."""
    H, W = 16, 64
    K = synth.spherical_K(H, W).astype(np.float64)
    pose = synth.keyframe_poses(3)[2]
    view, proj = synth.camera_matrices(K.astype(np.float32), pose)
    cam = oracle32.camera(H, W, view, proj)
    col, row = oracle32.ray_tables(cam)
    for (c, r, rho) in ((3, 2, 4.0), (60, 13, 11.0), (31, 8, 25.0)):
        d = np.array([col[c, 0] * row[r, 0], col[c, 1] * row[r, 0], row[r, 1]], np.float64)
        p_world = pose[:3, :3] @ (d * rho) + pose[:3, 3]
        n_world = pose[:3, :3] @ (-d)
        tu = np.cross(n_world, [0.3, 0.1, 1.0]); tu /= np.linalg.norm(tu)
        q = synth._quat_from_R(np.stack([tu, np.cross(n_world, tu), n_world], 1)[None])[0]
        st = oracle32.forward(cam, p_world[None], np.array([[0.05 * rho, 0.05 * rho]]), q[None], np.array([[0.9]]))
        am = st["allmap"]
        assert np.unravel_index(am[1].argmax(), am[1].shape) == (r, c)
        assert abs(am[0, r, c] / am[1, r, c] - rho) < 1e-3 * rho


def test_pixel_centre_offset_meets_the_reference_backprojection(oracle32):
    """With pix_offset = (-0.5, -0.5) the rasterizer's pixel (c, r) is the direction the reference's own
    back-projection gives that pixel (utils/graphic_utils.py:46-49: K^-1 [c - 0.5, r - 0.5, 1]): a surfel put there
    renders CENTRED on (r, c) — equal alpha left and right — while the default D1 render of the same surfel is
    lopsided by the half pixel DESIGN.md section 9 reports as a tracking bias."""
    H, W = 16, 64
    K = synth.spherical_K(H, W).astype(np.float64)
    view, proj = synth.camera_matrices(K.astype(np.float32), None)
    cam_h = oracle32.camera(H, W, view, proj, pix_offset=(-0.5, -0.5))
    cam_0 = oracle32.camera(H, W, view, proj)
    assert abs(cam_h.cx - (cam_0.cx + 0.5)) < 1e-6 and abs(cam_h.cy - (cam_0.cy + 0.5)) < 1e-6
    for (c, r, rho) in ((20, 6, 6.0), (41, 9, 15.0)):
        az, el = (c - 0.5 - K[0, 2]) / K[0, 0], (r - 0.5 - K[1, 2]) / K[1, 1]
        d = np.array([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)])
        n = -d
        tu = np.cross(n, [0.3, 0.1, 1.0]); tu /= np.linalg.norm(tu)
        q = synth._quat_from_R(np.stack([tu, np.cross(n, tu), n], 1)[None])[0]
        args = ((d * rho)[None], np.array([[0.04 * rho, 0.04 * rho]]), q[None], np.array([[0.9]]))
        a_h = oracle32.forward(cam_h, *args)["allmap"][1]
        a_0 = oracle32.forward(cam_0, *args)["allmap"][1]
        assert np.unravel_index(a_h.argmax(), a_h.shape) == (r, c)
        assert abs(a_h[r, c - 1] - a_h[r, c + 1]) < 2e-3 * a_h[r, c] and abs(a_h[r - 1, c] - a_h[r + 1, c]) < 2e-3 * a_h[r, c]
        assert abs(a_0[r, c - 1] - a_0[r, c + 1]) > 0.02 * a_0[r, c]        # D1: shifted by half a pixel


def test_sorted_list_invariants_and_seam(oracle32):
    N, H, W = 3000, 32, 256
    sc = synth.make_scene(N, H, W, seed=4)
    # put a few big surfels exactly on the azimuth seam (az = +-pi)
    sc["means"][:5] = np.array([[-6.0, 1e-4 * k - 2e-4, -0.5] for k in range(5)], np.float32)
    sc["scales"][:5] = 0.4
    view, proj = synth.camera_matrices(sc["K"])
    cam = oracle32.camera(H, W, view, proj)
    assert cam.wrap == 1
    st = oracle32.forward(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    b, pre = st["binned"], st["pre"]
    keys, vals = b["keys"], b["vals"]
    assert np.all(keys[1:] >= keys[:-1])
    same = keys[1:] == keys[:-1]
    assert np.all(vals[1:][same] > vals[:-1][same])                       # stable (D9)
    assert sorted(zip(b["keys_unsorted"].tolist(), b["vals_unsorted"].tolist())) == list(zip(keys.tolist(), vals.tolist()))
    assert np.array_equal(np.bincount(vals, minlength=N).astype(np.uint32), pre["tiles"])
    rng = b["ranges"].astype(np.int64)
    assert np.array_equal(rng[:, 1] - rng[:, 0], np.bincount((keys >> np.uint64(32)).astype(np.int64), minlength=cam.T))
    # seam surfels touch both the first and the last tile column
    for i in range(5):
        txlo, ncols = pre["rect"][i, 0], pre["rect"][i, 1]
        cols = {(txlo + k) % cam.GX for k in range(ncols)}
        assert 0 in cols and cam.GX - 1 in cols
    # and render on both image edges
    assert st["allmap"][1, :, 0].max() > 0.1 and st["allmap"][1, :, W - 1].max() > 0.1


def test_empty_culled_and_ragged(oracle32):
    H, W = 40, 200                                    # not multiples of 16; 120 deg => no wrap
    K = synth.spherical_K(H, W, hfov_deg=120.0)
    view, proj = synth.camera_matrices(K)
    cam = oracle32.camera(H, W, view, proj)
    assert cam.wrap == 0 and cam.GX == 13 and cam.GY == 3
    z = lambda *s: np.zeros(s, np.float32)
    st = oracle32.forward(cam, z(0, 3), z(0, 2), z(0, 4), z(0, 1))
    assert st["binned"]["R"] == 0 and not st["allmap"].any()
    sc = synth.make_scene(500, H, W, seed=2, range_lo=0.01, range_hi=0.19)      # all inside the near cut
    st = oracle32.forward(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    assert st["binned"]["R"] == 0 and not (st["radii"] > 0).any()
    sc = synth.make_scene(2000, H, W, seed=2)          # 360-degree cloud, 120-degree camera
    st = oracle32.forward(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    az = np.arctan2(sc["means"][:, 1], sc["means"][:, 0])
    assert not (st["radii"][np.abs(az) > np.radians(75)] > 0).any()            # off-image => culled
    assert (st["radii"][np.abs(az) < np.radians(50)] > 0).all()
    bw = oracle32.backward(st, np.ones((7, H, W), np.float32))
    assert np.isfinite(bw["dmeans"]).all() and not bw["dmeans"][st["radii"] == 0].any()


def test_f32_checker_tracks_f64(oracle32, oracle64):
    N, H, W = 1500, 32, 128
    sc = synth.make_scene(N, H, W, seed=9, range_lo=2.0, range_hi=20.0)
    view, proj = synth.camera_matrices(sc["K"])
    a = oracle32.forward(oracle32.camera(H, W, view, proj), sc["means"], sc["scales"], sc["rots"], sc["opac"])
    b = oracle64.forward(oracle64.camera(H, W, view, proj), sc["means"], sc["scales"], sc["rots"], sc["opac"])
    ok = ~(a["fwd"]["fragile"] | b["fwd"]["fragile"])
    for c in range(5):
        scale = np.abs(b["allmap"][c]).max()
        assert (np.abs(a["allmap"][c] - b["allmap"][c]) / scale)[ok].max() < 5e-5


def test_knn_bruteforce(oracle32):
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [10, 10, 10]], np.float32)
    d = oracle32.knn_dist2(pts)
    assert np.allclose(d[0], (1 + 4 + 9) / 3) and np.allclose(d[1], (1 + 5 + 10) / 3)
    assert oracle32.knn_dist2(pts[:3])[0] > 1e37      # fewer than 4 points: FLT_MAX sentinel (lineage)


def test_adam_matches_torch(oracle32):
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(1000, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=5e-3, eps=1e-15)
    p = p0.numpy().copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    for step in range(1, 5):
        gr = torch.randn(1000, generator=g)
        p_ref.grad = gr.clone(); opt.step()
        oracle32.adam(p, gr.numpy(), m, v, 5e-3, step)
    assert np.abs(p - p_ref.detach().numpy()).max() <= 2e-6 * np.abs(p).max()


def test_pure_torch_tile_rasterizer_matches_checker(oracle32):
    """oracle/torch_tiles.py — the pure-PyTorch, tile-batched CPU rasterizer bench.py times as the CPU baseline
    BASELINE.json names — computes the checker's function: same instance count, same image, and its autograd
    gradients equal the checker's hand-written backward (float32 torch vs float32 C: 5e-5)."""
    import torch
    from helpers import tangent
    from oracle import torch_tiles as tt
    N, H, W = 4000, 32, 256
    sc = synth.make_scene(N, H, W, seed=5, range_lo=2.0, range_hi=20.0)
    view, proj = synth.camera_matrices(sc["K"], synth.keyframe_poses(2)[1])
    # (the pure-torch rasterizer bins a surfel's whole tile rectangle; the checker's D10 test only removes
    #  instances that contribute nothing, so the images agree either way — compared with D10 on — and the
    #  instance count is the un-culled one)
    cam = oracle32.camera(H, W, view, proj)
    ost = oracle32.forward(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
    R_rect = oracle32.bin_sort(oracle32.camera(H, W, view, proj, tile_cull=False), oracle32.preprocess(
        oracle32.camera(H, W, view, proj, tile_cull=False), sc["means"], sc["scales"], sc["rots"], sc["opac"]))["R"]
    t = {k: torch.tensor(sc[k]).requires_grad_(True) for k in ("means", "scales", "rots", "opac")}
    stats = {}
    radii, am = tt.rasterize(tt.camera_dict(H, W, view, proj), t["means"], t["scales"], t["rots"], t["opac"], stats=stats)
    assert stats["R"] == R_rect >= ost["binned"]["R"] and np.array_equal(radii.numpy(), ost["radii"])
    ok = ~ost["fwd"]["fragile"]
    for ch in range(7):
        ref = ost["allmap"][ch]
        scale = max(np.abs(ref).max(), 1.0 if ch == 6 else 1e-12)
        assert (np.abs(am[ch].detach().numpy() - ref) / scale)[ok].max() <= 5e-5, ch
    dL = np.random.default_rng(0).normal(size=(7, H, W)).astype(np.float32)
    dL[:, ~ok] = 0
    (am * torch.tensor(dL)).sum().backward()
    ob = oracle32.backward(ost, dL)
    for k, r in (("means", "dmeans"), ("scales", "dscales"), ("opac", "dopac")):
        assert np.abs(t[k].grad.numpy() - ob[r]).max() <= 5e-5 * np.abs(ob[r]).max(), k
    rot = sc["rots"].astype(np.float64)
    g, ref = tangent(t["rots"].grad.numpy().astype(np.float64), rot), tangent(ob["drots"].astype(np.float64), rot)
    assert np.abs(g - ref).max() <= 5e-5 * np.abs(ref).max()
    # a tile subset renders exactly those tiles
    with torch.no_grad():
        _, part = tt.rasterize(tt.camera_dict(H, W, view, proj), t["means"], t["scales"], t["rots"], t["opac"], tiles=[3, 17])
    full = am.detach()
    assert torch.equal(part[:, 0:16, 48:64], full[:, 0:16, 48:64]) and torch.equal(part[:, 16:32, 16:32], full[:, 16:32, 16:32])
    assert float(part[:, :, 64:].abs().max()) == 0.0


def test_integer_outputs_reproduced_with_independent_math(oracle32):
    """the checker and the kernels share include/sls_det_math.h (polynomial atan2 / asin) and
    sls_spec.h, so a defect there is invisible to HIP-vs-checker.  oracle/torch_tiles.py computes the same projection
    and 3-sigma extents with torch's own atan2 / asin / norm and its own copy of the constants: on BASELINE config 2's
    scene and on a near, large-footprint scene every radius, every tile rectangle and every tile's instance count come
    out identical (the depth-sorted order inside a tile may differ where two ranges differ by an ulp)."""
    import torch
    from oracle import torch_tiles as tt
    for (N, H, W, seed, kw) in ((50000, 64, 1024, 0, {}),
                                (20000, 64, 512, 9, dict(range_lo=1.0, range_hi=8.0, scale_lo=0.05, scale_hi=0.5))):
        sc = synth.make_scene(N, H, W, seed=seed, **kw)
        view, proj = synth.camera_matrices(sc["K"], synth.keyframe_poses(2)[1])
        cam = oracle32.camera(H, W, view, proj, tile_cull=False)      # rectangles: D10 has its own tests (test_tile_cull.py)
        pre = oracle32.preprocess(cam, sc["means"], sc["scales"], sc["rots"], sc["opac"])
        b = oracle32.bin_sort(cam, pre)
        t = {k: torch.tensor(sc[k]) for k in ("means", "scales", "rots", "opac")}
        _, bb = tt.preprocess(tt.camera_dict(H, W, view, proj), t["means"], t["scales"], t["rots"], t["opac"])
        vals, ranges = tt.bin_sort(bb)
        vis = pre["radii"] > 0
        assert np.array_equal(bb["radii"].numpy(), pre["radii"])
        rect = torch.stack([bb["txlo"], bb["ncols"], bb["tylo"], bb["nrows"]], 1).numpy().astype(np.int32)
        assert np.array_equal(rect[vis], pre["rect"][vis])
        assert vals.numel() == b["R"] and np.array_equal(ranges.numpy().astype(np.uint32), b["ranges"])


def test_tiled_float64_formulation_equals_the_dense_one():
    """oracle/torch_ref.py: dense_forward_tiled (the tiles as a batch, every surfel only on the tiles of its rectangle) is
    the SAME function as dense_forward — image and gradients to float64 rounding — on a wrapping 32x128 image and on a
    ragged 40x72 one (tiles that overhang the image).  The tiled form is what lets the GPU suite hold the HIP path to the
    float64 formulation at 2 000 surfels (tests/test_gpu_parity.py)."""
    import torch
    from oracle import torch_ref
    from oracle.oracle import Oracle
    from splat_loam_amd import synth
    o = Oracle(np.float64)
    for N, H, W in ((60, 32, 128), (40, 40, 72)):
        sc = synth.make_scene(N, H, W, seed=3, range_lo=2.0, range_hi=8.0, scale_lo=0.05, scale_hi=0.4)
        view, proj = synth.camera_matrices(sc["K"])
        cam = o.camera(H, W, view.astype(np.float64), proj)
        a64 = [np.asarray(sc[k], np.float64) for k in ("means", "scales", "rots", "opac")]
        ost = o.forward(cam, *a64)
        la = [torch.tensor(a, requires_grad=True) for a in a64]
        lb = [torch.tensor(a, requires_grad=True) for a in a64]
        A = torch_ref.dense_forward(cam, ost["tables"], ost["pre"], *la)
        B = torch_ref.dense_forward_tiled(cam, ost["tables"], ost["pre"], *lb)
        assert float((A - B).detach().abs().max()) <= 1e-13
        dL = torch.tensor(np.random.default_rng(0).normal(size=(7, H, W)))
        (A * dL).sum().backward()
        (B * dL).sum().backward()
        for x, y in zip(la, lb):
            assert float((x.grad - y.grad).abs().max()) <= 1e-12 * float(x.grad.abs().max())
