"""Golden vectors produced by the REFERENCE's own Python in the build container
(tools/make_golden.py; only data is committed).  They pin the parts of the hot
path the reference tree does contain: camera / back-projection conventions (G1),
render()'s allmap post-processing and its gradient (G2), rotation / activation
helpers (G3) and Mapper.optimize — loss + Adam — as parameter trajectories (G5,
run there on top of the CPU checker injected as the rasterizer).
"""
import os

import numpy as np
import torch

from splat_loam_amd import renderer, synth
from splat_loam_amd.mapping import MappingConfig, optimize_step
from splat_loam_amd.scene import Camera, SurfelModel

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cam(K, pose, depth, valid=None):
    return Camera(K, depth, None, valid, pose, data_device="cpu")


def test_g1_camera_conventions_and_backprojection():
    g = np.load(os.path.join(GOLD, "g1_camera.npz"))
    for i in range(3):
        cam = _cam(g[f"K{i}"], g[f"pose{i}"], g[f"depth{i}"])
        assert np.allclose(cam.world_view_transform.numpy(), g[f"view{i}"], atol=1e-6)
        assert np.array_equal(cam.projection_matrix.numpy(), g[f"proj{i}"])
        d = torch.from_numpy(g[f"depth{i}"])
        assert np.allclose(renderer.depth_to_points(cam, d, False).numpy(), g[f"pts_sensor{i}"], rtol=1e-5, atol=1e-5)
        assert np.allclose(renderer.depth_to_points(cam, d, True).numpy(), g[f"pts_world{i}"], rtol=1e-5, atol=2e-5)
        assert np.allclose(renderer.depth_to_normal(cam, d).numpy(), g[f"normal{i}"], atol=2e-4)
        # the C-ABI's camera parsing agrees with the same matrices
        import ctypes as C
        from splat_loam_amd import _abi
        sc = _abi.SlsCamera()
        v = cam.world_view_transform.contiguous(); p = cam.projection_matrix.contiguous()
        _abi.check(_abi.lib().sls_camera_from_matrices(v.data_ptr(), p.data_ptr(), 8, 16, 1.0, C.byref(sc)), "cam")
        K = g[f"K{i}"]
        assert (sc.fx, sc.fy, sc.cx, sc.cy) == (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        Rvw = np.array(list(sc.Rvw)).reshape(3, 3); tvw = np.array(list(sc.tvw))
        Tinv = np.linalg.inv(g[f"pose{i}"].astype(np.float64))
        assert np.allclose(Rvw, Tinv[:3, :3], atol=1e-6) and np.allclose(tvw, Tinv[:3, 3], atol=1e-5)


def test_g2_render_postprocess_and_gradient():
    g = np.load(os.path.join(GOLD, "g2_render.npz"))
    H, W = g["allmap"].shape[1:]
    cam = _cam(g["K"], g["pose"], np.ones((1, H, W), np.float32))
    for ratio, tag in ((0.0, "_r0"), (0.3, "_r3")):
        am = torch.from_numpy(g["allmap"].copy()).requires_grad_(True)
        pkg = renderer.postprocess(cam, am, ratio)
        loss = 0
        for k in ("rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal"):
            assert np.allclose(pkg[k].detach().numpy(), g[k + tag], rtol=1e-5, atol=2e-4), k
            loss = loss + (pkg[k] * torch.from_numpy(g["w_" + k + tag])).sum()
        loss.backward()
        ref = g["grad_allmap" + tag]
        assert np.abs(am.grad.numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
        if ratio == 0.0:   # depth_ratio = 0: no gradient reaches the median / distortion channels except rend_dist's own weight
            assert not am.grad[5].any()


def test_g3_rotation_and_activation_helpers():
    g = np.load(os.path.join(GOLD, "g3_utils.npz"))
    from oracle.torch_ref import build_rotation
    q = torch.from_numpy(g["q"]).double()
    R = build_rotation(q / q.norm(dim=1, keepdim=True)).numpy()          # reference normalises first
    assert np.allclose(R, g["R"], atol=1e-6)
    from splat_loam_amd.scene import inverse_sigmoid
    assert np.allclose(inverse_sigmoid(torch.from_numpy(g["x"])).numpy(), g["inv_sigmoid"], rtol=1e-6, atol=1e-6)
    # quaternion order (w, x, y, z): quat_from_R of the reference round-trips through R(q)
    Rq = build_rotation(torch.from_numpy(g["quat_from_R"]).double()).numpy()
    assert np.allclose(Rq, g["R_from_dirs"], atol=1e-5)
    assert np.allclose(synth._quat_from_R(g["R_from_dirs"].astype(np.float64)),
                       g["quat_from_R"] * np.sign(g["quat_from_R"][:, :1] + 1e-30), atol=1e-5)


def test_g5_mapper_optimize_trajectory():
    """3 iterations of the reference's Mapper.optimize (its loss code + its
    GaussianModel.training_setup Adam) == 3 x this repo's optimize_step with
    torch.optim.Adam, both on top of the CPU checker as the rasterizer."""
    from oracle.torch_function import GaussianRasterizer as OracleRasterizer
    g = np.load(os.path.join(GOLD, "g5_mapper.npz"))
    cam = _cam(g["K"], g["pose"], g["depth"], g["valid"])
    model = SurfelModel(g["init_xyz"], g["init_scaling"], g["init_rotation"], g["init_opacity"], device="cpu")
    assert np.allclose(g["lr"], [5e-4, 5e-2, 5e-3, 1e-3])                # utils/config_utils.py:180-183
    model.training_setup(*[float(x) for x in g["lr"][[0, 1, 2, 3]]], fused=False)
    cfg = MappingConfig(opt_lambda_alpha=0.4, opt_lambda_normal=0.5, opt_scaling_max=0.1, opt_scaling_max_penalty=1.0)
    for _ in range(3):
        optimize_step(model, cam, cfg, rasterizer_cls=OracleRasterizer)
    for name in ("_xyz", "_scaling", "_rotation", "_opacity"):
        got = getattr(model, name).detach().numpy()
        ref, init = g["final" + name], g["init" + name]
        moved = np.abs(ref - init).max()
        assert moved > 0
        assert np.abs(got - ref).max() <= 2e-3 * moved + 1e-7, name


def test_g6_slam_rules_against_the_reference():
    """The rules that size the hot path's workload (splat_loam_amd/slam_rules.py) against the reference's own
    functions: compute_depth_gradient (utils/graphic_utils.py:91-106), sample_geometric
    (utils/sampling_utils.py:11-20; also the golden G3 cases) and Tracker.require_new_keyframe's truth table
    (slam/tracker.py:61-84, evaluated by the reference's method when the fixture was made)."""
    from splat_loam_amd import slam_rules as sr
    g = np.load(os.path.join(GOLD, "g6_slam_rules.npz"))
    dg = sr.compute_depth_gradient(torch.from_numpy(g["depth"]), torch.from_numpy(g["valid"])).numpy()
    assert np.array_equal(dg, g["depth_gradient"])
    g3 = np.load(os.path.join(GOLD, "g3_utils.npz"))
    for key, (n, p) in {"geom_1_4": (1, 0.4), "geom_5_4": (5, 0.4), "geom_8_7": (8, 0.7)}.items():
        assert np.allclose(sr.sample_geometric(n, p), g3[key], rtol=1e-12, atol=0)
    assert np.allclose(sr.sample_geometric(8, 0.4), g["geom_8_4"], rtol=1e-12)
    assert np.allclose(sr.keyframe_probabilities(2, 0.4), g["geom_2_4"], rtol=1e-12)
    assert np.allclose(sr.keyframe_probabilities(4, None), 0.25) and np.allclose(sr.keyframe_probabilities(4, -1.0), 0.25)
    for nfr, fit, dist, t_n, t_f, t_d, want in g["keyframe_rule"]:
        T = np.eye(4); T[0, 3] = dist
        got = sr.require_new_keyframe(int(nfr), fit, T, int(t_n), t_f, t_d)
        assert got == bool(want), (nfr, fit, dist, t_n, t_f, t_d)


def test_densify_selection_follows_the_mapper():
    """slam/mapper.py:51-102 restated: candidates = valid & alpha <= threshold (| the depth-error quantile mask),
    int(percentage * #candidates) of them drawn without replacement, weighted by the log-depth gradient."""
    from splat_loam_amd import slam_rules as sr
    g = np.load(os.path.join(GOLD, "g6_slam_rules.npz"))
    depth, valid = torch.from_numpy(g["depth"]), torch.from_numpy(g["valid"])
    rng = np.random.default_rng(0)
    alpha = torch.from_numpy(rng.uniform(size=depth.shape).astype(np.float32))
    cand = sr.densify_candidates(valid, alpha, depth * 1.1, depth, 0.5, -1.0)
    assert np.array_equal(cand.numpy(), (alpha[0].numpy() <= 0.5) & (g["valid"][0] == 1))
    first = sr.densify_candidates(valid, initialize_model=True)
    assert np.array_equal(first.numpy(), g["valid"][0] == 1)
    sd = depth.clone(); sd[0, 8, 20] = depth[0, 8, 20] + 30.0      # one pixel rendered far behind its measurement
    valid2 = valid.clone(); valid2[0, 8, 20] = 1
    cand_e = sr.densify_candidates(valid2, torch.ones_like(alpha), sd, depth, 0.5, 0.1)
    assert bool(cand_e[8, 20]) and int(cand_e.sum()) == 1
    gen = torch.Generator().manual_seed(1)
    m = sr.densify_sample(cand, depth, valid, 0.15, generator=gen)
    assert m is not None and int(m.sum()) == int(0.15 * int(cand.sum())) and not bool((m & ~cand).any())
    w = sr.compute_depth_gradient(depth, valid)[0]
    assert float(w[m].min()) > 0.0                                   # zero-gradient pixels are never drawn
    assert sr.densify_sample(cand & False, depth, valid, 0.15) is None
    keep = sr.prune_mask(torch.tensor([[0.05], [0.5]]), torch.tensor([[0.1, 0.1], [0.001, 0.001]]), 0.1, 0.01)
    assert keep.tolist() == [True, True] and not sr.prune_mask(torch.tensor([[0.05]]), torch.tensor([[1.0, 1.0]])).any()
