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
