"""render(): the allmap -> {rend_alpha, rend_normal, rend_dist, surf_depth,
surf_normal} stage around the rasterizer, and the spherical back-projection it
needs.  Own implementation of the contract of gaussian_renderer/__init__.py:11-93
and utils/graphic_utils.py:26-88 (same names, same dict keys, same half-pixel
convention `(c - 0.5, r - 0.5)` in the back-projection, D1) so the harness in
bench.py / tests reads like slam/mapper.py.  In a Splat-LOAM checkout the
reference's own gaussian_renderer/__init__.py is used unchanged on top of
`diff_surfel_spherical_rasterization`; this module exists
because reference code cannot travel to the GPU box.  Pinned against the
reference by tests/golden/g1_camera.npz and g2_render.npz.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, _stream, get_camera, half_pixel_tables


def pixel_rays(camera) -> torch.Tensor:
    """(H, W, 3) unit rays in the sensor frame for pixel (c, r) at image
    coordinate (c - 0.5, r - 0.5)  — utils/graphic_utils.py:41-59."""
    W, H = int(camera.image_width), int(camera.image_height)
    dev = camera.world_view_transform.device
    K = camera.projection_matrix[:3, :3].T
    Kinv = torch.linalg.inv(K)
    xs = torch.arange(W, dtype=torch.float32, device=dev) - 0.5
    ys = torch.arange(H, dtype=torch.float32, device=dev) - 0.5
    gx = xs[None, :].expand(H, W)
    gy = ys[:, None].expand(H, W)
    az = Kinv[0, 0] * gx + Kinv[0, 1] * gy + Kinv[0, 2]
    el = Kinv[1, 0] * gx + Kinv[1, 1] * gy + Kinv[1, 2]
    ce = torch.cos(el)
    return torch.stack([torch.cos(az) * ce, torch.sin(az) * ce, torch.sin(el)], dim=-1)


def depth_to_points(camera, depth: torch.Tensor, transform_in_world: bool = True) -> torch.Tensor:
    """[1,H,W] range image -> [3,H,W] points (utils/graphic_utils.py:26-66)."""
    rays = pixel_rays(camera)
    rng = depth.squeeze(0).unsqueeze(-1)
    if transform_in_world:
        c2w = torch.linalg.inv(camera.world_view_transform.T)
        pts = rng * (rays @ c2w[:3, :3].T) + c2w[:3, 3]
    else:
        pts = rng * rays
    return pts.permute(2, 0, 1)


def depth_to_normal(camera, depth: torch.Tensor) -> torch.Tensor:
    """Normals from central differences of the back-projected range image; zero
    on the 1-px border (utils/graphic_utils.py:69-88)."""
    pts = depth_to_points(camera, depth)
    out = torch.zeros((3, int(camera.image_height), int(camera.image_width)), dtype=torch.float32,
                      device=depth.device)
    d_row = pts[:, 2:, 1:-1] - pts[:, :-2, 1:-1]
    d_col = pts[:, 1:-1, 2:] - pts[:, 1:-1, :-2]
    out[:, 1:-1, 1:-1] = torch.nn.functional.normalize(torch.cross(d_row, d_col, dim=0), dim=0)
    return out


def render(camera, model, depth_ratio: float = 0.0, rasterizer_cls=GaussianRasterizer) -> dict:
    """Same contract as gaussian_renderer.render (gaussian_renderer/__init__.py:11-93).
    `rasterizer_cls` exists so the CPU tests can drive this function with the
    checker's rasterizer; the default (and only product path) is the HIP one, which
    refuses CPU tensors."""
    settings = GaussianRasterizationSettings(
        image_height=int(camera.image_height), image_width=int(camera.image_width), scale_modifier=1.0,
        viewmatrix=camera.world_view_transform, projmatrix=camera.projection_matrix,
        prefiltered=False, debug=False)
    rasterizer = rasterizer_cls(raster_settings=settings)
    means3D = model.get_xyz
    means2D = torch.zeros_like(means3D, dtype=torch.float32)
    radii, allmap = rasterizer(means3D=means3D, means2D=means2D, opacities=model.get_opacity,
                               scales=model.get_scaling, rotations=model.get_rotation, cov3D_precomp=None)
    if allmap.is_cuda and not allmap.requires_grad and rasterizer_cls is GaussianRasterizer:
        # nobody differentiates this call (Mapper.densify, the tracker's target, the logger, meshing): the maps in ONE launch
        return render_maps(camera, settings, allmap, depth_ratio, radii=radii, means2D=means2D)
    return postprocess(camera, allmap, depth_ratio, radii=radii, means2D=means2D)


def render_maps(camera, settings, allmap: torch.Tensor, depth_ratio: float = 0.0, radii=None, means2D=None) -> dict:
    """postprocess() as one HIP launch (sls_render_maps): same dict, same values to rounding (the surface normal is
    taken from sensor-frame differences and rotated once — the reference's world-frame differences carry the pose's
    translation through a cancellation), no autograd graph.  Pinned against postprocess(), which golden G2 pins against
    the reference (tests/test_fused_render.py::test_render_maps_match_postprocess)."""
    if not allmap.is_cuda:
        raise RuntimeError("render_maps needs a ROCm device tensor; there is no CPU fallback")
    dev = allmap.device
    am = allmap.detach()
    if am.dtype != torch.float32 or not am.is_contiguous():
        am = am.float().contiguous()
    _, H, W = am.shape
    ce = get_camera(settings, dev)
    col_h, row_h = half_pixel_tables(ce, dev)
    out = torch.empty((7, H, W), dtype=torch.float32, device=dev)      # rend_normal 3 | surf_depth 1 | surf_normal 3
    _abi.check(_abi.lib().sls_render_maps(H, W, am.data_ptr(), C.addressof(ce.rot9), col_h.data_ptr(), row_h.data_ptr(),
                                          float(depth_ratio), out[0:3].data_ptr(), out[3:4].data_ptr(), out[4:7].data_ptr(),
                                          _stream(dev)), "sls_render_maps")
    res = {"rend_alpha": am[1:2], "rend_normal": out[0:3], "rend_dist": am[6:7], "surf_depth": out[3:4], "surf_normal": out[4:7]}
    if radii is not None:
        res.update({"viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii})
    return res


def postprocess(camera, allmap: torch.Tensor, depth_ratio: float = 0.0, radii=None, means2D=None) -> dict:
    """allmap (7,H,W) -> the five maps (gaussian_renderer/__init__.py:48-93)."""
    alpha = allmap[1:2]
    hit = alpha > 0.0
    safe = torch.where(hit, alpha, torch.ones_like(alpha))
    R_wv = camera.world_view_transform[:3, :3]            # = R_vw^T as a row-vector operator
    normal_w = torch.einsum("chw,dc->dhw", allmap[2:5], R_wv)  # n_world = R_vw^T n_view
    rend_normal = torch.where(hit, normal_w / safe, normal_w)
    depth_expected = torch.where(hit, allmap[0:1] / safe, allmap[0:1])
    depth_median = allmap[5:6]
    surf_depth = depth_expected * (1.0 - depth_ratio) + depth_median * depth_ratio
    surf_normal = depth_to_normal(camera, surf_depth) * alpha
    out = {"rend_alpha": alpha, "rend_normal": rend_normal, "rend_dist": allmap[6:7],
           "surf_depth": surf_depth, "surf_normal": surf_normal}
    if radii is not None:
        out.update({"viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii})
    return out
