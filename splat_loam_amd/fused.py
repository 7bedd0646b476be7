"""Fused consumer of allmap: render()'s post-processing, depth_to_normal and the
per-pixel terms of the mapper loss as ONE C-ABI call (sls_consumer_fwd_bwd, two
small HIP kernels) instead of ~85 tiny torch kernels and several host syncs
(SURVEY.md §8f-1).  Same value and same gradient as
`mapping_loss(postprocess(camera, allmap), ...)` minus the keyframe-independent
scale regulariser, which stays a (sync-free) torch expression on the model.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, get_camera


class _CameraAux:
    """Per-keyframe constants of the loss: half-pixel ray tables on the device,
    contiguous measurement images and the number of valid pixels (one host sync,
    when the keyframe is first used)."""
    __slots__ = ("col_h", "row_h", "gt", "valid", "n_valid", "key")


def camera_aux(camera) -> _CameraAux:
    key = (camera.projection_matrix.data_ptr(), camera.projection_matrix._version, camera.image_depth.data_ptr(),
           camera.image_depth._version, tuple(camera.image_depth.shape), camera.image_depth.dtype,
           camera.image_valid.data_ptr(), camera.image_valid._version)
    aux = getattr(camera, "_sls_aux", None)
    if aux is not None and aux.key == key:
        return aux
    dev = camera.image_depth.device
    H, W = int(camera.image_height), int(camera.image_width)
    settings = GaussianRasterizationSettings(H, W, 1.0, camera.world_view_transform, camera.projection_matrix)
    ce = get_camera(settings, dev)
    col = torch.empty((W, 2), dtype=torch.float32)
    row = torch.empty((H, 2), dtype=torch.float32)
    _abi.check(_abi.lib().sls_ray_tables_at(C.byref(ce.cam), -0.5, -0.5, col.data_ptr(), row.data_ptr()),
               "sls_ray_tables_at")
    aux = _CameraAux()
    aux.col_h, aux.row_h = col.to(dev), row.to(dev)
    aux.gt = camera.image_depth.reshape(H, W).float().contiguous()
    valid = camera.image_valid.reshape(H, W)
    aux.valid = (valid == 1).to(torch.uint8).contiguous()
    aux.n_valid = int(aux.valid.sum().item())
    aux.key = key
    camera._sls_aux = aux
    return aux


class _FusedMappingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, aux, depth_ratio, lambda_normal, lambda_alpha):
        if not allmap.is_cuda:
            raise RuntimeError("fused mapping loss needs a ROCm device tensor; there is no CPU fallback")
        lib = _abi.lib()
        am = allmap.detach()
        if am.dtype != torch.float32 or not am.is_contiguous():
            am = am.float().contiguous()
        _, H, W = am.shape
        dev = am.device
        sums = torch.empty((4,), dtype=torch.float32, device=dev)
        grad = torch.empty_like(am)
        nbytes = int(lib.sls_consumer_scratch_bytes(H, W))
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _abi.check(lib.sls_consumer_fwd_bwd(H, W, am.data_ptr(), aux.gt.data_ptr(), aux.valid.data_ptr(),
                                            aux.col_h.data_ptr(), aux.row_h.data_ptr(), float(depth_ratio),
                                            float(lambda_normal), float(lambda_alpha), int(aux.n_valid),
                                            sums.data_ptr(), grad.data_ptr(), scratch.data_ptr(), nbytes,
                                            torch.cuda.current_stream(dev).cuda_stream), "sls_consumer_fwd_bwd")
        ctx.save_for_backward(grad)
        ctx.sums = sums
        return sums[3]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


def fused_pixel_loss(allmap: torch.Tensor, camera, cfg) -> torch.Tensor:
    """geom_l1 + normal_loss + alpha_loss of slam/mapper.py:174-187 from allmap."""
    return _FusedMappingLoss.apply(allmap, camera_aux(camera), cfg.depth_ratio, cfg.opt_lambda_normal,
                                   cfg.opt_lambda_alpha)


def scale_regulariser(model, cfg) -> torch.Tensor:
    """slam/mapper.py:190-195 without the boolean-mask gather (which costs a host
    sync): sum(relu(max_axis_scale - s_max)) has the same value, and the same
    gradient everywhere except exactly at max_axis_scale == s_max."""
    smax = model.get_scaling.max(dim=1).values
    return cfg.opt_scaling_max_penalty * torch.relu(smax - cfg.opt_scaling_max).sum()


def rasterize(camera, model):
    settings = GaussianRasterizationSettings(
        image_height=int(camera.image_height), image_width=int(camera.image_width), scale_modifier=1.0,
        viewmatrix=camera.world_view_transform, projmatrix=camera.projection_matrix, prefiltered=False, debug=False)
    means3D = model.get_xyz
    return GaussianRasterizer(raster_settings=settings)(
        means3D=means3D, means2D=means3D, opacities=model.get_opacity, scales=model.get_scaling,
        rotations=model.get_rotation, cov3D_precomp=None)


def fused_loss(model, camera, cfg, with_regulariser: bool = True) -> torch.Tensor:
    _, allmap = rasterize(camera, model)
    loss = fused_pixel_loss(allmap, camera, cfg)
    if with_regulariser and cfg.opt_scaling_max_penalty != 0.0:
        loss = loss + scale_regulariser(model, cfg)
    return loss
