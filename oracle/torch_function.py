"""CPU autograd wrapper around the C checker, exposing the rasterizer interface
(TEST INFRASTRUCTURE, "parity unpinned" — see sls_oracle.c).  Used (a) by
tools/make_golden.py to run the REFERENCE's render()/Mapper.optimize in the build
container on top of a working rasterizer, and (b) by the CPU tests to drive this
repo's renderer/mapping code without a GPU.  The product never imports it.
"""
from __future__ import annotations

from typing import NamedTuple

import numpy as np
import torch

from .oracle import Oracle

_ORACLE = {}


def _oracle(dtype):
    if dtype not in _ORACLE:
        _ORACLE[dtype] = Oracle(dtype)
    return _ORACLE[dtype]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


TILE = (16, 16)
BACKWARD_THREADS = 1     # threads of the checker's tile backward under autograd (a test at full size raises it)


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, scales, rotations, cov3D_precomp, settings):
        np_dtype = np.float64 if means3D.dtype == torch.float64 else np.float32
        o = _oracle(np_dtype)
        cam = o.camera(settings.image_height, settings.image_width, settings.viewmatrix.detach().cpu().numpy(),
                       settings.projmatrix.detach().cpu().numpy(), settings.scale_modifier, tile=TILE)
        st = o.forward(cam, means3D.detach().numpy(), scales.detach().numpy(), rotations.detach().numpy(),
                       opacities.detach().numpy())
        ctx.st, ctx.o, ctx.dtype = st, o, means3D.dtype
        radii = torch.from_numpy(st["radii"].copy())
        ctx.mark_non_differentiable(radii)
        return radii, torch.from_numpy(st["allmap"].copy())

    @staticmethod
    def backward(ctx, _g_radii, g_allmap):
        b = ctx.o.backward(ctx.st, g_allmap.detach().numpy(), threads=BACKWARD_THREADS, want_abs=False)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.dtype)
        return t(b["dmeans"]), None, t(b["dopac"]), t(b["dscales"]), t(b["drots"]), None, None


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, scales=None, rotations=None, cov3D_precomp=None):
        return _OracleRasterize.apply(means3D, means2D, opacities, scales, rotations, cov3D_precomp,
                                      self.raster_settings)
