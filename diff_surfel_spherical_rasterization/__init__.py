"""Import-compatible drop-in for the reference's (un-vendored) CUDA extension:
`from diff_surfel_spherical_rasterization import GaussianRasterizer,
GaussianRasterizationSettings` (gaussian_renderer/__init__.py:5-8) resolves to
the MI355X implementation.

`SLS_FUSED_MAPPER=1` in the environment additionally binds `slam.mapper.Mapper.optimize` to the fused iteration at
run time (splat_loam_amd/fused_mapper.py) — opt-in, no file of the checkout is touched.
`SLS_FUSED_RENDER=1` likewise gives `gaussian_renderer.render` a one-launch post-processing wherever autograd is off
(Mapper.densify, the tracker's target, the logger, meshing; splat_loam_amd/fused_render.py).
`SLS_AUTOGRAD_SINGLE_THREAD=1` calls `torch.autograd.set_multithreading_enabled(False)`: backward() then runs on the
calling thread instead of torch's autograd device thread — without it every mapping iteration hands over between two
threads, which costs up to 0.1 ms per iteration on a many-core host (profiles/r05j_autograd_thread.txt).  Process-wide,
hence opt-in."""
import os as _os
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

from splat_loam_amd import fused_mapper as _fused_mapper

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
_fused_mapper.maybe_install()
from splat_loam_amd import fused_render as _fused_render

_fused_render.maybe_install()
if _os.environ.get("SLS_AUTOGRAD_SINGLE_THREAD", "0") == "1":
    import torch as _torch
    _torch.autograd.set_multithreading_enabled(False)
