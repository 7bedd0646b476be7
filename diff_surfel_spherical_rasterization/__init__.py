"""Import-compatible drop-in for the reference's (un-vendored) CUDA extension:
`from diff_surfel_spherical_rasterization import GaussianRasterizer,
GaussianRasterizationSettings` (gaussian_renderer/__init__.py:5-8) resolves to
the MI355X implementation.

`SLS_FUSED_MAPPER=1` in the environment additionally binds `slam.mapper.Mapper.optimize` to the fused iteration at
run time (splat_loam_amd/fused_mapper.py) — opt-in, no file of the checkout is touched.
`SLS_FUSED_RENDER=1` likewise gives `gaussian_renderer.render` a one-launch post-processing wherever autograd is off
(Mapper.densify, the tracker's target, the logger, meshing; splat_loam_amd/fused_render.py).
With ONE visible GPU the first differentiated forward makes backward() run on the calling thread
(`torch.autograd.set_multithreading_enabled(False)`): otherwise every mapping iteration hands over twice between the
caller and torch's autograd worker, up to 0.1 ms per iteration on a many-core host (profiles/r05j_autograd_thread.txt;
splat_loam_amd/rasterizer.py: _autograd_policy).  `SLS_AUTOGRAD_SINGLE_THREAD=0` leaves torch as it is, `=1` forces it."""
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

from splat_loam_amd import fused_mapper as _fused_mapper

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
_fused_mapper.maybe_install()
from splat_loam_amd import fused_render as _fused_render

_fused_render.maybe_install()
