"""Import-compatible drop-in for the reference's (un-vendored) CUDA extension:
`from diff_surfel_spherical_rasterization import GaussianRasterizer,
GaussianRasterizationSettings` (gaussian_renderer/__init__.py:5-8) resolves to
the MI355X implementation.

`SLS_FUSED_MAPPER=1` in the environment additionally binds `slam.mapper.Mapper.optimize` to the fused iteration at
run time (splat_loam_amd/fused_mapper.py) — opt-in, no file of the checkout is touched."""
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

from splat_loam_amd import fused_mapper as _fused_mapper

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
_fused_mapper.maybe_install()
