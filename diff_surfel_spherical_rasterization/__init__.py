"""Import-compatible drop-in for the reference's (un-vendored) CUDA extension:
`from diff_surfel_spherical_rasterization import GaussianRasterizer,
GaussianRasterizationSettings` (gaussian_renderer/__init__.py:5-8) resolves to
the MI355X implementation."""
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
