"""splat_loam_amd — MI355X-native hot path of Splat-LOAM: the differentiable
spherical 2D-Gaussian-surfel rasterizer, simple-knn and fused Adam as
hand-written HIP behind a C-ABI (include/sls_abi.h, libsls_hip.so), plus the
thin host-side mirror of the reference's interface.  See DESIGN.md.
"""
__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "distCUDA2", "FusedAdam", "render"]


def __getattr__(name):
    if name in ("GaussianRasterizationSettings", "GaussianRasterizer"):
        from . import rasterizer
        return getattr(rasterizer, name)
    if name == "distCUDA2":
        from .knn import distCUDA2
        return distCUDA2
    if name == "FusedAdam":
        from .optim import FusedAdam
        return FusedAdam
    if name == "render":
        from .renderer import render
        return render
    raise AttributeError(name)
