from splat_loam_amd.knn import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
