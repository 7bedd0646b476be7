"""Import-compatible stand-in for the reference's `gsaligner` extension
(slam/tracker.py:4, utils/config_utils.py:7): `from gsaligner import GSAligner, GSAlignerParams`."""
from splat_loam_amd.aligner import GSAligner, GSAlignerParams  # noqa: F401

__all__ = ["GSAligner", "GSAlignerParams"]
