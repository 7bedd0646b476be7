"""Import-compatible drop-in for INRIA simple-knn: `from simple_knn._C import distCUDA2`
(slam/mapper.py:13, scene/gaussian_model.py:13)."""
