"""Drop-in for the `gsaligner` extension (slam/tracker.py:141-197, utils/config_utils.py:7,95):

    params = GSAlignerParams(); params.image_height = H; params.image_width = W
    aligner = GSAligner(**params.__dict__)
    aligner.set_reference(depth, points, projmatrix)     # rendered keyframe
    aligner.set_query(depth, points, projmatrix)         # new frame
    keyframe_T_frame, fitness, _ = aligner.align(iguess) # 4x4 torch tensor, float, info

`depth` is (1,H,W) or (H,W), `points` is (H*W,3) in the frame's own coordinates
(`depth_to_points(..., transform_in_world=False)`), `projmatrix` the camera's 4x4
projection matrix (K = projmatrix[:3,:3]^T, scene/cameras.py:47-50).

The reference's implementation is an un-vendored CUDA/Eigen submodule; the algorithm
behind this interface is this repository's own (HIP kernels in
csrc/sls_aligner.hip): projective association on the spherical range image,
point-to-plane + range-image residuals with Huber weights, Gauss-Newton on SE(3).
Device only: there is no CPU path in the product.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
import warnings

import torch

from . import _abi


@dataclasses.dataclass
class GSAlignerParams:
    """Parameter bag as the reference uses it: default-constructed, fields assigned, then
    `GSAligner(**params.__dict__)` (slam/tracker.py:146-160); a dataclass, because the reference also
    names it as a field type of its OmegaConf structured config (utils/config_utils.py:95:
    `gsaligner: Optional[GSAlignerParams]`), which accepts dataclass / attrs types only."""
    image_height: int = 64
    image_width: int = 1024
    num_iterations: int = 15
    min_inliers: int = 64
    max_distance: float = 1.0              # association gate (m)
    max_angle_deg: float = 80.0            # reference normal vs viewing ray
    huber_delta: float = 0.10              # point-to-plane residual (m)
    range_weight: float = 0.25             # range-image term (0 switches it off)
    range_huber: float = 0.30
    depth_min: float = 0.5
    depth_max: float = 100.0
    damping: float = 1e-6


def _f32(t, dev):
    t = t.detach()
    if not t.is_cuda:
        raise RuntimeError("GSAligner needs ROCm device tensors; there is no CPU fallback")
    return t.to(device=dev, dtype=torch.float32).contiguous()


class GSAligner:
    def __init__(self, **kw):
        # a YAML written for the original gsaligner may carry fields this implementation does not have
        # (its parameter list is not in the reference tree): they are ignored with a warning, not fatal
        known = {f.name for f in dataclasses.fields(GSAlignerParams)}
        unknown = sorted(set(kw) - known)
        if unknown:
            warnings.warn(f"GSAligner: ignoring unknown parameter(s) {unknown}", stacklevel=2)
        self.params = GSAlignerParams(**{k: v for k, v in kw.items() if k in known})
        self.H, self.W = int(self.params.image_height), int(self.params.image_width)
        self._ref = None
        self._query = None
        self._cam = None
        self._ws = None
        self.last = None

    def _native_params(self):
        p, n = self.params, _abi.SlsAlignerParams()
        n.num_iterations, n.min_inliers = int(p.num_iterations), int(p.min_inliers)
        n.max_distance, n.min_cos_angle = float(p.max_distance), math.cos(math.radians(float(p.max_angle_deg)))
        n.huber_delta, n.range_weight, n.range_huber = float(p.huber_delta), float(p.range_weight), float(p.range_huber)
        n.depth_min, n.depth_max, n.damping = float(p.depth_min), float(p.depth_max), float(p.damping)
        return n

    def _camera(self, projmatrix):
        cam = _abi.SlsCamera()
        proj = projmatrix.detach().to("cpu", torch.float32).contiguous()
        view = torch.eye(4, dtype=torch.float32)
        _abi.check(_abi.lib().sls_camera_from_matrices(view.data_ptr(), proj.data_ptr(), self.H, self.W,
                                                       C.c_float(1.0), C.byref(cam)), "sls_camera_from_matrices")
        return cam

    def _frame(self, depth, points):
        dev = depth.device
        d = _f32(depth, dev).reshape(-1)
        p = _f32(points, dev).reshape(-1, 3)
        if d.numel() != self.H * self.W or p.shape[0] != self.H * self.W:
            raise ValueError(f"expected {self.H}x{self.W} depth and {self.H * self.W} points")
        return d, p

    def set_reference(self, depth, points, projmatrix) -> None:
        d, p = self._frame(depth, points)
        self._cam = self._camera(projmatrix)
        n = torch.empty_like(p)
        st = torch.cuda.current_stream(d.device).cuda_stream
        _abi.check(_abi.lib().sls_aligner_normals(C.byref(self._cam), d.data_ptr(), p.data_ptr(),
                                                  C.c_float(float(self.params.depth_min)), n.data_ptr(), st),
                   "sls_aligner_normals")
        self._ref = (d, p, n)

    def set_query(self, depth, points, projmatrix) -> None:
        self._query = self._frame(depth, points)
        if self._cam is None:
            self._cam = self._camera(projmatrix)

    def _workspace(self, dev):
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.zeros((int(_abi.lib().sls_aligner_workspace_bytes()) + 64,), dtype=torch.uint8, device=dev)
        return self._ws

    def linearize(self, T) -> torch.Tensor:
        """The 6x6 system at pose T (tests): 32 doubles, H upper triangle | b | chi2 | inliers | valid."""
        (rd, rp, rn), (qd, qp) = self._ref, self._query
        dev = rd.device
        sys = torch.zeros((32,), dtype=torch.float64, device=dev)
        Th = T.detach().to("cpu", torch.float32).contiguous()
        prm = self._native_params()
        _abi.check(_abi.lib().sls_aligner_linearize(C.byref(self._cam), C.byref(prm), rd.data_ptr(), rp.data_ptr(),
                                                    rn.data_ptr(), qd.data_ptr(), qp.data_ptr(), Th.data_ptr(),
                                                    self._workspace(dev).data_ptr(), sys.data_ptr(),
                                                    torch.cuda.current_stream(dev).cuda_stream), "sls_aligner_linearize")
        return sys

    def align(self, iguess):
        """-> (ref_T_query 4x4 float32 tensor on the input's device, fitness, info dict)."""
        if self._ref is None or self._query is None:
            raise RuntimeError("set_reference() and set_query() first")
        (rd, rp, rn), (qd, qp) = self._ref, self._query
        dev = rd.device
        Th = iguess.detach().to("cpu", torch.float32).contiguous()
        res = torch.zeros((C.sizeof(_abi.SlsAlignerResult) // 4,), dtype=torch.int32, device=dev)
        prm = self._native_params()
        _abi.check(_abi.lib().sls_aligner_align(C.byref(self._cam), C.byref(prm), rd.data_ptr(), rp.data_ptr(),
                                                rn.data_ptr(), qd.data_ptr(), qp.data_ptr(), Th.data_ptr(),
                                                self._workspace(dev).data_ptr(), res.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream), "sls_aligner_align")
        h = res.cpu()                                   # the one sync of an alignment
        f = h.view(torch.float32)
        T = torch.eye(4, dtype=torch.float32)
        T[:3, :4] = f[:12].reshape(3, 4)
        info = {"chi2": float(f[13]), "last_step": float(f[14]), "inliers": int(h[15]), "valid_query": int(h[16]),
                "iterations": int(h[17])}
        self.last = info
        return T.to(iguess.device), float(f[12]), info
