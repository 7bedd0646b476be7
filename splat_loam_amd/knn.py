"""distCUDA2: mean squared distance to the 3 nearest neighbours (simple-knn).

Same name and contract as `simple_knn._C.distCUDA2` used at
slam/mapper.py:113-115 and scene/gaussian_model.py:77-81: (M,3) float32 device
tensor -> (M,) float32.  Runs in libsls_hip.so (sls_knn_dist2); no CPU path.
"""
from __future__ import annotations

import torch

from . import _abi


def distCUDA2(points: torch.Tensor, first: int | None = None) -> torch.Tensor:
    """`first` (an extension; the reference's call has one argument): only the first `first` points are queries — the
    neighbours are still searched among all of them — and the result has `first` entries, equal to `distCUDA2(points)[:first]`
    (Mapper.densify keeps exactly those, slam/mapper.py:109-117)."""
    from . import rasterizer
    for hook in list(rasterizer._PENDING_HOOKS.values()):      # (SLS_FUSED_MAPPER=1: Mapper.densify calls this before the first optimize)
        hook()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs a ROCm device tensor (libsls_hip.so); there is no CPU fallback")
    lib = _abi.lib()
    pts = points.detach()
    if pts.dtype != torch.float32:
        pts = pts.float()
    pts = pts.contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("points must be (M,3)")
    M = int(pts.shape[0])
    Mq = M if first is None else int(first)
    if Mq < 0 or Mq > M:
        raise ValueError("first must lie in [0, M]")
    out = torch.empty((Mq,), dtype=torch.float32, device=pts.device)
    if M == 0 or Mq == 0:
        return out
    nbytes = int(lib.sls_knn_scratch_bytes(M))
    scratch = torch.empty((nbytes + 256,), dtype=torch.uint8, device=pts.device)
    base = scratch.data_ptr()
    aligned = (base + 255) & ~255
    st = torch.cuda.current_stream(pts.device).cuda_stream
    if Mq < M:
        _abi.check(lib.sls_knn_dist2_first(M, Mq, pts.data_ptr(), out.data_ptr(), aligned, nbytes, st), "sls_knn_dist2_first")
    else:
        _abi.check(lib.sls_knn_dist2(M, pts.data_ptr(), out.data_ptr(), aligned, nbytes, st), "sls_knn_dist2")
    return out
