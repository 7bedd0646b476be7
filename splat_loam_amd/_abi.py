"""ctypes binding of libsls_hip.so (include/sls_abi.h).

This is the stub a Splat-LOAM maintainer adds to reach the HIP library
(INTEGRATION.md shows it in context).  There is NO fallback: if the shared
library is missing the import of any product entry point raises, and device
entry points refuse CPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsls_hip.so")

SLS_OK = 0


class SlsCamera(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("wrap", C.c_int32), ("tile_cull_min", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("scale_modifier", C.c_float), ("near_cut", C.c_float), ("far_cut", C.c_float), ("flags", C.c_uint32),
        ("Rvw", C.c_float * 9), ("tvw", C.c_float * 3),
        ("pix_offset", C.c_float * 2),
    ]


class SlsMappingConfig(C.Structure):
    _fields_ = [
        ("lambda_alpha", C.c_float), ("lambda_normal", C.c_float), ("scaling_max", C.c_float),
        ("scaling_max_penalty", C.c_float), ("depth_ratio", C.c_float),
        ("lr_xyz", C.c_float), ("lr_opacity", C.c_float), ("lr_scaling", C.c_float), ("lr_rotation", C.c_float),
        ("apply_adam", C.c_int32), ("reuse_depth_order", C.c_int32), ("keep_grads", C.c_int32), ("workspace_ready", C.c_int32),
        ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
        ("depth_order", C.c_void_p),
        ("status_mirror", C.c_void_p),
        ("void_flags_out", C.c_void_p),
        ("grad_chunk", C.c_uint32), ("grad_ranks", C.c_uint32),
        ("deterministic", C.c_int32), ("block_masks", C.c_int32),
        ("grad_bitmap", C.c_void_p),
        ("det_prev", C.c_void_p),
        ("phase", C.c_int32), ("reserved", C.c_int32),
        ("block_order", C.c_void_p),
        ("union_bitmap", C.c_void_p), ("union_prefix", C.c_void_p), ("grad_compact", C.c_void_p),
        ("grad_compact_index", C.c_void_p), ("grad_compact_capacity", C.c_uint32), ("reserved2", C.c_uint32),
    ]


class SlsAlignerParams(C.Structure):
    _fields_ = [("num_iterations", C.c_int32), ("min_inliers", C.c_int32), ("max_distance", C.c_float),
                ("min_cos_angle", C.c_float), ("huber_delta", C.c_float), ("range_weight", C.c_float),
                ("range_huber", C.c_float), ("depth_min", C.c_float), ("depth_max", C.c_float), ("damping", C.c_float)]


class SlsAlignerResult(C.Structure):
    _fields_ = [("pose", C.c_float * 12), ("fitness", C.c_float), ("chi2", C.c_float), ("last_step", C.c_float),
                ("inliers", C.c_int32), ("valid_query", C.c_int32), ("iterations", C.c_int32)]


class SlsMappingStatus(C.Structure):
    _fields_ = [("R", C.c_uint32), ("overflow", C.c_uint32), ("loss_sums", C.c_float * 4),
                ("loss_reg", C.c_float), ("exchange_count", C.c_uint32)]


class SlsAdamGroup(C.Structure):
    _fields_ = [
        ("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("numel", C.c_int64), ("lr", C.c_float), ("pad", C.c_float),
    ]


_VP = C.c_void_p
_PROTOS = {
    # name: (restype, argtypes)
    "sls_last_error": (C.c_char_p, []),
    "sls_version": (C.c_int, []),
    "sls_tile_w": (C.c_int, []),
    "sls_tile_h": (C.c_int, []),
    "sls_rec_stride": (C.c_int, []),
    "sls_grec_stride": (C.c_int, []),
    "sls_camera_from_matrices": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.POINTER(SlsCamera)]),
    "sls_ray_tables": (C.c_int, [C.POINTER(SlsCamera), _VP, _VP]),
    "sls_ray_tables_at": (C.c_int, [C.POINTER(SlsCamera), C.c_float, C.c_float, _VP, _VP]),
    "sls_render_maps": (C.c_int, [C.c_int, C.c_int, _VP, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _VP]),
    "sls_densify_rows": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_VP] * 10),
    "sls_densify_weights": (C.c_int, [C.c_int, C.c_int, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP]),
    "sls_consumer_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sls_consumer_fwd_bwd": (C.c_int, [C.c_int, C.c_int] + [_VP] * 5 + [C.c_float] * 3 + [C.c_int, _VP, _VP, _VP,
                                                                                         C.c_size_t, _VP]),
    "sls_stage1_scratch_bytes": (C.c_size_t, [C.c_int]),
    "sls_forward_stage1": (C.c_int, [C.POINTER(SlsCamera), C.c_int] + [_VP] * 16 + [_VP, C.c_size_t, _VP]),
    "sls_sort_scratch_bytes": (C.c_size_t, [C.c_uint64]),
    "sls_forward_stage2": (C.c_int, [C.POINTER(SlsCamera), C.c_int, C.c_uint64] + [_VP] * 13 +
                           [_VP, C.c_size_t, C.POINTER(C.c_int), _VP, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)] +
                           [_VP] * 8 + [C.POINTER(C.c_int), _VP]),
    "sls_block_mask_bytes": (C.c_size_t, [C.c_uint64, C.c_int, C.c_int]),
    "sls_mapping_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_uint64]),
    "sls_block_order_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sls_mapping_workspace_bytes_cfg": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(SlsMappingConfig)]),
    "sls_mapping_step": (C.c_int, [C.POINTER(SlsCamera), C.c_int] + [_VP] * 7 + [C.c_int64, _VP, _VP, C.c_int] +
                         [_VP] * 4 + [C.POINTER(SlsMappingConfig), C.c_uint64, _VP, C.c_size_t, _VP,
                                      C.POINTER(C.c_void_p), _VP]),
    "sls_backward": (C.c_int, [C.POINTER(SlsCamera), C.c_int, C.c_uint64] + [_VP] * 7 + [C.c_int] + [_VP] * 11 +
                     [C.c_int, _VP]),
    "sls_forward_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_uint64]),
    "sls_forward_ws": (C.c_int, [C.POINTER(SlsCamera), C.c_int] + [_VP] * 6 + [C.c_uint64, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP,
                                 _VP, C.c_size_t, _VP, _VP, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), _VP]),
    "sls_backward_ws": (C.c_int, [C.POINTER(SlsCamera), C.c_int] + [_VP] * 7 + [C.c_uint64, _VP, C.c_size_t, _VP, C.c_int, C.c_int] +
                        [_VP] * 6),
    "sls_backward_det_scratch_bytes": (C.c_size_t, [C.c_int]),
    "sls_backward_det": (C.c_int, [C.POINTER(SlsCamera), C.c_int, C.c_uint64] + [_VP] * 7 + [C.c_int] + [_VP] * 10 +
                         [C.c_int, _VP, C.c_size_t, _VP]),
    "sls_adam_step": (C.c_int, [C.POINTER(SlsAdamGroup), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int64, _VP]),
    "sls_adam_step_guarded": (C.c_int, [C.POINTER(SlsAdamGroup), C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_int64, _VP, _VP]),
    "sls_adam_step_reduced": (C.c_int, [C.POINTER(SlsAdamGroup), C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_int64, _VP, _VP, _VP, _VP]),
    "sls_grad_bitmap_words": (C.c_size_t, [C.c_int]),
    "sls_grad_union": (C.c_int, [C.c_int, _VP, C.c_int, _VP, C.c_uint32, _VP, _VP, _VP]),
    "sls_adam_step_union": (C.c_int, [C.c_int] + [_VP] * 6 + [C.c_uint32, _VP, _VP] + [C.c_float] * 4 + [C.c_double] * 3 + [C.c_int64, _VP, _VP, _VP]),
    "sls_grad_compact": (C.c_int, [C.c_int, _VP, C.c_int, _VP, _VP, _VP, C.c_uint32, _VP, _VP, _VP]),
    "sls_adam_step_sparse": (C.c_int, [C.c_int] + [_VP] * 9 + [C.c_float] * 4 + [C.c_double] * 3 + [C.c_int64, C.c_int, _VP, _VP, _VP]),
    "sls_projector_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sls_projector_prepare": (C.c_int, [C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "sls_projector_intrinsics": (C.c_int, [C.c_int, _VP, C.c_int, C.c_int, C.c_float, _VP, _VP, C.c_size_t, _VP]),
    "sls_projector_project": (C.c_int, [C.c_int, _VP, _VP, C.c_int, C.c_int, C.c_float, C.c_float, _VP, _VP, _VP, _VP,
                                        _VP, C.c_size_t, _VP]),
    "sls_knn_scratch_bytes": (C.c_size_t, [C.c_int]),
    "sls_knn_dist2": (C.c_int, [C.c_int, _VP, _VP, _VP, C.c_size_t, _VP]),
    "sls_wait_status_mirror": (C.c_int, [_VP, C.c_uint32, _VP]),
    "sls_knn_dist2_first": (C.c_int, [C.c_int, C.c_int, _VP, _VP, _VP, C.c_size_t, _VP]),
    "sls_mark_visible": (C.c_int, [C.POINTER(SlsCamera), C.c_int, _VP, _VP, _VP]),
    "sls_aligner_workspace_bytes": (C.c_size_t, []),
    "sls_aligner_normals": (C.c_int, [C.POINTER(SlsCamera), _VP, _VP, C.c_float, _VP, _VP]),
    "sls_aligner_linearize": (C.c_int, [C.POINTER(SlsCamera), C.POINTER(SlsAlignerParams)] + [_VP] * 8 + [_VP]),
    "sls_aligner_align": (C.c_int, [C.POINTER(SlsCamera), C.POINTER(SlsAlignerParams)] + [_VP] * 8 + [_VP]),
    "sls_timing_slots": (C.c_int, []),
    "sls_timing_name": (C.c_char_p, [C.c_int]),
    "sls_timing_enable": (C.c_int, [C.c_int]),
    "sls_timing_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "sls_debug_wave_cycles": (C.c_int, [_VP, _VP]),
    "sls_debug_variant": (C.c_int, [C.c_int, C.c_int]),
    "sls_selftest": (C.c_int, [_VP]),
}

EXPORTS = tuple(_PROTOS)

_lib = None


def lib():
    """Load libsls_hip.so once.  torch must be imported first so that the HIP
    runtime the process already uses (torch's libamdhip64) is the one our
    kernels launch on."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        import torch  # noqa: F401  (loads libamdhip64 first)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != SLS_OK:
        msg = lib().sls_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def tile_size() -> tuple[int, int]:
    l = lib()
    return int(l.sls_tile_w()), int(l.sls_tile_h())
