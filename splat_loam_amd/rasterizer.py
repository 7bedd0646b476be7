"""Host-side mirror of the reference's rasterizer interface.

`GaussianRasterizationSettings` and `GaussianRasterizer` keep the names,
argument meaning and call pattern of diff_surfel_spherical_rasterization as
used by gaussian_renderer/__init__.py:16-47 (settings built with exactly these
seven keywords; rasterizer built with `raster_settings=` and called with
`means3D, means2D, opacities, scales, rotations, cov3D_precomp`, returning
`(radii, allmap)`), so gaussian_renderer.render(), slam/mapper.py and
slam/tracker.py run unmodified.  All arithmetic happens in libsls_hip.so
(include/sls_abi.h); torch supplies device memory, the stream and autograd
plumbing only.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _abi


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    scale_modifier: float
    viewmatrix: torch.Tensor   # (4,4) = inv(world_T_lidar)^T   scene/cameras.py:43-46
    projmatrix: torch.Tensor   # (4,4), [:3,:3] = K^T            scene/cameras.py:47-50
    prefiltered: bool = False
    debug: bool = False
    # extension (not a reference field; the tree builds the settings with the seven keywords above): D1 as a
    # parameter.  Pixel (c, r) has image coordinate (c + ox, r + oy).  None -> the process default (pix_offset()):
    # (0, 0), the lineage convention; (-0.5, -0.5) is the convention of the reference's own back-projection and
    # projector (utils/graphic_utils.py:46-49), under which a rendered keyframe registers without the half-pixel bias
    pix_offset: Optional[tuple] = None
    # extension: threshold of D10's tile-level footprint test (SlsCamera.tile_cull_min: 0 default, 1 off, k >= 2)
    tile_cull_min: Optional[int] = None
    # extension: True = the caller neither reads allmap's median / distortion planes (5, 6) nor sends a gradient into
    # them (the reference's mapper and tracker at depth_ratio = 0): they come back as zeros and the tile kernels do
    # not track them — the kernels sls_mapping_step runs.  None -> the process default (SLS_LEAN_ALLMAP=1), False
    lean_allmap: Optional[bool] = None


_PIX_OFFSET = None


def pix_offset() -> tuple:
    """Process default of D1's pixel-centre offset: set_pix_offset(), else SLS_PIX_OFFSET="ox,oy", else (0, 0)."""
    global _PIX_OFFSET
    if _PIX_OFFSET is None:
        env = os.environ.get("SLS_PIX_OFFSET", "")
        parsed = tuple(float(v) for v in env.split(",")) if env else (0.0, 0.0)
        if len(parsed) != 2:                     # (validated before it is cached: a malformed value raises every time)
            raise ValueError("SLS_PIX_OFFSET must be 'ox,oy'")
        _PIX_OFFSET = parsed
    return _PIX_OFFSET


def set_pix_offset(ox: float, oy: float) -> None:
    """For a caller that cannot touch the settings (gaussian_renderer/__init__.py builds them itself)."""
    global _PIX_OFFSET
    _PIX_OFFSET = (float(ox), float(oy))


# ---------------------------------------------------------------------------
# camera cache: the two 4x4 matrices live on the device; copying them to the
# host costs a sync, so do it once per (tensor, version).  Entries hold strong
# references to the tensors, which keeps their addresses from being reused
# while the entry is alive.
# ---------------------------------------------------------------------------
class _CamEntry:
    __slots__ = ("cam", "col_cs", "row_cs", "view_ref", "proj_ref", "rot9", "half", "c2w")


_CAM_CACHE: "OrderedDict[tuple, _CamEntry]" = OrderedDict()
_CAM_CACHE_MAX = 64
_TABLE_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()


def _ray_tables(cam: _abi.SlsCamera, device: torch.device):
    key = (cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, cam.pix_offset[0], cam.pix_offset[1], str(device))
    hit = _TABLE_CACHE.get(key)
    if hit is not None:
        _TABLE_CACHE.move_to_end(key)
        return hit
    col = torch.empty((cam.W, 2), dtype=torch.float32)
    row = torch.empty((cam.H, 2), dtype=torch.float32)
    _abi.check(_abi.lib().sls_ray_tables(C.byref(cam), col.data_ptr(), row.data_ptr()), "sls_ray_tables")
    out = (col.to(device), row.to(device))
    _TABLE_CACHE[key] = out
    while len(_TABLE_CACHE) > 16:
        _TABLE_CACHE.popitem(last=False)
    return out


def get_camera(settings: GaussianRasterizationSettings, device: torch.device) -> _CamEntry:
    v, p = settings.viewmatrix, settings.projmatrix
    off = getattr(settings, "pix_offset", None)
    off = pix_offset() if off is None else (float(off[0]), float(off[1]))
    tcm = getattr(settings, "tile_cull_min", None)
    if tcm is None:        # SLS_TILE_CULL_MIN=k: D10's threshold for every camera built here (0 / 1: off, the default)
        tcm = int(os.environ.get("SLS_TILE_CULL_MIN", "0"))
    lean = getattr(settings, "lean_allmap", None)
    if lean is None:
        lean = os.environ.get("SLS_LEAN_ALLMAP", "0") == "1"
    key = (v.data_ptr(), v._version, p.data_ptr(), p._version, int(settings.image_height),
           int(settings.image_width), float(settings.scale_modifier), off, int(tcm), bool(lean), str(device))
    hit = _CAM_CACHE.get(key)
    if hit is not None:
        _CAM_CACHE.move_to_end(key)
        return hit
    vh = v.detach().to("cpu", torch.float32).contiguous()
    ph = p.detach().to("cpu", torch.float32).contiguous()
    if vh.shape != (4, 4) or ph.shape != (4, 4):
        raise ValueError("viewmatrix and projmatrix must be 4x4")
    e = _CamEntry()
    e.cam = _abi.SlsCamera()
    _abi.check(_abi.lib().sls_camera_from_matrices(vh.data_ptr(), ph.data_ptr(), int(settings.image_height),
                                                   int(settings.image_width), float(settings.scale_modifier),
                                                   C.byref(e.cam)), "sls_camera_from_matrices")
    e.cam.pix_offset[0], e.cam.pix_offset[1] = off
    e.cam.tile_cull_min = int(tcm)
    e.cam.flags = 1 if lean else 0            # SLS_CAM_LEAN_ALLMAP
    e.col_cs, e.row_cs = _ray_tables(e.cam, device)
    e.view_ref, e.proj_ref = v, p
    e.rot9 = (C.c_float * 9)(*[float(x) for x in vh[:3, :3].reshape(-1)])   # view frame -> world, as a matrix (render_maps)
    e.half = None                                                              # the (c - .5, r - .5) ray tables, made when asked for
    e.c2w = None                                                               # inv(view^T) on the device (fused_mapper's densify), likewise
    _CAM_CACHE[key] = e
    while len(_CAM_CACHE) > _CAM_CACHE_MAX:
        _CAM_CACHE.popitem(last=False)
    return e


def half_pixel_tables(e: _CamEntry, device: torch.device):
    """The ray tables of the consumer's back-projection convention, pixel (c, r) at (c - 0.5, r - 0.5)
    (utils/graphic_utils.py:46-49), cached with the camera entry."""
    if e.half is None:
        col = torch.empty((e.cam.W, 2), dtype=torch.float32)
        row = torch.empty((e.cam.H, 2), dtype=torch.float32)
        _abi.check(_abi.lib().sls_ray_tables_at(C.byref(e.cam), -0.5, -0.5, col.data_ptr(), row.data_ptr()), "sls_ray_tables_at")
        e.half = (col.to(device), row.to(device))
    return e.half


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _need_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the spherical surfel rasterizer runs only on a ROCm device "
            "(libsls_hip.so); there is no CPU fallback")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_DT = {"f32": (torch.float32, 4), "i32": (torch.int32, 4), "i64": (torch.int64, 8), "u8": (torch.uint8, 1)}


class _Arena:
    """ONE device allocation carved into the named buffers of a forward stage: a forward needs some twenty-five
    buffers, and twenty-five torch.empty calls were half of the drop-in path's host time at the mapper's real sizes.
    Views are made on demand (the tests look at everything, the mapper at two of them)."""
    __slots__ = ("base", "spec")

    def __init__(self, dev, spec):
        off, table = 0, {}
        for name, kind, shape in spec:
            dt, size = _DT[kind]
            n = 1
            for d in shape:
                n *= int(d)
            table[name] = (off, n * size, dt, tuple(int(d) for d in shape))
            off += (n * size + 255) & ~255
        self.spec = table
        self.base = torch.empty((max(off, 256),), dtype=torch.uint8, device=dev)

    def ptr(self, name) -> int:
        return self.base.data_ptr() + self.spec[name][0]

    def view(self, name) -> torch.Tensor:
        off, nbytes, dt, shape = self.spec[name]
        return self.base[off:off + nbytes].view(dt).view(shape)


class ForwardState:
    """Everything the backward (and the tests) need from one forward: two arenas (stage 1, stage 2) and the scalars.
    Buffers are reached as attributes (`st.radii`, `st.allmap`, ...), views created when asked for."""
    __slots__ = ("cam", "N", "R", "a1", "a2", "keys", "_vals", "vals_ptr", "vals_stride", "block_masks_shape", "_views")

    def __getattr__(self, name):
        # (only reached for names that are not slots: the arenas' buffers)
        views = object.__getattribute__(self, "_views")
        if name in views:
            return views[name]
        for arena in (object.__getattribute__(self, "a1"), object.__getattribute__(self, "a2")):
            if arena is not None and name in arena.spec:
                views[name] = arena.view(name)
                return views[name]
        if name == "vals":           # the sorted list of surfel indices
            return sorted_list(self)
        if name == "block_masks":
            return None
        raise AttributeError(name)


def list_pairs_mode() -> int:
    """Process default of sls_forward_stage2's `list_pairs` (SLS_BLOCK_MASKS, the engine's switch): 0 = the sorted
    list comes as (surfel, block mask) pairs — and the forward runs its dense rounds — where a tile's list averages
    1500 entries or more, 1 = whenever possible, 2 = never."""
    return int(os.environ.get("SLS_BLOCK_MASKS", "0"))


def rasterize_forward(settings: GaussianRasterizationSettings, means3D, opacities, scales, rotations,
                      want_keys: bool = False, list_pairs: Optional[int] = None) -> ForwardState:
    """preprocess -> depth order -> scan -> [host reads R] -> binning -> stable sort by tile -> ranges -> render.
    `want_keys` (tests): also the 64-bit keys (tile << 32 | depth bits) the list is ordered by, derived here from the
    list, the ranges and the depths, so that the kernels under test are the production ones."""
    for name, t in (("means3D", means3D), ("opacities", opacities), ("scales", scales), ("rotations", rotations)):
        _need_cuda(t, name)
    lib = _abi.lib()
    dev = means3D.device
    means3D, opacities, scales, rotations = map(_f32c, (means3D, opacities, scales, rotations))
    N = int(means3D.shape[0])
    if means3D.shape != (N, 3) or scales.shape != (N, 2) or rotations.shape != (N, 4) or opacities.numel() != N:
        raise ValueError("expected means3D (N,3), scales (N,2), rotations (N,4), opacities (N,1)")
    ce = get_camera(settings, dev)
    cam = ce.cam
    H, W = cam.H, cam.W
    tw, th = _abi.tile_size()
    T = ((W + tw - 1) // tw) * ((H + th - 1) // th)
    st = _stream(dev)
    debug = bool(settings.debug)

    def dbg():
        # debug=True: surface asynchronous kernel faults at the stage that caused them
        if debug:
            torch.cuda.synchronize(dev)

    s = ForwardState()
    s.cam, s.N, s.a2, s.keys, s._views = ce, N, None, None, {}
    # (uint32 buffers are carried as int32 tensors)
    sb = int(lib.sls_stage1_scratch_bytes(N))
    # (radii and allmap are what autograd hands to the caller, who may write into them in place: tensors of their own,
    #  never views of an arena — torch forbids in-place changes of a view made inside a custom Function)
    radii = s._views["radii"] = torch.empty((N,), dtype=torch.int32, device=dev)
    a1 = s.a1 = _Arena(dev, (("rec", "f32", (N, lib.sls_rec_stride())), ("rect", "i32", (N, 4)),
                             ("tiles", "i32", (N,)), ("tmask", "i64", (N,)),      # D10: which tiles of the rectangle are emitted
                             ("sbox", "i32", (N,)),                                # the surfels' block boxes (for the list's block masks)
                             ("depth", "f32", (N,)), ("order", "i32", (N,)), ("offsets", "i32", (N,)),
                             ("total", "i32", (4,)), ("scratch1", "u8", (max(sb, 4),))))
    _abi.check(lib.sls_forward_stage1(C.byref(cam), N, means3D.data_ptr(), scales.data_ptr(), rotations.data_ptr(),
                                      opacities.data_ptr(), ce.col_cs.data_ptr(), ce.row_cs.data_ptr(),
                                      a1.ptr("rec"), radii.data_ptr(), a1.ptr("rect"), a1.ptr("tiles"), a1.ptr("tmask"),
                                      a1.ptr("sbox"), a1.ptr("depth"), a1.ptr("order"), a1.ptr("offsets"),
                                      a1.ptr("total"), a1.ptr("scratch1"), sb, st), "sls_forward_stage1")
    dbg()
    R = int(a1.view("total")[0].item()) & 0xFFFFFFFF   # the one device->host sync of the forward (as in the lineage)
    s.R = R
    Ra = max(R, 1)
    ssb = int(lib.sls_sort_scratch_bytes(R))
    # forward -> backward hand-over: 128 B per instance of capacity (room for every list entry in each of a tile's 16
    # pixel blocks; only what contributes is written).  (Without it — block_masks = NULL at the C-ABI — the backward
    # culls the tiles' lists itself, about a third slower at the mapper's sizes.)
    hand_over = True
    allmap = s._views["allmap"] = torch.empty((7, H, W), dtype=torch.float32, device=dev)
    spec2 = [("ranges", "i32", (T, 2)), ("pix_state", "f32", (H * W, 4)),
             ("pix_contrib", "i32", (H * W, 2)), ("tile_consumed", "i32", (T,)),
             ("keys_a", "i32", (Ra,)), ("keys_b", "i32", (Ra,)), ("vals_a", "i32", (Ra,)), ("vals_b", "i32", (Ra,)),
             ("sort_scratch", "u8", (max(ssb, 4),))]
    if hand_over:
        spec2.append(("block_masks", "i64", (int(lib.sls_block_mask_bytes(R, H, W)) // 8,)))
    a2 = s.a2 = _Arena(dev, spec2)
    in_tmp, stride, shape = C.c_int(0), C.c_int(1), C.c_int(0)
    lst = C.c_void_p(0)
    _abi.check(lib.sls_forward_stage2(C.byref(cam), N, R, a1.ptr("rec"), a1.ptr("rect"), a1.ptr("tiles"),
                                      a1.ptr("tmask"), a1.ptr("sbox"), a1.ptr("depth"), a1.ptr("order"),
                                      a1.ptr("offsets"), a1.ptr("total"),
                                      a2.ptr("keys_a"), a2.ptr("vals_a"), a2.ptr("keys_b"), a2.ptr("vals_b"),
                                      a2.ptr("sort_scratch"), ssb, C.byref(in_tmp), None,
                                      list_pairs_mode() if list_pairs is None else int(list_pairs),
                                      C.byref(lst), C.byref(stride), a2.ptr("ranges"),
                                      ce.col_cs.data_ptr(), ce.row_cs.data_ptr(), allmap.data_ptr(), a2.ptr("pix_state"),
                                      a2.ptr("pix_contrib"), a2.ptr("tile_consumed"),
                                      a2.ptr("block_masks") if hand_over else None, C.byref(shape), st),
               "sls_forward_stage2")
    dbg()
    s.vals_stride, s.block_masks_shape = int(stride.value), int(shape.value)
    s.vals_ptr = int(lst.value) if R > 0 else a2.ptr("vals_a")
    s._vals = None
    if want_keys:
        _list_and_keys(s, T)
    return s


def sorted_list(s: ForwardState) -> torch.Tensor:
    """The sorted list of surfel indices (R entries) as a tensor: a plain array, or a strided view of the tile sort's
    (surfel, block mask) pairs inside the sort scratch."""
    if s._vals is None:
        Ra = max(s.R, 1)
        off = s.vals_ptr - s.a2.base.data_ptr()
        v = s.a2.base[off:off + 4 * s.vals_stride * Ra].view(torch.int32).view(Ra, s.vals_stride)[:, 0]
        s._vals = v[:0] if s.R == 0 else v
    return s._vals


def _list_and_keys(s: ForwardState, T: int) -> None:
    """(tile << 32 | depth bits) of every list entry, from the ranges, the list and the depths"""
    vals = sorted_list(s)
    dev = vals.device
    cnt = (s.ranges[:, 1].long() - s.ranges[:, 0].long()) & 0xFFFFFFFF
    tile = torch.repeat_interleave(torch.arange(T, device=dev, dtype=torch.int64), cnt)
    bits = s.depth.view(torch.int32)[vals.long()].long() & 0xFFFFFFFF
    s.keys = (tile << 32) | bits


def deterministic_mode() -> bool:
    """SLS_DETERMINISTIC=1 or 2: gradient records are accumulated with integer atomics (bit-identical gradients from
    run to run, about one extra tile-backward; the staged interface has the two-launch scheme only, so 2 — MappingEngine's
    one-launch scheme — means 1 here); default: float atomics, whose order changes between runs."""
    return os.environ.get("SLS_DETERMINISTIC", "0") in ("1", "2")


def rasterize_backward(state: ForwardState, means3D, scales, rotations, dL_dallmap, deterministic=None):
    lib = _abi.lib()
    dev = means3D.device
    N = state.N
    dL = _f32c(dL_dallmap)
    det = deterministic_mode() if deterministic is None else deterministic
    nbytes = int(lib.sls_backward_det_scratch_bytes(N)) if det else 0
    out = _Arena(dev, (("dmeans", "f32", (N, 3)), ("dscales", "f32", (N, 2)), ("drots", "f32", (N, 4)), ("dopac", "f32", (N, 1)))
                 + ((("scratch", "u8", (nbytes,)),) if det else (("grec", "f32", (N, lib.sls_grec_stride())),)))
    ce, a1, a2 = state.cam, state.a1, state.a2
    bm = a2.ptr("block_masks") if "block_masks" in a2.spec else None
    if det:
        _abi.check(lib.sls_backward_det(C.byref(ce.cam), N, state.R, means3D.data_ptr(), scales.data_ptr(),
                                        rotations.data_ptr(), state.radii.data_ptr(), a1.ptr("rec"),
                                        a2.ptr("ranges"), state.vals_ptr, state.vals_stride, ce.col_cs.data_ptr(),
                                        ce.row_cs.data_ptr(), a2.ptr("pix_state"), a2.ptr("pix_contrib"),
                                        dL.data_ptr(), out.ptr("dmeans"), out.ptr("dscales"), out.ptr("drots"),
                                        out.ptr("dopac"), bm, state.block_masks_shape,
                                        out.ptr("scratch"), nbytes, _stream(dev)), "sls_backward_det")
        return out.view("dmeans"), out.view("dscales"), out.view("drots"), out.view("dopac"), None
    _abi.check(lib.sls_backward(C.byref(ce.cam), N, state.R, means3D.data_ptr(), scales.data_ptr(),
                                rotations.data_ptr(), state.radii.data_ptr(), a1.ptr("rec"),
                                a2.ptr("ranges"), state.vals_ptr, state.vals_stride, ce.col_cs.data_ptr(),
                                ce.row_cs.data_ptr(), a2.ptr("pix_state"), a2.ptr("pix_contrib"),
                                dL.data_ptr(), out.ptr("grec"), out.ptr("dmeans"), out.ptr("dscales"),
                                out.ptr("drots"), out.ptr("dopac"), bm,
                                state.block_masks_shape, _stream(dev)),
               "sls_backward")
    return out.view("dmeans"), out.view("dscales"), out.view("drots"), out.view("dopac"), out.view("grec")


# ---------------------------------------------------------------------------
# The default path of GaussianRasterizer: sls_forward_ws / sls_backward_ws — one native call each, sized by a capacity
# instead of a host read of R in the middle of the forward, the camera's previous depth order repaired instead of
# sorted anew, gradient records cleared where they are read.  Same kernels as sls_mapping_step; same results as the
# staged calls (tests/test_gpu_parity.py::test_workspace_path_matches_staged_path).
# ---------------------------------------------------------------------------
class _WsEntry:
    """Per (device, N, H, W, flags): one workspace, the status block + its pinned mirror, the capacity guess; per
    camera (the matrix-cache key): the depth order of its last call."""
    __slots__ = ("ws", "ws_ptr", "ws_bytes", "cap", "ready", "busy", "status", "mirror", "mirror_np", "orders", "calls",
                 "rounds", "rounds_until", "stats", "stream", "cap_hint")


_WS_CACHE: "OrderedDict[tuple, _WsEntry]" = OrderedDict()
_WS_CACHE_MAX = 4
_WS_ORDERS_MAX = 64
_SENTINEL = -1          # 0xFFFFFFFF as int32: the device never writes it into words 0 and 7 of the status block


def workspace_path_enabled() -> bool:
    """SLS_STAGED_FORWARD=1: GaussianRasterizer goes through sls_forward_stage1/2 + sls_backward (A/B runs, and what
    callers of rasterize_forward() — the tests — look at buffer by buffer)."""
    return os.environ.get("SLS_STAGED_FORWARD", "0") != "1"


def _ws_entry(dev, N, H, W, lean) -> _WsEntry:
    key = (str(dev), N, H, W, bool(lean))
    e = _WS_CACHE.get(key)
    if e is not None:
        _WS_CACHE.move_to_end(key)
        return e
    e = _WsEntry()
    e.ws, e.cap, e.ready, e.busy = None, 0, False, False
    e.status = torch.zeros((8,), dtype=torch.int32, device=dev)
    e.mirror = torch.zeros((8,), dtype=torch.int32).pin_memory()
    e.mirror_np = e.mirror.numpy()
    e.orders, e.calls, e.rounds, e.rounds_until = OrderedDict(), 0, 1, 0
    e.stats = {"too_small": 0, "repair_failed": 0, "from_scratch": 0, "repaired": 0}
    e.stream, e.cap_hint = None, 0
    # (the surfel set changed — Mapper.densify / prune between keyframes: the old size's workspace will not be asked
    #  for again; one still held by a forward awaiting its backward stays until that has run.  What the old size
    #  learned about the scene — instances per surfel — sizes the new one: a first forward that overflows runs twice)
    for old in [k for k, v in _WS_CACHE.items() if k[0] == key[0] and k[2:] == key[2:] and k[1] != N]:
        seen = _WS_CACHE[old].stats.get("R")
        if seen:
            e.cap_hint = max(e.cap_hint, int(1.3 * seen * N / max(old[1], 1)) + 1024)
        if not _WS_CACHE[old].busy:
            del _WS_CACHE[old]
    _WS_CACHE[key] = e
    while len(_WS_CACHE) > _WS_CACHE_MAX:
        _WS_CACHE.popitem(last=False)
    return e


_WARNED = set()


def _warn_once(key, text) -> None:
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(text, RuntimeWarning, stacklevel=3)


def _ws_alloc(e: _WsEntry, dev, N, H, W, cap) -> None:
    nbytes = int(_abi.lib().sls_forward_ws_bytes(N, H, W, cap))
    e.ws = None
    e.ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
    e.ws_ptr = (e.ws.data_ptr() + 255) & ~255
    e.ws_bytes, e.cap, e.ready = nbytes, int(cap), False


class WsState:
    """What the backward needs from a workspace forward (the buffers themselves live in the entry's workspace)."""
    __slots__ = ("cam", "entry", "N", "R", "cap", "radii", "allmap", "list_ptr", "stride", "shape", "ws", "ws_ptr", "ws_bytes",
                 "block_order")


def rasterize_forward_ws(settings: GaussianRasterizationSettings, means3D, opacities, scales, rotations,
                         keep_for_backward: bool = True) -> Optional[WsState]:
    """Returns None where sls_forward_ws does not apply (more than 512 tiles, D10 on, the workspace of this size still
    held by a forward whose backward has not run): the caller takes the staged path."""
    lib = _abi.lib()
    dev = means3D.device
    N = int(means3D.shape[0])
    if N == 0:
        return None
    ce = get_camera(settings, dev)
    cam = ce.cam
    H, W = cam.H, cam.W
    tw, th = _abi.tile_size()
    T = ((W + tw - 1) // tw) * ((H + th - 1) // th)
    if T > 512 or cam.tile_cull_min >= 2:
        return None
    e = _ws_entry(dev, N, H, W, bool(cam.flags & 1))
    st = _stream(dev)
    if e.busy:
        # a forward of this size still waits for its backward (a loss kept alive, or two renders before one backward):
        # the staged path serves this call — correct, slower; say so once, it is easy to hold a graph by accident
        _warn_once("busy", "GaussianRasterizer: the workspace of this (N, H, W) is held by an earlier forward whose "
                           "backward has not run; this call takes the staged path (slower).  Drop or backward() the "
                           "earlier result to get the one-call path back.")
        return None
    if e.stream is not None and e.stream != st:
        # the workspace's buffers are ordered by ONE stream; a call on another one has no dependency on what still runs there
        _warn_once("stream", "GaussianRasterizer: called on another stream than the one its workspace belongs to; this "
                             "call takes the staged path.")
        return None
    e.stream = st
    if e.ws is None:
        _ws_alloc(e, dev, N, H, W, max(4 * N, 1 << 16, e.cap_hint))
    # the camera's previous depth order (the matrix cache's key says "same camera"): repaired while young enough, with
    # MappingEngine's ages (a repair that does not reach the exact order voids the forward, which is then repeated
    # from scratch — the caller never sees an inexact list)
    okey = id(ce)
    ent = e.orders.get(okey)
    if ent is None or ent[2] is not ce:
        # [depth order, call it was last written at, the camera entry, the tile backward's launch order (zeros: none yet)]
        ent = [torch.empty((N,), dtype=torch.int32, device=dev), None, ce,
               torch.zeros((int(lib.sls_block_order_bytes(H, W)) // 4,), dtype=torch.int32, device=dev)]
        e.orders[okey] = ent
        while len(e.orders) > _WS_ORDERS_MAX:
            e.orders.popitem(last=False)
    else:
        e.orders.move_to_end(okey)
    scale = min(max(500000.0 / N, 1.0), 3.0)
    age = None if ent[1] is None else e.calls - ent[1]
    if e.rounds > 1 and e.calls >= e.rounds_until:
        e.rounds, e.rounds_until = e.rounds - 1, e.calls + 256
    reuse = 0
    if age is not None and age <= int(48 * scale):
        reuse = min(e.rounds + (1 if age > int(4 * scale) else 0) + (1 if age > int(12 * scale) else 0), 4)
    radii = torch.empty((N,), dtype=torch.int32, device=dev)
    allmap = torch.empty((7, H, W), dtype=torch.float32, device=dev)
    lst, stride, shape = C.c_void_p(0), C.c_int(1), C.c_int(0)
    row = e.mirror_np
    while True:
        row[0] = _SENTINEL
        row[7] = _SENTINEL
        rc = lib.sls_forward_ws(C.byref(cam), N, means3D.data_ptr(), scales.data_ptr(), rotations.data_ptr(),
                                opacities.data_ptr(), ce.col_cs.data_ptr(), ce.row_cs.data_ptr(), e.cap,
                                ent[0].data_ptr(), reuse, list_pairs_mode(), 1 if e.ready else 0,
                                1 if keep_for_backward else 0, radii.data_ptr(),
                                allmap.data_ptr(), e.ws_ptr, e.ws_bytes, e.status.data_ptr(), e.mirror.data_ptr(),
                                C.byref(lst), C.byref(stride), C.byref(shape), st)
        if rc == -4:                      # SLS_E_UNSUPPORTED
            return None
        _abi.check(rc, "sls_forward_ws")
        e.ready = True
        # (the wait runs in the library: no interpreter lock held while the binning and the tile forward run)
        _abi.check(lib.sls_wait_status_mirror(e.mirror.data_ptr(), 0xFFFFFFFF, st), "sls_wait_status_mirror")
        R, flags = int(row[0]) & 0xFFFFFFFF, int(row[1])
        if flags == 0:
            break
        if flags & 1:                     # capacity too small: more room (the old workspace drains on the stream first)
            e.stats["too_small"] += 1
            _ws_alloc(e, dev, N, H, W, int(R * 1.3) + 1024)
        if flags & 2:                     # the repair did not reach the exact order: from scratch, and one more round for a while
            e.stats["repair_failed"] += 1
            e.rounds, e.rounds_until = min(e.rounds + 1, 3), e.calls + 256
        reuse = 0
    e.stats["repaired" if reuse else "from_scratch"] += 1
    e.stats["R"] = R
    e.calls += 1
    ent[1] = e.calls
    if settings.debug:
        torch.cuda.synchronize(dev)
        # the early mirror is a snapshot taken by bin_direct's first workgroup (words 0 and 1): nothing behind that point
        # may raise a void bit (sls_sort.hip states the invariant where the bits are set) — checked here against the
        # device's own status block once everything has run
        final = e.status.cpu().numpy()
        if (int(final[0]) & 0xFFFFFFFF) != R or int(final[1]) != 0:
            raise RuntimeError(f"sls_forward_ws: the status block changed after its early mirror (R {R} -> "
                               f"{int(final[0]) & 0xFFFFFFFF}, flags 0 -> {int(final[1])})")
    s = WsState()
    s.cam, s.entry, s.N, s.R, s.cap, s.radii, s.allmap = ce, e, N, R, e.cap, radii, allmap
    s.list_ptr, s.stride, s.shape = int(lst.value), int(stride.value), int(shape.value)
    s.ws, s.ws_ptr, s.ws_bytes, s.block_order = e.ws, e.ws_ptr, e.ws_bytes, ent[3]
    if keep_for_backward:
        e.busy = True                     # released by the backward (or when the autograd node dies without one)
    return s


def rasterize_backward_ws(state: WsState, means3D, scales, rotations, dL_dallmap):
    lib = _abi.lib()
    dev = means3D.device
    N, ce = state.N, state.cam
    dL = _f32c(dL_dallmap)
    out = _Arena(dev, (("dmeans", "f32", (N, 3)), ("dscales", "f32", (N, 2)), ("drots", "f32", (N, 4)), ("dopac", "f32", (N, 1))))
    try:
        _abi.check(lib.sls_backward_ws(C.byref(ce.cam), N, means3D.data_ptr(), scales.data_ptr(), rotations.data_ptr(),
                                       state.radii.data_ptr(), ce.col_cs.data_ptr(), ce.row_cs.data_ptr(), dL.data_ptr(),
                                       state.cap, state.ws_ptr, state.ws_bytes, state.list_ptr, state.stride, state.shape,
                                       state.block_order.data_ptr(), out.ptr("dmeans"), out.ptr("dscales"), out.ptr("drots"), out.ptr("dopac"),
                                       _stream(dev)), "sls_backward_ws")
    except Exception:
        state.entry.ready = False         # (whatever state the records are in: the next forward clears them)
        raise
    return out.view("dmeans"), out.view("dscales"), out.view("drots"), out.view("dopac")


class _WsLease:
    """Held by the autograd context of a workspace forward: gives the workspace back when the backward has run — or when
    the graph is dropped without one (render() under no_grad never takes a lease)."""
    __slots__ = ("state",)

    def __init__(self, state):
        self.state = state

    def release(self):
        st, self.state = self.state, None
        if st is not None and st.entry.ws is st.ws:
            st.entry.busy = False

    def __del__(self):
        self.release()


def frame_from_precomp(cov3D_precomp: torch.Tensor):
    """(scales (N,2), rotations (N,4) w,x,y,z) of surfels given as the precomputed transform the reference's model
    builds (scene/gaussian_model.py:20-36, `build_covariance_from_scaling_rotation`): (N,4,4) with
    `[:, :3, :3] = (R diag(s_u, s_v, 1))^T` — rows 0 and 1 the scaled tangents, row 2 the normal — and row 3 the
    centre; an (N,3,3) tensor (only the frame) is accepted too.  The reference never passes it to the rasterizer
    (gaussian_renderer/__init__.py:46), so this is the in-tree layout, not a pinned rasterizer contract.  Pure torch,
    any device; the surfels rendered from it are those the (scales, rotations) path renders."""
    t = cov3D_precomp
    if t.dim() != 3 or tuple(t.shape[1:]) not in ((4, 4), (3, 3)):
        raise ValueError(f"cov3D_precomp must be (N,4,4) or (N,3,3), got {tuple(t.shape)}")
    rs_t = t[:, :3, :3].to(torch.float32)
    su, sv = rs_t[:, 0].norm(dim=1), rs_t[:, 1].norm(dim=1)
    tu = rs_t[:, 0] / su.clamp_min(1e-30)[:, None]
    tv = rs_t[:, 1] / sv.clamp_min(1e-30)[:, None]
    tn = torch.cross(tu, tv, dim=1)                        # (row 2 is R's third column: recomputed, right-handed)
    R = torch.stack([tu, tv, tn], dim=2)                   # columns t_u, t_v, n
    # rotation matrix -> unit quaternion (w, x, y, z), the branch with the largest denominator per surfel
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22,
                                                1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=1), min=0.0))
    cand = torch.stack([
        torch.stack([q_abs[:, 0] ** 2, R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], dim=1),
        torch.stack([R[:, 2, 1] - R[:, 1, 2], q_abs[:, 1] ** 2, R[:, 1, 0] + R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0]], dim=1),
        torch.stack([R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] + R[:, 0, 1], q_abs[:, 2] ** 2, R[:, 2, 1] + R[:, 1, 2]], dim=1),
        torch.stack([R[:, 1, 0] - R[:, 0, 1], R[:, 2, 0] + R[:, 0, 2], R[:, 2, 1] + R[:, 1, 2], q_abs[:, 3] ** 2], dim=1),
    ], dim=1)                                              # (N, 4 candidates, 4 components), each = 2 q_k * q
    best = q_abs.argmax(dim=1)
    q = cand[torch.arange(R.shape[0], device=R.device), best]
    q = q / (2.0 * q_abs.gather(1, best[:, None]).clamp_min(1e-12))
    q = torch.nn.functional.normalize(q, dim=1)
    return torch.stack([su, sv], dim=1).contiguous(), q.contiguous()


_AUTOGRAD_POLICY_DONE = False


def _autograd_policy() -> None:
    """torch runs the backward of device tensors on a worker thread of its own: every mapping iteration then hands over
    twice between that thread and the caller's, and on a many-core host the two wake each other in 10 us or in 100, at
    random per process (profiles/r05j_autograd_thread.txt: the drop-in rows were bimodal, 0.18 / 0.28 ms at C3).  With ONE
    visible GPU the worker buys nothing — there is nothing to run beside it — so the first differentiated forward makes
    backward() run on the calling thread (`torch.autograd.set_multithreading_enabled(False)`, process-wide).
    SLS_AUTOGRAD_SINGLE_THREAD=0 leaves torch as it is, =1 does it whatever the number of GPUs."""
    global _AUTOGRAD_POLICY_DONE
    if _AUTOGRAD_POLICY_DONE:
        return
    _AUTOGRAD_POLICY_DONE = True
    want = os.environ.get("SLS_AUTOGRAD_SINGLE_THREAD", "")
    if want == "1" or (want == "" and torch.cuda.device_count() == 1):
        torch.autograd.set_multithreading_enabled(False)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, scales, rotations, cov3D_precomp, raster_settings):
        if scales is None or rotations is None:
            raise ValueError("scales and rotations are required")
        m, o, s_, r = map(_f32c, (means3D.detach(), opacities.detach(), scales.detach(), rotations.detach()))
        for name, t in (("means3D", m), ("opacities", o), ("scales", s_), ("rotations", r)):
            _need_cuda(t, name)
        if m.shape != (m.shape[0], 3) or s_.shape != (m.shape[0], 2) or r.shape != (m.shape[0], 4) or o.numel() != m.shape[0]:
            raise ValueError("expected means3D (N,3), scales (N,2), rotations (N,4), opacities (N,1)")
        needs_grad = any(ctx.needs_input_grad[:5])
        if needs_grad:
            _autograd_policy()
        ctx.debug = bool(raster_settings.debug)
        ctx.lease = None
        st = None
        # default: one native call against a capacity (sls_forward_ws); the staged calls where that does not apply, for
        # the deterministic accumulation (sls_backward_det works on the staged buffers) and with SLS_STAGED_FORWARD=1
        if workspace_path_enabled() and not (needs_grad and deterministic_mode()):
            st = rasterize_forward_ws(raster_settings, m, o, s_, r, keep_for_backward=needs_grad)
            if st is not None and needs_grad:
                ctx.lease = _WsLease(st)
        if st is None:
            st = rasterize_forward(raster_settings, m, o, s_, r)
        ctx.state = st
        ctx.settings = raster_settings
        ctx.save_for_backward(m, s_, r, o)
        ctx.mark_non_differentiable(st.radii)
        # allmap is returned as a fresh tensor the caller may overwrite in place
        # (gaussian_renderer/__init__.py:61-62,70-71); the backward never reads it.
        return st.radii, st.allmap

    @staticmethod
    def backward(ctx, _grad_radii, grad_allmap):
        m, s_, r, o = ctx.saved_tensors
        st = ctx.state
        if grad_allmap is None:
            grad_allmap = torch.zeros_like(st.allmap)
        if isinstance(st, WsState) and (ctx.lease is None or ctx.lease.state is None):
            # a second walk of the graph (retain_graph=True): the workspace went back with the first one.  The forward is
            # repeated through the staged calls, whose buffers this node then owns — later walks reuse them
            st = ctx.state = rasterize_forward(ctx.settings, m, o, s_, r)
        if isinstance(st, WsState):
            try:
                dmeans, dscales, drots, dopac = rasterize_backward_ws(st, m, s_, r, grad_allmap)
            finally:
                ctx.lease.release()
        else:
            dmeans, dscales, drots, dopac, _ = rasterize_backward(st, m, s_, r, grad_allmap)
        if ctx.debug:
            torch.cuda.synchronize(m.device)
        return dmeans, None, dopac, dscales, drots, None, None


# run-time bindings that wait for their target to exist: fused_mapper (SLS_FUSED_MAPPER=1: slam/mapper.py imports this
# module before its class statement runs) and fused_render (SLS_FUSED_RENDER=1: gaussian_renderer likewise); empty otherwise
_PENDING_HOOKS: dict = {}


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings
        for hook in list(_PENDING_HOOKS.values()):
            hook()

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        _need_cuda(positions, "positions")
        with torch.no_grad():
            p = _f32c(positions)
            ce = get_camera(self.raster_settings, p.device)
            vis = torch.empty((p.shape[0],), dtype=torch.uint8, device=p.device)
            _abi.check(_abi.lib().sls_mark_visible(C.byref(ce.cam), int(p.shape[0]), p.data_ptr(), vis.data_ptr(),
                                                   _stream(p.device)), "sls_mark_visible")
        return vis.bool()

    def forward(self, means3D, means2D, opacities, scales: Optional[torch.Tensor] = None,
                rotations: Optional[torch.Tensor] = None, cov3D_precomp: Optional[torch.Tensor] = None):
        if (scales is None or rotations is None) == (cov3D_precomp is None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if cov3D_precomp is not None:
            # the precomputed frame carries no gradient (autograd returns None for it, as for means2D)
            scales, rotations = frame_from_precomp(cov3D_precomp.detach())
        return _RasterizeGaussians.apply(means3D, means2D, opacities, scales, rotations, None,
                                         self.raster_settings)
