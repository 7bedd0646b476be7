"""The mapper's `update_model` on the fused iteration, and the opt-in runtime binding that gives an UNMODIFIED
`slam/mapper.py` the fast path.

`Mapper.update_model` (slam/mapper.py:33-47) is densify -> optimize -> prune.  Only `optimize` (slam/mapper.py:140-214)
is hot: num_iterations + 1 times render + loss + backward + Adam on a keyframe drawn at random.  `fused_optimize` is
that loop on `MappingEngine` (one `sls_mapping_step` per iteration): the same keyframe probabilities, the same calls
into NumPy's global generator (so a seeded run draws the same keyframes), the same Adam — its moments and step counts
are read from and written back to the model's `optimizer.state`, so whatever the reference's densify / prune helpers
(scene/gaussian_model.py:223-316) do with that state before and after keeps working, including their effective state
loss at every prune (`_prune_optimizer` re-files the state under the group's name: scene/gaussian_model.py:237-256).

`SLS_FUSED_MAPPER=1` + importing `diff_surfel_spherical_rasterization` (which gaussian_renderer/__init__.py:5 does)
installs `fused_optimize` as `slam.mapper.Mapper.optimize` — at run time, no file of the checkout is touched.
`densify_model` / `prune_model` restate the two cold stages for callers without a Splat-LOAM checkout (tests, tools);
golden G7 (`tools/make_golden.py`: the reference's own `Mapper` over three keyframes) pins all three.
"""
from __future__ import annotations

import importlib.abc
import importlib.util
import os
import sys

import numpy as np
import torch

from . import slam_rules
from .scene import inverse_sigmoid

GROUPS = ("xyz", "opacity", "scaling", "rotation")          # the flat bucket's order = training_setup's group order
_WIDTH = {"xyz": 3, "opacity": 1, "scaling": 2, "rotation": 4}
_ATTR = {"xyz": "_xyz", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}


# --------------------------------------------------------------------------------------------------------------
# cold stages (slam/mapper.py:49-138, 216-233): restated for callers that have no Splat-LOAM checkout
# --------------------------------------------------------------------------------------------------------------
def normal_aligned_quaternions(normals: torch.Tensor) -> torch.Tensor:
    """(M,3) directions -> (M,4) unit quaternions (w,x,y,z) of a frame whose third axis is the direction:
    first axis d x e_x (d x e_y where d is within 1e-3 of the x axis), second d x first — the frame
    `create_rotation_matrix_from_direction_vector_batch` builds (utils/general_utils.py:152-187) — written as the
    quaternion `matrix_to_quaternion` returns for it (utils/general_utils.py:85-150): read off the rotation matrix
    through its largest component (the best conditioned of the four ways), real part non-negative (pinned by golden
    G3 and G7; the sign matters: Adam moves the RAW quaternion)."""
    d = normals / normals.norm(dim=-1, keepdim=True)
    helper = torch.zeros_like(d)
    near_x = (d[:, 1].abs() < 1e-3) & (d[:, 2].abs() < 1e-3)
    helper[:, 0] = 1.0
    helper[near_x] = torch.tensor([0.0, 1.0, 0.0], dtype=d.dtype, device=d.device)
    a = torch.linalg.cross(d, helper)
    a = a / a.norm(dim=-1, keepdim=True)
    b = torch.linalg.cross(d, a)
    b = b / b.norm(dim=-1, keepdim=True)
    # R = [a | b | d] (columns); trace forms of the four components
    r00, r11, r22 = a[:, 0], b[:, 1], d[:, 2]
    four_sq = torch.stack([1 + r00 + r11 + r22, 1 + r00 - r11 - r22, 1 - r00 + r11 - r22, 1 - r00 - r11 + r22], dim=1)
    mag = torch.sqrt(four_sq.clamp_min(0.0))                  # 2 |q_k|
    r01, r02, r10, r12, r20, r21 = b[:, 0], d[:, 0], a[:, 1], d[:, 1], a[:, 2], b[:, 2]
    # row k = 4 q_k * (w, x, y, z)
    prod = torch.stack([
        torch.stack([four_sq[:, 0], r21 - r12, r02 - r20, r10 - r01], dim=1),
        torch.stack([r21 - r12, four_sq[:, 1], r10 + r01, r02 + r20], dim=1),
        torch.stack([r02 - r20, r10 + r01, four_sq[:, 2], r12 + r21], dim=1),
        torch.stack([r10 - r01, r20 + r02, r21 + r12, four_sq[:, 3]], dim=1)], dim=1)
    k = mag.argmax(dim=1)
    rows = torch.arange(d.shape[0], device=d.device)
    q = prod[rows, k] / (2.0 * mag[rows, k].clamp_min(0.1))[:, None]
    return torch.where(q[:, 0:1] < 0, -q, q)


def _densify_draw_hip(cam, rend_alpha, threshold_opacity: float, percentage: float, generator=None):
    """slam_rules.densify_candidates + densify_sample at densify_threshold_egeom <= 0 through sls_densify_weights: the
    candidates' weights, their count, the gradient's maximum and the weights' sum in ONE launch and ONE host read; the
    draw itself stays torch.multinomial (without replacement, proportional to the weights) — over the whole image with
    zeros outside the candidates instead of over the gathered candidates: the same distribution, another use of the
    random stream (a GPU's stream is not the reference's CUDA stream either way; the CPU path and golden G6 keep the
    reference's form).  Returns (drawn (H,W) bool or None, number of candidates)."""
    import ctypes as C
    from . import _abi
    from .fused import camera_aux
    from .rasterizer import _stream
    dev = cam.image_depth.device
    H, W = int(cam.image_height), int(cam.image_width)
    aux = camera_aux(cam)
    w = torch.empty((H * W,), dtype=torch.float32, device=dev)
    stats = torch.empty((4,), dtype=torch.int32, device=dev)
    alpha = None if rend_alpha is None else rend_alpha.reshape(-1).float().contiguous()
    _abi.check(_abi.lib().sls_densify_weights(H, W, aux.gt.data_ptr(), aux.valid.data_ptr(),
                                              None if alpha is None else alpha.data_ptr(), float(threshold_opacity),
                                              w.data_ptr(), stats.data_ptr(), _stream(dev)), "sls_densify_weights")
    host = stats.cpu().numpy()
    n_cand = int(host[0]) & 0xFFFFFFFF
    gmax, total = (float(v) for v in host[1:3].view(np.float32))
    no_samples = int(percentage * n_cand)
    if no_samples < 2 or not gmax > 0.0 or total / gmax <= 1e-5:
        return None, n_cand
    idx = torch.multinomial(w[None, :], no_samples, generator=generator)[0]
    drawn = torch.zeros((H * W,), dtype=torch.bool, device=dev)
    drawn[idx] = True
    return drawn.view(H, W), n_cand


def _densify_rows_hip(frame, drawn: torch.Tensor):
    """(centres (n,3), quaternions (n,4)) of the drawn pixels through sls_densify_rows: what depth_to_points + the normals'
    rotation + normal_aligned_quaternions compute with ~65 torch kernels (the torch form stays the CPU path and the
    pinned one: golden G3 / G7; tests/test_fused_mapper.py::test_densify_rows_kernel_matches_torch)."""
    import ctypes as C
    from . import _abi
    from .rasterizer import GaussianRasterizationSettings, _stream, get_camera, half_pixel_tables
    cam = frame.camera
    dev = cam.image_depth.device
    H, W = int(cam.image_height), int(cam.image_width)
    pix = drawn.reshape(-1).nonzero().reshape(-1)                # row-major, the order `[..., drawn]` gathers in
    n = int(pix.numel())
    xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    quat = torch.empty((n, 4), dtype=torch.float32, device=dev)
    if n == 0:
        return xyz, quat
    ce = get_camera(GaussianRasterizationSettings(H, W, 1.0, cam.world_view_transform, cam.projection_matrix), dev)
    col_h, row_h = half_pixel_tables(ce, dev)
    if ce.c2w is None:      # inv(world_view_transform^T), once per camera (torch.linalg.inv is 80 us of launches per call)
        ce.c2w = torch.linalg.inv(cam.world_view_transform.T.float()).contiguous()
    c2w = ce.c2w
    mTf = frame.model_T_frame.to(device=dev, dtype=torch.float32).contiguous()
    depth = cam.image_depth.reshape(-1).float().contiguous()
    normal = cam.image_normal.reshape(3, -1).float().contiguous()
    _abi.check(_abi.lib().sls_densify_rows(n, H, W, pix.data_ptr(), depth.data_ptr(), normal.data_ptr(), col_h.data_ptr(),
                                           row_h.data_ptr(), c2w.data_ptr(), mTf.data_ptr(), xyz.data_ptr(), quat.data_ptr(),
                                           _stream(dev)), "sls_densify_rows")
    return xyz, quat


@torch.no_grad()
def densify_model(gmodel, frame, drawn: torch.Tensor, opt_scaling_max: float, knn=None) -> int:
    """slam/mapper.py:104-137: one surfel per drawn pixel of the keyframe — centre = the pixel's measured point in the
    model frame, both scales = sqrt(mean squared distance to the 3 nearest of (new + existing) centres) clamped to
    [sqrt(1e-7), opt_scaling_max], third axis = the measured normal rotated into the model frame, opacity 0.9 —
    appended with `gmodel.densification_postfix`.  `drawn`: (H,W) bool.  Returns the number added."""
    from .renderer import depth_to_points
    own_knn = knn is None
    if own_knn:
        from .knn import distCUDA2 as knn
    cam = frame.camera
    quats = None
    if cam.image_depth.is_cuda:
        points, quats = _densify_rows_hip(frame, drawn)          # centres + rotations in one launch
    else:
        points = depth_to_points(cam, cam.image_depth)[..., drawn].T.contiguous()
    n_new = int(points.shape[0])
    if n_new == 0:
        return 0
    n_old = int(gmodel.get_xyz.shape[0])
    every = points if n_old == 0 else torch.cat((points, gmodel.get_xyz.detach()))
    if own_knn and n_old > 0:
        d2 = knn(every, first=n_new).clamp(1e-7, opt_scaling_max ** 2)       # (only the new surfels' distances are kept)
    else:
        d2 = knn(every).clamp(1e-7, opt_scaling_max ** 2)[:n_new]
    log_scales = torch.log(torch.sqrt(d2))[:, None].repeat(1, 2)
    if quats is None:
        normals = frame.model_T_frame[:3, :3].to(points) @ cam.image_normal[..., drawn]
        quats = normal_aligned_quaternions(normals.T.contiguous())
    raw_opacity = inverse_sigmoid(torch.full((n_new, 1), 0.9, dtype=torch.float32, device=points.device))
    gmodel.densification_postfix(new_xyz=points, new_opacity=raw_opacity, new_scaling=log_scales, new_rotation=quats)
    return n_new


@torch.no_grad()
def prune_model(gmodel, min_opacity: float = 0.0, min_size: float = 0.0) -> torch.Tensor:
    """slam/mapper.py:216-233.  Returns the mask of the surfels that were REMOVED."""
    if not (min_opacity and min_opacity > 0) and not (min_size and min_size > 0) and hasattr(gmodel, "_replace_parameters"):
        # the reference's defaults (utils/config_utils.py:105-106): nothing can be removed, yet the reference still runs
        # prune_points — new Parameter objects, the Adam state gone (SurfelModel.prune_points).  The same effect without
        # four boolean gathers (a host sync and a copy each): the new Parameters share the old storage
        gmodel._replace_parameters(lambda name, old: old, lambda state, new: None)
        return torch.zeros((gmodel._xyz.shape[0],), dtype=torch.bool, device=gmodel._xyz.device)
    mask = slam_rules.prune_mask(gmodel.get_opacity, gmodel.get_scaling, min_opacity, min_size)
    gmodel.prune_points(mask)
    return mask


# --------------------------------------------------------------------------------------------------------------
# the hot stage
# --------------------------------------------------------------------------------------------------------------
_ENGINE_ATTR = "_sls_fused_engine"      # (signature, MappingEngine) kept ON the model: it lives and dies with it


def engine_of(gmodel):
    """The MappingEngine fused_optimize last ran `gmodel` on (None: none yet) — its `.stats` count repeated iterations."""
    hit = getattr(gmodel, _ENGINE_ATTR, None)
    return hit[1] if hit is not None else None


def _group_of(optimizer, name):
    for g in optimizer.param_groups:
        if g.get("name") == name:
            return g
    raise RuntimeError(f"the model's optimizer has no parameter group named {name!r} "
                       "(scene/gaussian_model.py:97-121 builds xyz / opacity / scaling / rotation)")


def _engine_for(gmodel, cfg_map, depth_ratio):
    """One MappingEngine per model object and optimiser setting; a changed surfel count resizes it in place
    (MappingEngine.resize: the pinned mirrors, the workspace and the keyframes' launch orders survive)."""
    from .engine import MappingEngine
    from .mapping import MappingConfig
    params = tuple(getattr(gmodel, _ATTR[g]) for g in GROUPS)
    groups = [_group_of(gmodel.optimizer, g) for g in GROUPS]
    lrs = tuple(float(g["lr"]) for g in groups)
    betas = tuple(float(b) for b in groups[0]["betas"])
    eps = float(groups[0]["eps"])
    for g in groups[1:]:
        if tuple(float(b) for b in g["betas"]) != betas or float(g["eps"]) != eps:
            raise RuntimeError("fused_optimize: the four parameter groups must share betas and eps")
    for g in groups:      # (what the fused Adam does not implement must not be dropped silently)
        if float(g.get("weight_decay", 0.0)) != 0.0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise RuntimeError("fused_optimize: weight_decay, amsgrad and maximize are not supported "
                               "(scene/gaussian_model.py:121 sets none of them)")
    mc = MappingConfig(opt_lambda_alpha=float(cfg_map.opt_lambda_alpha), opt_lambda_normal=float(cfg_map.opt_lambda_normal),
                       opt_scaling_max=float(cfg_map.opt_scaling_max),
                       opt_scaling_max_penalty=float(cfg_map.opt_scaling_max_penalty), depth_ratio=float(depth_ratio))
    sig = (lrs, betas, eps, tuple(sorted(mc.__dict__.items())), str(params[0].device))
    for p in params:        # (the engine updates these tensors in place: it needs them as it finds them)
        if not p.is_cuda:
            raise RuntimeError("fused_optimize needs the model on a ROCm device; there is no CPU fallback")
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("fused_optimize needs contiguous float32 parameters")
    hit = getattr(gmodel, _ENGINE_ATTR, None)
    if hit is not None and hit[0] == sig:
        # the same model under the same settings: the engine reads the parameter tensors from the model at every step
        # (Mapper.densify / prune replace them between keyframes), only their COUNT is the engine's own
        if hit[1].N != int(params[0].shape[0]):
            hit[1].resize(int(params[0].shape[0]))
        return hit[1]
    eng = MappingEngine(gmodel, mc, lrs=lrs, betas=betas, eps=eps)
    eng.resize(eng.N)       # (buckets with head room from the start: the next keyframe's surfels fit)
    setattr(gmodel, _ENGINE_ATTR, (sig, eng))      # (model -> engine -> model: a cycle the collector frees with the model;
    return eng                                     #  a module-level table keyed by the model would keep both alive for ever)


def _adam_state_in(eng, optimizer, params) -> None:
    """optimizer.state -> the engine's flat moment buckets and step count (zeros / 0 where a parameter has none)."""
    N, off, steps = eng.N, 0, []
    for name, p in zip(GROUPS, params):
        n = _WIDTH[name] * N
        st = optimizer.state.get(p, None)
        if st is not None and "exp_avg" in st:
            eng.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            eng.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.append(int(float(st["step"])))
        else:
            eng.exp_avg[off:off + n].zero_()
            eng.exp_avg_sq[off:off + n].zero_()
            steps.append(0)
        off += n
    if len(set(steps)) != 1:
        raise RuntimeError(f"fused_optimize: the parameter groups' Adam step counts differ ({steps})")
    eng.t = steps[0]


def _adam_state_out(eng, optimizer, params) -> None:
    N, off = eng.N, 0
    for name, p in zip(GROUPS, params):
        n = _WIDTH[name] * N
        st = optimizer.state[p]
        st["exp_avg"] = eng.exp_avg[off:off + n].view(p.shape).clone()
        st["exp_avg_sq"] = eng.exp_avg_sq[off:off + n].view(p.shape).clone()      # (clones: the buckets are the engine's to overwrite)
        old = st.get("step", None)
        if old is not None and not torch.is_tensor(old):
            st["step"] = int(eng.t)
        else:
            # where torch keeps the count: with the old tensor if there is one, else on the parameter's device for a fused /
            # capturable optimizer (torch/optim/adam.py: _init_group) and on the host otherwise — a later optimizer.step()
            # of torch's own must find it where its kernels expect it
            group = _group_of(optimizer, name)
            on_device = bool(group.get("fused")) or bool(group.get("capturable"))
            dev = old.device if old is not None else (p.device if on_device else torch.device("cpu"))
            st["step"] = torch.tensor(float(eng.t), dtype=old.dtype if old is not None else torch.float32, device=dev)
        off += n


@torch.no_grad()
def fused_optimize(gmodel, keyframes, cfg, logger=None, rng=None, marks=None):
    """`Mapper.optimize` (slam/mapper.py:140-214) on MappingEngine.  `gmodel`: the surfel model (`_xyz`, `_opacity`,
    `_scaling`, `_rotation` as contiguous float32 device Parameters and `optimizer` with the four named groups);
    `keyframes`: the local model's list (objects with `.camera`); `cfg`: the reference's Configuration (`.mapping`,
    `.opt.depth_ratio`).  `rng`: None = NumPy's global generator, as the reference uses it.
    Returns the loss's exponential moving average (the reference computes it for its log line and returns None)."""
    m = cfg.mapping
    n_kf = len(keyframes)
    probabilities = slam_rules.keyframe_probabilities(n_kf, m.prob_view_last_keyframe)
    choose = (np.random if rng is None else rng).choice
    eng = _engine_for(gmodel, m, cfg.opt.depth_ratio)
    params = tuple(getattr(gmodel, _ATTR[g]) for g in GROUPS)
    optimizer = gmodel.optimizer
    optimizer.zero_grad(set_to_none=True)
    _adam_state_in(eng, optimizer, params)
    if marks is not None:
        marks("engine_and_adam_state_in")
    ema, weight = None, 0.1
    seen = 0

    def account(status):
        nonlocal ema, seen
        seen += 1
        ema = status["loss"] if ema is None else weight * status["loss"] + (1 - weight) * ema
        if logger is not None and seen % 100 == 0:
            logger.debug(f"it={seen} l_ema={ema:.3f}")

    for _ in range(int(m.num_iterations) + 1):
        keyframe = keyframes[int(choose(n_kf, p=probabilities))]
        status = eng.step(keyframe.camera, sync="lagged")      # the status of the iteration BEFORE this one
        if status is not None:
            account(status)
    eng.flush()
    for status in eng.flushed:
        account(status)
    if marks is not None:
        marks("iterations")
    _adam_state_out(eng, optimizer, params)
    return ema


@torch.no_grad()
def update_model(gmodel, keyframes, frame, cfg, initialize_model: bool = False, drawn=None, generator=None,
                 logger=None, rng=None, timings: bool = False):
    """`Mapper.update_model` (slam/mapper.py:33-47) for callers without a Splat-LOAM checkout: densify `frame`
    (already in `keyframes`), optimise over `keyframes`, prune.  `drawn`: the (H,W) mask of densified pixels, instead
    of drawing it (`slam_rules.densify_sample`, torch.multinomial with `generator`) from the candidates.
    Returns dict(added, removed (mask), loss_ema, candidates (None where the draw ran in one launch)); `timings=True` (bench_extras.update_model) synchronises
    the device between the stages and adds `timings_ms`."""
    import time
    from .renderer import render
    m = cfg.mapping
    cam = frame.camera
    marks = []

    def mark(name):
        if timings:
            torch.cuda.synchronize(gmodel._xyz.device)
            marks.append((name, time.perf_counter()))
    mark("start")
    pkg = None if initialize_model else render(cam, gmodel, cfg.opt.depth_ratio)
    if drawn is None and cam.image_depth.is_cuda and not (m.densify_threshold_egeom and m.densify_threshold_egeom > 0.0):
        # (the usual configuration — configs/*: densify_threshold_egeom = -1: candidates, weights and their sums in one launch)
        candidates = None
        drawn, _ = _densify_draw_hip(cam, None if pkg is None else pkg["rend_alpha"], m.densify_threshold_opacity,
                                     m.densify_percentage, generator)
    else:
        candidates = slam_rules.densify_candidates(cam.image_valid, None if pkg is None else pkg["rend_alpha"],
                                                   None if pkg is None else pkg["surf_depth"], cam.image_depth,
                                                   m.densify_threshold_opacity, m.densify_threshold_egeom, initialize_model)
        if drawn is None:
            drawn = slam_rules.densify_sample(candidates, cam.image_depth, cam.image_valid, m.densify_percentage, generator)
    mark("densify_render_and_draw")
    added = 0 if drawn is None else densify_model(gmodel, frame, drawn, m.opt_scaling_max)
    mark("densify_knn_and_append")
    ema = fused_optimize(gmodel, keyframes, cfg, logger=logger, rng=rng, marks=mark if timings else None)
    mark("adam_state_out")
    removed = prune_model(gmodel, m.pruning_min_opacity, m.pruning_min_size or 0.0)
    mark("prune")
    out = {"added": added, "removed": removed, "loss_ema": ema, "candidates": candidates}
    if timings:
        out["timings_ms"] = {b[0]: (b[1] - a[1]) * 1e3 for a, b in zip(marks[:-1], marks[1:])}
    return out


# --------------------------------------------------------------------------------------------------------------
# run-time binding:  SLS_FUSED_MAPPER=1  ->  slam.mapper.Mapper.optimize = fused_optimize
# --------------------------------------------------------------------------------------------------------------
_TARGET = "slam.mapper"
_PENDING = False          # a binding was asked for and slam.mapper.Mapper did not exist yet
_INSTALLED = None         # the class that was patched


def _bound_optimize(original):
    def optimize(self):
        gmodel = self.model.get_gmodel
        if not gmodel.get_xyz.is_cuda:
            return original(self)                     # (a CPU model: nothing here can run it; the reference's loop says why)
        mod = sys.modules.get(_TARGET)
        fused_optimize(gmodel, self.model.keyframes, self.cfg, logger=getattr(mod, "logger", None))
    optimize.__doc__ = "Mapper.optimize through MappingEngine (SLS_FUSED_MAPPER=1; splat_loam_amd/fused_mapper.py)"
    optimize._sls_original = original
    return optimize


def install(mapper_cls=None) -> bool:
    """Patches `Mapper.optimize` of `slam.mapper` (or of the class given).  True if the class is patched now."""
    global _PENDING, _INSTALLED
    if mapper_cls is None:
        mod = sys.modules.get(_TARGET)
        mapper_cls = getattr(mod, "Mapper", None) if mod is not None else None
    if mapper_cls is None:
        return False
    if not hasattr(mapper_cls.optimize, "_sls_original"):
        mapper_cls.optimize = _bound_optimize(mapper_cls.optimize)
    _PENDING, _INSTALLED = False, mapper_cls
    from . import rasterizer
    rasterizer._PENDING_HOOKS.pop("mapper", None)
    return True


def uninstall() -> None:
    global _INSTALLED, _PENDING
    if _INSTALLED is not None and hasattr(_INSTALLED.optimize, "_sls_original"):
        _INSTALLED.optimize = _INSTALLED.optimize._sls_original
    _INSTALLED, _PENDING = None, False
    from . import rasterizer
    rasterizer._PENDING_HOOKS.pop("mapper", None)


class _AfterImport(importlib.abc.MetaPathFinder):
    """slam.mapper not imported yet: let the regular finders locate it, patch the class once its module has run."""
    _busy = False

    def find_spec(self, name, path=None, target=None):
        if name != _TARGET or _AfterImport._busy or not _PENDING:
            return None
        _AfterImport._busy = True
        try:
            spec = importlib.util.find_spec(name)
        finally:
            _AfterImport._busy = False
        if spec is None or spec.loader is None or not hasattr(spec.loader, "exec_module"):
            return None
        inner = spec.loader

        class _Loader(importlib.abc.Loader):
            def create_module(self, s):
                return inner.create_module(s)

            def exec_module(self, module):
                inner.exec_module(module)
                install(getattr(module, "Mapper", None))
        spec.loader = _Loader()
        return spec


def poll() -> None:
    """Called (when a binding is pending) from the two entry points every mapper run passes before its first
    `optimize` — `GaussianRasterizer.__init__` and `distCUDA2` (slam/mapper.py:52,113): `slam.mapper` imports
    `gaussian_renderer` — and with it this package — BEFORE its `class Mapper` statement runs, so the class cannot be
    patched at import time in that order."""
    if _PENDING:
        install()


def maybe_install() -> None:
    """SLS_FUSED_MAPPER=1: bind now if `slam.mapper.Mapper` exists, else after its import / at the first poll()."""
    global _PENDING
    if os.environ.get("SLS_FUSED_MAPPER", "0") != "1" or _INSTALLED is not None or _PENDING:
        return
    if install():
        return
    _PENDING = True
    from . import rasterizer
    rasterizer._PENDING_HOOKS["mapper"] = poll
    if not any(isinstance(f, _AfterImport) for f in sys.meta_path):
        sys.meta_path.insert(0, _AfterImport())
