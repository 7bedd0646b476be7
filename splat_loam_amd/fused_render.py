"""`SLS_FUSED_RENDER=1`: the reference's `gaussian_renderer.render`, when NOBODY differentiates the call — Mapper.densify
(slam/mapper.py:52-54), the tracker's target (slam/tracker.py:173-175), the logger (slam/slam.py:81-82), meshing
(scene/postprocessing.py:162) — runs as the rasterizer's forward + ONE launch for the five maps
(`renderer.render` -> `sls_render_maps`) instead of the ~30 torch kernels of gaussian_renderer/__init__.py:48-93.

The callers hold the function itself (`from gaussian_renderer import render`), so rebinding the module attribute would
not reach them: the function object's CODE is exchanged in place — same object, same name, same defaults — for a shim
that hands a call with autograd ENABLED to the reference's own code (kept as a function of its own) and everything else
to the fast path.  `gaussian_renderer` is still being imported when this package is (it imports the rasterizer before
its `def render` runs), so the exchange happens at the first `GaussianRasterizer(...)`, i.e. inside the first render()
call, and holds from the second call on.  Opt-in, process-wide, no file of the checkout is touched; `uninstall()` puts
the code back.  Pinned by tests/test_fused_render.py (CPU: the mechanics; GPU: same maps as the reference-shaped path)."""
from __future__ import annotations

import os
import sys
import types

_TARGET = "gaussian_renderer"
_ORIGINAL = None          # the reference's render as a function object of its own (its code, its globals)
_PATCHED = None           # the function object whose code was exchanged
CALLS = {"fast": 0, "reference": 0}


def _fast(camera, model, depth_ratio):
    from . import renderer
    CALLS["fast"] += 1
    return renderer.render(camera, model, depth_ratio)


def _shim(camera, model, depth_ratio=0.0):
    # (runs with gaussian_renderer's globals: everything it needs is imported here)
    import torch as _torch
    from splat_loam_amd import fused_render as _fr
    if _torch.is_grad_enabled() or not model.get_xyz.is_cuda:
        _fr.CALLS["reference"] += 1
        return _fr._ORIGINAL(camera, model, depth_ratio)
    return _fr._fast(camera, model, depth_ratio)


def install(fn=None) -> bool:
    """Exchanges the code of `gaussian_renderer.render` (or of the function given).  True if it is exchanged now."""
    global _ORIGINAL, _PATCHED
    if _PATCHED is not None:
        return True
    if fn is None:
        mod = sys.modules.get(_TARGET)
        fn = getattr(mod, "render", None) if mod is not None else None
    if fn is None or not isinstance(fn, types.FunctionType):
        return False
    if fn.__code__.co_freevars or fn.__code__.co_argcount != 3:
        raise RuntimeError("SLS_FUSED_RENDER: gaussian_renderer.render is not the plain render(camera, model, depth_ratio) "
                           "this binding was written for")
    _ORIGINAL = types.FunctionType(fn.__code__, fn.__globals__, fn.__name__, fn.__defaults__, fn.__closure__)
    fn.__code__ = _shim.__code__
    _PATCHED = fn
    from . import rasterizer
    rasterizer._PENDING_HOOKS.pop("render", None)
    return True


def uninstall() -> None:
    global _ORIGINAL, _PATCHED
    if _PATCHED is not None and _ORIGINAL is not None:
        _PATCHED.__code__ = _ORIGINAL.__code__
    _ORIGINAL = _PATCHED = None
    from . import rasterizer
    rasterizer._PENDING_HOOKS.pop("render", None)


def maybe_install() -> None:
    """SLS_FUSED_RENDER=1: exchange now if `gaussian_renderer.render` exists, else at the first GaussianRasterizer(...)."""
    if os.environ.get("SLS_FUSED_RENDER", "0") != "1" or _PATCHED is not None:
        return
    if install():
        return
    from . import rasterizer
    rasterizer._PENDING_HOOKS["render"] = install
