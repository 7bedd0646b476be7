"""Shared helpers of the parity tests: run the HIP path and the CPU checker on
the same seeded scene and compare."""
import numpy as np
import torch

from splat_loam_amd import synth
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_backward, rasterize_forward

# Parity bar (BASELINE.json north_star): <= 1e-5 relative on rendered
# depth/normal/alpha and gradients, integers bit-exact.  "Relative" is taken
# against the scale of the quantity: |hip - ref| <= RTOL * max(|ref|, scale).
RTOL = 1e-5


def scene_and_camera(N, H, W, seed=0, pose=None, hfov_deg=360.0, **kw):
    sc = synth.make_scene(N, H, W, seed=seed, **kw)
    if hfov_deg != 360.0:
        sc["K"] = synth.spherical_K(H, W, hfov_deg=hfov_deg)
    view, proj = synth.camera_matrices(sc["K"], pose)
    return sc, view, proj


def hip_forward(device, sc, view, proj, H, W, scale_modifier=1.0, pix_offset=None, tile_cull_min=None, list_pairs=None, lean_allmap=None):
    """The drop-in interface's forward (sls_forward_stage1/2) as GaussianRasterizer runs it, plus the 64-bit keys.
    list_pairs: None = the production rule (dense forward rounds on (surfel, block mask) pairs where the lists are
    long), 1 = pairs whenever possible, 2 = plain lists."""
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, scale_modifier=scale_modifier,
        viewmatrix=torch.tensor(view, device=device), projmatrix=torch.tensor(proj, device=device),
        prefiltered=False, debug=True, pix_offset=pix_offset, tile_cull_min=tile_cull_min, lean_allmap=lean_allmap)
    t = {k: torch.tensor(sc[k], device=device) for k in ("means", "scales", "rots", "opac")}
    st = rasterize_forward(settings, t["means"], t["opac"], t["scales"], t["rots"], want_keys=True, list_pairs=list_pairs)
    torch.cuda.synchronize()
    return st, t


def hip_backward(st, t, dL):
    dL_t = torch.tensor(np.ascontiguousarray(dL, dtype=np.float32), device=t["means"].device)
    out = rasterize_backward(st, t["means"], t["scales"], t["rots"], dL_t)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def rel_err(a, ref, scale=None):
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    s = np.abs(ref)
    if scale is not None:
        s = np.maximum(s, scale)
    return np.abs(a - ref) / np.maximum(s, 1e-30)


def tangent(g, q):
    """Component of a quaternion gradient orthogonal to q: the part that survives
    F.normalize's backward."""
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    return g - (g * q).sum(1, keepdims=True) * q
