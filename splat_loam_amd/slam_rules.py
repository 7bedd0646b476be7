"""The three host-side rules of the reference's SLAM loop that decide WHAT the hot path is asked to do —
how many surfels a keyframe adds and where (Mapper.densify), which keyframe an iteration renders
(Mapper.optimize's sampling) and when a frame becomes a keyframe (Tracker.require_new_keyframe).

They are not GPU work and the orchestration around them stays out of scope, but BASELINE config 4
("full tracker+mapper loop") is only the reference's workload if N and the keyframe cadence come out of the
reference's own rules; `tools/slam_demo.py` and `bench.py` use these.  Plain torch / NumPy, any device.
"""
from __future__ import annotations

import numpy as np
import torch


def sample_geometric(n_frames: int, last_frame_probability: float) -> np.ndarray:
    """utils/sampling_utils.py:11-20: p_i ~ (1 - p)^i * p over the keyframe LIST ORDER (i = 0 is the first
    keyframe of the list — the reference's name notwithstanding), normalised.  One frame: [1.0]."""
    if n_frames == 1:
        return np.array([1.0])
    k = np.arange(1, n_frames + 1)
    probs = np.power(1.0 - last_frame_probability, k - 1) * last_frame_probability
    return probs / probs.sum()


def keyframe_probabilities(n_frames: int, prob_view_last_keyframe) -> np.ndarray:
    """slam/mapper.py:142-149: uniform (float32, as the reference builds it) when the probability is None or
    negative, else sample_geometric."""
    if prob_view_last_keyframe is None or prob_view_last_keyframe < 0.0:
        return np.float32([1.0 / n_frames] * n_frames)
    return sample_geometric(n_frames, prob_view_last_keyframe)


def compute_depth_gradient(depth: torch.Tensor, valid_mask: torch.Tensor) -> torch.Tensor:
    """utils/graphic_utils.py:91-106: magnitude of the central differences of log(depth) (non-finite logs -> 0),
    each difference masked by the validity of its two end points, zero on the 1-pixel border.
    depth (1,H,W) float, valid_mask (1,H,W) bool/uint8."""
    logd = torch.nan_to_num(torch.log(depth), nan=0.0, posinf=0.0, neginf=0.0)
    v = valid_mask.bool()
    res = torch.zeros_like(logd)
    dx = (logd[..., 2:, 1:-1] - logd[..., :-2, 1:-1]) * (v[..., 2:, 1:-1] & v[..., :-2, 1:-1])
    dy = (logd[..., 1:-1, 2:] - logd[..., 1:-1, :-2]) * (v[..., 1:-1, 2:] & v[..., 1:-1, :-2])
    res[..., 1:-1, 1:-1] = torch.sqrt(dx ** 2 + dy ** 2)
    return res


def densify_candidates(image_valid: torch.Tensor, rend_alpha=None, surf_depth=None, image_depth=None,
                       threshold_opacity: float = 0.5, threshold_egeom: float = -1.0,
                       initialize_model: bool = False) -> torch.Tensor:
    """slam/mapper.py:51-76: the pixels a keyframe may add surfels at, (H,W) bool.  First keyframe: every valid
    pixel.  Afterwards: valid pixels the model renders with alpha <= threshold_opacity, plus (threshold_egeom > 0)
    pixels whose rendered depth lies BEHIND the measurement with an error above the 95 % quantile of the masked
    depth error."""
    valid = image_valid[0] == 1
    if initialize_model:
        return valid.clone()
    mask = (rend_alpha[0] <= threshold_opacity) & valid
    if threshold_egeom > 0.0:
        geom_loss = torch.abs(image_depth - surf_depth)
        geom_loss[..., ~valid] = 0.0
        mask_depth = (surf_depth > image_depth) & (geom_loss > geom_loss.quantile(0.95))
        mask = mask | mask_depth[0]
    return mask


def densify_sample(candidates: torch.Tensor, image_depth: torch.Tensor, image_valid: torch.Tensor,
                   percentage: float = 0.15, generator: torch.Generator | None = None):
    """slam/mapper.py:78-102: int(percentage * #candidates) pixels drawn WITHOUT replacement from the candidates
    with probability proportional to the (max-normalised) log-depth gradient — surfels go where the range image
    has structure.  Returns the (H,W) bool mask of the drawn pixels, or None where the reference returns without
    densifying (fewer than 2 samples, or no gradient mass on the candidates)."""
    cand = candidates.nonzero()
    no_samples = int(percentage * cand.shape[0])
    if no_samples < 2:
        return None
    grad = compute_depth_gradient(image_depth, image_valid)
    grad = grad / grad.max()
    w = grad[..., candidates]                      # (1, #candidates), row-major candidate order as .nonzero()
    if float(w.sum()) <= 1e-5:
        return None
    idx = torch.multinomial(w, no_samples, generator=generator)[0]
    out = torch.zeros_like(candidates, dtype=torch.bool)
    out[cand[idx, 0], cand[idx, 1]] = True
    return out


def require_new_keyframe(num_frames_tracked: int, fitness: float, keyframe_T_frame,
                         threshold_nframes: int = -1, threshold_fitness: float = -1.0,
                         threshold_distance: float = 1.0) -> bool:
    """slam/tracker.py:61-84: a new keyframe when (each test only if its threshold is > 0) more than
    threshold_nframes frames were tracked on this keyframe, OR the registration fitness fell below
    threshold_fitness, OR the frame is further than threshold_distance from the keyframe."""
    t = keyframe_T_frame[:3, -1]
    dist = float(torch.linalg.norm(t)) if torch.is_tensor(t) else float(np.linalg.norm(np.asarray(t)))
    ret = False
    if threshold_nframes and threshold_nframes > 0:
        ret = ret or (num_frames_tracked > threshold_nframes)
    if threshold_fitness and threshold_fitness > 0:
        ret = ret or (fitness < threshold_fitness)
    if threshold_distance and threshold_distance > 0:
        ret = ret or (dist > threshold_distance)
    return bool(ret)


def prune_mask(opacity: torch.Tensor, scaling: torch.Tensor, min_opacity: float = 0.0, min_size: float = 0.0):
    """slam/mapper.py:216-233: surfels to REMOVE — activated opacity below min_opacity (if > 0) or
    |scaling| below min_size (if > 0).  (N,) bool."""
    mask = torch.zeros((opacity.shape[0],), dtype=torch.bool, device=opacity.device)
    if min_opacity > 0:
        mask |= (opacity.reshape(-1) < min_opacity)
    if min_size > 0:
        mask |= (torch.linalg.norm(scaling, dim=-1) < min_size)
    return mask
