"""Minimal host-side mirrors of the two reference classes the render path
touches: `Camera` (scene/cameras.py:10-50) and the parameter container /
activation getters of `GaussianModel` (scene/gaussian_model.py:39-69,97-121).
Only what the hot path reads is reproduced (attribute names identical), so the
harness code in mapping.py reads like slam/mapper.py.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .optim import FusedAdam


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    return torch.log(x / (1 - x))


class Camera:
    """One LiDAR keyframe: range / normal / valid images + the two matrices
    (world_view_transform = inv(world_T_lidar)^T, projection_matrix[:3,:3] = K^T)."""

    def __init__(self, K, image_depth, image_normal=None, image_valid=None, world_T_lidar=None,
                 data_device="cuda"):
        dev = torch.device(data_device)
        self.data_device = dev

        def put(a, dtype=None):
            # NumPy (the reference's scene/cameras.py:26-39) or tensors already on the device
            # (projector.DeviceProjector): the latter are taken as they are, no host round trip
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=dtype if dtype is not None else a.dtype).contiguous()
            return torch.from_numpy(np.asarray(a, dtype=dtype and {torch.float32: np.float32}[dtype])).to(dev)

        self.image_depth = put(image_depth, torch.float32)
        self.image_height, self.image_width = int(self.image_depth.shape[1]), int(self.image_depth.shape[2])
        if image_normal is None:
            image_normal = np.zeros((3, self.image_height, self.image_width), np.float32)
        if image_valid is None:
            image_valid = np.ones((1, self.image_height, self.image_width), np.uint8)
        self.image_normal = put(image_normal, torch.float32)
        self.image_valid = put(image_valid)
        if world_T_lidar is None:
            world_T_lidar = np.eye(4)
        view = np.linalg.inv(np.asarray(world_T_lidar, dtype=np.float64)).astype(np.float32)
        self.world_view_transform = torch.tensor(view).transpose(0, 1).contiguous().to(dev)
        self.projection_matrix = torch.eye(4, dtype=torch.float32, device=dev)
        self.projection_matrix[:3, :3] = put(K, torch.float32).reshape(3, 3).transpose(0, 1)


class SurfelModel:
    """Raw (pre-activation) parameters + the reference's activations:
    xyz identity, opacity sigmoid, scaling exp, rotation F.normalize."""

    def __init__(self, xyz, scaling_raw, rotation_raw, opacity_raw, device="cuda"):
        def par(a):
            return nn.Parameter(torch.as_tensor(a, dtype=torch.float32).to(device).contiguous().requires_grad_(True))
        self._xyz, self._scaling = par(xyz), par(scaling_raw)
        self._rotation, self._opacity = par(rotation_raw), par(opacity_raw)
        self.optimizer = None

    @classmethod
    def from_activated(cls, means, scales, rots, opac, device="cuda"):
        means, scales, rots, opac = (torch.as_tensor(a, dtype=torch.float32) for a in (means, scales, rots, opac))
        return cls(means, torch.log(scales), rots, inverse_sigmoid(opac.reshape(-1, 1)), device)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    # -- the surfel set changes between keyframes (Mapper.densify / Mapper.prune) ---------------------------------
    _GROUP_ATTR = {"xyz": "_xyz", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}

    def _replace_parameters(self, make_new, carry_state) -> None:
        """Every group's tensor is replaced by `make_new(name, old)`; `carry_state(old_state, new_param)` returns the
        Adam state the new tensor starts with (None: none — the optimizer initialises it lazily at step 0)."""
        for group in (self.optimizer.param_groups if self.optimizer is not None else
                      [{"name": n, "params": [getattr(self, a)]} for n, a in self._GROUP_ATTR.items()]):
            old = group["params"][0]
            new = nn.Parameter(make_new(group["name"], old.detach()).contiguous().requires_grad_(True))
            if self.optimizer is not None:
                state = self.optimizer.state.pop(old, None)
                kept = carry_state(state, new) if state else None
                if kept is not None:
                    self.optimizer.state[new] = kept
                group["params"][0] = new
            setattr(self, self._GROUP_ATTR[group["name"]], new)

    def densification_postfix(self, new_xyz, new_opacity, new_scaling, new_rotation) -> None:
        """Appends surfels (scene/gaussian_model.py:258-316, `cat_tensors_to_optimizer`): a parameter that has Adam
        moments keeps them, the new rows' moments start at zero, the step count goes on."""
        extra = {"xyz": new_xyz, "opacity": new_opacity, "scaling": new_scaling, "rotation": new_rotation}

        def carry(state, new):
            pad = lambda m: torch.cat((m, torch.zeros((new.shape[0] - m.shape[0],) + tuple(m.shape[1:]), dtype=m.dtype,
                                                      device=m.device)))
            return {"step": state["step"], "exp_avg": pad(state["exp_avg"]), "exp_avg_sq": pad(state["exp_avg_sq"])}
        self._replace_parameters(lambda name, old: torch.cat((old, extra[name].to(old))), carry)

    def prune_points(self, mask: torch.Tensor) -> None:
        """Removes the surfels marked in `mask` (scene/gaussian_model.py:237-256, 258-265).  The Adam state does NOT
        survive: the reference's `_prune_optimizer` files the new parameter's state under the group's NAME, where
        torch.optim.Adam never looks, so every prune — i.e. every keyframe's `update_model` — restarts Adam at step 0
        with zero moments (golden G7 pins this through the parameter trajectories)."""
        keep = ~mask.reshape(-1).bool()
        self._replace_parameters(lambda name, old: old[keep], lambda state, new: None)

    def training_setup(self, position_lr=5e-4, opacity_lr=5e-2, scaling_lr=5e-3, rotation_lr=1e-3, fused=True):
        """4 groups, eps 1e-15 (scene/gaussian_model.py:97-121; lrs utils/config_utils.py:180-183)."""
        groups = [
            {"params": [self._xyz], "lr": position_lr, "name": "xyz"},
            {"params": [self._opacity], "lr": opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": rotation_lr, "name": "rotation"},
        ]
        self.optimizer = (FusedAdam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
        return self.optimizer
