"""The mapper's optimisation step (loss of slam/mapper.py:158-201 + Adam step
:204) restated on top of renderer.render, plus the keyframe-parallel variant
that shards independent keyframe renders one-per-GPU and all-reduces the
gradients (not present in the single-GPU reference).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist

from .renderer import render


@dataclass
class MappingConfig:
    # configs/kitti/kitti-00-odom.yaml:18-21
    opt_lambda_alpha: float = 0.4
    opt_lambda_normal: float = 0.5
    opt_scaling_max: float = 0.1
    opt_scaling_max_penalty: float = 1.0
    depth_ratio: float = 0.0


def mapping_loss(render_pkg: dict, camera, model, cfg: MappingConfig) -> torch.Tensor:
    """The mapper's objective on render()'s maps — the four terms of slam/mapper.py:158-199, pinned by golden G2 / G5:
    range error, agreement of the blended normals with the normals of the rendered range image, occupancy where the
    sensor measured something, and a linear price on surfels longer than the cap."""
    measured = camera.image_valid[0] == 1.0                    # (H, W): pixels that carry a measurement
    alpha = render_pkg["rend_alpha"]
    # mean over ALL pixels of the absolute range error at the measured ones
    range_term = (measured * (render_pkg["surf_depth"] - camera.image_depth)).abs().mean()
    cosine = (render_pkg["rend_normal"][..., measured] * render_pkg["surf_normal"][..., measured]).sum(dim=0)
    normal_term = cfg.opt_lambda_normal * (1 - cosine).mean()
    occupancy_term = cfg.opt_lambda_alpha * torch.nn.functional.binary_cross_entropy(
        alpha[..., measured], camera.image_valid[..., measured].float(), reduction="mean")
    longest = model.get_scaling.max(dim=1).values              # the larger of a surfel's two axes
    size_term = cfg.opt_scaling_max_penalty * (longest[longest >= cfg.opt_scaling_max] - cfg.opt_scaling_max).sum()
    return range_term + occupancy_term + normal_term + size_term


def optimize_step(model, camera, cfg: MappingConfig, **render_kw) -> torch.Tensor:
    """One iteration of Mapper.optimize (slam/mapper.py:150-204) for one keyframe."""
    model.optimizer.zero_grad(set_to_none=True)
    pkg = render(camera, model, cfg.depth_ratio, **render_kw)
    loss = mapping_loss(pkg, camera, model, cfg)
    loss.backward()
    with torch.no_grad():
        model.optimizer.step()
    return loss.detach()


def optimize_step_fused(model, camera, cfg: MappingConfig, group=None, average: bool = False) -> torch.Tensor:
    """The same iteration with the consumer fused into HIP (fused.py): rasterizer
    forward -> sls_consumer_fwd_bwd -> rasterizer backward -> fused Adam.  With a
    process group: keyframe-parallel, gradients all-reduced, regulariser on rank 0."""
    from .fused import fused_loss
    model.optimizer.zero_grad(set_to_none=True)
    sharded = dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if sharded else 0
    loss = fused_loss(model, camera, cfg, with_regulariser=(rank == 0))
    loss.backward()
    if sharded:
        flat_grad_allreduce(model, group, average)
    with torch.no_grad():
        model.optimizer.step()
    return loss.detach()


def flat_grad_allreduce(model, group=None, average: bool = False) -> None:
    """all-reduce(SUM) of the four gradient tensors as ONE flat bucket
    (40 B/surfel: 20 MB at 500k) — a single large collective suits xGMI's
    point-to-point links better than four small ones."""
    params = [model._xyz, model._opacity, model._scaling, model._rotation]
    grads = [p.grad for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def flat_grad_allreduce_sparse(model, group=None, average: bool = False) -> int:
    """The same sum with only the TOUCHED set on the wire (the layout MappingEngine's dp_mode "sparse" moves with
    sls_grad_compact / sls_adam_step_sparse, here in torch): a surfel takes part iff any of its 10 gradient
    values is non-zero on some rank — the bitmaps are combined with a MAX all-reduce (N bytes here, N/8 in the native
    path), the union's rows [xyz 3 | opacity | scaling 2 | rotation 4] are packed in surfel order, SUM-reduced and
    scattered back.  Bit-identical to flat_grad_allreduce (same operands per element).  Returns the rows sent."""
    params = [model._xyz, model._opacity, model._scaling, model._rotation]
    rows = torch.cat([p.grad.reshape(p.grad.shape[0], -1) for p in params], dim=1)          # (N, 10)
    touched = (rows != 0).any(dim=1).to(torch.uint8)
    dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=group)
    idx = touched.nonzero().reshape(-1)
    compact = rows[idx].contiguous()
    dist.all_reduce(compact, op=dist.ReduceOp.SUM, group=group)
    if average:
        compact /= dist.get_world_size(group)
    rows.zero_()
    rows[idx] = compact
    off = 0
    for p in params:
        w = p.grad.reshape(p.grad.shape[0], -1).shape[1]
        p.grad.copy_(rows[:, off:off + w].reshape(p.grad.shape))
        off += w
    return int(idx.numel())


def optimize_step_sharded(model, my_camera, cfg: MappingConfig, group=None, average: bool = False,
                          sparse: bool = False, **render_kw) -> torch.Tensor:
    """Keyframe-parallel iteration: every rank renders ITS keyframe against the
    replicated model, gradients are summed over ranks, every rank applies the
    same Adam step (replicas stay bit-identical because the all-reduce result
    is).  The keyframe-independent scale regulariser is counted once: rank 0
    keeps it, the others drop it."""
    model.optimizer.zero_grad(set_to_none=True)
    pkg = render(my_camera, model, cfg.depth_ratio, **render_kw)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    local_cfg = cfg if rank == 0 else MappingConfig(**{**cfg.__dict__, "opt_scaling_max_penalty": 0.0})
    loss = mapping_loss(pkg, my_camera, model, local_cfg)
    loss.backward()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        (flat_grad_allreduce_sparse if sparse else flat_grad_allreduce)(model, group, average)
    with torch.no_grad():
        model.optimizer.step()
    return loss.detach()
