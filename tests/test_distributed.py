"""Keyframe-parallel mapping with world_size 2 over gloo on CPU:
the all-reduced gradients equal the sum of the per-keyframe gradients, replicas
stay bit-identical, and the scale regulariser is counted once.  The rasterizer
under the harness is the CPU checker (rasterizer_cls test seam); on the GPU box
the same code path runs over RCCL with the HIP rasterizer (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from splat_loam_amd import synth
from splat_loam_amd.mapping import MappingConfig, mapping_loss, optimize_step_sharded
from splat_loam_amd.renderer import render
from splat_loam_amd.scene import Camera, SurfelModel

N, H, W = 300, 16, 64


def _setup(rank):
    from oracle.torch_function import GaussianRasterizer as OracleRasterizer
    sc = synth.make_scene(N, H, W, seed=21, range_lo=2.0, range_hi=10.0, scale_lo=0.05, scale_hi=0.25)
    depth, valid = synth.make_targets(H, W, sc)
    poses = synth.keyframe_poses(2)
    cams = [Camera(sc["K"], depth, None, valid, p, data_device="cpu") for p in poses]
    model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cpu")
    model.training_setup(fused=False)
    return cams, model, OracleRasterizer


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cams, model, R = _setup(rank)
    cfg = MappingConfig()
    for _ in range(2):
        optimize_step_sharded(model, cams[rank], cfg, rasterizer_cls=R)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), xyz=model._xyz.detach().numpy(),
             scaling=model._scaling.detach().numpy(), rotation=model._rotation.detach().numpy(),
             opacity=model._opacity.detach().numpy(), grad_xyz=model._xyz.grad.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_matches_summed_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f"replicas diverged in {k}"
    # single-process emulation: sum of per-keyframe gradients, regulariser once
    cams, model, R = _setup(0)
    cfg = MappingConfig()
    for _ in range(2):
        model.optimizer.zero_grad(set_to_none=True)
        total = 0
        for g, cam in enumerate(cams):
            c = cfg if g == 0 else MappingConfig(**{**cfg.__dict__, "opt_scaling_max_penalty": 0.0})
            total = total + mapping_loss(render(cam, model, cfg.depth_ratio, rasterizer_cls=R), cam, model, c)
        total.backward()
        last_grad = model._xyz.grad.numpy().copy()
        model.optimizer.step()
    assert np.abs(r0["grad_xyz"] - last_grad).max() <= 1e-5 * np.abs(last_grad).max()
    for k, p in (("xyz", model._xyz), ("scaling", model._scaling), ("rotation", model._rotation), ("opacity", model._opacity)):
        ref = p.detach().numpy()
        assert np.abs(r0[k] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-6), k
