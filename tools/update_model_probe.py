#!/usr/bin/env python3
"""Where a keyframe's update_model spends its HOST time outside the iterations (cProfile of three updates on a C4-sized
local model; the iterations are cut to 2 so that the cold stages dominate)."""
import cProfile
import os
import pstats
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from splat_loam_amd import fused_mapper, synth
from splat_loam_amd.renderer import depth_to_points
from splat_loam_amd.scene import Camera, SurfelModel

dev = "cuda:0"
n0, H, W, n_kf = 150_000, 128, 1024, 8
sc = synth.make_scene(n0, H, W, seed=0)
d, v = synth.make_targets(H, W, sc)
poses = synth.keyframe_poses(n_kf + 6)


def frame(k):
    cam = Camera(sc["K"], d, None, v, poses[k], data_device=dev)
    pts = depth_to_points(cam, cam.image_depth)
    cam.image_normal = (-pts / pts.norm(dim=0, keepdim=True).clamp_min(1e-9)).contiguous()
    return SimpleNamespace(camera=cam, model_T_frame=torch.tensor(poses[k], dtype=torch.float32, device=dev))


mapping = SimpleNamespace(num_iterations=int(os.environ.get("ITERS", "2")), densify_threshold_egeom=-1.0, densify_threshold_opacity=0.5,
                          densify_percentage=0.15, prob_view_last_keyframe=0.4, pruning_min_opacity=0.0, pruning_min_size=0.0,
                          opt_lambda_alpha=0.1, opt_lambda_normal=0.1, opt_scaling_max=0.5, opt_scaling_max_penalty=0.2)
cfg = SimpleNamespace(mapping=mapping, opt=SimpleNamespace(depth_ratio=0.0))
model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device=dev)
model.training_setup(fused=True)
frames = [frame(k) for k in range(n_kf + 6)]
kfs = frames[:n_kf]
gen = torch.Generator(device=dev); gen.manual_seed(0)
for k in range(n_kf, n_kf + 2):
    kfs = kfs[1:] + [frames[k]]
    fused_mapper.update_model(model, kfs, frames[k], cfg, generator=gen)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(n_kf + 2, n_kf + 6):
    kfs = kfs[1:] + [frames[k]]
    fused_mapper.update_model(model, kfs, frames[k], cfg, generator=gen)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
