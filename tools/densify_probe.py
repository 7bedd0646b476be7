#!/usr/bin/env python3
"""Sub-stage times of update_model's densify (render / candidates / draw / rows / knn / append), device synchronised
between them, median of 20 on a C4-sized model."""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from splat_loam_amd import fused_mapper, slam_rules, synth
from splat_loam_amd.renderer import depth_to_points, render
from splat_loam_amd.scene import Camera, SurfelModel
from splat_loam_amd.knn import distCUDA2
dev = "cuda:0"
n0, H, W = 150_000, 128, 1024
sc = synth.make_scene(n0, H, W, seed=0)
d, v = synth.make_targets(H, W, sc)
pose = synth.keyframe_poses(3)[2]
cam = Camera(sc["K"], d, None, v, pose, data_device=dev)
pts = depth_to_points(cam, cam.image_depth)
cam.image_normal = (-pts / pts.norm(dim=0, keepdim=True).clamp_min(1e-9)).contiguous()
frame = SimpleNamespace(camera=cam, model_T_frame=torch.tensor(pose, dtype=torch.float32, device=dev))
model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device=dev)
model.training_setup(fused=True)
gen = torch.Generator(device=dev); gen.manual_seed(0)
T = {}
def lap(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
with torch.no_grad():
    for it in range(25):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pkg = render(cam, model, 0.0); lap("render", t0); t0 = time.perf_counter()
        cand = slam_rules.densify_candidates(cam.image_valid, pkg["rend_alpha"], pkg["surf_depth"], cam.image_depth, 0.5, -1.0, False); lap("candidates", t0); t0 = time.perf_counter()
        drawn = slam_rules.densify_sample(cand, cam.image_depth, cam.image_valid, 0.15, gen); lap("sample", t0); t0 = time.perf_counter()
        xyz, q = fused_mapper._densify_rows_hip(frame, drawn); lap("rows", t0); t0 = time.perf_counter()
        every = torch.cat((xyz, model.get_xyz.detach())); d2 = distCUDA2(every); lap("knn", t0); t0 = time.perf_counter()
for k, vals in T.items():
    print(f"{k:12s} median {np.median(vals[5:]):.3f} ms  min {np.min(vals[5:]):.3f}")
