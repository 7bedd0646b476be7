#!/usr/bin/env python3
"""Diagnostic: how far does a surfel move in the depth order between two mapping iterations?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from splat_loam_amd import synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel
N, H, W = 500000, 64, 2048
sc = synth.make_scene(N, H, W, seed=0)
depth, valid = synth.make_targets(H, W, sc)
cam = Camera(sc["K"], depth, None, valid, None, data_device="cuda:0")
model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cuda:0")
eng = MappingEngine(model, MappingConfig())
def ranks():
    p = model._xyz.detach()
    Rm = cam.world_view_transform[:3, :3].T.to(p.device); t = cam.world_view_transform[3, :3].to(p.device)
    d = (p @ Rm.T + t).norm(dim=1)
    o = torch.argsort(d, stable=True)
    r = torch.empty_like(o); r[o] = torch.arange(N, device=o.device)
    return r
prev = ranks()
for it in range(12):
    eng.step(cam)
    cur = ranks()
    disp = (cur - prev).abs()
    print("iter %2d: moved %6d surfels, max displacement %5d, p99.9 %4d, mean %.2f" % (
        it, int((disp > 0).sum()), int(disp.max()), int(torch.quantile(disp.float()[:100000], 0.999)), float(disp.float().mean())))
    prev = cur
