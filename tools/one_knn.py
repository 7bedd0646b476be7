#!/usr/bin/env python3
"""One distCUDA2 call per size (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_loam_amd import synth
from splat_loam_amd.knn import distCUDA2
M = int(sys.argv[1]) if len(sys.argv) > 1 else 170_000
pts = torch.tensor(synth.make_scene(M, 64, 2048, seed=1)["means"], device="cuda:0")
for _ in range(10):
    distCUDA2(pts)
torch.cuda.synchronize()
