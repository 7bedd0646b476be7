#!/usr/bin/env python3
"""Timing of distCUDA2 (sls_knn_dist2) on the bench scene's surfel centres and on a LiDAR-like
scan, with SciPy's cKDTree on the host cores beside it (a baseline, not a target)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from splat_loam_amd import synth
from splat_loam_amd.knn import distCUDA2

dev = torch.device("cuda:0")
for M in (50_000, 170_000, 500_000, 2_000_000):
    sc = synth.make_scene(M, 64, 2048, seed=1)
    pts = torch.tensor(sc["means"], device=dev)
    for _ in range(3):
        out = distCUDA2(pts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        out = distCUDA2(pts)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    line = f"M={M}: {ms * 1e3:.1f} us/call, {M / ms * 1e-3:.1f} Mpoints/s"
    if M <= 500_000:
        from scipy.spatial import cKDTree
        p = sc["means"].astype(np.float64)
        t0 = time.perf_counter()
        d, _ = cKDTree(p).query(p, k=4, workers=-1)
        t1 = time.perf_counter()
        ref = (d[:, 1:] ** 2).mean(1)
        err = np.abs(out.cpu().numpy() - ref).max() / ref.max()
        line += f" | cKDTree ({os.cpu_count()} threads) {1e3 * (t1 - t0):.1f} ms, max rel diff {err:.1e}"
    print(line)
