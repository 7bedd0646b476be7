#!/usr/bin/env python3
"""Timing of one frame-to-keyframe alignment (15 Gauss-Newton iterations) on a synthetic room scan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_aligner import two_scans, pose_error
from gsaligner import GSAligner, GSAlignerParams
dev = torch.device("cuda:0")
for (H, W) in ((64, 1024), (64, 2048), (128, 1024)):
    K, H, W, (dA, pA), (dB, pB), Tgt = two_scans(H, W)
    proj = np.eye(4, dtype=np.float32); proj[:3, :3] = K.T
    t = lambda a: torch.tensor(a, device=dev)
    p = GSAlignerParams(image_height=H, image_width=W)
    al = GSAligner(**p.__dict__)
    al.set_reference(t(dA)[None], t(pA.reshape(-1, 3)), t(proj))
    al.set_query(t(dB)[None], t(pB.reshape(-1, 3)), t(proj))
    ig = torch.eye(4, device=dev)
    for _ in range(3):
        T, fit, info = al.align(ig)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        T, fit, info = al.align(ig)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    e = pose_error(T.cpu().numpy().astype(np.float64), Tgt)
    print(f"{H}x{W}: {dt * 1e3:.3f} ms per align ({p.num_iterations} iterations, {dt / (p.num_iterations + 1) * 1e6:.1f} us each), "
          f"fitness {fit:.3f}, error {e[0] * 1e3:.2f} mm / {np.degrees(e[1]):.3f} deg")
