#!/usr/bin/env python3
"""Is a mapping iteration CPU-bound (enqueue rate) or GPU-bound at a given size?
    python tools/enqueue_rate.py N H W [iters]
Prints: host time to ENQUEUE an iteration (sync=False: nothing read back, no waiting), host time per lagged step,
and the GPU time per iteration (events around the whole run)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_loam_amd import synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel
N, H, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (50000, 64, 1024)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 400
sc = synth.make_scene(N, H, W, seed=0)
depth, valid = synth.make_targets(H, W, sc)
cam = Camera(sc["K"], depth, None, valid, None, data_device="cuda:0")
model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cuda:0")
eng = MappingEngine(model, MappingConfig())
for _ in range(20):
    eng.step(cam)
torch.cuda.synchronize()
for mode in ("lagged", False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters):
        eng.step(cam, sync=mode)
    t_enq = time.perf_counter() - t0
    if mode == "lagged":
        eng.flush()
    e1.record(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{N} {H}x{W} sync={mode}: host enqueue loop {t_enq / iters * 1e6:.1f} us/iter, wall incl. drain {t_all / iters * 1e6:.1f} us/iter, "
          f"GPU span {e0.elapsed_time(e1) / iters * 1e3:.1f} us/iter")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    eng.step(cam, sync=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
