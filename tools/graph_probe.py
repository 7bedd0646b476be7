#!/usr/bin/env python3
"""Does replaying the iteration's launch chain from a HIP graph shorten the GPU-side time per iteration?
Experiment only: the captured kernel arguments (Adam step number, status slot) are frozen, so the replayed
iterations are not a valid optimisation — only their TIMING is looked at.
    python tools/graph_probe.py N H W [iters]"""
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
for _ in range(30):
    eng.step(cam)
torch.cuda.synchronize()


def span(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def plain():
    eng._enqueue(cam, apply_adam=True, with_regulariser=True)


print(f"{N} {H}x{W}: stream launches (repair path, no status read): {span(plain, iters):.1f} us/iter")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        plain()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        plain()
    print(f"{N} {H}x{W}: graph replay of the same chain: {span(g.replay, iters):.1f} us/iter")
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        for _ in range(4):
            plain()
    print(f"{N} {H}x{W}: graph of 4 iterations: {span(g4.replay, iters // 4) / 4:.1f} us/iter")
except Exception as e:
    print("capture failed:", e)
