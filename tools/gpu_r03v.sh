#!/bin/bash
# visit v: where do the block masks start to pay?  whole iterations, one keyframe, 600 timed
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
cat > /tmp/one.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from splat_loam_amd import synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel
N, H, W = (int(x) for x in sys.argv[1:4])
sc = synth.make_scene(N, H, W, seed=0)
depth, valid = synth.make_targets(H, W, sc)
cam = Camera(sc["K"], depth, None, valid, None, data_device="cuda:0")
for mode in (2, 1, 2, 1):
    model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cuda:0")
    eng = MappingEngine(model, MappingConfig()); eng.block_masks = mode
    for _ in range(100): eng.step(cam, sync="lagged")
    eng.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(600): eng.step(cam, sync="lagged")
    eng.flush(); torch.cuda.synchronize()
    print(N, H, W, "block_masks", mode, round((time.perf_counter() - t0) / 600 * 1e6, 2), "us/iter", flush=True)
PY
for shape in "20000 64 1024" "50000 64 1024" "100000 64 1024" "50000 128 1024" "170000 64 1024"; do python /tmp/one.py $shape 2>&1 | grep block_masks; done
