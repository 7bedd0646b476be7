#!/bin/bash
# visit x: A/B of library builds on whole iterations at small sizes (one keyframe, 600 timed) + the GPU suite on the last
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
tag = sys.argv[1]
for N, H, W in ((20000, 64, 1024), (50000, 64, 1024), (50000, 128, 1024)):
    sc = synth.make_scene(N, H, W, seed=0)
    depth, valid = synth.make_targets(H, W, sc)
    cam = Camera(sc["K"], depth, None, valid, None, data_device="cuda:0")
    for rep in range(2):
        model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cuda:0")
        eng = MappingEngine(model, MappingConfig())
        for _ in range(100): eng.step(cam, sync="lagged")
        eng.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(600): eng.step(cam, sync="lagged")
        eng.flush(); torch.cuda.synchronize()
        print(tag, N, H, W, round((time.perf_counter() - t0) / 600 * 1e6, 2), "us/iter", flush=True)
PY
for v in $VARIANTS; do cp gpurun_tmp_$v.so splat_loam_amd/libsls_hip.so; python /tmp/one.py $v 2>&1 | grep "us/iter"; done
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
