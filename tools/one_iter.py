#!/usr/bin/env python3
"""A few engine iterations on the bench scene (target for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_loam_amd import synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel
N, H, W = 500000, 64, 2048
sc = synth.make_scene(N, H, W, seed=0)
depth, valid = synth.make_targets(H, W, sc)
cam = Camera(sc["K"], depth, None, valid, None, data_device="cuda:0")
model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cuda:0")
if os.environ.get("SLS_VARIANT"):      # experiment: sls_debug_variant(fwd, bwd)
    from splat_loam_amd import _abi
    _abi.lib().sls_debug_variant(*[int(x) for x in os.environ["SLS_VARIANT"].split(",")])
eng = MappingEngine(model, MappingConfig())
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    st = eng.step(cam)
print(st)
# calibration kernel for tools/pmc_traffic.sh: one stand-alone fused Adam step over 10*N floats
# (16 B read + 12 B written per element, known exactly)
from splat_loam_amd.optim import FusedAdam
p = torch.nn.Parameter(torch.zeros(10 * N, device="cuda:0"))
p.grad = torch.ones_like(p)
opt = FusedAdam([{"params": [p], "lr": 1e-3, "name": "calib"}], lr=1e-3, eps=1e-15)
opt.step()
torch.cuda.synchronize()
