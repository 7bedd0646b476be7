#!/usr/bin/env python3
"""Lane-use statistics of the tile forward (DBG instantiation through sls_debug_wave_cycles): entries staged, entries past
the block cull, steps, live lanes, entry slots with a live lane, entry slots with a geometric hit.
    python tools/fwd_stats.py N H W"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from splat_loam_amd import _abi, synth
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_forward

N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
sc = synth.make_scene(N, H, W, seed=0)
view, proj = synth.camera_matrices(sc["K"])
settings = GaussianRasterizationSettings(H, W, 1.0, torch.tensor(view, device=dev), torch.tensor(proj, device=dev))
t = {k: torch.tensor(sc[k], device=dev) for k in ("means", "scales", "rots", "opac")}
T = ((H + 15) // 16) * ((W + 15) // 16)
fwd_c = torch.zeros(T * 16 + 64, dtype=torch.int32, device=dev)
lib = _abi.lib()
lib.sls_debug_wave_cycles(C.c_void_p(fwd_c.data_ptr()), C.c_void_p(0))
st = rasterize_forward(settings, t["means"], t["opac"], t["scales"], t["rots"], list_pairs=1)      # (dense rounds at every size: the counters live there)
torch.cuda.synchronize()
lib.sls_debug_wave_cycles(C.c_void_p(0), C.c_void_p(0))
s = fwd_c.cpu().numpy()[T * 16:T * 16 + 6].astype(np.int64)
bm = st.block_masks.cpu().numpy().view(np.uint32)
counts = bm[2:2 + T * 16].astype(np.int64)
staged, passed, steps, lanes, slots, geom = s
print(f"N={N} {H}x{W}: R={st.R} staged {staged} passed-cull {passed} steps {steps} (x4 = {4*steps} entry slots)")
print(f"  live lanes/step {lanes/steps:.1f} of 64; slots with a live lane {slots} ({slots/(4*steps):.2%} of slots, {slots/passed:.2%} of passed)")
print(f"  slots with a geometric hit (finished pixels included) {geom} ({geom/passed:.2%} of passed); compact entries {counts.sum()}")
h = fwd_c.cpu().numpy()[T * 16 + 6:T * 16 + 15].astype(np.int64)
print(f"  finer blocks: contributing entries {h[0]}; (4x2 half, entry) pairs {h[1]} ({h[1]/h[0]:.2f} per entry), (2x2 quarter, entry) pairs {h[2]} ({h[2]/h[0]:.2f})")
print(f"  steps of 4 entries per block, summed: 8x2 blocks {h[3]}; two half-waves in lockstep {h[4]} ({h[4]/h[3]:.2%}); four quarter-waves {h[5]} ({h[5]/h[3]:.2%}); longest block {h[6]} / {h[7]} / {h[8]}")
