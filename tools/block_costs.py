#!/usr/bin/env python3
"""Per-block cost data of the tile backward (GPU): the forward's compact-list counts per 16-pixel block and the shader
clocks each backward wave took (sls_debug_wave_cycles), for offline scheduling simulations (tools/sim_schedule.py).
    python tools/block_costs.py N H W out.npz"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from splat_loam_amd import _abi, synth
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_backward, rasterize_forward

N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
sc = synth.make_scene(N, H, W, seed=0)
view, proj = synth.camera_matrices(sc["K"])
settings = GaussianRasterizationSettings(H, W, 1.0, torch.tensor(view, device=dev), torch.tensor(proj, device=dev))
t = {k: torch.tensor(sc[k], device=dev) for k in ("means", "scales", "rots", "opac")}
T = ((H + 15) // 16) * ((W + 15) // 16)
fwd_c = torch.zeros(T * 16 + 64, dtype=torch.int32, device=dev)
bwd_c = torch.zeros(T * 16 + 64, dtype=torch.int32, device=dev)
lib = _abi.lib()
dL = torch.randn(7, H, W, device=dev)
dL[5:] = 0
for it in range(3):
    if it == 2:
        lib.sls_debug_wave_cycles(C.c_void_p(0), C.c_void_p(bwd_c.data_ptr()))
    st = rasterize_forward(settings, t["means"], t["opac"], t["scales"], t["rots"])
    rasterize_backward(st, t["means"], t["scales"], t["rots"], dL)
torch.cuda.synchronize()
lib.sls_debug_wave_cycles(C.c_void_p(0), C.c_void_p(0))
bm = st.block_masks.cpu().numpy().view(np.uint32)
counts = bm[2:2 + T * 16].astype(np.int64)
ranges = st.ranges.cpu().numpy().reshape(T, 2) if hasattr(st, "ranges") else None
np.savez_compressed(sys.argv[4], counts=counts, bwd_cycles=bwd_c.cpu().numpy()[:T * 16], T=T, R=int(st.R),
                    ranges=ranges if ranges is not None else np.zeros(0))
print("blocks", T * 16, "entries", counts.sum(), "max", counts.max(), "mean", counts.mean(),
      "cycles mean", bwd_c.float().mean().item(), "max", bwd_c.max().item())
