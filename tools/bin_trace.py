#!/usr/bin/env python3
"""Experiment (needs a -DSLS_TRACE build: tools/build_variant.sh trace ...): where the direct binning's kernels spend
their time — per wave of bin_direct_kernel the 100 MHz wall clock at start / records loaded / counted / cursors ready /
end, its rounds and instances; per workgroup of the counting merge start / network done / counted / table stored.
python tools/bin_trace.py [N H W]   (the library under test must be splat_loam_amd/libsls_hip.so)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from splat_loam_amd import _abi, synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel
N, H, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (500000, 64, 2048)
sc = synth.make_scene(N, H, W, seed=0)
depth, valid = synth.make_targets(H, W, sc)
cam = Camera(sc["K"], depth, None, valid, None, data_device="cuda:0")
m = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cuda:0")
e = MappingEngine(m, MappingConfig())
for _ in range(6):
    e.step(cam)
torch.cuda.synchronize()
lib = _abi.lib()
bb = (C.c_uint32 * (32768 * 8))(); mb = (C.c_uint32 * (1024 * 4))()
lib.sls_debug_read_bin_trace.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sls_debug_read_bin_trace(bb, mb) == 0
b = np.frombuffer(bb, dtype=np.uint32).reshape(-1, 8).astype(np.int64)
b = b[b[:, 4] != 0]
t0 = b[:, 0].min()
us = lambda x: (x - t0) * 0.01
print(f"== {N} surfels {H}x{W}: bin_direct_kernel, {len(b)} waves; span {us(b[:, 4].max()):.1f} us")
ph = np.stack([b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2], b[:, 4] - b[:, 3]], 1) * 0.01
for name, col in zip(("loads + scan + first barrier", "count pass", "digit bases + cursors (2 barriers)", "emit rounds"), ph.T):
    print(f"  {name:38s} mean {col.mean():6.2f}  p50 {np.median(col):6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f} us")
rounds, S = b[:, 5], b[:, 6]
print(f"  rounds per wave mean {rounds.mean():.2f} p90 {np.percentile(rounds, 90):.0f} max {rounds.max()}; instances {S.sum()}")
heavy = rounds >= 8
if heavy.any():
    print(f"  waves with >= 8 rounds: {int(heavy.sum())}: emit {ph[heavy, 3].mean():.2f} us = {ph[heavy, 3].sum() / rounds[heavy].sum():.3f} us per round; "
          f"start {us(b[heavy, 0]).mean():.1f}, end mean {us(b[heavy, 4]).mean():.1f} max {us(b[heavy, 4]).max():.1f} us")
light = rounds <= 2
print(f"  waves with <= 2 rounds: {int(light.sum())}: start mean {us(b[light, 0]).mean():.1f} max {us(b[light, 0]).max():.1f}, end mean {us(b[light, 4]).mean():.1f} max {us(b[light, 4]).max():.1f} us; "
      f"emit {ph[light, 3].mean():.2f} us")
print("  waves running every 2 us:", [int(((us(b[:, 0]) <= t) & (us(b[:, 4]) > t)).sum()) for t in np.arange(0, us(b[:, 4].max()), 2.0)])
last = np.argsort(b[:, 4])[-8:]
print("  last waves to end: " + "; ".join(f"start {us(b[i, 0]):.1f} end {us(b[i, 4]):.1f} rounds {rounds[i]} phases {np.round(ph[i], 1).tolist()}" for i in last))
mt = np.frombuffer(mb, dtype=np.uint32).reshape(-1, 4).astype(np.int64)
mt = mt[mt[:, 3] != 0]
if len(mt):
    m0 = mt[:, 0].min()
    d = np.stack([mt[:, 1] - mt[:, 0], mt[:, 2] - mt[:, 1], mt[:, 3] - mt[:, 2]], 1) * 0.01
    print(f"== counting merge: {len(mt)} workgroups, span {(mt[:, 3].max() - m0) * 0.01:.1f} us; start spread {(mt[:, 0].max() - m0) * 0.01:.1f} us")
    for name, col in zip(("load + merge network", "gather + count + records", "count table column"), d.T):
        print(f"  {name:28s} mean {col.mean():6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f} us")
