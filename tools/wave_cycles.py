#!/usr/bin/env python3
"""Diagnostic: per-wave cycle distribution of the tile kernels on the bench scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from splat_loam_amd import _abi, synth
from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_forward, rasterize_backward
N, H, W = 500000, 64, 2048
dev = torch.device("cuda:0")
sc = synth.make_scene(N, H, W, seed=0)
view, proj = synth.camera_matrices(sc["K"])
s = GaussianRasterizationSettings(H, W, 1.0, torch.tensor(view, device=dev), torch.tensor(proj, device=dev))
t = {k: torch.tensor(sc[k], device=dev) for k in ("means", "scales", "rots", "opac")}
fv = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_abi.lib().sls_debug_variant(fv, bv)
tw, th = _abi.tile_size(); T = (W // tw) * (H // th); wpt = tw * th // 16
f = torch.zeros(T * wpt + 8, dtype=torch.int32, device=dev); b = torch.zeros_like(f)
_abi.lib().sls_debug_wave_cycles(f.data_ptr(), b.data_ptr())
for it in range(2):
    f.zero_()
    st = rasterize_forward(s, t["means"], t["opac"], t["scales"], t["rots"])
    rasterize_backward(st, t["means"], t["scales"], t["rots"], torch.randn(7, H, W, device=dev))
torch.cuda.synchronize()
_abi.lib().sls_debug_wave_cycles(None, None)
cons = st.tile_consumed.cpu().numpy().view(np.uint32)
for name, a in (("fwd", f), ("bwd", b)):
    per = tw * th // (16 if (fv if name == 'fwd' else bv) >= 2 else 64)
    c = a.cpu().numpy().astype(np.int64)[:T * per].reshape(T, per)
    tile = c.max(1)
    print(name, "wave cycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max()))
    print(name, "per tile-row max:", tile.reshape(H // th, W // tw).max(1), "mean:", tile.reshape(H // th, W // tw).mean(1).astype(int))
    k = np.argsort(tile)[-5:]
    print(name, "slowest tiles", k, "cycles", tile[k], "consumed", cons[k])
print("consumed mean", cons.mean(), "max", cons.max(), "sum", cons.sum())
per = tw * th // (16 if fv >= 2 else 64)
st = f.cpu().numpy()[T * per:T * per + 6].astype(np.int64)
if fv >= 2:
    print("fwd block waves: staged %d, passed the box cull %d (%.1f%%), steps %d (slot fill %.2f of 4), live lanes per step %.1f"
          % (st[0], st[1], 100.0 * st[1] / st[0], st[2], st[1] / max(st[2], 1), st[3] / max(st[2], 1)))
    print("   (block, surfel) pairs evaluated %d, with >= 1 live pixel %d (%.1f%%), with >= 1 pixel above 1/255 incl. finished ones %d (%.1f%%)"
          % (st[1], st[4], 100.0 * st[4] / max(st[1], 1), st[5], 100.0 * st[5] / max(st[1], 1)))
else:
    print("fwd waves: staged %d, passed the box cull %d (%.1f%%), with >=1 contributing pixel %d (%.1f%% of passed), "
          "contributing lanes per evaluated surfel %.1f" % (st[0], st[1], 100.0 * st[1] / st[0], st[2], 100.0 * st[2] / st[1], st[3] / st[1]))
