#!/usr/bin/env python3
"""How much of the engine-vs-checker difference at BASELINE config 3 is float32 accumulation noise?  Runs the engine
(float atomics x3, deterministic x1) on the bench scene and the checker chain in float32 and float64, prints the
max-norm relative differences per tensor between every pair.  python tools/c3_noise.py [N H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import tangent
import test_timed_path as T
import oracle.torch_function as otf
from oracle.oracle import Oracle
from splat_loam_amd import synth
from splat_loam_amd.mapping import MappingConfig

N, H, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (500000, 64, 2048)
dev = torch.device("cuda:0")
o32 = Oracle(np.float32); o32.set_threads(o32.max_threads()); otf.BACKWARD_THREADS = o32.max_threads()
sc, raw, depth, valid = T._raw_scene(N, H, W, seed=0)
pose = synth.keyframe_poses(2)[1]
view, proj = synth.camera_matrices(sc["K"], pose)
cfg = MappingConfig()
runs = {}
for name, det in (("atomics#0", False), ("atomics#1", False), ("atomics#2", False), ("deterministic", True)):
    os.environ["SLS_DETERMINISTIC"] = "1" if det else "0"
    st, g, am, eng, model, cam = T._engine_once(dev, raw, sc["K"], pose, depth, valid, cfg, 0)
    runs[name] = g
    del eng, model
refs = {}
for name, dt in (("checker32", np.float32), ("checker64", np.float64)):
    r = T.reference_iteration(raw, sc["K"], view, proj, H, W, depth[0], valid[0] == 1, cfg, allmap_value=am, dtype=dt)
    refs[name] = {k: v.astype(np.float64) for k, v in r["grads"].items()}
allg = {**{k: {kk: vv.astype(np.float64) for kk, vv in v.items()} for k, v in runs.items()}, **refs}
rot = raw["rotation"].astype(np.float64)
names = list(allg)
for k in ("xyz", "opacity", "scaling", "rotation"):
    print(f"== d{k}: max-norm relative difference (to the float64 checker's max)")
    f = (lambda a: tangent(a, rot)) if k == "rotation" else (lambda a: a)
    scale = np.abs(f(allg["checker64"][k])).max()
    for i, a in enumerate(names):
        print(f"  {a:14s}", "  ".join(f"{np.abs(f(allg[a][k]) - f(allg[b][k])).max() / scale:8.1e}" for b in names[:i + 1]))
    d = np.abs(f(allg["atomics#0"][k]) - f(allg["checker64"][k]))
    i = np.unravel_index(d.argmax(), d.shape)
    print(f"  worst element of atomics#0 vs checker64: surfel {i[0]} comp {i[1]}: {f(allg['atomics#0'][k])[i]:.6e} vs {f(allg['checker64'][k])[i]:.6e}"
          f" (|max| {scale:.3e}); its range {np.linalg.norm(raw['xyz'][i[0]]):.2f} m, scale {np.exp(raw['scaling'][i[0]])}")
