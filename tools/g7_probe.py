#!/usr/bin/env python3
"""How far do 21 Adam iterations on the GPU land from the reference's (golden G7, keyframe k) — through the engine, through
the drop-in autograd path with torch's Adam, and run to run?  (Adam with eps = 1e-15 turns a gradient's sign into a step
of +-lr whatever its size: entries whose gradient cancels to ~0 are chaotic.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_fused_mapper as t
from splat_loam_amd import fused_mapper, slam_rules
from splat_loam_amd.mapping import MappingConfig, optimize_step

g = t._g7(); lrs = tuple(float(v) for v in g["lr"]); c = g["cfg"]; dev = "cuda:0"
mc = MappingConfig(opt_lambda_alpha=float(c[3]), opt_lambda_normal=float(c[4]), opt_scaling_max=float(c[5]), opt_scaling_max_penalty=float(c[6]))
frames = [t._frame(g, k, dev) for k in range(3)]


def stats(tag, got, want, start):
    for name, cols in t.COLS.items():
        moved = np.abs(want[:, cols] - start[:, cols]).max()
        e = np.abs(got[:, cols] - want[:, cols]) / moved
        print(f"  {tag:28s} {name:10s} max {e.max():.2e}  p99.9 {np.quantile(e, 0.999):.2e}  p99 {np.quantile(e, 0.99):.2e}  median {np.median(e):.2e}  "
              f"entries > 1e-2: {(e > 1e-2).sum()} of {e.size}")


for k in range(3):
    start = np.concatenate([np.zeros((0, 10), np.float32) if k == 0 else t._survivors(g, k - 1), g[f"added_k{k}"]])
    want = g[f"after_optimize_k{k}"]
    print(f"keyframe {k}: {start.shape[0]} surfels")
    runs = {}
    for tag in ("engine", "engine again", "engine deterministic", "autograd + torch Adam"):
        model = t._model(start, dev, lrs, fused=False)
        np.random.seed(100 + k)
        if tag.startswith("engine"):
            cfg = t._cfg(g)
            eng = fused_mapper._engine_for(model, cfg.mapping, 0.0)
            eng.deterministic = tag.endswith("deterministic")
            fused_mapper.fused_optimize(model, frames[:k + 1], cfg)
        else:
            p = slam_rules.keyframe_probabilities(k + 1, float(c[2]))
            for _ in range(21):
                optimize_step(model, frames[int(np.random.choice(k + 1, p=p))].camera, mc)
        runs[tag] = t._rows(model)
        stats(tag + " vs G7", runs[tag], want, start)
    stats("engine vs engine again", runs["engine"], runs["engine again"], start)
    stats("engine vs autograd", runs["engine"], runs["autograd + torch Adam"], start)
    op = 1 / (1 + np.exp(-runs["engine"][:, 3])); opw = 1 / (1 + np.exp(-want[:, 3]))
    thr = float(g[f"prune_threshold_k{k}"])
    print(f"  prune decisions that differ (engine vs G7): {int(((op < thr) != (opw < thr)).sum())}; |opacity difference| max {np.abs(op - opw).max():.2e}, "
          f"entries > 1e-3: {(np.abs(op - opw) > 1e-3).sum()}")
