#!/usr/bin/env python3
"""Soak run of MappingEngine: many lagged iterations over a window of keyframes sampled at random (the
reference's schedule, slam/mapper.py:143-156), with a deliberately small instance capacity at the start so that
the grow-and-repeat path, the depth-order repair / fallback path and the lagged status protocol all get
exercised together.  Checks: finite parameters, decreasing loss EMA, every iteration accounted for.

    python tools/soak.py [N H W n_keyframes n_iterations max_order_age]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splat_loam_amd import synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel


def run(N=60000, H=64, W=1024, n_kf=5, n_iter=600, max_age=None, seed=0, dev="cuda:0", verbose=True):
    sc = synth.make_scene(N, H, W, seed=seed)
    depth, valid = synth.make_targets(H, W, sc)
    poses = synth.keyframe_poses(n_kf)
    cams = [Camera(sc["K"], depth, None, valid, poses[k], data_device=dev) for k in range(n_kf)]
    model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device=dev)
    eng = MappingEngine(model, MappingConfig())
    if max_age is not None:
        eng.max_order_age = max_age
    eng.capacity = N // 2                      # too small: the first iterations must grow and repeat
    rng = np.random.default_rng(seed)
    prob = np.array([0.4] + [0.6 / (n_kf - 1)] * (n_kf - 1)) if n_kf > 1 else np.array([1.0])
    ema, first, seen = None, None, 0
    t0 = time.perf_counter()
    for it in range(n_iter):
        if it == n_iter // 2:
            torch.cuda.synchronize(); t_half = time.perf_counter()
        st = eng.step(cams[rng.choice(n_kf, p=prob)], sync="lagged")
        if st is not None:
            seen += 1
            assert not st["overflow"] and np.isfinite(st["loss"])
            ema = st["loss"] if ema is None else 0.1 * st["loss"] + 0.9 * ema
            first = ema if seen == 20 else first
    st = eng.flush(); seen += len(eng.flushed)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert seen == n_iter and eng.t == n_iter, (seen, eng.t, n_iter)
    for p in (model._xyz, model._scaling, model._rotation, model._opacity):
        assert bool(torch.isfinite(p).all())
    assert eng.stats["repeated_too_small"] >= 1
    assert ema < first, (first, ema)
    if verbose:
        print(f"{n_iter} iterations over {n_kf} keyframes in {dt * 1e3:.0f} ms ({dt / n_iter * 1e3:.3f} ms each, second half {(time.perf_counter() - t_half) / (n_iter - n_iter // 2) * 1e3:.3f}), "
              f"loss EMA {first:.4f} -> {ema:.4f}, capacity {eng.capacity}, stats {eng.stats}")
    return eng.stats


if __name__ == "__main__":
    run(*[int(a) for a in sys.argv[1:]])
