#!/usr/bin/env python3
"""bench_extras.py — the secondary measurements behind bench.py's `extras` (1 GPU; the headline stays the driver's
command).  Called by bench.py unless --no-extras; `python bench_extras.py [names...]` runs them alone and prints one JSON
object.  Every figure is a whole mapping iteration (or what its key says) on the bench's scene and keyframe sampling
unless stated otherwise.

  single_keyframe      one keyframe re-rendered every iteration (rounds 1-2's headline)
  uniform_keyframes    the window's keyframes drawn uniformly (prob_view_last_keyframe: null, as half the reference's configs)
  full_sort            depth order sorted from scratch every iteration
  iterations_400_800   400 timed iterations after 400 un-timed ones (the optimisation changes the workload)
  deterministic        integer accumulation of the gradient records: two launches / one launch with predicted scales
  real_sizes           the sizes the reference's mapper meets: 50 k and 170 k surfels at 64x1024
  sparse_union         size of the union of the touched sets for G = 2, 4, 8 ranks (what dp_mode "sparse" moves)
  dropin               what an UNMODIFIED slam/mapper.py gets: GaussianRasterizer under torch autograd (sls_forward_ws /
                       sls_backward_ws; `staged`: sls_forward_stage1/2 + sls_backward with the host read of R), the
                       reference-shaped torch glue around it, the HIP loss consumer instead of the glue, and
                       `hooked_mapper`: fused_mapper.fused_optimize — what SLS_FUSED_MAPPER=1 binds Mapper.optimize to
  dp_world1            the keyframe-parallel iteration through RCCL in a ONE-rank group (collectives move a rank's data
                       onto itself): what the exchange adds on one GPU, per scheme; no scaling figure
  update_model         the reference's unit of work, Mapper.update_model (densify -> 201 iterations -> prune) per keyframe
                       on a C4-sized local model through fused_mapper.update_model: wall time split into its stages
                       against 201 x the steady-state iteration
"""
from __future__ import annotations

import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

from bench_support import log


def _timed(dev, fn, n_w, n_t):
    """ms per call: the best of five timed runs of n_t calls (host-bound loops: one allocator or scheduler hiccup in a
    run of a hundred calls would otherwise be the figure; between boxes the host-bound rows still differ by a factor)"""
    for _ in range(n_w):
        fn()
    best = float("inf")
    for _ in range(5):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_t):
            fn()
        torch.cuda.synchronize(dev)
        best = min(best, (time.perf_counter() - t0) / n_t * 1e3)
    return best


def single_keyframe(c):
    m, e = c.fresh()
    d, _ = c.run(m, e, [c.cams[0]], c.args.warmup * c.ips, c.n_iters)
    return {"ms_per_iteration": round(d / c.n_iters * 1e3, 4), "Msplats_per_s": round(c.N / (d / c.n_iters) / 1e6, 1),
            "repeated_iterations": dict(e.stats)}


def uniform_keyframes(c):
    """The window's keyframes drawn UNIFORMLY — `prob_view_last_keyframe: null`, which half of the reference's configs set
    (configs/kitti/kitti-00-odom.yaml:14, ncd/quad-easy-*.yaml:14; slam/mapper.py:142-149) — instead of the geometric
    draw of kitti.yaml / ncd.yaml the headline uses: every keyframe's depth order is ~8 iterations old when it is visited
    again, and the window's far end (surfels next to the sensor: the heavy keyframes) is visited as often as its near end."""
    m, e = c.fresh()
    pick = np.random.default_rng(4).integers(0, c.n_kf, size=(c.args.warmup * c.ips + c.n_iters) * 2) if c.n_kf > 1 else None
    d, _ = c.run(m, e, c.window, c.args.warmup * c.ips, c.n_iters, pick=pick)
    return {"ms_per_iteration": round(d / c.n_iters * 1e3, 4), "Msplats_per_s": round(c.N / (d / c.n_iters) / 1e6, 1),
            "repeated_iterations": dict(e.stats)}


def full_sort(c):
    m, e = c.fresh(full_sort=True)
    d, _ = c.run(m, e, c.window, c.args.warmup * c.ips, c.n_iters, pick=c.pick_rank)
    return {"ms_per_iteration": round(d / c.n_iters * 1e3, 4)}


def iterations_400_800(c):
    m, e = c.fresh()
    pick = np.random.default_rng(1).choice(c.n_kf, size=800, p=c.kf_p) if c.n_kf > 1 else None
    d, _ = c.run(m, e, c.window, 400, 400, pick=pick)
    return {"ms_per_iteration": round(d / 400 * 1e3, 4), "repeated_iterations": dict(e.stats)}


def deterministic(c):
    out = {"float_atomics_ms_per_iteration": round(c.ms_per_iter, 4)}
    for name, mode in (("two_launches", True), ("one_launch_predicted_scales", 2)):
        m, e = c.fresh()
        e.deterministic = mode
        d, _ = c.run(m, e, c.window, c.args.warmup * c.ips, c.n_iters, pick=c.pick_rank)
        out[name + "_ms_per_iteration"] = round(d / c.n_iters * 1e3, 4)
        out[name + "_repeated"] = dict(e.stats)
    return out


def _scene_at(c, n2, h2, w2):
    from splat_loam_amd import synth
    if (n2, h2, w2) == (c.N, c.H, c.W):
        return c.scene, c.depth, c.valid
    sc2 = synth.make_scene(n2, h2, w2, seed=0)
    d2, v2 = synth.make_targets(h2, w2, sc2)
    return sc2, d2, v2


def real_sizes(c):
    from splat_loam_amd.engine import MappingEngine
    from splat_loam_amd.scene import Camera, SurfelModel
    out = {}
    for n2, h2, w2 in ((50_000, 64, 1024), (170_000, 64, 1024)):
        sc2, d2, v2 = _scene_at(c, n2, h2, w2)
        cams2 = [Camera(sc2["K"], d2, None, v2, c.poses[k], data_device=str(c.dev)) for k in range(c.n_kf)]
        res = {}
        for name, cs, pk in (("single_keyframe", [cams2[0]], None),
                             ("sampled_keyframes", cams2, np.random.default_rng(2).choice(c.n_kf, size=600, p=c.kf_p) if c.n_kf > 1 else None)):
            mdl = SurfelModel.from_activated(sc2["means"], sc2["scales"], sc2["rots"], sc2["opac"], device=str(c.dev))
            mdl.training_setup(fused=True)
            d, _ = c.run(mdl, MappingEngine(mdl, c.cfg), cs, 100, 400, pick=pk)
            res[name + "_ms_per_iteration"] = round(d / 400 * 1e3, 4)
        out[f"{n2}_{h2}x{w2}"] = res
    return out


def sparse_union(c):
    """rows = surfels with a non-zero gradient on at least one of the G ranks (rank 0 carries the scale regulariser);
    window: rank g renders keyframe g (BASELINE config 5); sampled: every rank draws its keyframe with the mapper's
    probabilities, 50 draws; bytes = 40 B per row SUM-reduced + the G bitmaps all-gathered; ONE GPU, no collective"""
    from splat_loam_amd.fused import fused_loss
    from splat_loam_amd.scene import SurfelModel
    N = c.N
    m = SurfelModel.from_activated(c.scene["means"], c.scene["scales"], c.scene["rots"], c.scene["opac"], device=str(c.dev))
    with torch.no_grad():      # the model as the timed iterations left it
        for dst, src in zip((m._xyz, m._scaling, m._rotation, m._opacity),
                            (c.model._xyz, c.model._scaling, c.model._rotation, c.model._opacity)):
            dst.copy_(src)
    params = (m._xyz, m._opacity, m._scaling, m._rotation)
    sets = []
    for k in range(min(8, c.n_kf)):
        for p_ in params:
            p_.grad = None
        fused_loss(m, c.cams[k], c.cfg, with_regulariser=(k == 0)).backward()
        sets.append(torch.cat([p_.grad.reshape(N, -1) for p_ in params], dim=1).ne(0).any(dim=1))
    out = {"per_keyframe_rows": [int(x.sum().item()) for x in sets]}
    rng_u = np.random.default_rng(3)
    for G in (2, 4, 8):
        if G > len(sets):
            continue
        rows = int(torch.stack(sets[:G]).any(dim=0).sum().item())
        p = c.kf_p[:len(sets)] / c.kf_p[:len(sets)].sum()
        drawn = [int(torch.stack([sets[int(k_)] for k_ in rng_u.choice(len(sets), size=G, p=p)]).any(dim=0).sum().item())
                 for _ in range(50)]
        out[f"G{G}"] = {"union_rows_window": rows, "bytes_per_rank_window": 40 * rows + G * ((N + 63) // 64) * 8,
                        "union_rows_sampled_mean": int(np.mean(drawn)), "union_rows_sampled_max": int(np.max(drawn)),
                        "dense_bytes_per_rank": 40 * N}
    return out


def _dropin_at(c, n2, h2, w2, iters, warm, rows_only=False):
    from splat_loam_amd import fused_mapper, rasterizer
    from splat_loam_amd.mapping import optimize_step, optimize_step_fused
    from splat_loam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from splat_loam_amd.scene import Camera, SurfelModel
    dev, lib = c.dev, c.lib
    sc2, d2, v2 = _scene_at(c, n2, h2, w2)
    cam2 = Camera(sc2["K"], d2, None, v2, c.poses[0], data_device=str(dev))
    new_model = lambda: SurfelModel.from_activated(sc2["means"], sc2["scales"], sc2["rots"], sc2["opac"], device=str(dev))

    def rasterizer_alone(lean, staged):
        """forward (no_grad) and forward + backward through autograd, dL/dallmap given"""
        os.environ["SLS_STAGED_FORWARD"] = "1" if staged else "0"
        try:
            mdl = new_model()
            with torch.no_grad():
                leaves = [t.detach().clone().requires_grad_(True) for t in (mdl.get_xyz, mdl.get_opacity, mdl.get_scaling, mdl.get_rotation)]
            rast = GaussianRasterizer(raster_settings=GaussianRasterizationSettings(
                h2, w2, 1.0, cam2.world_view_transform, cam2.projection_matrix, False, False, lean_allmap=lean))
            dL = torch.randn((7, h2, w2), device=dev)
            if lean:
                dL[5:7] = 0

            def fb():
                for t in leaves:
                    t.grad = None
                _, am = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), opacities=leaves[1],
                             scales=leaves[2], rotations=leaves[3], cov3D_precomp=None)
                am.backward(dL)

            def fwd_only():
                with torch.no_grad():
                    rast(means3D=leaves[0], means2D=leaves[0], opacities=leaves[1], scales=leaves[2], rotations=leaves[3])
            # (forward + backward FIRST, behind a long warm-up: a mapper calls backward() every iteration, and torch's
            #  autograd device thread answers slowly — +0.05…0.1 ms per call — for a while after a stretch without one)
            t_fb = _timed(dev, fb, 4 * warm, iters)
            res = {"rasterizer_fwd_ms": round(_timed(dev, fwd_only, warm, iters), 4), "rasterizer_fwd_bwd_ms": round(t_fb, 4)}
            lib.sls_timing_enable(1)
            for _ in range(10):
                fb()
            torch.cuda.synchronize(dev)
            res["kernels_us"] = {k: round(ms / n * 1e3, 2) for k, (ms, n) in c.collect().items()}
            lib.sls_timing_enable(0)
            return res
        finally:
            os.environ.pop("SLS_STAGED_FORWARD", None)

    def iterations(lean):
        """a whole iteration of Mapper.optimize on one keyframe (render() builds its settings itself, as
        gaussian_renderer/__init__.py does: the lean kernels are chosen by the process default)"""
        res = {}
        os.environ["SLS_LEAN_ALLMAP"] = "1" if lean else "0"
        try:
            for name, fn in (("iteration_torch_glue_ms", lambda m: float(optimize_step(m, cam2, c.cfg))),
                             ("iteration_hip_consumer_ms", lambda m: float(optimize_step_fused(m, cam2, c.cfg)))):
                mdl = new_model()
                mdl.training_setup(fused=True)
                res[name] = round(_timed(dev, lambda: fn(mdl), warm, iters), 4)
        finally:
            os.environ.pop("SLS_LEAN_ALLMAP", None)
        res["Msplats_per_s_torch_glue"] = round(n2 / (res["iteration_torch_glue_ms"] * 1e-3) / 1e6, 1)
        return res

    rasterizer_alone(False, False)       # (discarded: whatever the process does once — module loads, allocator growth — lands here)
    if rows_only:
        out = {"all_planes": rasterizer_alone(False, False), "lean_allmap": rasterizer_alone(True, False)}
        rasterizer._WS_CACHE.clear()
        return out
    out = {"all_planes": {**rasterizer_alone(False, False), **iterations(False)},
           "lean_allmap": {**rasterizer_alone(True, False), **iterations(True)},
           "staged_all_planes": rasterizer_alone(False, True)}
    # what SLS_FUSED_MAPPER=1 binds Mapper.optimize to: fused_optimize on the model's own optimizer state, one keyframe,
    # 200 iterations per call (configs/kitti/kitti-00-odom.yaml) — next to the engine driven directly
    mdl = new_model()
    mdl.training_setup(fused=True)
    frames = [SimpleNamespace(camera=cam2)]
    kcfg = SimpleNamespace(mapping=SimpleNamespace(num_iterations=199, prob_view_last_keyframe=0.4,
                                                   opt_lambda_alpha=c.cfg.opt_lambda_alpha, opt_lambda_normal=c.cfg.opt_lambda_normal,
                                                   opt_scaling_max=c.cfg.opt_scaling_max, opt_scaling_max_penalty=c.cfg.opt_scaling_max_penalty),
                           opt=SimpleNamespace(depth_ratio=c.cfg.depth_ratio))
    hooked = _timed(dev, lambda: fused_mapper.fused_optimize(mdl, frames, kcfg), 1, 2) / 200
    from splat_loam_amd.engine import MappingEngine
    mdl2 = new_model()
    eng = MappingEngine(mdl2, c.cfg)

    def direct():
        for _ in range(200):
            eng.step(cam2, sync="lagged")
        eng.flush()
    engine_ms = _timed(dev, direct, 1, 2) / 200
    out["hooked_mapper"] = {"ms_per_iteration": round(hooked, 4), "engine_ms_per_iteration": round(engine_ms, 4),
                            "ratio": round(hooked / engine_ms, 3), "Msplats_per_s": round(n2 / (hooked * 1e-3) / 1e6, 1)}
    rasterizer._WS_CACHE.clear()
    del mdl, mdl2, eng
    torch.cuda.empty_cache()     # (the next size starts from a clean allocator: no 440 MB blocks to carve 4 MB tensors from)
    return out


def _pin_threads(cores):
    """Every thread of this process onto `cores` (None: back to what each had).  The drop-in rows are a ping-pong between
    the Python thread and torch's autograd device thread; on a 256-core host the two land on cores that wake each other
    in 10 us or in 100, at random per process — the rows were bimodal (0.18 / 0.28 ms at C3).  Four neighbouring cores
    keep the measurement about the path, not about the scheduler."""
    saved = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            saved[int(tid)] = os.sched_getaffinity(int(tid))
            os.sched_setaffinity(int(tid), cores if not isinstance(cores, dict) else cores.get(int(tid), saved[int(tid)]))
        except OSError:
            pass
    return saved


def dropin(c):
    """The reported rows are UNPINNED — what a user's process gets (round 6: backward() on the calling thread by default
    with one GPU, rasterizer._autograd_policy); `pinned_to_four_cores` repeats the rasterizer-alone rows with every thread
    of the process on four neighbouring cores, round 5's way of taking them."""
    out = {f"{c.N}_{c.H}x{c.W}": _dropin_at(c, c.N, c.H, c.W, 30, 10), "50000_64x1024": _dropin_at(c, 50_000, 64, 1024, 50, 20)}
    out["autograd_multithreading"] = bool(torch.autograd.is_multithreading_enabled())
    saved = None
    try:
        mine = sorted(os.sched_getaffinity(0))
        saved = _pin_threads(set(mine[:4]))
    except (AttributeError, OSError):
        pass
    try:
        if saved:
            pinned = _dropin_at(c, c.N, c.H, c.W, 30, 10, rows_only=True)
            out["pinned_to_four_cores"] = {"cores": sorted(set(mine[:4])), f"{c.N}_{c.H}x{c.W}": pinned}
    finally:
        if saved:
            _pin_threads(saved)
    return out


def dp_world1(c):
    """The keyframe-parallel iteration with its collectives issued through RCCL in a ONE-rank group: the exchange's
    launches, copies and the separate Adam on one GPU — DESIGN.md section 6's "+17 us" term as a measured number.  No
    link is crossed: NOT a scaling figure."""
    import torch.distributed as dist
    import socket
    if dist.is_initialized():
        return {"skipped": "a process group exists already"}
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=c.dev)
    out = {"single_gpu_ms_per_iteration": round(c.ms_per_iter, 4)}
    try:
        for name, mode in (("sparse", "sparse"), ("allreduce", "allreduce"), ("rs_ag", "rs_ag")):
            m, e = c.fresh(dp_mode=mode)
            e.exchange_at_world_1 = True
            d, _ = c.run(m, e, c.window, c.args.warmup * c.ips, c.n_iters, pick=c.pick_rank)
            out[name + "_ms_per_iteration"] = round(d / c.n_iters * 1e3, 4)
            out[name + "_bytes_per_rank"] = int(e.exchanged_bytes)
            del m, e
    finally:
        dist.destroy_process_group()
    return out


def update_model(c, n0=150_000, h2=128, w2=1024, n_kf=8, new_keyframes=4, num_iterations=200):
    """The reference's unit of work: `Mapper.update_model` (slam/mapper.py:33-47) = densify -> num_iterations + 1 iterations
    -> prune, per keyframe, through fused_mapper.update_model on a C4-sized local model (NCD 128x1024,
    <= 150 k surfels per local model, configs/ncd: densify 15 % of the candidates).  Wall time per keyframe, split into
    its stages (device synchronised between them), against (num_iterations + 1) x the steady-state iteration."""
    from splat_loam_amd import fused_mapper, synth
    from splat_loam_amd.engine import MappingEngine
    from splat_loam_amd.renderer import depth_to_points
    from splat_loam_amd.scene import Camera, SurfelModel
    dev = str(c.dev)
    sc2 = synth.make_scene(n0, h2, w2, seed=0)
    d2, v2 = synth.make_targets(h2, w2, sc2)
    poses = synth.keyframe_poses(n_kf + new_keyframes)

    def frame(k):
        cam = Camera(sc2["K"], d2, None, v2, poses[k], data_device=dev)
        pts = depth_to_points(cam, cam.image_depth)                   # sensor frame: the measured normal faces the sensor
        cam.image_normal = (-pts / pts.norm(dim=0, keepdim=True).clamp_min(1e-9)).contiguous()
        return SimpleNamespace(camera=cam, model_T_frame=torch.tensor(poses[k], dtype=torch.float32, device=dev))
    mapping = SimpleNamespace(num_iterations=num_iterations, densify_threshold_egeom=-1.0, densify_threshold_opacity=0.5,
                              densify_percentage=0.15, prob_view_last_keyframe=0.4, pruning_min_opacity=0.0,
                              pruning_min_size=0.0, opt_lambda_alpha=c.cfg.opt_lambda_alpha,
                              opt_lambda_normal=c.cfg.opt_lambda_normal, opt_scaling_max=c.cfg.opt_scaling_max,
                              opt_scaling_max_penalty=c.cfg.opt_scaling_max_penalty)
    cfg = SimpleNamespace(mapping=mapping, opt=SimpleNamespace(depth_ratio=0.0))
    model = SurfelModel.from_activated(sc2["means"], sc2["scales"], sc2["rots"], sc2["opac"], device=dev)
    model.training_setup(fused=True)
    frames = [frame(k) for k in range(n_kf + new_keyframes)]
    keyframes = frames[:n_kf]
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    np.random.seed(0)
    # warm-up: two whole updates (allocator, kernels' first launches — the first update of a process pays ~0.25 s of them), not reported
    for _ in range(2):
        fused_mapper.update_model(model, keyframes, keyframes[-1], cfg, generator=gen)
    rows = []
    for k in range(n_kf, n_kf + new_keyframes):
        keyframes = keyframes[1:] + [frames[k]]
        torch.cuda.synchronize(c.dev)
        t0 = time.perf_counter()
        res = fused_mapper.update_model(model, keyframes, frames[k], cfg, generator=gen, timings=True)
        torch.cuda.synchronize(c.dev)
        wall = (time.perf_counter() - t0) * 1e3
        tm = res["timings_ms"]
        rows.append({"N": int(model._xyz.shape[0]), "added": int(res["added"]), "wall_ms": round(wall, 3),
                     **{kk: round(vv, 3) for kk, vv in tm.items()}})
    # the steady-state iteration on the same model and window (engine alone, keyframes drawn as the mapper draws them)
    eng = MappingEngine(model, fused_mapper.engine_of(model).cfg)
    pick = np.random.default_rng(3).choice(n_kf, size=600, p=c.kf_p) if n_kf == c.n_kf else None
    d, _ = c.run(model, eng, [f.camera for f in keyframes], 200, 400, pick=pick)
    it_ms = d / 400 * 1e3
    mean_wall = float(np.mean([r["wall_ms"] for r in rows]))
    return {"workload": f"{n0} surfels + 15 % of the candidates per keyframe, {h2}x{w2}, window of {n_kf}, num_iterations {num_iterations}",
            "per_keyframe": rows, "steady_state_ms_per_iteration": round(it_ms, 4),
            "iterations_alone_ms": round((num_iterations + 1) * it_ms, 3), "mean_wall_ms": round(mean_wall, 3),
            "wall_over_iterations": round(mean_wall / ((num_iterations + 1) * it_ms), 4)}


ALL = {"single_keyframe": single_keyframe, "uniform_keyframes": uniform_keyframes, "full_sort": full_sort, "iterations_400_800": iterations_400_800,
       "deterministic": deterministic, "real_sizes": real_sizes, "sparse_union": sparse_union, "dropin": dropin,
       "dp_world1": dp_world1, "update_model": update_model}


def collect_extras(c, names=None):
    out = {}
    for name in (names or ALL):
        try:
            out[name] = ALL[name](c)
        except Exception as e:      # (a report, never a reason to lose the bench line)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        log(f"extras: {name} done")
    out["note"] = "keys: bench_extras.py's docstring"
    return out


if __name__ == "__main__":
    import bench
    names = [a for a in sys.argv[1:] if a in ALL]
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a not in ALL]
    bench.main(extras_only=names or list(ALL))
