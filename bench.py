#!/usr/bin/env python3
"""bench.py — forward+backward throughput of the spherical surfel rasterizer
on MI355X (BASELINE.json: "fwd+bwd Msplats/s @ 500k Gaussians, 64x2048").

One step = one mapping iteration of Splat-LOAM's hot loop (slam/mapper.py:150-204)
for ONE keyframe per GPU on the synthetic scene of SURVEY.md §8d:
    render()  [HIP: preprocess, depth order, binning, tile render]
    + mapper loss + loss.backward()  [HIP: loss stage in the tile backward, preprocess backward]
    + Adam step on the 4 parameter tensors  [HIP: fused Adam]
with every input already resident in HBM.  Nothing is skipped inside the
timed region.  At N GPUs every rank renders its own keyframe of the shared
model and the 40 B/surfel gradients are exchanged over RCCL (weak scaling:
value = n_gpus * N / step time).

    python bench.py                       # 1 GPU, 20 steps, 3 warm-up
    python bench.py --gpus 8              # starts 8 ranks itself (torch.distributed.run, RCCL over xGMI)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3     # the driver's form

Rank 0 prints ONE JSON line.  This file is the headline: scene, timed loop, per-kernel HIP events; the roofline
object, the replayed counters and the CPU baseline leg live in bench_support.py, the secondary measurements
(`extras`) in bench_extras.py.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from bench_support import cpu_baselines, kernel_table, log, roofline_block


def main(extras_only=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=500_000, help="surfels (BASELINE config 3: 500k)")
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip bench_extras.py's secondary measurements")
    ap.add_argument("--no-timing", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--mode", choices=("engine", "fused", "unfused"), default="engine",
                    help="engine: one sls_mapping_step per iteration (default); fused: torch autograd around the HIP "
                         "rasterizer + HIP loss consumer; unfused: torch render() / loss glue")
    ap.add_argument("--status-read", choices=("lagged", "sync", "async"), default="lagged",
                    help="how an iteration's status (loss, R, void bits) reaches the host — the reference reads its loss once "
                         "per iteration (slam/mapper.py:206-209).  lagged: after the NEXT iteration has been enqueued; "
                         "sync: before; async: never")
    ap.add_argument("--async-steps", action="store_true", help="alias of --status-read async")
    ap.add_argument("--full-sort", action="store_true", help="sort the depth order from scratch every iteration")
    ap.add_argument("--dp-mode", choices=("auto", "rs_ag", "allreduce", "sparse"), default=os.environ.get("SLS_DP_MODE", "auto"),
                    help="N > 1: the gradient exchange (DESIGN.md section 6); auto: time all three un-timed, take the fastest")
    ap.add_argument("--iters-per-step", type=int, default=10,
                    help="mapping iterations inside ONE bench step (--steps 20 then times 200 iterations; 20 alone are "
                         "4 ms of GPU time); ms_per_step is a whole step, config.ms_per_iteration one iteration")
    ap.add_argument("--keyframes", type=int, default=8,
                    help="window an iteration draws its keyframe from, as Mapper.optimize does (slam/mapper.py:142-156)")
    ap.add_argument("--prob-view-last-keyframe", type=float, default=0.4, help="configs/kitti/kitti.yaml:22")
    ap.add_argument("--variant", type=int, nargs=2, default=None, metavar=("FWD", "BWD"),
                    help="tuning: tile-kernel pixel-block shapes (sls_debug_variant: 2 = 4x4, 3 = 8x2)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL over xGMI)
        import socket
        import subprocess
        backend = os.environ.get("SLS_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend == "nccl" and ndev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible (SLS_BENCH_BACKEND=gloo "
                             "runs several ranks on one device for a functional check)")
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the ONE JSON line and nothing else: whatever a library prints there (gloo's connection banner,
    # RCCL warnings) goes to stderr — file descriptor 1 is pointed at 2, the line is written to a saved duplicate
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # ranks > devices only in the single-GPU self-test (gloo)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SLS_BENCH_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            if world > ndev:
                raise SystemExit(f"{world} ranks but {ndev} GPUs visible")
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         "(python bench.py --gpus N does it by itself)")

    from splat_loam_amd import _abi, synth
    from splat_loam_amd.mapping import MappingConfig, optimize_step_fused, optimize_step_sharded
    from splat_loam_amd.scene import Camera, SurfelModel

    lib = _abi.lib()
    if args.variant:
        _abi.check(lib.sls_debug_variant(args.variant[0], args.variant[1]), "sls_debug_variant")
    N, H, W = args.n, args.height, args.width
    scene = synth.make_scene(N, H, W, seed=0)
    kf0 = int(os.environ.get("SLS_BENCH_FIRST_KEYFRAME", "0"))          # (experiments: the window starts at this keyframe of the row)
    poses = synth.keyframe_poses(max(world, 8) + kf0)[kf0:]
    depth, valid = synth.make_targets(H, W, scene)
    cfg = MappingConfig()
    status_read = {"sync": True, "async": False, "lagged": "lagged"}["async" if args.async_steps else args.status_read]

    def camera(k):
        return Camera(scene["K"], depth, None, valid, poses[k], data_device=str(dev))

    def fresh(full_sort=False, dp_mode=None):
        """A new model (the scene's initial surfels) and, in engine mode, its engine."""
        model = SurfelModel.from_activated(scene["means"], scene["scales"], scene["rots"], scene["opac"], device=str(dev))
        model.training_setup(fused=True)
        engine = None
        if args.mode == "engine":
            from splat_loam_amd.engine import MappingEngine
            engine = MappingEngine(model, cfg)
            engine.reuse_depth_order = not full_sort
            if dp_mode:
                engine.dp_mode = dp_mode
        return model, engine

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    def stepper(model, engine, cams, pick=None):
        def one(i):
            cam = cams[pick[i] if pick is not None else 0]
            if engine is not None:
                return engine.step(cam, sync=status_read)
            if args.mode == "unfused":
                return optimize_step_sharded(model, cam, cfg)
            return optimize_step_fused(model, cam, cfg)
        return one

    def run(model, engine, cams, n_warm, n_steps, pick=None, after_warmup=None):
        """n_warm un-timed + n_steps timed iterations, barrier + synchronize on both sides; seconds, max over ranks."""
        one = stepper(model, engine, cams, pick)
        for i in range(n_warm):
            one(i)
        if engine is not None:
            engine.flush()
        barrier()
        if after_warmup:
            after_warmup()
        t0 = time.perf_counter()
        for i in range(n_steps):
            one(n_warm + i)
        if engine is not None:
            engine.flush()          # the last iteration's status is read inside the timed region too
        barrier()
        return max_over_ranks(time.perf_counter() - t0), one

    from splat_loam_amd.slam_rules import keyframe_probabilities
    ips = max(1, args.iters_per_step)
    n_kf = max(1, args.keyframes)
    # the keyframe window: poses 0.5 m apart.  Every iteration draws its keyframe as the mapper
    # does; with N ranks each rank takes its own draw of the same seeded sequence
    cams = [camera(k) for k in range(n_kf)]
    kf_p = keyframe_probabilities(n_kf, args.prob_view_last_keyframe)
    n_draws = (args.warmup + 2 * args.steps + 4) * ips * world
    draws = np.random.default_rng(0).choice(n_kf, size=n_draws, p=kf_p)
    pick_rank = draws[rank::world] if n_kf > 1 else None
    cam = cams[0] if n_kf > 1 else camera(rank)

    # ---- N > 1: which gradient exchange?  (un-timed calibration on throw-away models, same verdict on every rank)
    dp_mode, dp_cal = None, None
    if world > 1 and args.mode == "engine":
        dp_mode = args.dp_mode
        if dp_mode == "auto":
            dp_cal = {}
            for m in ("rs_ag", "allreduce", "sparse"):
                mdl, eng = fresh(dp_mode=m)
                try:
                    dt_m, _ = run(mdl, eng, cams if n_kf > 1 else [cam], 3, 10, pick=pick_rank)
                    ok = 1.0
                except Exception as e:          # (a collective the backend does not offer: the other scheme stays)
                    log(f"dp_mode {m} failed: {e}")
                    dt_m, ok = float("inf"), 0.0
                # every rank must reach the same verdict
                ok = -max_over_ranks(-ok)
                dp_cal[m] = round(dt_m / 10 * 1e3, 4) if ok > 0 else None
                del mdl, eng
            usable = {k: v for k, v in dp_cal.items() if v is not None}
            dp_mode = min(usable, key=usable.get) if usable else "allreduce"

    # ---- the headline run: the driver's command ---------------------------------------------------------------
    log(f"scene ready; dp_mode={dp_mode} calibration={dp_cal}")
    window = cams if n_kf > 1 else [cam]
    model, engine = fresh(full_sort=args.full_sort, dp_mode=dp_mode)
    timing = not args.no_timing and not extras_only
    # events around the dominant kernel only, and around one launch in eight of it (mode 4): the kernel is timed
    # live inside the timed region, and the iteration — host-bound at this size — pays for 1/8 of an event pair
    n_iters = args.steps * ips
    dt, step = run(model, engine, window, args.warmup * ips, n_iters, pick=pick_rank,
                   after_warmup=(lambda: lib.sls_timing_enable(4)) if timing else None)
    if engine is not None and status_read is False:
        assert not engine._read_status()["overflow"], "instance buffers overflowed during the timed region"

    def collect():
        ns = lib.sls_timing_slots()
        tot = (C.c_double * ns)()
        cnt = (C.c_int64 * ns)()
        lib.sls_timing_collect(tot, cnt)
        return {lib.sls_timing_name(s).decode(): (tot[s], int(cnt[s])) for s in range(ns) if cnt[s]}

    log(f"headline: {dt / args.steps * 1e3:.4f} ms/step = {dt / n_iters * 1e3:.4f} ms/iteration")
    kernels, live, comm = {}, {}, None
    if timing:
        live = collect()                     # the dominant kernel, measured inside the timed region
        lib.sls_timing_enable(1)             # every launch, in an extra un-timed pass of the same steps
        if engine is not None and world > 1:
            engine.comm_events = []
        for i in range(n_iters):
            step(args.warmup * ips + n_iters + i)
        if engine is not None:
            engine.flush()
        barrier()
        kernels = collect()
        lib.sls_timing_enable(0)
        for name, (ms, c) in live.items():   # the live average over the sampled launches, at the pass's launch count
            c_all = kernels.get(name, (0.0, c))[1] or c
            kernels[name] = (ms / c * c_all, c_all)
        if engine is not None and engine.comm_events:
            ev, engine.comm_events = engine.comm_events, None
            avg = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in ev) / len(ev) * 1e3
            ex, ad, ag = avg(0, 1), avg(1, 2), avg(2, 3)
            sharded = engine._dp is not None
            comm = {"mode": "sparse" if engine._sx is not None else ("rs_ag" if sharded else "allreduce"),
                    ("reduce_scatter_us" if sharded else "all_reduce_us"): round(ex, 1),
                    "adam_us": round(ad, 1), "all_gather_us": round(ag, 1) if sharded else 0.0,
                    "exchange_us": round(ex + ag, 1), "bytes_per_rank": int(engine.exchanged_bytes),
                    "dense_bytes_per_rank": 40 * N,
                    "union_rows": (engine.last or {}).get("exchange_count") if engine._sx is not None else None,
                    "note": "HIP events on the compute stream around each collective / the Adam kernel, un-timed "
                            "extra pass of the same steps (rank 0)"}
    if world > 1:
        # the collectives really went through RCCL with `world` ranks: every rank reports its backend and group size
        assert backend != "nccl" or (dist.get_backend() == "nccl" and dist.get_world_size() == world)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # workload statistics from un-timed forwards of the model as the timed iterations left it (R, R_eff, surfels the
    # backward reaches): one per keyframe of the window, averaged with the sampling probabilities
    from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_forward

    def workload(c):
        with torch.no_grad():
            st = rasterize_forward(GaussianRasterizationSettings(H, W, 1.0, c.world_view_transform, c.projection_matrix, False, False),
                                   model.get_xyz, model.get_opacity, model.get_scaling, model.get_rotation)
            cons = st.tile_consumed.long() & 0xFFFFFFFF
            touched = 0
            if st.R > 0:
                start = st.ranges.view(-1, 2)[:, 0].long() & 0xFFFFFFFF
                delta = torch.zeros((st.R + 1,), dtype=torch.int32, device=dev)
                delta.index_add_(0, start, torch.ones_like(start, dtype=torch.int32))
                delta.index_add_(0, start + cons, -torch.ones_like(start, dtype=torch.int32))
                touched = int(torch.unique(st.vals[:st.R][torch.cumsum(delta[:st.R], 0) > 0]).numel())
            return st.R, int(cons.sum().item()), touched
    per_kf = [workload(c) for c in window]
    wts = kf_p if n_kf > 1 else np.array([1.0])
    R, R_eff, N_touched = (int(round(float(np.dot(wts, [w[k] for w in per_kf])))) for k in range(3))
    tw, th = _abi.tile_size()
    P = H * W
    ms_per_step, ms_per_iter = dt / args.steps * 1e3, dt / n_iters * 1e3
    value = world * N * ips / (dt / args.steps) / 1e6
    roofline = roofline_block(kernels, live, N, R, R_eff, P, N_touched, H, W) if kernels else None
    breakdown = kernel_table(kernels, n_iters, N, R, R_eff, P, N_touched)
    log("per-kernel pass done")

    extras = None
    if world == 1 and engine is not None and (extras_only or not args.no_extras):
        import bench_extras
        ctx = SimpleNamespace(args=args, dev=dev, lib=lib, scene=scene, poses=poses, depth=depth, valid=valid, cfg=cfg,
                              cams=cams, window=window, kf_p=kf_p, n_kf=n_kf, pick_rank=pick_rank, N=N, H=H, W=W, ips=ips,
                              n_iters=n_iters, model=model, ms_per_iter=ms_per_iter, fresh=fresh, run=run, collect=collect)
        extras = bench_extras.collect_extras(ctx, extras_only)
        if extras_only:
            print(json.dumps(extras), file=result_out, flush=True)
            return
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baselines(scene, poses, depth, valid, cfg, N, H, W, (tw, th))
    log("cpu baseline done")

    out = {
        "metric": "fwd+bwd Msplats/s", "value": round(value, 3), "unit": "Msplats/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{N} surfels, {H}x{W} spherical, 1 keyframe/GPU per iteration drawn from a window of "
                               f"{n_kf} keyframes as Mapper.optimize does (sample_geometric, p={args.prob_view_last_keyframe}): "
                               "render fwd + mapper loss + bwd + fused Adam"
                               + {"engine": " (one native sls_mapping_step per iteration)",
                                  "fused": " (torch autograd + HIP loss consumer)",
                                  "unfused": " (torch loss glue)"}[args.mode],
                   "iterations_per_step": ips, "ms_per_iteration": round(ms_per_iter, 5),
                   "step": f"one bench step = {ips} mapping iterations ({args.steps} steps = {n_iters} timed iterations)",
                   "N": N, "H": H, "W": W, "tile": [tw, th], "R": R, "R_eff": R_eff, "N_touched": N_touched,
                   "keyframes": n_kf, "keyframe_probabilities": [round(float(x), 4) for x in kf_p],
                   "parallelism": f"keyframe-dp{world}",
                   "status_read": {True: "sync", False: "async", "lagged": "lagged-1"}[status_read]
                   if engine is not None else "torch",
                   "depth_order": (f"per keyframe: repaired from the keyframe's last order while it is at most {engine.max_order_age_extra} "
                                   "iterations old (verified exact), else sorted from scratch"
                                   if (engine is not None and engine.reuse_depth_order and status_read is not False)
                                   else "sorted from scratch"),
                   "repeated_iterations": dict(engine.stats) if engine is not None else None},
        "rccl_ranks": world if backend == "nccl" else 0, "collective_backend": backend,
        "dp_mode": (comm["mode"] if comm else dp_mode), "dp_overlap": None,
        "dp_calibration_ms": dp_cal,
        "allreduce_us": comm["exchange_us"] if comm else None, "adam_us": comm["adam_us"] if comm else None,
        "comm": comm,
        "roofline": roofline, "cpu_baseline": cpu, "extras": extras, "kernels": breakdown,
    }
    print(json.dumps(out), file=result_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
