#!/usr/bin/env python3
"""bench.py — forward+backward throughput of the spherical surfel rasterizer
on MI355X (BASELINE.json: "fwd+bwd Msplats/s @ 500k Gaussians, 64x2048").

One step = one mapping iteration of Splat-LOAM's hot loop (slam/mapper.py:150-204)
for ONE keyframe per GPU on the synthetic scene of SURVEY.md §8d:
    render()  [HIP: preprocess, scan, keys, radix sort, ranges, tile render]
    + mapper loss + loss.backward()  [HIP: tile backward, preprocess backward]
    + Adam step on the 4 parameter tensors  [HIP: fused Adam]
with every input already resident in HBM.  Nothing is skipped inside the
timed region.  At N GPUs every rank renders its own keyframe of the shared
model and the 40 B/surfel gradients are all-reduced over RCCL (weak scaling:
value = n_gpus * N / step time).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

Rank 0 prints ONE JSON line.  `roofline` is for the kernel with the largest
total time in the step, from HIP events recorded on the launch stream inside
the timed region; `cpu_baseline` times the CPU checker (oracle/, C + OpenMP) on
the same workload on this host's cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(name, N, R, R_eff, P, n_sort_passes):
    """Compulsory HBM bytes per launch (records 80 B, gradient
    records 64 B, key/value pair 12 B, per-pixel outputs 52 B)."""
    return {
        "preprocess_fwd": N * (40 + 80 + 28),
        "scan": N * 8,
        "emit_keys": N * 28 + R * 12,
        "sort_hist": R * 8,
        "sort_rowscan": 0,
        "sort_scatter": R * 24,
        "tile_ranges": R * 8,
        "render_fwd": R_eff * 84 + P * 52,
        "grec_memset": N * 64,
        "render_bwd": R_eff * (84 + 64) + P * (28 + 24),
        "preprocess_bwd": N * (40 + 4 + 64 + 40),
        "adam": N * 10 * 28,
        "consumer": P * (2 * 28 + 5 + 4 + 2 * 48 + 28),
    }.get(name, 0)


def pmc_traffic(slot, N, H, W):
    """HBM bytes per launch of the kernel behind a timing slot, from the committed PMC pass
    (tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, FETCH_SIZE x2 as
    MI355X_MICROARCH.md prescribes for gfx950, cross-checked on adam_kernel's known byte count).
    rocprofv3 cannot wrap the process it is called from, so bench.py reports the figure of the
    last committed pass for the default workload and null otherwise."""
    if (N, H, W) != (500_000, 64, 2048):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
    for name, v in k.items():
        if name.startswith(slot):
            return int(v["hbm_bytes_corrected"])
    return None


def pmc_valu(slot, N, H, W):
    """VALU issue utilisation of the kernel behind a timing slot from the committed SQ counter pass
    (tools/pmc_sq.sh); null when there is none for this workload."""
    if (N, H, W) != (500_000, 64, 2048):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_sq.json")))
    if not files:
        return None
    try:
        for name, v in json.load(open(files[-1]))["kernels"].items():
            if name.startswith(slot):
                return {"valu_issue_busy": v["valu_issue_busy"], "valu_insts": v["SQ_INSTS_VALU"],
                        "source": os.path.basename(files[-1])}
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=500_000, help="surfels (BASELINE config 3: 500k)")
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timing", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--mode", choices=("engine", "fused", "unfused"), default="engine",
                    help="engine: one native sls_mapping_step per iteration (default); fused: torch autograd around "
                         "the HIP rasterizer + HIP loss consumer; unfused: torch render()/loss glue (same maths)")
    ap.add_argument("--status-read", choices=("lagged", "sync", "async"), default="lagged",
                    help="engine mode, how the per-iteration status (loss terms, R, overflow) reaches the host.  The "
                         "reference reads its loss once per iteration (slam/mapper.py:206-209).  lagged (default): "
                         "every iteration's status is read, but after the NEXT iteration has been enqueued, so the GPU "
                         "queue never drains; sync: read before enqueuing the next one; async: never read")
    ap.add_argument("--async-steps", action="store_true", help="alias of --status-read async")
    ap.add_argument("--full-sort", action="store_true",
                    help="engine mode: sort the depth order from scratch every iteration instead of repairing the "
                         "previous iteration's order (windowed re-sort + exactness check, DESIGN.md section 4)")
    ap.add_argument("--variant", type=int, nargs=2, default=None, metavar=("FWD", "BWD"),
                    help="tuning: tile-kernel variants (sls_debug_variant)")
    ap.add_argument("--pad-lds", type=int, nargs=2, default=None, metavar=("FWD", "BWD"),
                    help="tuning: unused dynamic LDS bytes for the tile kernels (caps workgroups per CU)")
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
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # ranks > devices only in the single-GPU self-test (gloo)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SLS_BENCH_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            assert world <= ndev, f"{world} ranks but {ndev} GPUs visible"
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
    if args.pad_lds:
        lib.sls_debug_pad_lds(args.pad_lds[0], args.pad_lds[1])
    if args.variant:
        lib.sls_debug_variant(args.variant[0], args.variant[1])
    N, H, W = args.n, args.height, args.width
    scene = synth.make_scene(N, H, W, seed=0)
    poses = synth.keyframe_poses(max(world, 1))
    depth, valid = synth.make_targets(H, W, scene)
    cam = Camera(scene["K"], depth, None, valid, poses[rank], data_device=str(dev))
    model = SurfelModel.from_activated(scene["means"], scene["scales"], scene["rots"], scene["opac"], device=str(dev))
    model.training_setup(fused=True)
    cfg = MappingConfig()

    engine = None
    if args.mode == "engine":
        from splat_loam_amd.engine import MappingEngine
        engine = MappingEngine(model, cfg)
        engine.reuse_depth_order = not args.full_sort

    status_read = {"sync": True, "async": False, "lagged": "lagged"}["async" if args.async_steps else args.status_read]

    def step():
        if engine is not None:
            return engine.step(cam, sync=status_read)
        if args.mode == "unfused":
            return optimize_step_sharded(model, cam, cfg)
        return optimize_step_fused(model, cam, cfg)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    if engine is not None:
        engine.flush()
    barrier()
    timing = not args.no_timing
    if timing:
        # events around the dominant kernel only (2 per step, ~10 us): it is timed live inside the
        # timed region without the ~0.2 ms/step that 60 event records per step would add
        lib.sls_timing_enable(3)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if engine is not None:
        engine.flush()          # the last iteration's status is read inside the timed region too
    barrier()
    dt = time.perf_counter() - t0
    if engine is not None and status_read is False:
        assert not engine._read_status()["overflow"], "instance buffers overflowed during the timed region"
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def collect():
        ns = lib.sls_timing_slots()
        tot = (C.c_double * ns)()
        cnt = (C.c_int64 * ns)()
        lib.sls_timing_collect(tot, cnt)
        return {lib.sls_timing_name(s).decode(): (tot[s], int(cnt[s])) for s in range(ns) if cnt[s]}

    kernels, live = {}, {}
    if timing:
        live = collect()                     # render_bwd, measured inside the timed region
        lib.sls_timing_enable(1)             # every launch, in an extra un-timed pass of the same steps
        for _ in range(args.steps):
            step()
        barrier()
        kernels = collect()
        lib.sls_timing_enable(0)
        kernels.update(live)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # workload statistics from one un-timed forward (R, R_eff)
    from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_forward
    with torch.no_grad():
        st = rasterize_forward(GaussianRasterizationSettings(H, W, 1.0, cam.world_view_transform,
                                                             cam.projection_matrix, False, False),
                               model.get_xyz, model.get_opacity, model.get_scaling, model.get_rotation)
        R = st.R
        R_eff = int(st.tile_consumed.cpu().numpy().view(np.uint32).astype(np.int64).sum())
    tw, th = _abi.tile_size()
    T = ((W + tw - 1) // tw) * ((H + th - 1) // th)
    n_pass = (32 + max(T - 1, 1).bit_length() + 7) // 8
    P = H * W

    ms_per_step = dt / args.steps * 1e3
    value = world * N / (dt / args.steps) / 1e6

    roofline = None
    breakdown = {}
    if kernels:
        for name, (ms, c) in kernels.items():
            avg_us = ms / c * 1e3
            b = algorithmic_bytes(name, N, R, R_eff, P, n_pass)
            breakdown[name] = {"launches_per_step": c / args.steps, "avg_us": round(avg_us, 2),
                               "us_per_step": round(ms / args.steps * 1e3, 2),
                               "alg_bytes_per_launch": int(b),
                               "GBps": round(b / (avg_us * 1e-6) / 1e9, 1) if avg_us > 0 else None}
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, c = kernels[dom]
        b = algorithmic_bytes(dom, N, R, R_eff, P, n_pass)
        ach = b / (ms / c * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom, N, H, W),
                    "avg_launch_us": round(ms / c * 1e3, 2), "alg_bytes_per_launch": int(b),
                    # the tile kernels are VALU-issue bound, not HBM bound: what the
                    # SQ counters say about the dominant kernel's real limiter
                    "valu": pmc_valu(dom, N, H, W)}
        fb_ms = sum(kernels[k][0] / kernels[k][1] for k in ("render_fwd", "render_bwd") if k in kernels)
        if fb_ms > 0:
            fb_b = algorithmic_bytes("render_fwd", N, R, R_eff, P, n_pass) + algorithmic_bytes("render_bwd", N, R, R_eff, P, n_pass)
            roofline["tile_fwd_bwd_GBps"] = round(fb_b / (fb_ms * 1e-3) / 1e9, 2)
            roofline["tile_fwd_bwd_us"] = round(fb_ms * 1e3, 2)

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            from oracle.oracle import Oracle
            o = Oracle(np.float32)
            cores = o.max_threads()
            view, proj = synth.camera_matrices(scene["K"], poses[0])
            ocam = o.camera(H, W, view, proj, tile=(tw, th))
            dL = np.random.default_rng(0).normal(size=(7, H, W)).astype(np.float32)
            reps, tt = 0, 0.0
            o.forward(ocam, scene["means"], scene["scales"], scene["rots"], scene["opac"], frag_tol=0.0)  # warm-up
            while tt < 10.0 and reps < 8:
                t1 = time.perf_counter()
                ost = o.forward(ocam, scene["means"], scene["scales"], scene["rots"], scene["opac"], frag_tol=0.0)
                o.backward(ost, dL, threads=cores, want_abs=False)
                tt += time.perf_counter() - t1
                reps += 1
            cpu = {"value": round(N / (tt / reps) / 1e6, 4), "unit": "Msplats/s", "cores": cores, "kind": "port",
                   "sample": f"{reps} x full rasterizer forward+backward (no loss/Adam) of the same {N}-surfel "
                             f"{H}x{W} scene, oracle/sls_oracle.c with OpenMP"}
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            cpu = {"value": None, "unit": "Msplats/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    out = {
        "metric": "fwd+bwd Msplats/s", "value": round(value, 3), "unit": "Msplats/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{N} surfels, {H}x{W} spherical, 1 keyframe/GPU: render fwd + mapper loss + bwd + fused Adam"
                               + {"engine": " (one native sls_mapping_step per iteration)",
                                  "fused": " (torch autograd + HIP loss consumer)",
                                  "unfused": " (torch loss glue)"}[args.mode],
                   "N": N, "H": H, "W": W, "tile": [tw, th], "R": R, "R_eff": R_eff,
                   "parallelism": f"keyframe-dp{world}",
                   "status_read": {True: "sync", False: "async", "lagged": "lagged-1"}[status_read]
                   if engine is not None else "torch",
                   "depth_order": ("repaired from the previous iteration, verified exact"
                                   if (engine is not None and engine.reuse_depth_order and status_read is not False)
                                   else "sorted from scratch"),
                   "repeated_iterations": dict(engine.stats) if engine is not None else None},
        "roofline": roofline, "cpu_baseline": cpu, "kernels": breakdown,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
