#!/usr/bin/env python3
"""bench.py — forward+backward throughput of the spherical surfel rasterizer
on MI355X (BASELINE.json: "fwd+bwd Msplats/s @ 500k Gaussians, 64x2048").

One step = one mapping iteration of Splat-LOAM's hot loop (slam/mapper.py:150-204)
for ONE keyframe per GPU on the synthetic scene of SURVEY.md §8d:
    render()  [HIP: preprocess, depth order, binning, tile sort, tile render]
    + mapper loss + loss.backward()  [HIP: consumer, tile backward, preprocess backward]
    + Adam step on the 4 parameter tensors  [HIP: fused Adam]
with every input already resident in HBM.  Nothing is skipped inside the
timed region.  At N GPUs every rank renders its own keyframe of the shared
model and the 40 B/surfel gradients are exchanged over RCCL (weak scaling:
value = n_gpus * N / step time).

    python bench.py                       # 1 GPU, 20 steps, 3 warm-up
    python bench.py --gpus 8              # starts 8 ranks itself (torch.distributed.run, RCCL over xGMI)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3     # the driver's form

Rank 0 prints ONE JSON line.  `roofline` is for the kernel with the largest
total time in the step, from HIP events recorded on the launch stream inside
the timed region; `cpu_baseline` times, on this host's cores, the pure-PyTorch
tile rasterizer (oracle/torch_tiles.py, incl. loss + Adam) and the C/OpenMP
checker (oracle/sls_oracle.c) on the same scene (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
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
_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes(name, N, R, R_eff, P, N_touched):
    """Compulsory HBM bytes per launch (records 80 B, gradient records 64 B, instance word 4 B,
    per-pixel outputs 52 B).  `preprocess_bwd` is the fused kernel of the timed path: gradient chain + Adam
    (parameters 40 B read + 40 B written, radii 4, touched flag 1, moments 80 + 80) + the records of the surfels
    the backward reached (64 B read + 64 B cleared)."""
    return {
        "preprocess_fwd": N * (40 + 80 + 28),
        "scan": N * 8,
        "emit_keys": N * 28 + R * 4,
        "sort_hist": R * 4,
        "sort_rowscan": 0,
        "sort_scatter": R * 12,
        # depth-order repair; with the direct binning its merge also gathers the 8-byte emission records, stores them by
        # depth position and writes its column of the count table
        "resort": N * 28 + N * 16,
        "bin_count": N * 20,                 # (from-scratch iterations: order 4 + record gather 8 + record store 8)
        "bin_direct": N * 12 + R * 8,        # order 4 + record 8 per position, one (surfel, block mask) pair per instance
        "tile_ranges": R * 8,
        "render_fwd": R_eff * 84 + P * 52,
        "grec_memset": N * 64,
        "render_bwd": R_eff * (84 + 64) + P * (28 + 24),
        "preprocess_bwd": N * 245 + N_touched * 128,
        "adam": N * 10 * 28,
        "consumer": P * (28 + 5 + 4 + 3 * 16 + 16),
    }.get(name, 0)


def _latest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def kernel_source_hash():
    """sha256 over the kernel sources (splat_loam_amd/csrc/*.hip|*.hpp|Makefile, include/*.h): what the replayed PMC
    counters must have been measured on.  tools/pmc_*.sh store it in their JSON; bench.py compares (`stale`)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "splat_loam_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "splat_loam_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "splat_loam_amd", "csrc", "Makefile")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def valu_calibration():
    """Peak reading of the VALU counters on this hardware (tools/micro/valu_calib.hip, committed as
    profiles/*valu_calibration.json): what `SQ_ACTIVE_INST_VALU / SIMD quad-cycles` shows for a stream of independent
    v_fma_f32 at 4 resident waves per SIMD (the tile kernels' occupancy).  A kernel's calibrated VALU fraction is its
    own reading divided by this."""
    f = _latest_profile("*valu_calibration.json")
    if not f:
        return None
    try:
        rows = json.load(open(f))["rows"]
        peak = [r for r in rows if r["class"] == "v_fma_f32" and r["waves_per_simd"] == 4][0]
        return {"peak_valu_issue_busy_quad": peak["valu_issue_busy_quad"],
                "fma_ns_per_inst_per_simd": peak["ns_per_inst_per_simd"],
                "source": "profiles/" + os.path.basename(f)}
    except Exception:
        return None


def pmc_traffic(slot, N, H, W):
    """HBM bytes per launch of the kernel behind a timing slot, REPLAYED from the newest committed PMC pass
    (tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, FETCH_SIZE x2 as
    MI355X_MICROARCH.md prescribes for gfx950, cross-checked on adam_kernel's known byte count).
    rocprofv3 cannot wrap the process it is called from, so this is not measured in this run: the line
    carries the file it came from; null for any other workload."""
    f = _latest_profile("*pmc_traffic.json")
    if (N, H, W) != (500_000, 64, 2048) or not f:
        return None, None, None
    try:
        d = json.load(open(f))
        for name, v in d["kernels"].items():
            if name.startswith(slot):
                return int(v["hbm_bytes_corrected"]), "replayed from profiles/" + os.path.basename(f), d.get("kernel_source_hash")
    except Exception:
        pass
    return None, None, None


def pmc_valu(slot, N, H, W):
    """VALU issue utilisation of the kernel behind a timing slot, REPLAYED from the newest committed SQ counter
    pass (tools/pmc_sq.sh); null when there is none for this workload."""
    f = _latest_profile("*pmc_sq.json")
    if (N, H, W) != (500_000, 64, 2048) or not f:
        return None
    try:
        d = json.load(open(f))
        for name, v in d["kernels"].items():
            if name.startswith(slot):
                return {"valu_issue_busy_quad": v["valu_issue_busy"], "valu_insts": v["SQ_INSTS_VALU"],
                        "valu_active_quad_cycles": v.get("SQ_ACTIVE_INST_VALU"),
                        "avg_waves_per_simd": v.get("avg_waves_per_simd"),
                        "source": "replayed from profiles/" + os.path.basename(f),
                        "kernel_source_hash": d.get("kernel_source_hash")}
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
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (full sort, steps "
                    "200-400, 8 sampled keyframes) that follow the headline run at 1 GPU")
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
    ap.add_argument("--dp-mode", choices=("auto", "rs_ag", "allreduce", "sparse"), default=os.environ.get("SLS_DP_MODE", "auto"),
                    help="N > 1: gradient exchange.  rs_ag: reduce-scatter -> Adam on the rank's 1/N -> all-gather of "
                         "the parameters; allreduce: one all-reduce -> Adam everywhere; sparse: only the touched set "
                         "(bitmap OR + the union's rows SUM); auto: time all three for a few un-timed iterations and "
                         "take the fastest")
    ap.add_argument("--dp-overlap", action="store_true", default=os.environ.get("SLS_DP_OVERLAP", "0") == "1",
                    help="N > 1, dp_mode sparse: issue the two collectives from a side stream, behind the projection's "
                         "backward and the Adam update of the surfels outside the union (MappingEngine.overlap)")
    ap.add_argument("--iters-per-step", type=int, default=10,
                    help="mapping iterations inside ONE bench step (the driver's --steps 20 then times 200 iterations: "
                         "20 alone are 5 ms of GPU time, too few for a stable figure); ms_per_step is the time of a "
                         "whole step, config.ms_per_iteration the time of one iteration")
    ap.add_argument("--keyframes", type=int, default=8,
                    help="size of the keyframe window an iteration draws its keyframe from, as Mapper.optimize does "
                         "(slam/mapper.py:142-156: np.random.choice with sample_geometric(prob_view_last_keyframe)); "
                         "1 = re-render one keyframe every iteration (round 1/2's headline, now extras.single_keyframe)")
    ap.add_argument("--prob-view-last-keyframe", type=float, default=0.4,
                    help="configs/kitti/kitti.yaml:22, utils/config_utils.py:104")
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
    poses = synth.keyframe_poses(max(world, 8))
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
            engine.overlap = bool(args.dp_overlap)
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
    model, engine = fresh(full_sort=args.full_sort, dp_mode=dp_mode)
    timing = not args.no_timing
    # events around the dominant kernel only, and around one launch in eight of it (mode 4): the kernel is timed
    # live inside the timed region, and the iteration — host-bound at this size — pays for 1/8 of an event pair
    # instead of the ~60 event records that timing every launch would add
    n_iters = args.steps * ips
    dt, step = run(model, engine, cams if n_kf > 1 else [cam], args.warmup * ips, n_iters, pick=pick_rank,
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
        live = collect()                     # render_bwd, measured inside the timed region
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
            ev = engine.comm_events
            engine.comm_events = None

            def avg(a, b):
                return sum(e[a].elapsed_time(e[b]) for e in ev) / len(ev) * 1e3
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

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # workload statistics from un-timed forwards of the model as the timed iterations left it (R, R_eff, surfels the
    # backward reaches): one per keyframe of the window, averaged with the sampling probabilities
    from splat_loam_amd.rasterizer import GaussianRasterizationSettings, rasterize_forward

    def workload(c):
        with torch.no_grad():
            st = rasterize_forward(GaussianRasterizationSettings(H, W, 1.0, c.world_view_transform,
                                                                 c.projection_matrix, False, False),
                                   model.get_xyz, model.get_opacity, model.get_scaling, model.get_rotation)
            cons = st.tile_consumed.long() & 0xFFFFFFFF
            touched = 0
            if st.R > 0:
                start = st.ranges.view(-1, 2)[:, 0].long() & 0xFFFFFFFF
                delta = torch.zeros((st.R + 1,), dtype=torch.int32, device=dev)
                delta.index_add_(0, start, torch.ones_like(start, dtype=torch.int32))
                delta.index_add_(0, start + cons, -torch.ones_like(start, dtype=torch.int32))
                in_prefix = torch.cumsum(delta[:st.R], 0) > 0
                touched = int(torch.unique(st.vals[:st.R][in_prefix]).numel())
            return st.R, int(cons.sum().item()), touched
    per_kf = [workload(c) for c in (cams if n_kf > 1 else [cam])]
    wts = kf_p if n_kf > 1 else np.array([1.0])
    R, R_eff, N_touched = (int(round(float(np.dot(wts, [w[k] for w in per_kf])))) for k in range(3))
    tw, th = _abi.tile_size()
    P = H * W

    ms_per_step = dt / args.steps * 1e3
    ms_per_iter = dt / n_iters * 1e3
    value = world * N * ips / (dt / args.steps) / 1e6

    roofline = None
    breakdown = {}
    if kernels:
        for name, (ms, c) in kernels.items():
            avg_us = ms / c * 1e3
            b = algorithmic_bytes(name, N, R, R_eff, P, N_touched)
            breakdown[name] = {"launches_per_iteration": round(c / n_iters, 3), "avg_us": round(avg_us, 2),
                               "us_per_iteration": round(ms / n_iters * 1e3, 2),
                               "alg_bytes_per_launch": int(b),
                               "GBps": round(b / (avg_us * 1e-6) / 1e9, 1) if avg_us > 0 else None}
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, c = kernels[dom]
        b = algorithmic_bytes(dom, N, R, R_eff, P, N_touched)
        ach = b / (ms / c * 1e-3) / 1e9
        traffic, traffic_source, traffic_hash = pmc_traffic(dom, N, H, W)
        valu = pmc_valu(dom, N, H, W)
        cal = valu_calibration()
        hbm_frac = ach / HBM_PEAK_GBS
        src_hash = kernel_source_hash()
        # counters are REPLAYED from committed rocprofv3 passes (rocprofv3 cannot wrap the process it is called from):
        # they describe this build only if the kernel sources hash to what the pass was measured on
        stale = bool((traffic is not None and traffic_hash != src_hash) or
                     (valu is not None and valu.get("kernel_source_hash") != src_hash))
        # VALU side, calibrated: the raw reading SQ_ACTIVE_INST_VALU / SIMD quad-cycles is 1.6-1.8 —
        # not 1.0 — when a SIMD issues independent v_fma_f32 back to back (tools/micro/valu_calib.hip: one wave64
        # VALU instruction per ~2.3 clocks), so the kernel's reading is divided by that peak
        valu_frac = None
        if valu and cal:
            valu_frac = round(valu["valu_issue_busy_quad"] / cal["peak_valu_issue_busy_quad"], 4)
        bound = "valu" if (valu_frac is not None and valu_frac > hbm_frac) else "hbm"
        roofline = {"kernel": dom, "bound": bound, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(hbm_frac, 5), "hbm_frac": round(hbm_frac, 5),
                    "valu_frac": valu_frac, "valu_calibration": cal,
                    "traffic": traffic, "traffic_source": traffic_source, "stale": stale,
                    "kernel_source_hash": src_hash,
                    "avg_launch_us": round(ms / c * 1e3, 2), "alg_bytes_per_launch": int(b), "valu": valu,
                    "live_launches_timed": live.get(dom, (0.0, 0))[1],
                    "note": "achieved/frac: algorithmic bytes / live HIP-event time of this run (one launch in eight of the timed region bracketed) against the HBM peak "
                            "(the contract's figure).  valu_frac: the kernel's VALU counter reading relative to what "
                            "the same counter shows at the measured peak issue rate of plain FP32 instructions; the "
                            "step loops are made of half- and quarter-rate classes (DPP, v_cmp, packed, lane swaps, "
                            "exp/rcp), weighted by their measured rates the issue port is ~0.75 busy (DESIGN.md section 4). "
                            "stale: the replayed counters were measured on other kernel sources than this build's"}
        fb_ms = sum(kernels[k][0] / kernels[k][1] for k in ("render_fwd", "render_bwd") if k in kernels)
        if fb_ms > 0:
            fb_b = (algorithmic_bytes("render_fwd", N, R, R_eff, P, N_touched)
                    + algorithmic_bytes("render_bwd", N, R, R_eff, P, N_touched))
            roofline["tile_fwd_bwd_GBps"] = round(fb_b / (fb_ms * 1e-3) / 1e9, 2)
            roofline["tile_fwd_bwd_us"] = round(fb_ms * 1e3, 2)

    # ---- secondary measurements (1 GPU): the headline above stays the driver's command -------------------------
    extras = None
    log("per-kernel pass done")
    if world == 1 and engine is not None and not args.no_extras:
        extras = {}
        # round 1/2's headline: ONE keyframe re-rendered every iteration (its depth order repaired each time)
        m1, e1 = fresh()
        d1, _ = run(m1, e1, [cams[0]], args.warmup * ips, n_iters)
        extras["single_keyframe"] = {"ms_per_iteration": round(d1 / n_iters * 1e3, 4),
                                     "Msplats_per_s": round(N / (d1 / n_iters) / 1e6, 1),
                                     "repeated_iterations": dict(e1.stats)}
        del m1, e1
        m2, e2 = fresh(full_sort=True)
        d2, _ = run(m2, e2, cams if n_kf > 1 else [cam], args.warmup * ips, n_iters, pick=pick_rank)
        extras["ms_per_iteration_full_sort"] = round(d2 / n_iters * 1e3, 4)
        del m2, e2
        m3, e3 = fresh()
        d3, _ = run(m3, e3, cams if n_kf > 1 else [cam], 400, 400, pick=np.random.default_rng(1).choice(n_kf, size=800, p=kf_p) if n_kf > 1 else None)
        extras["ms_per_iteration_400_800"] = round(d3 / 400 * 1e3, 4)
        extras["repeated_iterations_0_800"] = dict(e3.stats)
        del m3, e3
        # deterministic accumulation of the gradient records (integer atomics): what bit-reproducible gradients cost
        det = {}
        for name, mode in (("two_launches", True), ("one_launch_predicted_scales", 2)):
            md, ed = fresh()
            ed.deterministic = mode
            dd, _ = run(md, ed, cams if n_kf > 1 else [cam], args.warmup * ips, n_iters, pick=pick_rank)
            det[name + "_ms_per_iteration"] = round(dd / n_iters * 1e3, 4)
            det[name + "_repeated"] = dict(ed.stats)
            del md, ed
        det["float_atomics_ms_per_iteration"] = round(ms_per_iter, 4)
        extras["deterministic"] = det
        # the sizes the reference's mapper really meets (BASELINE config 2 and a grown map at its geometry): there an
        # iteration is a chain of launches, not of bandwidth — tracked since VERDICT r02 (targets 0.100 / 0.18 ms)
        def real_size(n2, h2, w2):
            from splat_loam_amd.engine import MappingEngine
            sc2 = synth.make_scene(n2, h2, w2, seed=0)
            d2_, v2_ = synth.make_targets(h2, w2, sc2)
            cams2 = [Camera(sc2["K"], d2_, None, v2_, poses[k], data_device=str(dev)) for k in range(n_kf)]
            res = {}
            for name, cs, pk in (("single_keyframe", [cams2[0]], None),
                                 ("sampled_keyframes", cams2, np.random.default_rng(2).choice(n_kf, size=600, p=kf_p) if n_kf > 1 else None)):
                mdl = SurfelModel.from_activated(sc2["means"], sc2["scales"], sc2["rots"], sc2["opac"], device=str(dev))
                mdl.training_setup(fused=True)
                eng = MappingEngine(mdl, cfg)
                dd, _ = run(mdl, eng, cs, 100, 400, pick=pk)
                res[name + "_ms_per_iteration"] = round(dd / 400 * 1e3, 4)
                del mdl, eng
            return res
        extras["real_sizes"] = {"50000_64x1024": real_size(50_000, 64, 1024), "170000_64x1024": real_size(170_000, 64, 1024),
                                "note": "whole mapping iterations (engine, lagged status read), 400 timed after 100 un-timed"}
        # the DROP-IN path: what an unmodified slam/mapper.py gets (gaussian_renderer/__init__.py:26,40-47 ->
        # GaussianRasterizer under torch autograd -> sls_forward_stage1/2 + sls_backward, the host read of R included;
        # render() post-processing and the loss as ~85 torch kernels; FusedAdam behind optimizer.step();
        # loss.item() once per iteration as slam/mapper.py:206-209 reads it) next to the headline's sls_mapping_step
        def dropin(n2, h2, w2, iters, warm):
            from splat_loam_amd.mapping import optimize_step
            from splat_loam_amd.rasterizer import GaussianRasterizer as GR
            sc2 = scene if (n2, h2, w2) == (N, H, W) else synth.make_scene(n2, h2, w2, seed=0)
            d2_, v2_ = (depth, valid) if (n2, h2, w2) == (N, H, W) else synth.make_targets(h2, w2, sc2)
            cam2 = Camera(sc2["K"], d2_, None, v2_, poses[0], data_device=str(dev))

            def timed(fn, n_w, n_t):
                """ms per call: the best of three timed runs of n_t calls (a host-bound loop: one allocator or
                scheduler hiccup in a run of a hundred calls would otherwise be the figure)"""
                for _ in range(n_w):
                    fn()
                best = float("inf")
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(n_t):
                        fn()
                    torch.cuda.synchronize(dev)
                    best = min(best, (time.perf_counter() - t0) / n_t * 1e3)
                return best

            def one(lean):
                res = {}
                # (i) the rasterizer alone, forward + backward through autograd, dL/dallmap given
                mdl = SurfelModel.from_activated(sc2["means"], sc2["scales"], sc2["rots"], sc2["opac"], device=str(dev))
                with torch.no_grad():
                    leaves = [t.detach().clone().requires_grad_(True) for t in
                              (mdl.get_xyz, mdl.get_opacity, mdl.get_scaling, mdl.get_rotation)]
                st_ = GaussianRasterizationSettings(h2, w2, 1.0, cam2.world_view_transform, cam2.projection_matrix,
                                                    False, False, lean_allmap=lean)
                rast = GR(raster_settings=st_)
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
                        rast(means3D=leaves[0], means2D=leaves[0], opacities=leaves[1], scales=leaves[2],
                             rotations=leaves[3], cov3D_precomp=None)
                res["rasterizer_fwd_ms"] = round(timed(fwd_only, warm, iters), 4)
                res["rasterizer_fwd_bwd_ms"] = round(timed(fb, warm, iters), 4)
                lib.sls_timing_enable(1)
                for _ in range(10):
                    fb()
                torch.cuda.synchronize(dev)
                res["kernels_us"] = {k: round(ms / c * 1e3, 2) for k, (ms, c) in collect().items()}
                lib.sls_timing_enable(0)
                del mdl, leaves
                # (ii) a whole iteration of Mapper.optimize on one keyframe (render() builds its settings itself, as
                # gaussian_renderer/__init__.py does: the lean kernels are chosen by the process default)
                os.environ["SLS_LEAN_ALLMAP"] = "1" if lean else "0"
                try:
                    for name, fn in (("iteration_torch_glue_ms", lambda m: float(optimize_step(m, cam2, cfg))),
                                     ("iteration_hip_consumer_ms", lambda m: float(optimize_step_fused(m, cam2, cfg)))):
                        mdl = SurfelModel.from_activated(sc2["means"], sc2["scales"], sc2["rots"], sc2["opac"], device=str(dev))
                        mdl.training_setup(fused=True)
                        res[name] = round(timed(lambda: fn(mdl), warm, iters), 4)
                        del mdl
                finally:
                    os.environ.pop("SLS_LEAN_ALLMAP", None)
                res["Msplats_per_s_torch_glue"] = round(n2 / (res["iteration_torch_glue_ms"] * 1e-3) / 1e6, 1)
                return res
            return {"all_planes": one(False), "lean_allmap": one(True)}
        # keyframe-parallel readiness that one GPU can establish: the size of the UNION of the
        # touched sets — what dp_mode "sparse" puts on the wire — for G = 2, 4, 8 ranks, by rendering the ranks'
        # keyframes one after another on the same model and OR-ing their non-zero-gradient sets
        def sparse_union():
            from splat_loam_amd.fused import fused_loss
            m = SurfelModel.from_activated(scene["means"], scene["scales"], scene["rots"], scene["opac"], device=str(dev))
            with torch.no_grad():      # the model as the timed iterations left it
                for dst, src in zip((m._xyz, m._scaling, m._rotation, m._opacity),
                                    (model._xyz, model._scaling, model._rotation, model._opacity)):
                    dst.copy_(src)
            params = (m._xyz, m._opacity, m._scaling, m._rotation)
            sets = []
            for k in range(min(8, n_kf)):
                for p_ in params:
                    p_.grad = None
                fused_loss(m, cams[k], cfg, with_regulariser=(k == 0)).backward()
                sets.append(torch.cat([p_.grad.reshape(N, -1) for p_ in params], dim=1).ne(0).any(dim=1))
            out = {"per_keyframe_rows": [int(x.sum().item()) for x in sets]}
            rng_u = np.random.default_rng(3)
            for G in (2, 4, 8):
                if G > len(sets):
                    continue
                u = torch.stack(sets[:G]).any(dim=0)
                rows = int(u.sum().item())
                drawn = []
                for _ in range(50):      # ranks drawing their keyframes as the mapper does
                    ks = rng_u.choice(len(sets), size=G, p=kf_p[:len(sets)] / kf_p[:len(sets)].sum())
                    drawn.append(int(torch.stack([sets[int(k_)] for k_ in ks]).any(dim=0).sum().item()))
                out[f"G{G}"] = {"union_rows_window": rows, "bytes_per_rank_window": 40 * rows + G * ((N + 63) // 64) * 8,
                                "union_rows_sampled_mean": int(np.mean(drawn)), "union_rows_sampled_max": int(np.max(drawn)),
                                "dense_bytes_per_rank": 40 * N}
            out["note"] = ("rows = surfels with a non-zero gradient on at least one of the G ranks (rank 0 carries the scale "
                           "regulariser); window: rank g renders keyframe g of the window (BASELINE config 5); sampled: every "
                           "rank draws its keyframe with the mapper's probabilities, 50 draws; bytes = 40 B per row SUM-reduced "
                           "+ the G bitmaps all-gathered; measured on ONE GPU, no collective involved")
            return out
        try:
            extras["sparse_union"] = sparse_union()
        except Exception as e:      # (a report, never a reason to lose the bench line)
            extras["sparse_union"] = {"error": str(e)}
        log("extras: real sizes done")
        extras["dropin"] = {f"{N}_{H}x{W}": dropin(N, H, W, 30, 10), "50000_64x1024": dropin(50_000, 64, 1024, 50, 20),
                            "note": "the path an unmodified slam/mapper.py runs: GaussianRasterizer under torch autograd "
                                    "(sls_forward_stage1/2 + sls_backward, incl. the host read of R), one re-rendered "
                                    "keyframe; rasterizer_*: the rasterizer alone (dL/dallmap given); iteration_torch_glue: "
                                    "render() post-processing + mapper loss as torch ops + FusedAdam + loss.item(); "
                                    "iteration_hip_consumer: the same with sls_consumer_fwd_bwd instead of the torch glue; "
                                    "kernels_us: HIP-event averages of the library's kernels in rasterizer_fwd_bwd (every launch "
                                    "bracketed: a few us above rocprof's); all_planes: the seven-plane contract; lean_allmap: "
                                    "SLS_LEAN_ALLMAP=1 / settings.lean_allmap — planes 5, 6 not tracked, the kernels of "
                                    "sls_mapping_step.  "
                                    "Compare with extras.single_keyframe (sls_mapping_step on one keyframe)"}
        extras["note"] = ("same scene, size and keyframe sampling as the headline unless said otherwise; single_keyframe: one "
                          "keyframe re-rendered every iteration; full_sort: depth order sorted from scratch every "
                          "iteration; 400_800: 400 timed iterations after 400 un-timed ones (the optimisation changes "
                          "the workload as it proceeds)")

    cpu = None
    log("extras done")
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
        "dp_mode": (comm["mode"] if comm else dp_mode), "dp_overlap": bool(args.dp_overlap) if world > 1 else None,
        "dp_calibration_ms": dp_cal,
        "allreduce_us": comm["exchange_us"] if comm else None, "adam_us": comm["adam_us"] if comm else None,
        "comm": comm,
        "roofline": roofline, "cpu_baseline": cpu, "extras": extras, "kernels": breakdown,
    }
    print(json.dumps(out), file=result_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baselines(scene, poses, depth, valid, cfg, N, H, W, tile):
    """The CPU legs (rank 0, 1 GPU only).  Primary = the baseline BASELINE.json names: the pure-PyTorch tile
    rasterizer (oracle/torch_tiles.py), one WHOLE mapping iteration (activations, render, render() post-processing
    + mapper loss in torch, autograd backward, torch.optim.Adam).
      * thread count: swept over 16 / 32 / 64 (and all host threads when there are at most 96: with 256 the pool's
        warm-up alone took 144 s, profiles/r03c_bench_cpu_thread_sweep.err) on a small tile subset, the fastest is
        used and reported;
      * the headline workload (500k surfels, 64x2048): a stated subset of the tiles, extrapolated by the tile count
        (SURVEY.md section 8d allows it for N = 500k);
      * SURVEY.md section 8d's mandatory case, 50k surfels at 64x1024 with EVERY tile: one warm-up, then the median
        of 5 whole iterations.
    Also reported: the C/OpenMP checker (rasterizer forward + backward only, every core)."""
    from splat_loam_amd import synth
    host = os.cpu_count() or 1
    out = {"value": None, "unit": "Msplats/s", "cores": None, "host_cores": host, "kind": "port", "sample": None}
    try:
        from oracle import torch_tiles as tt
        from splat_loam_amd.mapping import mapping_loss
        from splat_loam_amd.renderer import postprocess
        from splat_loam_amd.scene import Camera, SurfelModel

        def torch_iteration(sc, Hh, Ww, tiles):
            view, proj = synth.camera_matrices(sc["K"], poses[0])
            dpt, vld = (depth, valid) if (Hh, Ww) == (H, W) else synth.make_targets(Hh, Ww, sc)
            cam = Camera(sc["K"], dpt, None, vld, poses[0], data_device="cpu")
            model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cpu")
            model.training_setup(fused=False)
            c = tt.camera_dict(Hh, Ww, view, proj)
            t0 = time.perf_counter()
            model.optimizer.zero_grad(set_to_none=True)
            _, am = tt.rasterize(c, model.get_xyz, model.get_scaling, model.get_rotation, model.get_opacity, tiles=tiles)
            loss = mapping_loss(postprocess(cam, am, cfg.depth_ratio), cam, model, cfg)
            loss.backward()
            model.optimizer.step()
            return time.perf_counter() - t0

        sc2 = synth.make_scene(50_000, 64, 1024, seed=0)
        T2 = ((1024 + tile[0] - 1) // tile[0]) * ((64 + tile[1] - 1) // tile[1])
        # ---- thread sweep (the per-tile tensors are (entries x 256): a larger pool mostly adds fork/join cost)
        sweep = {}
        probe = sorted(set(int(i * T2 / 12) for i in range(12)))
        t_leg = time.perf_counter()
        # every host thread was tried once (profiles/r03c_bench_cpu_thread_sweep.err: 256 threads need 144 s for the
        # two warm-up tiles alone, 32 threads 0.31 s for 12 tiles, 64 threads 0.69 s): pools beyond 96 threads are
        # not probed again in the default run, which has to finish within minutes
        for th in sorted(set(t for t in (16, 32, 64, host) if t <= host and t <= 96)):
            torch.set_num_threads(th)
            t_w = time.perf_counter()
            torch_iteration(sc2, 64, 1024, probe[:2])                      # warm the pool
            if time.perf_counter() - t_w > 5.0 and sweep:                  # a pool this slow cannot win: skip its probe
                sweep[th] = float("inf")
                log(f"cpu baseline: {th} threads: warm-up alone took {time.perf_counter() - t_w:.1f} s, skipped")
                continue
            sweep[th] = round(torch_iteration(sc2, 64, 1024, probe), 3)
            log(f"cpu baseline: {th} threads: {sweep[th]} s for {len(probe)} tiles")
        cores = min(sweep, key=sweep.get)
        torch.set_num_threads(cores)
        out["cores"] = cores
        out["thread_sweep_s"] = {str(k): v for k, v in sweep.items()}
        log(f"cpu baseline: thread sweep {sweep} -> {cores}")

        def timed_subset(sc, Hh, Ww, n_sc, budget_s):
            """iteration time extrapolated from as many evenly spaced tiles as fit the budget."""
            T = ((Ww + tile[0] - 1) // tile[0]) * ((Hh + tile[1] - 1) // tile[1])
            torch_iteration(sc, Hh, Ww, [T // 2])                          # warm-up (thread pool, allocator)
            base = torch_iteration(sc, Hh, Ww, [])                         # preprocess / binning / loss / Adam: whole model
            one = max(torch_iteration(sc, Hh, Ww, [T // 3]) - base, 1e-3)
            k = int(max(2, min(T, budget_s / one)))
            sub = sorted(set(int(i * T / k) for i in range(k)))
            secs = torch_iteration(sc, Hh, Ww, sub)
            full = base + max(secs - base, 0.0) * (T / len(sub))
            return n_sc / full / 1e6, (f"{len(sub)} of {T} tiles blended in {secs:.2f} s (of which {base:.2f} s for the "
                                       f"un-subsampled preprocess / binning / loss / Adam), tile part scaled by {T}/{len(sub)}")

        v, how = timed_subset(scene, H, W, N, 4.0)
        out.update(value=round(v, 5), sample=f"pure-PyTorch tile rasterizer (oracle/torch_tiles.py, float32, {cores} torch "
                   f"threads, the fastest of {sorted(sweep)}): one whole mapping iteration (render + loss + autograd "
                   f"backward + torch Adam) of the same {N}-surfel {H}x{W} scene; {how}")
        log("cpu baseline: torch 500k subset done")
        # ---- 50k / 64x1024, every tile, warm-up + median of 5
        t_w = time.perf_counter()
        torch_iteration(sc2, 64, 1024, None)
        log(f"cpu baseline: 50k warm-up iteration {time.perf_counter() - t_w:.1f} s (leg so far {time.perf_counter() - t_leg:.0f} s)")
        times = []
        for _ in range(5):
            times.append(torch_iteration(sc2, 64, 1024, None))
            log(f"cpu baseline: 50k iteration {len(times)}: {times[-1]:.2f} s")
        times.sort()
        med = times[2]
        out["torch_50k_64x1024"] = {"value": round(50_000 / med / 1e6, 5), "unit": "Msplats/s", "cores": cores,
                                    "seconds_median_of_5": round(med, 3), "seconds_all": [round(t, 3) for t in times],
                                    "sample": "the same iteration, 50k surfels at 64x1024, EVERY tile: one warm-up, "
                                              "then 5 whole iterations, median"}
        log("cpu baseline: torch 50k (every tile, median of 5) done")
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        out["sample"] = f"pure-PyTorch baseline failed: {e}"
    try:
        from oracle.oracle import Oracle
        o = Oracle(np.float32)
        threads = o.max_threads()
        view, proj = synth.camera_matrices(scene["K"], poses[0])
        ocam = o.camera(H, W, view, proj, tile=tile)
        dL = np.random.default_rng(0).normal(size=(7, H, W)).astype(np.float32)
        reps, tt_ = 0, 0.0
        o.forward(ocam, scene["means"], scene["scales"], scene["rots"], scene["opac"], frag_tol=0.0)  # warm-up
        while tt_ < 5.0 and reps < 6:
            t1 = time.perf_counter()
            ost = o.forward(ocam, scene["means"], scene["scales"], scene["rots"], scene["opac"], frag_tol=0.0)
            o.backward(ost, dL, threads=threads, want_abs=False)
            tt_ += time.perf_counter() - t1
            reps += 1
        out["c_openmp_port"] = {"value": round(N / (tt_ / reps) / 1e6, 4), "unit": "Msplats/s", "cores": threads,
                                "sample": f"{reps} x rasterizer forward+backward only (no loss / Adam) of the same scene, "
                                          "oracle/sls_oracle.c with OpenMP"}
    except Exception as e:
        out["c_openmp_port"] = {"value": None, "sample": f"failed: {e}"}
    return out


if __name__ == "__main__":
    main()
