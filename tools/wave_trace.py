#!/usr/bin/env python3
"""Experiment (needs a -DSLS_TRACE build of libsls_hip.so): timeline of the tile kernels' waves — when each of
the 8192 waves started and ended (100 MHz wall clock) and on which XCD / CU / SIMD — to see how full the SIMDs are
over the life of a launch.  python tools/wave_trace.py [lib.so]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from splat_loam_amd import _abi, synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel
N, H, W = [int(x) for x in os.environ.get("SLS_TRACE_SHAPE", "500000,64,2048").split(",")]
NB = (H // 16) * (W // 16) * 16          # blocks (= waves) of a tile-kernel launch, at most 8192 recorded
sc = synth.make_scene(N, H, W, seed=0)
depth, valid = synth.make_targets(H, W, sc)
cam = Camera(sc["K"], depth, None, valid, None, data_device="cuda:0")
m = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cuda:0")
if os.environ.get("SLS_VARIANT"):
    _abi.lib().sls_debug_variant(*[int(x) for x in os.environ["SLS_VARIANT"].split(",")])
e = MappingEngine(m, MappingConfig())
for _ in range(6):
    e.step(cam)
torch.cuda.synchronize()
lib = _abi.lib()
buf = (C.c_uint32 * (2 * 8192 * 4))()
lib.sls_debug_read_trace.argtypes = [C.c_void_p]
assert lib.sls_debug_read_trace(buf) == 0
tr = np.frombuffer(buf, dtype=np.uint32).reshape(2, 8192, 4).astype(np.int64)[:, :NB]
if hasattr(lib, "sls_debug_read_trace_phases"):
    pb = (C.c_uint32 * (8192 * 4))()
    lib.sls_debug_read_trace_phases.argtypes = [C.c_void_p]
    if lib.sls_debug_read_trace_phases(pb) == 0:
        ph = np.frombuffer(pb, dtype=np.uint32).reshape(8192, 4).astype(np.float64)[:NB]
        rounds_f = (tr[0, :, 3] >> 16).astype(np.float64)
        steps_f = (tr[0, :, 3] & 0xFFFF).astype(np.float64)
        dur_f = (tr[0, :, 1] - tr[0, :, 0]) * 0.01
        for name_, sel in (("all", rounds_f > 0), (">= 12 rounds", rounds_f >= 12), ("the 10 longest", dur_f >= np.sort(dur_f)[-10])):
            tot = ph[sel].sum(0)
            print(f"  forward phases [{name_}: {int(sel.sum())} waves, {rounds_f[sel].sum():.0f} rounds, {steps_f[sel].sum():.0f} steps]: shader clocks per round - "
                  f"wait for the staged records + store {tot[0] / rounds_f[sel].sum():.0f}, cull + compaction {tot[1] / rounds_f[sel].sum():.0f}, "
                  f"steps {tot[2] / rounds_f[sel].sum():.0f} ({tot[2] / max(steps_f[sel].sum(), 1):.0f} per step), round end {tot[3] / rounds_f[sel].sum():.0f}")
if hasattr(lib, "sls_debug_read_trace_phases_b"):
    pb = (C.c_uint32 * (8192 * 4))()
    lib.sls_debug_read_trace_phases_b.argtypes = [C.c_void_p]
    if lib.sls_debug_read_trace_phases_b(pb) == 0:
        ph = np.frombuffer(pb, dtype=np.uint32).reshape(8192, 4).astype(np.float64)[:NB]
        rounds_b = (tr[1, :, 3] >> 16).astype(np.float64)
        steps_b = (tr[1, :, 3] & 0xFFFF).astype(np.float64)
        sel = rounds_b > 0
        tot = ph[sel].sum(0)
        print(f"  backward phases [{int(sel.sum())} waves, {rounds_b[sel].sum():.0f} rounds, {steps_b[sel].sum():.0f} steps]: shader clocks per round - "
              f"wait for the staged records + store {tot[0] / rounds_b[sel].sum():.0f}, mask + list {tot[1] / rounds_b[sel].sum():.0f}, "
              f"steps {tot[2] / rounds_b[sel].sum():.0f} ({tot[2] / max(steps_b[sel].sum(), 1):.0f} per step)")
marks = None
if hasattr(lib, "sls_debug_read_trace_marks"):
    mb = (C.c_uint32 * (2 * 8192 * 4))()
    lib.sls_debug_read_trace_marks.argtypes = [C.c_void_p]
    if lib.sls_debug_read_trace_marks(mb) == 0:
        marks = np.frombuffer(mb, dtype=np.uint32).reshape(2, 8192, 4).astype(np.float64)[:, :NB] * 0.01
        names = (("list range known", "pixel rays loaded", "first records in LDS", "rounds done"),
                 ("block known (order lookup)", "pixel state loaded (tmax)", "consumer gradient ready", "first records in LDS"))
        for k in range(2):
            dur_k = (tr[k, :, 1] - tr[k, :, 0]) * 0.01
            st_k = (tr[k, :, 0] - tr[k, :, 0].min()) * 0.01
            for nm, sel in (("all waves", dur_k > 0), ("waves started in the first 2 us", st_k < 2.0), ("waves started after 60 % of the span", st_k > 0.6 * st_k.max())):
                if sel.sum() == 0:
                    continue
                print(f"  {'fwd' if k == 0 else 'bwd'} marks [{nm}: {int(sel.sum())}] us from the wave's start (median): "
                      + ", ".join(f"{names[k][q]} {np.median(marks[k][sel, q]):.2f}" for q in range(4)) + f", end {np.median(dur_k[sel]):.2f}")
for k, name in enumerate(("fwd", "bwd")):
    t0, t1, hw, xcc = tr[k, :, 0], tr[k, :, 1], tr[k, :, 2], 0 * tr[k, :, 3]
    rounds, steps = tr[k, :, 3] >> 16, tr[k, :, 3] & 0xFFFF
    if k == 0:
        sp1, sp2, sp4 = tr[k, :, 2] & 1023, (tr[k, :, 2] >> 10) & 1023, (tr[k, :, 2] >> 20) & 1023
        long_ = rounds >= 12
        print("  forward waves with >= 12 rounds: %d; of their %d rounds, %d have <= 1 active pixel, %d <= 2, %d <= 4; all waves: %d of %d rounds with <= 1, %d <= 2"
              % (long_.sum(), rounds[long_].sum(), sp1[long_].sum(), sp2[long_].sum(), sp4[long_].sum(), sp1.sum(), rounds.sum(), sp2.sum()))
        dsorted = np.sort((tr[k, :, 1] - tr[k, :, 0]) * 0.01)[::-1]
        print("  longest forward waves (us):", np.round(dsorted[:12], 1), " waves > 45 us:", int((dsorted > 45).sum()), " > 38 us:", int((dsorted > 38).sum()))
        hw = 0 * hw
    base = t0.min()
    s, f = (t0 - base) * 0.01, (t1 - base) * 0.01            # us
    dur = f - s
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xc = xcc & 15
    uid = (((xc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    print(f"{name}: span {f.max():.1f} us; wave duration mean {dur.mean():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us; "
          f"distinct SIMDs {len(np.unique(uid))}; waves per SIMD min/mean/max {np.bincount(np.unique(uid, return_inverse=True)[1]).min()}/{NB / len(np.unique(uid)):.1f}/{np.bincount(np.unique(uid, return_inverse=True)[1]).max()}")
    grid = np.arange(0, f.max(), 1.0)
    occ = [(int(((s <= t) & (f > t)).sum())) for t in grid]
    print("  resident waves every 4 us:", occ[::4])
    print("  start times: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(s, [10, 50, 90, 100])))
    first = s < 2.0
    print("  waves started in the first 2 us: %d; their mean duration %.1f us; later waves %.1f us" % (first.sum(), dur[first].mean(), dur[~first].mean()))
    order = np.argsort(s)
    print("  per-XCD wave count", np.bincount(xc, minlength=8)[:8])
    # who makes the tail?  (block -> tile as tile_of_block does for T % 32 == 0, 16 blocks per tile)
    b = np.arange(NB); xcd_ = b % 8; i_ = b // 8; ts = i_ // 16
    tile = ((ts >> 2) * 8 + xcd_) * 4 + (ts & 3); row = tile // (W // 16)
    late = f > 0.75 * f.max()
    print("  waves alive in the last quarter: %d; their start p10/p50/p90 %.1f/%.1f/%.1f us, duration p10/p50/p90 %.1f/%.1f/%.1f us; by tile row %s"
          % (late.sum(), *np.percentile(s[late], [10, 50, 90]), *np.percentile(dur[late], [10, 50, 90]), np.bincount(row[late], minlength=4)))
    print("  rounds by row", [round(float(rounds[row == r].mean()), 1) for r in range(4)], "steps by row", [round(float(steps[row == r].mean()), 1) for r in range(4)],
          "rounds p50/p90/max %d/%d/%d steps p50/p90/max %d/%d/%d" % (*np.percentile(rounds, [50, 90, 100]), *np.percentile(steps, [50, 90, 100])))
    A = np.stack([rounds, steps, np.ones_like(rounds)], 1).astype(np.float64)
    for nm, sel in (("full phase (started < 2 us)", s < 2.0), ("late (started > 25 us)", s > 25.0)):
        coef = np.linalg.lstsq(A[sel], dur[sel], rcond=None)[0]
        print("  %s: duration ~ %.3f us/round + %.3f us/step + %.2f us  (n=%d)" % (nm, *coef, sel.sum()))
    print("  duration by tile row (mean us):", [round(float(dur[row == r].mean()), 1) for r in range(4)], "start by row (mean):", [round(float(s[row == r].mean()), 1) for r in range(4)])
    # list scheduling on 4096 slots with the measured durations: as dispatched vs longest first
    import heapq
    def makespan(order):
        h = [0.0] * 4096
        heapq.heapify(h)
        for j in order:
            t = heapq.heappop(h); heapq.heappush(h, t + dur[j])
        return max(h)
    print("  list-scheduling makespan with these durations: in block order %.1f us, longest first %.1f us, ideal %.1f us"
          % (makespan(range(NB)), makespan(np.argsort(-dur)), dur.sum() / 4096))
if os.environ.get("SLS_TRACE_DUMP"):
    np.save(os.environ["SLS_TRACE_DUMP"], tr)
