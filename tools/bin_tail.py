#!/usr/bin/env python3
"""Which launches of bin_direct are the slow ones?  Reads a rocprofv3 --kernel-trace CSV of
`bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-timing`, splits it into mapping iterations, replays
the bench's seeded keyframe draws beside them and prints, per keyframe of the window, the launch times of the binning and
tile kernels; then (CPU checker) the instance totals of the depth-order chunks per keyframe.
    python tools/bin_tail.py gpurun_out/r05a_kernel_trace.csv > profiles/r05a_bin_tail.txt"""
import csv, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splat_loam_amd import slam_rules, synth

NAMES = ('render_bwd', 'render_fwd', 'bin_direct', 'preprocess_bwd', 'preprocess_fwd_resort', 'preprocess_fwd',
         'resort_merge', 'sort_scatter', 'sort_hist', 'sort_rowscan', 'gather_count')


def short(n):
    for k in NAMES:
        if k in n:
            return k
    return 'other'


rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
its, cur = [], None
for s, e, n in ev:
    if n in ('preprocess_fwd_resort', 'preprocess_fwd'):
        cur = {'scratch': n == 'preprocess_fwd', 'k': {}}
        its.append(cur)
    if cur is not None:
        cur['k'][n] = cur['k'].get(n, 0) + (e - s) / 1e3
draws = np.random.default_rng(0).choice(8, size=len(its), p=slam_rules.keyframe_probabilities(8, 0.4))   # bench.py
per = {}
for it, k in zip(its[:230], draws):      # (the timed loop: warm-up 30 + 200 iterations; what follows is another pass)
    per.setdefault(int(k), []).append(it)
print("keyframe  visits  from-scratch | bin_direct avg / max | counting merge or gather_count avg | render_fwd avg | render_bwd avg   [us]")
for k in sorted(per):
    v = per[k]
    f = lambda name: np.array([i['k'].get(name, 0.0) for i in v])
    cnt = f('resort_merge') + f('gather_count')
    print(f"   {k}      {len(v):4d}     {sum(i['scratch'] for i in v):4d}      | {f('bin_direct').mean():7.1f} / {f('bin_direct').max():6.1f}   |"
          f" {cnt.mean():7.1f}                           | {f('render_fwd').mean():7.1f}        | {f('render_bwd').mean():7.1f}")
allb = np.array([i['k'].get('bin_direct', 0.0) for i in its[:230]])
print(f"all 230: bin_direct avg {allb.mean():.1f} us, sigma {allb.std():.1f}, max {allb.max():.1f}; without keyframes 5-7: "
      f"avg {np.mean([i['k'].get('bin_direct', 0.0) for i, k in zip(its[:230], draws) if k < 5]):.1f}, "
      f"sigma {np.std([i['k'].get('bin_direct', 0.0) for i, k in zip(its[:230], draws) if k < 5]):.1f}")
try:
    from oracle.oracle import Oracle
    o = Oracle(np.float32); o.set_threads(o.max_threads())
    N, H, W = 500000, 64, 2048
    sc = synth.make_scene(N, H, W, seed=0)
    print("\nCPU checker, bench scene: instances per chunk of 1024 depth positions (bin_direct: one workgroup per chunk, its 16 waves 64 positions each)")
    for k, pose in enumerate(synth.keyframe_poses(8)):
        view, proj = synth.camera_matrices(sc["K"], pose)
        pre = o.preprocess(o.camera(H, W, view, proj, tile=(16, 16)), sc["means"], sc["scales"], sc["rots"], sc["opac"])
        tiles = pre["tiles"].astype(np.int64)
        order = np.lexsort((np.arange(N), pre["depth"].view(np.uint32)))
        t = tiles[order]
        ch = np.add.reduceat(t, np.arange(0, N, 1024)); wv = np.add.reduceat(t, np.arange(0, N, 64))
        print(f"  keyframe {k} (x = {pose[0, 3]:.1f} m): R {tiles.sum()}, largest rectangle {tiles.max()} tiles, surfels > 64 tiles: {(tiles > 64).sum()}, "
              f"heaviest chunk {ch.max()} (median {int(np.median(ch))}), heaviest wave {wv.max()} instances = {-(-wv.max() // 64)} rounds (median {int(np.median(wv))})")
except Exception as e:       # (no checker: the trace part stands on its own)
    print("checker part skipped:", e)
