import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from splat_loam_amd import synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.scene import Camera, SurfelModel
N, H, W, n_kf, n_iter = 60000, 64, 1024, 5, 2000
dev = "cuda:0"
sc = synth.make_scene(N, H, W, seed=0)
depth, valid = synth.make_targets(H, W, sc)
poses = synth.keyframe_poses(n_kf)
cams = [Camera(sc["K"], depth, None, valid, poses[k], data_device=dev) for k in range(n_kf)]
model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device=dev)
eng = MappingEngine(model, MappingConfig())
rng = np.random.default_rng(0)
prob = np.array([0.4] + [0.6 / (n_kf - 1)] * (n_kf - 1))
Rs = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(n_iter):
    st = eng.step(cams[rng.choice(n_kf, p=prob)], sync="lagged")
    if st is not None: Rs.append(st["R"])
    if (it + 1) % 200 == 0:
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sca = model.get_scaling.detach(); op = model.get_opacity.detach()
        print(f"it {it+1}: {(t1-t0)/200*1e3:.4f} ms/it  R mean {np.mean(Rs[-200:]):.0f}  scale mean {sca.mean().item():.4f} max {sca.max().item():.3f}  opacity mean {op.mean().item():.3f}  stats {eng.stats}", flush=True)
        t0 = time.perf_counter()
eng.flush()
