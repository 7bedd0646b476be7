#!/usr/bin/env python3
"""Mapping + tracking on a synthetic sequence, every hot component in its reference role:
scan (room ray caster) -> point cloud -> range image on the device (projector.DeviceProjector, the job of
scene/preprocessing.py:42-64) -> surfels initialised as Mapper.densify does
(slam/mapper.py:100-135: one surfel per sampled pixel, scale from distCUDA2, normal-aligned,
opacity 0.9) -> MappingEngine iterations on the keyframe -> for the following scans:
render the keyframe from the model (render()), register the scan against it
(GSAligner, as slam/tracker.py:141-197 does) and compare with the ground-truth motion.

    python tools/slam_demo.py [H W n_frames n_iterations]
"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_aligner import room_scan, pose_of, pose_error
from gsaligner import GSAligner, GSAlignerParams
from simple_knn._C import distCUDA2
from splat_loam_amd import synth
from splat_loam_amd.engine import MappingEngine
from splat_loam_amd.mapping import MappingConfig
from splat_loam_amd.projector import DeviceProjector
from splat_loam_amd.renderer import depth_to_points, render
from splat_loam_amd.scene import Camera, SurfelModel


def surfels_from_scan(depth, points_sensor, world_T_sensor, stride, smax, dev):
    """One surfel per `stride`-th valid pixel, as densify() builds them."""
    H, W = depth.shape
    sel = np.zeros((H, W), bool); sel[::stride[0], ::stride[1]] = True
    sel &= depth > 0.5
    p = points_sensor[sel].astype(np.float64)
    pw = p @ world_T_sensor[:3, :3].T + world_T_sensor[:3, 3]
    n = -p / np.linalg.norm(p, axis=1, keepdims=True)              # scene/preprocessing.py:112: normals = -unit(point)
    nw = n @ world_T_sensor[:3, :3].T
    helper = np.where(np.abs(nw[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    t0 = np.cross(nw, helper); t0 /= np.linalg.norm(t0, axis=1, keepdims=True)
    t1 = np.cross(nw, t0)
    rots = synth._quat_from_R(np.stack([t0, t1, nw], 2))
    xyz = torch.tensor(pw, dtype=torch.float32, device=dev)
    d2 = torch.clamp(distCUDA2(xyz), 1e-7, smax ** 2)
    scales = torch.sqrt(d2)[:, None].repeat(1, 2)
    return SurfelModel.from_activated(xyz, scales, torch.tensor(rots, dtype=torch.float32),
                                      torch.full((xyz.shape[0], 1), 0.9), device=str(dev))


def run(H=64, W=1024, n_frames=6, n_iter=60, verbose=True, dev="cuda:0"):
    dev = torch.device(dev)
    K = synth.spherical_K(H, W).astype(np.float64)
    poses = [pose_of([0.30 * k, 0.05 * k, 0.0], yaw_deg=1.5 * k) for k in range(n_frames)]
    scans = [room_scan(K, H, W, P) for P in poses]
    cfg = MappingConfig()
    d0, p0 = scans[0]
    model = surfels_from_scan(d0, p0, poses[0], (1, 2), cfg.opt_scaling_max, dev)
    cam0 = Camera(K, d0[None], None, (d0 > 0.5)[None].astype(np.uint8), poses[0], data_device=str(dev))
    eng = MappingEngine(model, cfg)
    t0 = time.perf_counter()
    losses = []
    for it in range(n_iter):
        st = eng.step(cam0, sync="lagged")
        if st is not None:
            losses.append(st["loss"])
    losses.append(eng.flush()["loss"])
    torch.cuda.synchronize(); t_map = time.perf_counter() - t0
    # tracker: reference = the keyframe as the MODEL renders it
    with torch.no_grad():
        pkg = render(cam0, model, cfg.depth_ratio)
        ref_depth = pkg["surf_depth"]
        ref_points = depth_to_points(cam0, ref_depth, transform_in_world=False).permute(1, 2, 0).reshape(-1, 3)
    prm = GSAlignerParams(image_height=H, image_width=W)
    al = GSAligner(**prm.__dict__)
    al.set_reference(ref_depth, ref_points, cam0.projection_matrix)
    rd = ref_depth[0].cpu().numpy()
    depth_err = float(np.abs(rd - d0)[(d0 > 0.5) & (rd > 0.5)].mean())
    kf_T_frame = torch.eye(4, device=dev)
    errs, fits, t_track = [], [], 0.0
    proj = DeviceProjector(H, W, 0.5, 100.0, device=dev)
    Kd = torch.tensor(K.reshape(-1), dtype=torch.float32, device=dev)
    for k in range(1, n_frames):
        dk, pk = scans[k]
        # the scan arrives as an unordered point cloud; the projector turns it into the query images on the device
        cloud = pk[dk > 0.5].astype(np.float32)
        cloud = torch.tensor(cloud[np.random.default_rng(k).permutation(len(cloud))], device=dev)
        lut, q_depth, _, q_valid = proj.project(cloud, Kd)
        q_points = cloud[lut.reshape(-1).clamp_min(0).long()] * q_valid.reshape(-1, 1)
        if k == 1:
            assert float((q_depth.cpu() - torch.tensor(dk)).abs().max()) < 1e-4, "projector disagrees with the ray caster"
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        al.set_query(q_depth[None], q_points, cam0.projection_matrix)
        kf_T_frame, fitness, info = al.align(kf_T_frame)
        torch.cuda.synchronize(); t_track += time.perf_counter() - t1
        gt = np.linalg.inv(poses[0]) @ poses[k]
        e = pose_error(kf_T_frame.cpu().numpy().astype(np.float64), gt)
        errs.append(e); fits.append(fitness)
        if verbose:
            print(f"frame {k}: |t_gt| {np.linalg.norm(gt[:3, 3]):.2f} m  error {e[0] * 100:.2f} cm / {math.degrees(e[1]):.3f} deg  fitness {fitness:.3f}")
    if verbose:
        print(f"surfels {eng.N}, mapping {n_iter} iterations in {t_map * 1e3:.1f} ms (loss {losses[0]:.4f} -> {losses[-1]:.4f}), "
              f"rendered-vs-scan depth {depth_err * 100:.2f} cm, tracking {t_track / (n_frames - 1) * 1e3:.2f} ms per frame")
    return dict(losses=losses, errs=errs, fits=fits, depth_err=depth_err, N=eng.N)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    run(*a)
