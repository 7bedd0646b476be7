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
from splat_loam_amd import slam_rules, synth
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


def _with_pix_offset(fn):
    """pix_offset=(ox, oy): run with that pixel-centre convention in the rasterizer (SlsCamera.pix_offset; D1 default
    (0, 0), the reference's own back-projection convention (-0.5, -0.5)), restoring the process default after."""
    import functools
    from splat_loam_amd import rasterizer

    @functools.wraps(fn)
    def wrapped(*a, pix_offset=None, **kw):
        old = rasterizer.pix_offset()
        if pix_offset is not None:
            rasterizer.set_pix_offset(*pix_offset)
        try:
            return fn(*a, **kw)
        finally:
            rasterizer.set_pix_offset(*old)
    return wrapped


@_with_pix_offset
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


@_with_pix_offset
def run_sequence(H=64, W=1024, n_frames=13, kf_every=None, n_iter=60, verbose=True, dev="cuda:0", out_dir=None,
                 first_stride=None, el_deg=None, densify_percentage=0.15, densify_threshold_opacity=0.5,
                 densify_threshold_egeom=-1.0, prob_view_last_keyframe=0.4, keyframe_threshold_distance=0.9,
                 keyframe_threshold_fitness=0.30, keyframe_threshold_nframes=-1, pruning_min_opacity=0.1,
                 step=(0.25, 0.04, 1.2)):
    """Odometry + mapping over a sequence, the reference's per-frame loop with this repository's
    components: every scan is registered against the latest keyframe as the MODEL renders it (tracker); a frame
    becomes a keyframe — at its ESTIMATED pose — when Tracker.require_new_keyframe says so (slam/tracker.py:61-84:
    distance / fitness / frame-count thresholds; `kf_every` overrides the rule with a fixed cadence); a keyframe is
    densified as Mapper.densify does (slam/mapper.py:51-135: candidates = valid pixels the model renders with
    alpha <= densify_threshold_opacity [+ the depth-error quantile mask], densify_percentage of them drawn with
    probability proportional to the log-depth gradient; the first keyframe draws from every valid pixel;
    `first_stride` replaces the draw of the FIRST keyframe by a regular column stride — a denser model than the
    reference's rule builds, kept as a stress option), then engine.remap, `n_iter` iterations over keyframes
    sampled as Mapper.optimize does (sample_geometric over the keyframe list), pruning (Mapper.prune) — all three through
    fused_mapper.update_model, the code behind the SLS_FUSED_MAPPER=1 binding.
    step: (dx, dy, yaw_deg) of the generating trajectory per frame."""
    dev = torch.device(dev)
    rng = np.random.default_rng(0)
    K = (synth.spherical_K(H, W) if el_deg is None else synth.spherical_K(H, W, el_deg[0], el_deg[1])).astype(np.float64)
    gt = [pose_of([step[0] * k, step[1] * k, 0.0], yaw_deg=step[2] * k) for k in range(n_frames)]
    cfg = MappingConfig()
    proj = DeviceProjector(H, W, 0.5, 100.0, device=dev)
    Kd = torch.tensor(K.reshape(-1), dtype=torch.float32, device=dev)

    def frame(k):      # unordered cloud -> images on the device
        dk, pk = room_scan(K, H, W, gt[k])
        cloud = torch.tensor(pk[dk > 0.5][rng.permutation(int((dk > 0.5).sum()))].astype(np.float32), device=dev)
        lut, depth, normals, valid = proj.project(cloud, Kd)
        points = cloud[lut.reshape(-1).clamp_min(0).long()] * valid.reshape(-1, 1)
        return depth, normals, valid, points

    from types import SimpleNamespace
    from splat_loam_amd import fused_mapper
    mcfg = SimpleNamespace(mapping=SimpleNamespace(
        num_iterations=n_iter - 1, densify_threshold_egeom=densify_threshold_egeom,
        densify_threshold_opacity=densify_threshold_opacity, densify_percentage=densify_percentage,
        prob_view_last_keyframe=prob_view_last_keyframe, pruning_min_opacity=pruning_min_opacity, pruning_min_size=0.0,
        opt_lambda_alpha=cfg.opt_lambda_alpha, opt_lambda_normal=cfg.opt_lambda_normal, opt_scaling_max=cfg.opt_scaling_max,
        opt_scaling_max_penalty=cfg.opt_scaling_max_penalty), opt=SimpleNamespace(depth_ratio=cfg.depth_ratio))
    totals = {}

    def add_keyframe(model, est_pose, depth, normals, valid, first):
        """Mapper.update_model (slam/mapper.py:33-47) for this keyframe: fused_mapper.update_model — densify, n_iter
        iterations over the keyframes drawn as Mapper.optimize draws them, prune (pinned by golden G7)."""
        cam = Camera(K, depth[None], normals.permute(2, 0, 1), valid[None], est_pose, data_device=str(dev))
        frm = SimpleNamespace(camera=cam, model_T_frame=torch.tensor(est_pose, dtype=torch.float32, device=dev))
        kfs.append(frm)
        drawn = None
        if first and first_stride:
            drawn = valid.clone().bool()
            if first_stride > 1:
                keep_cols = torch.zeros(drawn.shape[1], dtype=torch.bool, device=dev); keep_cols[::first_stride] = True
                drawn &= keep_cols[None, :]
        res = fused_mapper.update_model(model, kfs, frm, mcfg, initialize_model=first, drawn=drawn, rng=rng,
                                        generator=torch.Generator(device=dev).manual_seed(len(kfs) - 1))
        eng = fused_mapper.engine_of(model)
        for k_, v_ in (eng.stats.items() if eng else ()):
            totals[k_] = totals.get(k_, 0) + v_
        return cam, res["added"], int(res["removed"].sum())

    kfs, est = [], [gt[0].copy()]
    t0 = time.perf_counter()
    empty = lambda w: torch.zeros((0, w), dtype=torch.float32)
    model = SurfelModel(empty(3), empty(2), empty(4), empty(1), device=str(dev))
    model.training_setup(fused=True)
    kf_cam, n_new, n_pruned = add_keyframe(model, est[0], *frame(0)[:3], first=True)
    kf_pose = est[0]
    log = [(0, n_new, n_pruned, int(model._xyz.shape[0]))]
    prm = GSAlignerParams(image_height=H, image_width=W)
    al = GSAligner(**prm.__dict__)

    def set_reference(cam):
        with torch.no_grad():
            ref_depth = render(cam, model, cfg.depth_ratio)["surf_depth"]
            ref_points = depth_to_points(cam, ref_depth, transform_in_world=False).permute(1, 2, 0).reshape(-1, 3)
        al.set_reference(ref_depth, ref_points, cam.projection_matrix)

    set_reference(kf_cam)
    kf_T_frame = torch.eye(4, device=dev)
    errs = [(0.0, 0.0)]
    tracked = 0
    for k in range(1, n_frames):
        depth, normals, valid, points = frame(k)
        al.set_query(depth[None], points, kf_cam.projection_matrix)
        kf_T_frame, fitness, _ = al.align(kf_T_frame)
        tracked += 1
        pose = kf_pose @ kf_T_frame.cpu().numpy().astype(np.float64)
        est.append(pose)
        errs.append(pose_error(pose, gt[k]))
        new_kf = (k % kf_every == 0) if kf_every else slam_rules.require_new_keyframe(
            tracked, float(fitness), kf_T_frame, keyframe_threshold_nframes, keyframe_threshold_fitness,
            keyframe_threshold_distance)
        if new_kf:
            tracked = 0
            kf_cam, n_new, n_pruned = add_keyframe(model, pose, depth, normals, valid, first=False)
            kf_pose, kf_T_frame = pose, torch.eye(4, device=dev)
            set_reference(kf_cam)
            log.append((k, n_new, n_pruned, int(model._xyz.shape[0])))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if out_dir is not None:
        # what SLAM.save_results leaves behind (slam/slam.py:130-170): odom.txt, graph.yaml, models/%04d.ply
        from splat_loam_amd import ply_io, traj_io
        os.makedirs(os.path.join(out_dir, "models"), exist_ok=True)
        traj_io.write_tum(os.path.join(out_dir, "odom.txt"), est, [0.1 * k for k in range(n_frames)])
        ply_io.save_ply(os.path.join(out_dir, "models", "0000.ply"), model._xyz.detach(), model._opacity.detach(),
                        model._scaling.detach(), model._rotation.detach())
        traj_io.write_graph(os.path.join(out_dir, "graph.yaml"),
                            [dict(id=0, world_T_model=np.eye(4), filename="models/0000.ply", frame_ids=list(range(n_frames)))],
                            [dict(id=k, timestamp=0.1 * k, model_T_frame=est[k], projmatrix=kf_cam.projection_matrix.cpu().numpy(),
                                  model_id=0) for k in range(n_frames)])
    if verbose:
        for k, n_new, n_pruned, n in log:
            print(f"keyframe at frame {k}: +{n_new} surfels, -{n_pruned} pruned, model {n}")
        print("pose error per frame [cm]:", " ".join(f"{e[0] * 100:.1f}" for e in errs))
        print(f"{n_frames} frames, {len(kfs)} keyframes in {dt * 1e3:.0f} ms; final error {errs[-1][0] * 100:.2f} cm / {math.degrees(errs[-1][1]):.3f} deg "
              f"after {np.linalg.norm(gt[-1][:3, 3]):.2f} m; engine stats {totals}")
    return dict(errs=errs, log=log, N=int(model._xyz.shape[0]), est=est, gt=gt, seconds=dt, stats=dict(totals))


if __name__ == "__main__":
    off = None
    if "--half-pixel" in sys.argv:
        sys.argv.remove("--half-pixel"); off = (-0.5, -0.5)
    if len(sys.argv) > 1 and sys.argv[1] == "sequence":
        run_sequence(*[int(x) for x in sys.argv[2:]], pix_offset=off)
    else:
        run(*[int(x) for x in sys.argv[1:]], pix_offset=off)
