#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python
(/root/reference, importable only in the build container) and recording inputs
and outputs.  Only data is committed; no reference source travels.

    python tools/make_golden.py            # rewrites tests/golden/g*.npz

G1  camera conventions + spherical back-projection: scene/cameras.py:10-50,
    utils/graphic_utils.py:26-88
G2  gaussian_renderer.render() post-processing on a seeded allmap
    (gaussian_renderer/__init__.py:48-93) and its gradient for a weighted sum
G3  build_rotation / inverse_sigmoid / matrix_to_quaternion /
    create_rotation_matrix_from_direction_vector_batch / sample_geometric
    (utils/general_utils.py, utils/sampling_utils.py)
G5  Mapper.optimize (slam/mapper.py:140-214) for 3 Adam iterations with the
    reference's GaussianModel.training_setup, on top of the CPU checker injected
    as `diff_surfel_spherical_rasterization` -> parameter trajectories
G7  Mapper.update_model stage by stage (slam/mapper.py:33-47: densify -> optimize -> prune) over THREE keyframes of a
    ray-cast room, 21 iterations each, on the CPU checker, with `distCUDA2` = a brute-force 3-NN: inputs, the
    rendered alpha the densification looks at, the drawn pixels, the drawn keyframes, the surfel set after every
    stage.  Replayed by tests/test_fused_mapper.py (CPU: the cold stages; GPU: the whole thing on MappingEngine).
G8  on-disk formats: the table GaussianModel.save_ply hands to plyfile (recording stand-in), what GaussianModel.load_ply
    reads from a file written by ply_io.save_ply, ResultGraph.from_slam's dataclass fields (scene/postprocessing.py)
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
OUT = os.path.join(ROOT, "tests", "golden")


def brute_force_dist2(points):
    """simple-knn's contract (slam/mapper.py:113-115): per point the mean squared distance to its 3 nearest others"""
    p = points.detach().double()
    d2 = torch.cdist(p, p) ** 2
    d2.fill_diagonal_(float("inf"))
    return d2.topk(3, dim=1, largest=False).values.mean(dim=1).float()


def install_stubs():
    from oracle import torch_function as tf
    m = types.ModuleType("diff_surfel_spherical_rasterization")
    m.GaussianRasterizer = tf.GaussianRasterizer
    m.GaussianRasterizationSettings = tf.GaussianRasterizationSettings
    sys.modules["diff_surfel_spherical_rasterization"] = m
    knn = types.ModuleType("simple_knn")
    knn_c = types.ModuleType("simple_knn._C")
    knn_c.distCUDA2 = brute_force_dist2
    sys.modules["simple_knn"], sys.modules["simple_knn._C"] = knn, knn_c
    gs = types.ModuleType("gsaligner")

    class GSAlignerParams:  # dataclass field default in utils/config_utils.py
        pass

    class GSAligner:
        pass
    gs.GSAlignerParams, gs.GSAligner = GSAlignerParams, GSAligner
    sys.modules["gsaligner"] = gs
    for name in ("plyfile", "rerun"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["rerun"].__path__ = []
    sys.modules["rerun.blueprint"] = types.ModuleType("rerun.blueprint")
    sys.modules["rerun"].blueprint = sys.modules["rerun.blueprint"]
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    try:
        import omegaconf  # noqa: F401
    except ImportError:
        oc = types.ModuleType("omegaconf")

        class OmegaConf:
            pass
        oc.OmegaConf = OmegaConf
        sys.modules["omegaconf"] = oc


def g1():
    from scene.cameras import Camera
    from utils.graphic_utils import depth_to_normal, depth_to_points
    from splat_loam_amd import synth
    rng = np.random.default_rng(1)
    out = {}
    H, W = 8, 16
    for i, (hfov, pose) in enumerate(((360.0, np.eye(4)), (120.0, synth.keyframe_poses(3)[2]), (360.0, None))):
        K = synth.spherical_K(H, W, hfov_deg=hfov)
        if pose is None:
            a = 0.7
            pose = np.eye(4)
            pose[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
            pose[:3, 3] = [1.0, -2.0, 0.5]
        depth = rng.uniform(2, 30, (1, H, W)).astype(np.float32)
        cam = Camera(K, depth, np.zeros((3, H, W), np.float32), np.ones((1, H, W), np.uint8),
                     world_T_lidar=pose.astype(np.float32), data_device="cpu")
        d = torch.from_numpy(depth)
        out[f"K{i}"], out[f"pose{i}"], out[f"depth{i}"] = K, pose.astype(np.float32), depth
        out[f"view{i}"] = cam.world_view_transform.numpy()
        out[f"proj{i}"] = cam.projection_matrix.numpy()
        out[f"pts_sensor{i}"] = depth_to_points(cam, d, False).numpy()
        out[f"pts_world{i}"] = depth_to_points(cam, d, True).numpy()
        out[f"normal{i}"] = depth_to_normal(cam, d).numpy()
    np.savez_compressed(os.path.join(OUT, "g1_camera.npz"), **out)


class _FixedAllmapRasterizer(torch.nn.Module):
    allmap = None
    radii = None

    def __init__(self, raster_settings):
        super().__init__()

    def forward(self, **kw):
        return _FixedAllmapRasterizer.radii, _FixedAllmapRasterizer.allmap


def g2():
    import gaussian_renderer
    from scene.cameras import Camera
    from splat_loam_amd import synth
    rng = np.random.default_rng(2)
    H, W = 8, 16
    K = synth.spherical_K(H, W)
    pose = synth.keyframe_poses(2)[1]
    cam = Camera(K, np.ones((1, H, W), np.float32), np.zeros((3, H, W), np.float32), np.ones((1, H, W), np.uint8),
                 world_T_lidar=pose.astype(np.float32), data_device="cpu")
    alpha = rng.uniform(0.2, 1.0, (H, W))
    alpha[rng.uniform(size=(H, W)) < 0.15] = 0.0      # some empty pixels
    allmap = np.zeros((7, H, W), np.float32)
    allmap[1] = alpha
    allmap[0] = alpha * rng.uniform(3, 20, (H, W))
    nrm = rng.normal(size=(3, H, W)); nrm /= np.linalg.norm(nrm, axis=0, keepdims=True)
    allmap[2:5] = nrm * alpha
    allmap[5] = rng.uniform(3, 20, (H, W)) * (alpha > 0)
    allmap[6] = rng.uniform(0, 0.01, (H, W)) * (alpha > 0)
    out = {"K": K, "pose": pose.astype(np.float32), "allmap": allmap}

    class M:
        get_xyz = torch.zeros(5, 3)
        get_opacity = get_scaling = get_rotation = None
    saved = gaussian_renderer.GaussianRasterizer
    gaussian_renderer.GaussianRasterizer = _FixedAllmapRasterizer
    try:
        for ratio in (0.0, 0.3):
            am = torch.from_numpy(allmap.copy()).requires_grad_(True)
            _FixedAllmapRasterizer.allmap = am * 1.0       # non-leaf, so the in-place writes are legal
            _FixedAllmapRasterizer.radii = torch.ones(5, dtype=torch.int32)
            pkg = gaussian_renderer.render(cam, M, ratio)
            wts = {k: torch.from_numpy(rng.normal(size=tuple(pkg[k].shape)).astype(np.float32))
                   for k in ("rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal")}
            loss = sum((pkg[k] * wts[k]).sum() for k in wts)
            loss.backward()
            tag = f"_r{int(ratio * 10)}"
            for k in wts:
                out[k + tag] = pkg[k].detach().numpy()
                out["w_" + k + tag] = wts[k].numpy()
            out["grad_allmap" + tag] = am.grad.numpy()
    finally:
        gaussian_renderer.GaussianRasterizer = saved
    np.savez_compressed(os.path.join(OUT, "g2_render.npz"), **out)


def g3():
    from utils import general_utils as gu
    from utils import sampling_utils as su
    rng = np.random.default_rng(3)
    q = rng.normal(size=(16, 4)).astype(np.float32)
    out = {"q": q, "R": gu.build_rotation(torch.from_numpy(q)).numpy()}
    x = rng.uniform(0.01, 0.99, 32).astype(np.float32)
    out["x"], out["inv_sigmoid"] = x, gu.inverse_sigmoid(torch.from_numpy(x)).numpy()
    d = rng.normal(size=(16, 3)).astype(np.float32)
    Rm = gu.create_rotation_matrix_from_direction_vector_batch(torch.from_numpy(d))
    out["dirs"], out["R_from_dirs"] = d, Rm.numpy()
    out["quat_from_R"] = gu.matrix_to_quaternion(Rm).numpy()
    for n, p in ((1, 0.4), (5, 0.4), (8, 0.7)):
        out[f"geom_{n}_{int(p * 10)}"] = np.asarray(su.sample_geometric(list(range(n)), p), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g3_utils.npz"), **out)


def g5():
    from scene.cameras import Camera
    from scene.frame import Frame
    from scene.gaussian_model import GaussianModel
    from slam.local_model import LocalModel
    from slam.mapper import Mapper
    from utils.config_utils import Configuration
    from splat_loam_amd import synth
    N, H, W = 200, 16, 64
    sc = synth.make_scene(N, H, W, seed=5, range_lo=2.0, range_hi=12.0, scale_lo=0.05, scale_hi=0.3)
    depth, valid = synth.make_targets(H, W, sc)
    valid = valid.copy(); valid[0, :2, :5] = 0            # a few invalid pixels
    pose = synth.keyframe_poses(2)[1].astype(np.float32)
    cfg = Configuration()
    cfg.device = "cpu"
    cfg.logging.enable = False
    cfg.mapping.num_iterations = 2                         # loop runs num_iterations + 1 = 3 times
    cfg.mapping.opt_lambda_alpha, cfg.mapping.opt_lambda_normal = 0.4, 0.5
    cfg.mapping.opt_scaling_max, cfg.mapping.opt_scaling_max_penalty = 0.1, 1.0
    cfg.mapping.prob_view_last_keyframe = -1.0
    cfg.opt.depth_ratio = 0.0
    cam = Camera(sc["K"], depth, np.zeros((3, H, W), np.float32), valid, world_T_lidar=pose, data_device="cpu")
    frame = Frame(cam, 0.0, "cpu")
    gm = GaussianModel("cpu")
    gm._xyz = torch.nn.Parameter(torch.tensor(sc["means"]))
    gm._scaling = torch.nn.Parameter(torch.log(torch.tensor(sc["scales"])))
    gm._rotation = torch.nn.Parameter(torch.tensor(sc["rots"]) * 1.3)      # un-normalised raw quaternions
    gm._opacity = torch.nn.Parameter(torch.log(torch.tensor(sc["opac"]) / (1 - torch.tensor(sc["opac"]))))
    init = {k: getattr(gm, k).detach().numpy().copy() for k in ("_xyz", "_scaling", "_rotation", "_opacity")}
    gm.training_setup(cfg)
    lm = LocalModel(cfg)
    lm.gmodel = gm if hasattr(lm, "gmodel") else gm
    for attr in ("gmodel", "_gmodel", "model"):
        if hasattr(lm, attr):
            setattr(lm, attr, gm)
    lm.keyframes = [frame]
    mapper = Mapper(cfg)
    mapper.register_model(lm)
    assert lm.get_gmodel is gm
    np.random.seed(0)
    mapper.optimize()
    out = {"K": sc["K"], "pose": pose, "depth": depth, "valid": valid,
           "lr": np.array([cfg.opt.position_lr, cfg.opt.opacity_lr, cfg.opt.scaling_lr, cfg.opt.rotation_lr])}
    for k, v in init.items():
        out["init" + k] = v
        out["final" + k] = getattr(gm, k).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "g5_mapper.npz"), **out)


def g6():
    """The SLAM rules that size the hot path's workload: compute_depth_gradient (utils/graphic_utils.py:91-106)
    on a seeded range image with holes, sample_geometric for the bench's 8-keyframe window, and
    Tracker.require_new_keyframe's truth table (slam/tracker.py:61-84) evaluated by the reference's own method
    on a stand-in object carrying the three fields it reads."""
    from utils import graphic_utils as gu
    from utils import sampling_utils as su
    rng = np.random.default_rng(6)
    H, W = 12, 40
    depth = rng.uniform(2.0, 40.0, size=(1, H, W)).astype(np.float32)
    depth[0, 3:6, 10:14] = 0.0                                   # log(0) = -inf -> 0
    valid = (rng.uniform(size=(1, H, W)) > 0.15)
    valid[0, 3:6, 10:14] = False
    out = {"depth": depth, "valid": valid.astype(np.uint8),
           "depth_gradient": gu.compute_depth_gradient(torch.from_numpy(depth), torch.from_numpy(valid)).numpy(),
           "geom_8_4": np.asarray(su.sample_geometric(list(range(8)), 0.4), dtype=np.float64),
           "geom_2_4": np.asarray(su.sample_geometric(list(range(2)), 0.4), dtype=np.float64)}
    from slam.tracker import Tracker
    from types import SimpleNamespace
    rows = []
    for nfr, fit, dist in ((3, 0.9, 0.2), (12, 0.9, 0.2), (3, 0.2, 0.2), (3, 0.9, 6.0), (12, 0.1, 9.0)):
        for thr in ((-1, -1.0, 1.0), (10, -1.0, -1.0), (-1, 0.3, 5.0), (10, 0.3, 5.0), (0, 0.0, 0.0)):
            T = torch.eye(4); T[0, 3] = dist
            fake = SimpleNamespace(
                cfg=SimpleNamespace(tracking=SimpleNamespace(keyframe_threshold_nframes=thr[0],
                                                             keyframe_threshold_fitness=thr[1],
                                                             keyframe_threshold_distance=thr[2])),
                num_frames_tracked=nfr, aligner=SimpleNamespace(fitness=lambda f=fit: f), keyframe_T_frame=T)
            rows.append([nfr, fit, dist, thr[0], thr[1], thr[2], float(bool(Tracker.require_new_keyframe(fake)))])
    out["keyframe_rule"] = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g6_slam_rules.npz"), **out)


class _Recorder:
    """stands in for Mapper.data_logger: keeps the one image densify() logs — the drawn pixels (slam/mapper.py:97)"""
    def __init__(self):
        self.images = {}

    def log_image(self, name, img):
        self.images[name] = img.clone()

    def __getattr__(self, _):
        return lambda *a, **k: None


def g7():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gaussian_renderer
    from test_aligner import pose_of, room_scan
    from scene.cameras import Camera
    from scene.frame import Frame
    from slam.local_model import LocalModel
    from slam.mapper import Mapper
    from utils.config_utils import Configuration
    from splat_loam_amd import slam_rules, synth
    H, W, n_kf = 32, 256, 3
    K = synth.spherical_K(H, W)
    cfg = Configuration()
    cfg.device = "cpu"
    cfg.logging.enable = False
    m = cfg.mapping                                        # configs/kitti/kitti-00-odom.yaml, fewer iterations
    m.num_iterations = 20                                  # the loop runs num_iterations + 1 = 21 times
    m.densify_threshold_egeom, m.densify_threshold_opacity, m.densify_percentage = -1.0, 0.2, 0.3
    m.prob_view_last_keyframe = 0.4                        # (kitti.yaml's; kitti-00 draws uniformly)
    m.pruning_min_opacity, m.pruning_min_size = 0.0, 0.0   # (the opacity threshold is set per keyframe below)
    m.opt_lambda_alpha, m.opt_lambda_normal, m.opt_scaling_max, m.opt_scaling_max_penalty = 0.4, 0.5, 0.1, 1.0
    cfg.opt.depth_ratio = 0.0
    lm = LocalModel(cfg)                                   # GaussianModel("cpu") + training_setup(cfg)
    gm = lm.get_gmodel
    mapper = Mapper(cfg)
    mapper.register_model(lm)
    rec = mapper.data_logger = _Recorder()
    snap = lambda: np.concatenate([getattr(gm, a).detach().numpy().reshape(gm._xyz.shape[0], -1)
                                   for a in ("_xyz", "_opacity", "_scaling", "_rotation")], axis=1).astype(np.float32)
    out = {"K": K, "H": H, "W": W, "n_keyframes": n_kf, "num_iterations": m.num_iterations,
           "lr": np.array([cfg.opt.position_lr, cfg.opt.opacity_lr, cfg.opt.scaling_lr, cfg.opt.rotation_lr]),
           "cfg": np.array([m.densify_threshold_opacity, m.densify_percentage, m.prob_view_last_keyframe,
                            m.opt_lambda_alpha, m.opt_lambda_normal, m.opt_scaling_max, m.opt_scaling_max_penalty])}
    torch.manual_seed(7)
    for k in range(n_kf):
        pose = pose_of([0.45 * k, 0.10 * k, 0.0], yaw_deg=4.0 * k).astype(np.float32)
        depth, pts = room_scan(K.astype(np.float64), H, W, pose.astype(np.float64))
        valid = (depth > 0.5)
        valid[:2, 10 * k:10 * k + 7] = False                # a few pixels without a measurement
        depth = np.where(valid, depth, 0.0).astype(np.float32)
        normal = np.where(valid[..., None], -pts / np.maximum(np.linalg.norm(pts, axis=-1, keepdims=True), 1e-9), 0.0)
        normal = np.ascontiguousarray(normal.transpose(2, 0, 1), np.float32)     # scene/preprocessing.py:112: -unit(point)
        cam = Camera(K, depth[None], normal, valid[None].astype(np.uint8), world_T_lidar=pose, data_device="cpu")
        frame = Frame(cam, float(k), "cpu", model_T_frame=pose)
        lm.insert_keyframe(frame)
        tag = f"_k{k}"
        out.update({"pose" + tag: pose, "depth" + tag: depth[None], "normal" + tag: normal,
                    "valid" + tag: valid[None].astype(np.uint8)})
        if k > 0:     # what densify() looks at (slam/mapper.py:52-61), rendered by the reference's render() just before
            with torch.no_grad():
                out["alpha" + tag] = gaussian_renderer.render(cam, gm, cfg.opt.depth_ratio)["rend_alpha"].numpy().copy()
        mapper.densify(frame, initialize_model=(k == 0))
        out["drawn" + tag] = rec.images["frame/densify_mask"].numpy().astype(bool)
        out["added" + tag] = snap()[-int(out["drawn" + tag].sum()):]      # (the rows in front = the set after the last prune)
        np.random.seed(100 + k)
        mapper.optimize()
        np.random.seed(100 + k)      # the same draws once more, for the record (np.random.choice over n consumes the same stream)
        p = slam_rules.keyframe_probabilities(k + 1, m.prob_view_last_keyframe)
        out["kf_draws" + tag] = np.array([np.random.choice(k + 1, p=p) for _ in range(m.num_iterations + 1)])
        out["after_optimize" + tag] = snap()
        # prune below an opacity threshold placed in the widest gap of the opacities between 0.80 and 0.88, so that the
        # decision does not hang on the last bits of a trajectory
        op = np.sort(gm.get_opacity.detach().numpy().reshape(-1))
        band = op[(op > 0.80) & (op < 0.88)]
        gaps = np.diff(band)
        j = int(np.argmax(gaps))
        m.pruning_min_opacity = float(0.5 * (band[j] + band[j + 1]))
        out["prune_threshold" + tag] = np.float64(m.pruning_min_opacity)
        out["prune_margin" + tag] = np.float64(0.5 * gaps[j])
        before = gm.get_opacity.detach().numpy().reshape(-1)
        mapper.prune()
        out["pruned" + tag] = before < np.float32(m.pruning_min_opacity)
        assert np.array_equal(snap(), out["after_optimize" + tag][~out["pruned" + tag]])     # (so it is not stored)
        states = [gm.optimizer.state.get(g["params"][0]) for g in gm.optimizer.param_groups]
        assert all(s is None or len(s) == 0 for s in states), "the prune leaves the new parameters without Adam state"
        print(f"  keyframe {k}: +{int(out['drawn' + tag].sum())} -> {out['after_optimize' + tag].shape[0]} surfels, "
              f"pruned {int(out['pruned' + tag].sum())} below {m.pruning_min_opacity:.5f} (margin {out['prune_margin' + tag]:.1e}), "
              f"draws {out['kf_draws' + tag].tolist()}")
    np.savez_compressed(os.path.join(OUT, "g7_update_model.npz"), **out)


def g8():
    """On-disk formats, pinned by what the reference's own writers build:
    (a) GaussianModel.save_ply (scene/gaussian_model.py:123-168) run with a RECORDING stand-in for plyfile
        (`PlyElement.describe` keeps the structured array it is handed, `PlyData.write` the file name): property
        names, dtypes and values of the table the reference gives to plyfile;
    (b) GaussianModel.load_ply (:170-221) run on a file written by THIS repo's ply_io.save_ply, through a stand-in
        `PlyData.read` (a 20-line header parser of its own, below; `device="cuda"` redirected to the CPU): what the
        reference reads back from our bytes;
    (c) ResultGraph.from_slam (scene/postprocessing.py:44-83) on stand-in local models: the dataclass fields.
    NOT covered: the TUM / KITTI trajectory writers (utils/trajectory_utils.py:185-242) — that module imports
    pytransform3d at load time, which is not installed here; traj_io's odom.txt is checked against its own reader only."""
    import dataclasses
    import hashlib
    import tempfile
    from pathlib import Path
    from types import SimpleNamespace
    import scene.gaussian_model as gmod
    from splat_loam_amd import ply_io
    rec = {}

    class PlyElement:
        @staticmethod
        def describe(arr, name):
            rec["elements"], rec["name"] = np.array(arr, copy=True), name
            return ("element", name)

    class _Prop:
        def __init__(self, name):
            self.name = name

    class _Element:
        def __init__(self, table):
            self.table = table
            self.properties = [_Prop(n) for n in table.dtype.names]

        def __getitem__(self, name):
            return self.table[name]

    class PlyData:
        def __init__(self, elements=None):
            self.elements = elements

        def write(self, filename):
            rec["filename"] = str(filename)

        @staticmethod
        def read(path):
            blob = open(path, "rb").read()
            end = blob.index(b"end_header\n") + 11
            head = blob[:end].decode("ascii").split("\n")
            assert head[0] == "ply" and head[1] == "format binary_little_endian 1.0"
            n = int(head[2].split()[2])
            props = [ln.split() for ln in head if ln.startswith("property")]
            assert all(pr[1] == "float" for pr in props)
            table = np.frombuffer(blob, dtype=[(pr[2], "<f4") for pr in props], count=n, offset=end)
            return PlyData([_Element(table)])

    gmod.PlyElement, gmod.PlyData = PlyElement, PlyData
    rng = np.random.default_rng(8)
    n = 37
    raw = {"xyz": rng.normal(size=(n, 3)), "opacity": rng.normal(size=(n, 1)), "scaling": rng.normal(size=(n, 2)),
           "rotation": rng.normal(size=(n, 4))}
    raw = {k: v.astype(np.float32) for k, v in raw.items()}
    gm = gmod.GaussianModel("cpu")
    gm._xyz = torch.nn.Parameter(torch.tensor(raw["xyz"]))
    gm._opacity = torch.nn.Parameter(torch.tensor(raw["opacity"]))
    gm._scaling = torch.nn.Parameter(torch.tensor(raw["scaling"]))
    gm._rotation = torch.nn.Parameter(torch.tensor(raw["rotation"]))
    tmp = Path(tempfile.mkdtemp())
    gm.save_ply(tmp / "models" / "0000.ply")
    el = rec["elements"]
    out = {"in_" + k: v for k, v in raw.items()}
    out["ply_element_name"] = np.array(rec["name"])
    out["ply_names"] = np.array(list(el.dtype.names))
    out["ply_dtypes"] = np.array([el.dtype[nm].str for nm in el.dtype.names])
    out["ply_table"] = np.stack([el[nm] for nm in el.dtype.names], axis=1)          # (n, 13), the table's own dtype (f4)
    assert rec["filename"].endswith("0000.ply")
    # (b) the reference's reader on OUR writer's bytes
    ours = tmp / "ours.ply"
    ply_io.save_ply(ours, raw["xyz"], raw["opacity"], raw["scaling"], raw["rotation"])
    out["ours_sha256"] = np.array(hashlib.sha256(open(ours, "rb").read()).hexdigest())

    class _TorchOnCpu:                      # load_ply builds its tensors with device="cuda" (gaussian_model.py:205-220)
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def tensor(data, dtype=None, device=None):
            return torch.tensor(data, dtype=dtype)
    gmod.torch = _TorchOnCpu()
    gm2 = gmod.GaussianModel("cpu")
    gm2.load_ply(str(ours))
    gmod.torch = torch
    for k, attr in (("xyz", "_xyz"), ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation")):
        out["ref_loaded_" + k] = getattr(gm2, attr).detach().numpy()
        assert np.array_equal(out["ref_loaded_" + k], raw[k]), k
    # (c) ResultGraph.from_slam
    for name in ("open3d",):
        sys.modules.setdefault(name, types.ModuleType(name))
    import scene.postprocessing as pp
    from splat_loam_amd import synth
    poses = synth.keyframe_poses(5)
    K = synth.spherical_K(64, 1024)
    proj = np.eye(4, dtype=np.float32)
    proj[:3, :3] = K.T

    def frame(i):
        return SimpleNamespace(timestamp=1.7e9 + 0.1 * i, model_T_frame=torch.tensor(poses[i], dtype=torch.float32),
                               camera=SimpleNamespace(projection_matrix=torch.tensor(proj)))
    wTm = np.eye(4, dtype=np.float32)
    wTm[:3, 3] = (3.0, -1.0, 0.25)
    models = [SimpleNamespace(world_T_model=torch.eye(4), keyframes=[frame(0), frame(1), frame(2)]),
              SimpleNamespace(world_T_model=torch.tensor(wTm), keyframes=[frame(3), frame(4)])]
    graph = pp.ResultGraph.from_slam(None, models, Path("results/run0"))
    d = dataclasses.asdict(graph)
    assert [f.name for f in dataclasses.fields(pp.ResultModel)] == ["id", "world_T_model", "filename", "frame_ids"]
    assert [f.name for f in dataclasses.fields(pp.ResultFrame)] == ["id", "timestamp", "model_T_frame", "projmatrix", "model_id"]
    out["graph_model_fields"] = np.array([f.name for f in dataclasses.fields(pp.ResultModel)])
    out["graph_frame_fields"] = np.array([f.name for f in dataclasses.fields(pp.ResultFrame)])
    out["graph_in_poses"] = np.stack(poses).astype(np.float32)
    out["graph_in_proj"], out["graph_in_wTm"] = proj, wTm
    out["graph_model_id"] = np.array([m["id"] for m in d["models"]])
    out["graph_model_world_T_model"] = np.array([m["world_T_model"] for m in d["models"]], dtype=np.float64)
    out["graph_model_filename"] = np.array([str(m["filename"]) for m in d["models"]])
    out["graph_model_frame_ids"] = np.array([",".join(str(i) for i in m["frame_ids"]) for m in d["models"]])
    out["graph_frame_id"] = np.array([f["id"] for f in d["frames"]])
    out["graph_frame_timestamp"] = np.array([f["timestamp"] for f in d["frames"]], dtype=np.float64)
    out["graph_frame_model_T_frame"] = np.array([f["model_T_frame"] for f in d["frames"]], dtype=np.float64)
    out["graph_frame_projmatrix"] = np.array([f["projmatrix"] for f in d["frames"]], dtype=np.float64)
    out["graph_frame_model_id"] = np.array([f["model_id"] for f in d["frames"]])
    np.savez_compressed(os.path.join(OUT, "g8_formats.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.manual_seed(0)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g5", "g6", "g7", "g8"]
    for name in which:
        globals()[name]()
        print("wrote", name)
