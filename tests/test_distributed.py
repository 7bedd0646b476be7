"""Keyframe-parallel mapping with world_size 2 over gloo on CPU:
the all-reduced gradients equal the sum of the per-keyframe gradients, replicas
stay bit-identical, and the scale regulariser is counted once.  The rasterizer
under the harness is the CPU checker (rasterizer_cls test seam); on the GPU box
the same code path runs over RCCL with the HIP rasterizer (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from splat_loam_amd import synth
from splat_loam_amd.mapping import MappingConfig, mapping_loss, optimize_step_sharded
from splat_loam_amd.renderer import render
from splat_loam_amd.scene import Camera, SurfelModel

N, H, W = 300, 16, 64


def _setup(rank):
    from oracle.torch_function import GaussianRasterizer as OracleRasterizer
    sc = synth.make_scene(N, H, W, seed=21, range_lo=2.0, range_hi=10.0, scale_lo=0.05, scale_hi=0.25)
    depth, valid = synth.make_targets(H, W, sc)
    poses = synth.keyframe_poses(2)
    cams = [Camera(sc["K"], depth, None, valid, p, data_device="cpu") for p in poses]
    model = SurfelModel.from_activated(sc["means"], sc["scales"], sc["rots"], sc["opac"], device="cpu")
    model.training_setup(fused=False)
    return cams, model, OracleRasterizer


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cams, model, R = _setup(rank)
    cfg = MappingConfig()
    for _ in range(2):
        optimize_step_sharded(model, cams[rank], cfg, rasterizer_cls=R)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), xyz=model._xyz.detach().numpy(),
             scaling=model._scaling.detach().numpy(), rotation=model._rotation.detach().numpy(),
             opacity=model._opacity.detach().numpy(), grad_xyz=model._xyz.grad.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_matches_summed_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f"replicas diverged in {k}"
    # single-process emulation: sum of per-keyframe gradients, regulariser once
    cams, model, R = _setup(0)
    cfg = MappingConfig()
    for _ in range(2):
        model.optimizer.zero_grad(set_to_none=True)
        total = 0
        for g, cam in enumerate(cams):
            c = cfg if g == 0 else MappingConfig(**{**cfg.__dict__, "opt_scaling_max_penalty": 0.0})
            total = total + mapping_loss(render(cam, model, cfg.depth_ratio, rasterizer_cls=R), cam, model, c)
        total.backward()
        last_grad = model._xyz.grad.numpy().copy()
        model.optimizer.step()
    assert np.abs(r0["grad_xyz"] - last_grad).max() <= 1e-5 * np.abs(last_grad).max()
    for k, p in (("xyz", model._xyz), ("scaling", model._scaling), ("rotation", model._rotation), ("opacity", model._opacity)):
        ref = p.detach().numpy()
        assert np.abs(r0[k] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-6), k


# ---------------------------------------------------------------------------
# The engine's second exchange scheme (engine.py: dp_mode "rs_ag"): gradients in G chunks -> reduce-scatter ->
# Adam on the rank's own chunk with ITS shard of the moments -> all-gather of the parameters.  Emulated here
# with torch ops on the CPU over gloo (the native step / Adam kernels need the GPU: tests/test_gpu_parity.py
# ::test_engine_keyframe_parallel_two_ranks runs the real thing): the layout helpers the engine uses
# (dp_chunk, dp_pieces, dp_chunked) must reproduce, bit for bit, what one all-reduce + Adam everywhere gives.
# ---------------------------------------------------------------------------
def _adam_torch(p, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-15):
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def _rsag_worker(rank, world, port, out_dir, n):
    from splat_loam_amd.engine import dp_chunk, dp_chunked, dp_pieces
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lrs = (5e-4, 5e-2, 5e-3, 1e-3)
    g0 = torch.Generator().manual_seed(7)
    params = torch.randn(10 * n, generator=g0)
    pa, pb = params.clone(), None
    C = dp_chunk(n, world)
    flat = torch.zeros(world * C); flat[:10 * n] = params
    ma, va = torch.zeros(10 * n), torch.zeros(10 * n)
    lo, hi = min(rank * C, 10 * n), min((rank + 1) * C, 10 * n)
    ms, vs = torch.zeros(C), torch.zeros(C)
    for t in (1, 2, 3):
        grad = torch.randn(10 * n, generator=torch.Generator().manual_seed(100 * t + rank))
        void = torch.tensor([0.0, 0.0])
        # (a) all-reduce of the flat bucket + Adam on everything
        ga = torch.cat([grad, void]); dist.all_reduce(ga)
        for a, b, lr in dp_pieces(n, 0, 10 * n, lrs):
            _adam_torch(pa[a:b], ga[a:b], ma[a:b], va[a:b], lr, t)
        # (b) chunked layout -> reduce-scatter -> Adam on the own chunk -> all-gather
        phys = dp_chunked(grad, C, world)
        phys.view(world, C + 4)[:, C:C + 2] = void
        shard = torch.empty(C + 4)
        dist.reduce_scatter_tensor(shard, phys)
        assert float(shard[C]) == 0.0 and float(shard[C + 1]) == 0.0
        for a, b, lr in dp_pieces(n, lo, hi, lrs):
            _adam_torch(flat[a:b], shard[a - lo:b - lo], ms[a - lo:b - lo], vs[a - lo:b - lo], lr, t)
        dist.all_gather_into_tensor(flat, flat[rank * C:(rank + 1) * C].clone())
        assert torch.equal(shard[:hi - lo], ga[lo:hi]), "reduce-scattered chunk = slice of the all-reduced bucket"
    assert torch.equal(flat[:10 * n], pa), "both exchange schemes walk the same parameter trajectory"
    assert torch.equal(ms[:hi - lo], ma[lo:hi]) and torch.equal(vs[:hi - lo], va[lo:hi])
    np.save(os.path.join(out_dir, f"p{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [2, 6, 300, 1234])        # incl. chunks that end inside a group / ranks with a short chunk
def test_reduce_scatter_adam_allgather_equals_allreduce_adam(tmp_path, n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rsag_worker, args=(2, port, str(tmp_path), n), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")), "replicas identical"


def test_dp_layout_helpers():
    from splat_loam_amd.engine import dp_chunk, dp_chunked, dp_pieces
    for n, G in ((2, 8), (6, 2), (300, 2), (500000, 8), (1234, 3)):
        C = dp_chunk(n, G)
        assert C % 4 == 0 and G * C >= 10 * n and (G * (C - 4) < 10 * n or C == 4)
        flat = torch.arange(10 * n, dtype=torch.float32)
        phys = dp_chunked(flat, C, G).view(G, C + 4)
        e = torch.arange(10 * n)
        assert torch.equal(phys.reshape(-1)[e + 4 * (e // C)], flat), "element e lives at e + 4 * (e // C)"
        assert float(phys[:, C:].abs().max()) == 0.0
        cover = []
        for r in range(G):
            for a, b, lr in dp_pieces(n, min(r * C, 10 * n), min((r + 1) * C, 10 * n), (1.0, 2.0, 3.0, 4.0)):
                cover.append((a, b))
                assert lr == (1.0 if b <= 3 * n else 2.0 if b <= 4 * n else 3.0 if b <= 6 * n else 4.0)
        assert cover[0][0] == 0 and cover[-1][1] == 10 * n and all(x[1] == y[0] for x, y in zip(cover, cover[1:]))


def _worker_sparse(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    out = {}
    for sparse in (False, True):
        cams, model, R = _setup(rank)
        for _ in range(2):
            optimize_step_sharded(model, cams[rank], MappingConfig(), sparse=sparse, rasterizer_cls=R)
        tag = "s" if sparse else "d"
        for k in ("_xyz", "_scaling", "_rotation", "_opacity"):
            out[tag + k] = getattr(model, k).detach().numpy()
            out[tag + "g" + k] = getattr(model, k).grad.numpy().copy()
    # how much of the model the two keyframes reach together
    from splat_loam_amd.mapping import flat_grad_allreduce_sparse
    cams, model, R = _setup(rank)
    model.optimizer.zero_grad(set_to_none=True)
    mapping_loss(render(cams[rank], model, 0.0, rasterizer_cls=R), cams[rank], model, MappingConfig()).backward()
    out["rows_sent"] = flat_grad_allreduce_sparse(model)
    np.savez(os.path.join(out_dir, f"sparse{rank}.npz"), **out)
    dist.destroy_process_group()


def test_touched_set_exchange_equals_the_dense_all_reduce(tmp_path):
    """dp_mode "sparse" at the torch level (mapping.flat_grad_allreduce_sparse; 2 ranks, gloo, CPU): OR of the touched
    bitmaps, the union's rows packed in surfel order, SUM, scatter — gradients and the parameters after two Adam steps
    equal the dense all-reduce's bit for bit on both ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_sparse, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "sparse0.npz"), np.load(tmp_path / "sparse1.npz")
    for k in ("_xyz", "_scaling", "_rotation", "_opacity"):
        for pre in ("", "g"):
            assert np.array_equal(r0["s" + pre + k], r0["d" + pre + k]), f"sparse != dense: {pre}{k}"
            assert np.array_equal(r0["s" + pre + k], r1["s" + pre + k]), f"replicas diverged: {pre}{k}"
    assert 0 < int(r0["rows_sent"]) == int(r1["rows_sent"]) <= N      # (this small scene is reached almost entirely)
