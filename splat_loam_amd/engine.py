"""MappingEngine: one native call per mapping iteration.

`Mapper.optimize` (slam/mapper.py:150-204) spends its time in ~120 small torch
kernels, a dozen allocations and several device->host syncs per iteration around
the rasterizer.  The engine keeps the model's four raw parameter tensors where
they are (torch Parameters), owns the flat gradient / Adam-state buckets and one
workspace, and issues `sls_mapping_step` — activations, rasterizer forward,
loss, backward, regulariser and Adam enqueued back to back on the stream with no
host sync; the iteration's status (instance count, overflow flag, loss terms) is
read once at the end, which is the same single sync per iteration the reference
has (`loss_total.item()`, slam/mapper.py:206-209).

Keyframe-parallel mode: every rank calls step() on ITS keyframe
with the same replicated model.  Two exchange schemes (`dp_mode`):
  "rs_ag" (default, even N): the gradients leave the native step in G chunks, a
      reduce-scatter gives every rank the summed chunk it owns, the rank applies
      Adam to THAT 1/G of the parameters (its Adam moments are the only ones it
      keeps: optimiser state sharded G ways), an all-gather republishes the
      parameters — the bytes of one all-reduce, the optimiser pass divided by G,
      replicas identical by construction;
  "allreduce": the flat 10*N bucket (+2 void words) is all-reduced in one RCCL
      collective and every rank applies the same guarded Adam update;
  "sparse" (even N): only the TOUCHED set travels (a keyframe reaches ~10 % of the
      surfels).  Behind the tile backward every rank knows which surfels CAN have a
      gradient (reached by its keyframe, or pushed by the scale regulariser): these
      bitmaps are all-gathered (N/8 bytes each) and OR-ed (sls_grad_union); the
      projection's backward then writes the union's 40-byte rows straight into their
      slots of the collective's buffer and applies Adam to every surfel OUTSIDE the
      union right there (zero gradient on every rank: the one-GPU update); the rows
      are SUM-reduced and sls_adam_step_sparse(part 2) updates the union.  No flat
      bucket, no packing launch, no second pass over the whole model.  The
      collective's size is a host-side capacity that follows the union's measured
      size; a union larger than it voids the iteration on every rank (bit 2 of the
      status word) and it is repeated with more room.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
import torch.distributed as dist

from . import _abi
from .fused import camera_aux
from .rasterizer import GaussianRasterizationSettings, get_camera


def carry_bucket(buf: torch.Tensor, keep: torch.Tensor, n_new: int) -> torch.Tensor:
    """A flat 10*N_old bucket [xyz 3N | opacity N | scaling 2N | rotation 4N] -> the 10*n_new bucket of the surfel
    set that keeps the rows marked in `keep` (in order) and appends n_new - keep.sum() zero rows per group."""
    n_old = int(keep.numel())
    n_keep = int(keep.sum().item())
    if buf.numel() != 10 * n_old or n_new < n_keep:
        raise ValueError("bucket / mask / new size do not fit together")
    out = torch.zeros((10 * n_new,), dtype=buf.dtype, device=buf.device)
    src_off = dst_off = 0
    for width in (3, 1, 2, 4):
        src = buf[src_off:src_off + width * n_old].view(n_old, width)
        out[dst_off:dst_off + width * n_keep].view(n_keep, width).copy_(src[keep])
        src_off += width * n_old
        dst_off += width * n_new
    return out


def dp_chunk(N: int, G: int) -> int:
    """Elements per rank of the reduce-scatter layout: a multiple of 4 floats with G * C >= 10 N."""
    return -(-(10 * N) // (4 * G)) * 4


def dp_pieces(N: int, lo: int, hi: int, lrs):
    """The flat range [lo, hi) of [xyz 3N | opacity N | scaling 2N | rotation 4N] cut at the parameter-group
    boundaries: [(start, stop, learning rate)], at most four pieces."""
    out = []
    for g0, g1, lr in ((0, 3 * N, lrs[0]), (3 * N, 4 * N, lrs[1]), (4 * N, 6 * N, lrs[2]), (6 * N, 10 * N, lrs[3])):
        a, b = max(lo, g0), min(hi, g1)
        if b > a:
            out.append((a, b, lr))
    return out


def dp_chunked(flat: torch.Tensor, C: int, G: int) -> torch.Tensor:
    """Flat bucket -> the physical reduce-scatter layout (element e at e + 4 * (e // C); 4 spare words per chunk,
    the first two of which carry the void flags).  The native step writes this layout directly; this helper is
    the layout's definition in torch (tests, CPU emulation)."""
    out = torch.zeros((G, C + 4), dtype=flat.dtype, device=flat.device)
    pad = torch.zeros((G * C,), dtype=flat.dtype, device=flat.device)
    pad[:flat.numel()] = flat
    out[:, :C] = pad.view(G, C)
    return out.reshape(-1)


class MappingEngine:
    def __init__(self, model, cfg, lrs=(5e-4, 5e-2, 5e-3, 1e-3), betas=(0.9, 0.999), eps=1e-15,
                 capacity_factor: float = 1.3):
        self.model, self.cfg = model, cfg
        self.lrs, self.betas, self.eps = tuple(lrs), tuple(betas), float(eps)
        self.capacity_factor = float(capacity_factor)
        p = model._xyz
        if not p.is_cuda:
            raise RuntimeError("MappingEngine needs ROCm device parameters; there is no CPU fallback")
        self.dev = p.device
        self.N = int(p.shape[0])
        n10 = 10 * self.N
        # flat buckets; two extra words at the end of grads carry the two "iteration void" flags
        # (instance buffers too small, depth-order repair failed) through the all-reduce
        self.grads = torch.zeros((n10 + 2,), dtype=torch.float32, device=self.dev)
        self.exp_avg = torch.zeros((n10,), dtype=torch.float32, device=self.dev)
        self.exp_avg_sq = torch.zeros((n10,), dtype=torch.float32, device=self.dev)
        self.status = torch.zeros((8,), dtype=torch.int32, device=self.dev)
        self.t = 0
        self.capacity = 0
        self.workspace = None
        self.allmap_ptr = C.c_void_p(0)
        self.last = None
        # lagged status read (sync="lagged"): two status slots on the device, pinned mirrors, events
        self._lag_dev = torch.zeros((2, 8), dtype=torch.int32, device=self.dev)
        self._lag_host = torch.zeros((2, 8), dtype=torch.int32).pin_memory()
        self._lag_ev = [torch.cuda.Event(), torch.cuda.Event()]
        # With the status mirror (the iteration's last kernel stores the status block into pinned host memory, word 7
        # last) the host needs no event to know that an iteration has finished: it arms words 0 and 7 of the slot with a
        # value the device never writes and polls them — no event record on the stream, no event wait
        # (status_poll = False: events; kept as an attribute for the equality tests, its verdict is in HISTORY.md)
        self._lag_np = self._lag_host.numpy()
        self.status_poll = True
        self._lag_polled = [False, False]
        self._group = None
        self._lag_ready = []              # statuses of finished iterations not handed out yet (lagged mode)
        self.flushed = []
        self._lag_pending = None          # (slot, camera) of the iteration whose status was not read yet
        # temporal re-sort: the workspace keeps the depth order of the last iteration; it is repaired
        # instead of recomputed when the next iteration renders the same keyframe (reuse_depth_order)
        self.reuse_depth_order = True
        # the loss stage inside the tile backward (no launch of its own; the backward's launch order then comes from the
        # keyframe's previous iteration): same gradients bit for bit
        self.inline_loss_stage = True
        self.repair_span = 256            # iterations an extra repair round stays on after a repair that did not reach
        self._repair_rounds, self._repair_until = 1, 0
        # single GPU: Adam is applied inside the backward; the flat gradient bucket is only filled when asked
        # for (the reference drops its gradients right after optimizer.step() as well)
        self.keep_grads = False
        self._ws_ready = False
        self._ws_hw = None
        self._ws_det = None
        self.status_mirror = True          # (False: the status through a device -> host copy; lagged mode)
        # one depth-order buffer per keyframe (the mapper samples keyframes at random, slam/mapper.py:152-156):
        # id(camera) -> [order tensor, iteration it was last written, weak reference to the camera (an id can be
        # reused by a new object once the old keyframe is gone)]; an order older than max_order_age
        # iterations gets an extra repair round, one older than max_order_age_extra is rebuilt from scratch
        self._orders = {}
        self._set_order_ages()
        self.max_cached_orders = 64
        self.stats = {"repeated_too_small": 0, "repeated_resort": 0, "repeated_exchange": 0, "repeated_det": 0}
        self._enq = 0                     # iterations enqueued so far (age of the cached depth orders)
        # keyframe-parallel exchange (set up at the first sharded step)
        self.dp_mode = os.environ.get("SLS_DP_MODE", "rs_ag")
        self._dp = None                   # dict(G, rank, C, flat, gshard) once the reduce-scatter layout is in place
        self._dp_agreed, self._dp_use_rs, self._dp_scheme = None, False, 0    # (G, rank) the scheme was agreed for; the agreed verdict
        from .rasterizer import deterministic_mode
        # integer-atomic gradient accumulation: False / True (SLS_DETERMINISTIC=1: two tile-backward launches) / 2
        # (SLS_DETERMINISTIC=2: one launch with scales predicted from the keyframe's previous iteration; the first
        # iteration on a workspace, and any iteration after a misprediction, run the two-launch scheme)
        self.deterministic = 2 if os.environ.get("SLS_DETERMINISTIC", "0") == "2" else deterministic_mode()
        self._det_prev = {}               # id(camera) -> [uint8 (N, 16) predicted scales, weak reference to the camera]
        self._det_two_pass_next = True    # the next deterministic iteration runs the two-launch scheme
        self.block_masks = int(os.environ.get("SLS_BLOCK_MASKS", "0"))   # 0: auto (long lists), 1: always, 2: never (SlsMappingConfig.block_masks)
        self._sx = None                   # sparse exchange: dict(bitmap, prefix, compact, cap, send) once set up
        self.exchanged_bytes = 0          # bytes this rank handed to collectives in the last keyframe-parallel step
        self.exchange_at_world_1 = False  # take the keyframe-parallel path (collectives + separate Adam) in a 1-rank group too
        self.comm_events = None           # list -> (start, after exchange, after Adam[, after all-gather]) events per step
        self._last_call = None            # (arguments, config) of the last sls_mapping_step: phase 2 repeats them

    # views of the flat gradient bucket in the optimiser's group order (single GPU: only filled
    # when keep_grads is set; keyframe-parallel mode always fills and all-reduces it)
    def grad_views(self):
        N = self.N
        g = self.grads
        if self._dp is not None:          # reduce-scatter layout -> flat copy (diagnostics / tests only)
            C, G = self._dp["C"], self._dp["G"]
            g = g.view(G, C + 4)[:, :C].reshape(-1)[:10 * N].clone()
        return {"xyz": g[0:3 * N].view(N, 3), "opacity": g[3 * N:4 * N].view(N, 1),
                "scaling": g[4 * N:6 * N].view(N, 2), "rotation": g[6 * N:10 * N].view(N, 4)}

    def _ensure_workspace(self, H, W, capacity):
        lib = _abi.lib()
        if (self.workspace is None or capacity > self.capacity or (H, W) != self._ws_hw
                or self._ws_det != bool(self.deterministic) or self.N > getattr(self, "_ws_n", 0)):
            self._ws_det = bool(self.deterministic)
            self.capacity, self._ws_hw = int(max(capacity, self.capacity)), (H, W)   # (keyframes of another size: re-carve)
            wcfg = _abi.SlsMappingConfig()
            wcfg.deterministic = 1 if self.deterministic else 0     # (the fixed-point accumulators only when asked for)
            # (sized for a quarter more surfels than there are: resize() keeps it while the model grows into that room)
            self._ws_n = self.N + (self.N // 4 if getattr(self, "_bucket_store", None) is not None else 0)
            nbytes = int(lib.sls_mapping_workspace_bytes_cfg(self._ws_n, H, W, self.capacity, C.byref(wcfg)))
            self.workspace = None     # release before re-allocating
            self.workspace = torch.empty((nbytes + 256,), dtype=torch.uint8, device=self.dev)
            self._ws_ready = False
        base = self.workspace.data_ptr()
        return (base + 255) & ~255, self.workspace.numel() - 256

    def _config(self, apply_adam, with_regulariser, reuse_order=False):
        c, cfg = _abi.SlsMappingConfig(), self.cfg
        c.reuse_depth_order = int(reuse_order)      # 0: from scratch, 1 / 2: repair with that many rounds
        c.keep_grads = 1 if self.keep_grads else 0
        c.workspace_ready = 1 if self._ws_ready else 0
        self._ws_ready = True
        c.lambda_alpha, c.lambda_normal = cfg.opt_lambda_alpha, cfg.opt_lambda_normal
        c.scaling_max = cfg.opt_scaling_max
        c.scaling_max_penalty = cfg.opt_scaling_max_penalty if with_regulariser else 0.0
        c.depth_ratio = cfg.depth_ratio
        c.lr_xyz, c.lr_opacity, c.lr_scaling, c.lr_rotation = self.lrs
        c.apply_adam = 1 if apply_adam else 0
        c.beta1, c.beta2, c.eps = self.betas[0], self.betas[1], self.eps
        if self._dp is not None and not apply_adam:
            c.grad_chunk, c.grad_ranks = self._dp["C"], self._dp["G"]
        c.deterministic = 1 if self.deterministic else 0      # (_enqueue raises it to 2 where the one-launch scheme applies)
        c.block_masks = int(self.block_masks)
        if self._sx is not None and not apply_adam:
            c.grad_bitmap = self._sx["mine"].data_ptr()
        return c

    def _order_entry(self, camera):
        ent = self._orders.get(id(camera))
        return ent if ent is not None and ent[2]() is camera else None

    def _forget_order(self, camera, failed_repair=False):
        ent = self._order_entry(camera)
        if ent is not None:
            ent[1] = None
        if failed_repair:
            self._repair_rounds = min(self._repair_rounds + 1, 3)
            self._repair_until = self._enq + self.repair_span

    def _set_order_ages(self):
        """The ages (iterations since the keyframe was last rendered) that decide how a keyframe's depth order is
        brought up to date: one repair round up to `max_order_age`, two up to `order_age_round3`, three up to
        `max_order_age_extra` (four beyond `order_age_round4`), a rebuild beyond.  A round reaches 512 POSITIONS, and
        how many positions a surfel travels for the same change of depth grows with the number of surfels: measured
        at 500 k surfels (4 / 12 / 48; looser values fail repairs, each costs a void iteration + a rebuild) and at
        170 k / 50 k (12 / 32 / 200 without a failure, -1.5 % / -2.5 % per iteration with the mapper's keyframe
        sampling), hence the scale with 500 k / N, capped at 3.  A further repair round costs 9 us, the radix sort 90.
        (The four ages are plain attributes; what other values cost is in HISTORY.md #44, #50, #51.)"""
        scale = min(max(500000.0 / max(self.N, 1), 1.0), 3.0)
        self.max_order_age = int(4 * scale)
        self.order_age_round3 = int(12 * scale)
        self.max_order_age_extra = int(48 * scale)
        self.order_age_round4 = 1000000

    def _params(self):
        m = self.model
        ps = (m._xyz, m._scaling, m._rotation, m._opacity)
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.shape[0] != self.N:
                raise RuntimeError("model parameters must stay contiguous float32 of the engine's size")
        return ps

    def _enqueue(self, camera, apply_adam, with_regulariser, status=None, mirror=None, allow_reuse=True, phase=0):
        lib = _abi.lib()
        H, W = int(camera.image_height), int(camera.image_width)
        settings = GaussianRasterizationSettings(H, W, 1.0, camera.world_view_transform, camera.projection_matrix)
        ce = get_camera(settings, self.dev)
        aux = camera_aux(camera)
        if self.capacity == 0:
            self.capacity = max(4 * self.N, 1 << 16)
        ws_ptr, ws_bytes = self._ensure_workspace(H, W, self.capacity)
        xyz, scaling, rotation, opacity = self._params()
        ent = self._order_entry(camera)
        if ent is None:
            stale = [k for k, e in self._orders.items() if e[2]() is None]
            for k in stale:
                del self._orders[k]
            if len(self._orders) >= self.max_cached_orders:
                self._orders.pop(next(iter(self._orders)))
            # + the keyframe's launch order of the tile backward (SlsMappingConfig.block_order; zeros: none yet)
            ent = [torch.empty((self.N,), dtype=torch.int32, device=self.dev), None, weakref.ref(camera),
                   torch.zeros((int(lib.sls_block_order_bytes(H, W)) // 4,), dtype=torch.int32, device=self.dev)]
            self._orders[id(camera)] = ent
        age = self._enq - ent[1] if ent[1] is not None else None
        reuse = allow_reuse and self.reuse_depth_order and age is not None and age <= self.max_order_age_extra
        # after a failed repair the following iterations repair with one more round (one more window of reach,
        # +16 us); every `repair_span` iterations without a failure the number of rounds steps down again
        if self._repair_rounds > 1 and self._enq >= self._repair_until:
            self._repair_rounds -= 1
            self._repair_until = self._enq + self.repair_span
        # an order older than max_order_age iterations gets one more round (the surfels have drifted further)
        reuse = min(self._repair_rounds + (1 if age is not None and age > self.max_order_age else 0)
                    + (1 if age is not None and age > self.order_age_round3 else 0)
                    + (1 if age is not None and age > self.order_age_round4 else 0), 4) if reuse else 0
        self._enq += 1
        ent[1] = self._enq
        cfg = self._config(apply_adam, with_regulariser, reuse)
        if self.deterministic == 2:
            dent = self._det_prev.get(id(camera))
            first_visit = dent is None or dent[1]() is not camera
            if first_visit:
                for k in [k for k, e in self._det_prev.items() if e[1]() is None]:
                    del self._det_prev[k]
                # (bounded like the depth orders: 16 N bytes per keyframe; an evicted keyframe's next visit is a
                #  two-launch iteration, as a first visit is)
                while len(self._det_prev) >= self.max_cached_orders:
                    self._det_prev.pop(next(iter(self._det_prev)))
                dent = [torch.zeros((self.N, 16), dtype=torch.uint8, device=self.dev), weakref.ref(camera)]
                self._det_prev[id(camera)] = dent
            cfg.det_prev = dent[0].data_ptr()
            # two launches where there is nothing to predict from: a workspace's first deterministic iteration (it sets
            # the fields' default scales) and a keyframe's first visit (measured: predicting a new view from the
            # defaults alone voids the iteration almost every time)
            cfg.deterministic = 1 if (self._det_two_pass_next or first_visit or not cfg.workspace_ready) else 2
            self._det_two_pass_next = False
        cfg.depth_order = ent[0].data_ptr()
        cfg.block_order = ent[3].data_ptr() if self.inline_loss_stage else None
        cfg.status_mirror = mirror
        # keyframe-parallel mode: the void bits leave the step as two floats behind the gradient bucket
        cfg.void_flags_out = None if (apply_adam or self._dp is not None) else self.grads.data_ptr() + 4 * 10 * self.N
        cfg.phase = int(phase)
        args = (C.byref(ce.cam), self.N, xyz.data_ptr(), scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(),
                self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.t + 1,
                aux.gt.data_ptr(), aux.valid.data_ptr(), aux.n_valid, ce.col_cs.data_ptr(), ce.row_cs.data_ptr(),
                aux.col_h.data_ptr(), aux.row_h.data_ptr(), C.byref(cfg), self.capacity, ws_ptr, ws_bytes,
                (self.status if status is None else status).data_ptr(), C.byref(self.allmap_ptr),
                torch.cuda.current_stream(self.dev).cuda_stream)
        self._last_call = (args, cfg) if phase == 1 else None      # (cfg and the camera entry stay alive with it)
        _abi.check(lib.sls_mapping_step(*args), "sls_mapping_step")

    @staticmethod
    def _void_reason(st):
        if st["too_small"]:
            return "repeated_too_small"
        if st["resort_failed"]:
            return "repeated_resort"
        return "repeated_det" if st.get("det_mispredicted") else "repeated_exchange"

    def _sharded(self, group):
        """Keyframe-parallel path?  World size > 1 — or 1 with `exchange_at_world_1` (the collectives then move
        a rank's data onto itself: how the RCCL path is exercised on a one-GPU box)."""
        if not dist.is_initialized():
            return False
        return dist.get_world_size(group) > 1 or self.exchange_at_world_1

    def _read_status(self):
        return self._note(self._parse_status(self.status.cpu()))   # the one sync of the iteration

    def _note(self, st):
        """Host-side bookkeeping driven by a status every rank sees identically: the sparse exchange's collective
        size follows the measured size of the union of the touched sets: up at once (50 % + 4096 slots of head room
        — keyframes of a window reach sets of different size, and an iteration voided by a union that outgrew the
        collective costs more than a few hundred KB on the wire), down by 3 % per iteration."""
        # One-launch deterministic mode: the repeat of ANY voided iteration runs the two-launch scheme.  A misprediction
        # (bit 3) is what calls for it, but the keyframe-parallel verdicts fold every bit above bit 0 into bit 1 on their
        # way through the collectives, so the host cannot tell it from a failed repair there — and a one-launch repeat
        # with the same parameters and the same predictions would mispredict again, for ever.
        if st["overflow"] and self.deterministic == 2:
            self._det_two_pass_next = True            # (and with it the defaults' refresh)
        if st.get("outside_union"):
            raise RuntimeError("a surfel outside the agreed union of the touched sets had a non-zero gradient: the early "
                               "gradient bitmap was no superset (sparse exchange) — the iteration's update is void")
        if st.get("handover_mismatch"):
            raise RuntimeError("the tile backward found a forward -> backward hand-over written by another tile-kernel variant "
                               "(sls_debug_variant switched between the two launches): the iteration's gradients are void")
        if self._sx is not None and (not st["overflow"] or st["exchange_too_small"]):
            want = int(st["exchange_count"] * 1.5) + 4096
            self._sx["send"] = int(min(self.N, max(want, int(self._sx["send"] * 0.97))))
        return st

    @staticmethod
    def _parse_status(h):
        R, flags = int(h[0].item()) & 0xFFFFFFFF, int(h[1].item())
        f = h.view(torch.float32)
        # "overflow": the iteration is void (Adam was skipped) and must be repeated; bit 0 = the
        # instance buffers were too small, bit 1 = the repaired depth order was not exact
        return {"R": R, "overflow": bool(flags), "too_small": bool(flags & 1), "resort_failed": bool(flags & 2),
                "exchange_too_small": bool(flags & 4), "det_mispredicted": bool(flags & 8),
                "handover_mismatch": bool(flags & 16), "outside_union": bool(flags & 32),
                "exchange_count": int(h[7].item()) & 0xFFFFFFFF,
                "loss_pixel": float(f[5]), "loss_reg": float(f[6]),
                "loss": float(f[5]) + float(f[6]), "sums": [float(f[2]), float(f[3]), float(f[4])]}

    @staticmethod
    def _parse_status_np(h):
        """_parse_status on a NumPy row (int32 x 8) of the pinned mirror: no tensor ops on the host's critical path"""
        import numpy as np
        f = h.view(np.float32)
        R, flags = int(h[0]) & 0xFFFFFFFF, int(h[1])
        return {"R": R, "overflow": bool(flags), "too_small": bool(flags & 1), "resort_failed": bool(flags & 2),
                "exchange_too_small": bool(flags & 4), "det_mispredicted": bool(flags & 8),
                "handover_mismatch": bool(flags & 16), "outside_union": bool(flags & 32),
                "exchange_count": int(h[7]) & 0xFFFFFFFF,
                "loss_pixel": float(f[5]), "loss_reg": float(f[6]),
                "loss": float(f[5]) + float(f[6]), "sums": [float(f[2]), float(f[3]), float(f[4])]}

    @torch.no_grad()
    def step(self, camera, group=None, sync: bool = True):
        """One mapping iteration on `camera`.  Returns the status dict (sync=True)
        or None (sync=False: fire-and-forget — nobody reads the status, so an iteration
        that overflowed its instance buffers is DROPPED, not repeated (its Adam update
        was skipped on the device, the model stays intact), and `self.t` counts enqueued
        iterations, which can then run ahead of the updates applied; use "lagged" when
        every iteration has to count).
        sync="lagged": the iteration is enqueued BEFORE the status of
        the previous one is read, so the GPU queue never runs dry while the host
        waits for a loss value; returns the PREVIOUS iteration's status (None on
        the first call) — finish with flush()."""
        sharded = self._sharded(group)
        if self._dp is not None and not sharded:
            raise RuntimeError("this engine's optimiser state is sharded over a process group (dp_mode rs_ag): "
                               "it cannot take a single-process step")
        self._group = group
        if sync == "lagged":
            return self._step_lagged(camera)
        if self._lag_pending is not None:
            self.flush()
        # Fire-and-forget iterations sort from scratch: nobody would notice a repair that did not reach, and every
        # later repair of that keyframe would start from the broken order and void its iteration as well.
        reuse_ok = sync is not False
        while True:
            if not sharded:
                self._enqueue(camera, apply_adam=True, with_regulariser=True, allow_reuse=reuse_ok)
            else:
                rank = dist.get_rank(group)
                self._ensure_dp(group)
                self._enqueue(camera, apply_adam=False, with_regulariser=(rank == 0), allow_reuse=reuse_ok,
                              phase=self._phase_for_exchange())
                # the step left its void flags behind the gradients: no torch glue kernels; any rank's flag voids
                # the iteration everywhere: Adam reads the reduced flags and stores the group's verdict into the
                # local status word, so the one status read below is the only sync
                self._exchange_and_adam(group, self.status, None)
            if not sync:
                self.t += 1
                return None
            st = self._read_status()
            if not st["overflow"]:
                self.t += 1
                self.last = st
                return st
            # repeat the iteration (parameters were not touched): with the full sort, and with more
            # room if the instance buffers were too small
            if st["too_small"] or st["resort_failed"]:
                self._forget_order(camera, failed_repair=st["resort_failed"])
            self.stats[self._void_reason(st)] += 1
            if st["too_small"]:
                need = st["R"]
                if sharded:
                    t = torch.tensor([need], dtype=torch.int64, device=self.dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                    need = int(t.item())
                self.capacity = int(max(need, self.capacity) * self.capacity_factor) + 1024
                self.workspace = None

    def _step_lagged(self, camera):
        slot = 0 if self._lag_pending is None else self._lag_pending[0] ^ 1
        group = self._group
        polled = False
        if self._sharded(group):
            # keyframe-parallel: the group's verdict is known on the device only (the void flags ride the
            # gradient all-reduce and guard Adam), so the host can lag here exactly as on one GPU
            self._ensure_dp(group)
            self._enqueue(camera, apply_adam=False, with_regulariser=(dist.get_rank(group) == 0),
                          status=self._lag_dev[slot], phase=self._phase_for_exchange())
            self._exchange_and_adam(group, self._lag_dev[slot],
                                    self._lag_host[slot].data_ptr() if self.status_mirror else None)
            if not self.status_mirror:
                self._lag_host[slot].copy_(self._lag_dev[slot], non_blocking=True)
        elif self.status_mirror:
            # the iteration's last kernel mirrors the status block into pinned host memory: no copy kernel
            polled = self.status_poll
            if polled:
                self._lag_np[slot, 0] = -1
                self._lag_np[slot, 7] = -1
            self._enqueue(camera, apply_adam=True, with_regulariser=True, status=self._lag_dev[slot],
                          mirror=self._lag_host[slot].data_ptr())
        else:
            self._enqueue(camera, apply_adam=True, with_regulariser=True, status=self._lag_dev[slot])
            self._lag_host[slot].copy_(self._lag_dev[slot], non_blocking=True)
        self.t += 1
        self._lag_polled[slot] = polled
        if not polled:
            self._lag_ev[slot].record(torch.cuda.current_stream(self.dev))
        prev, self._lag_pending = self._lag_pending, (slot, camera)
        if prev is not None:
            self._lag_collect(prev, redo_current=True)
        # every iteration's status is handed out exactly once, oldest first (a repeated iteration makes two
        # of them available at once: the next call returns the second)
        return self._lag_ready.pop(0) if self._lag_ready else None

    def _lag_collect(self, prev, redo_current):
        pslot, pcam = prev
        if self._lag_polled[pslot]:
            row, spins = self._lag_np[pslot], 0
            while row[7] == -1 or row[0] == -1:          # (the device writes word 7 last, and never this value)
                spins += 1
                if spins > 2_000_000:                    # ~1 s: something else is wrong — let the stream say what
                    torch.cuda.current_stream(self.dev).synchronize()
                    if row[7] == -1 or row[0] == -1:
                        raise RuntimeError("the mapping iteration's status never reached its host mirror")
            st = self._note(self._parse_status_np(row.copy()))
        else:
            self._lag_ev[pslot].synchronize()
            st = self._note(self._parse_status(self._lag_host[pslot].clone()))
        if not st["overflow"]:
            self.last = st
            self._lag_ready.append(st)
            return st
        # The instance buffers were too small: that iteration skipped its Adam update on the
        # device, and so did the one enqueued after it (same capacity).  Drain, grow, redo.
        cur, self._lag_pending = self._lag_pending, None
        torch.cuda.current_stream(self.dev).synchronize()
        cur_st = self._note(self._parse_status(self._lag_dev[cur[0]].cpu())) if cur is not None else None
        cur_void = cur_st is not None and cur_st["overflow"]
        # both in-flight iterations are taken off the step count; the one that did run is put back after the
        # redo below, so that the repeated iteration uses the Adam step number it was meant to have
        self.t -= 2 if cur is not None else 1
        if st["too_small"] or st["resort_failed"]:
            self._forget_order(pcam, failed_repair=st["resort_failed"])     # repeat with the full sort
        if cur_void and (cur_st["too_small"] or cur_st["resort_failed"]):
            self._forget_order(cur[1], failed_repair=cur_st["resort_failed"])
        self.stats[self._void_reason(st)] += 1
        if st["too_small"] or (cur_void and cur_st["too_small"]):
            need = max(st["R"], cur_st["R"] if cur_void else 0, self.capacity)
            self.capacity = int(need * self.capacity_factor) + 1024
            self.workspace = None
        if cur_st is not None and not cur_void:
            self._lag_ready.append(cur_st)   # (another keyframe that fitted: it did run, nothing to repeat)
        st = self.step(pcam, group=self._group, sync=True)
        self._lag_ready.append(st)
        if cur is not None and not cur_void:
            self.t += 1
        if cur_void:
            if redo_current:
                # (not through _step_lagged: its return value would take a status off the queue)
                self._redo_pending(cur[1])
            else:
                self._lag_ready.append(self.step(cur[1], group=self._group, sync=True))
        return st

    def _redo_pending(self, camera):
        ready, self._lag_ready = self._lag_ready, []
        self._step_lagged(camera)                    # enqueues it again; nothing older is pending
        self._lag_ready = ready + self._lag_ready

    def flush(self):
        """Drains the lagged pipeline: returns the status of the LAST iteration (None if there was none);
        `self.flushed` lists every status that had not been handed out yet, oldest first.
        NOT a stream synchronisation: with the polled status mirror the host learns an iteration's status from the
        FIRST thread of its last kernel, so that kernel (the Adam update) may still be running when this returns.
        Whatever reads the parameters next through torch is ordered behind it on the stream; a caller that hands
        the pointers to something else synchronises the stream itself."""
        if self._lag_pending is not None:
            prev, self._lag_pending = self._lag_pending, None
            self._lag_collect(prev, redo_current=False)
        self.flushed, self._lag_ready = self._lag_ready, []
        return self.flushed[-1] if self.flushed else None

    def _ensure_dp(self, group):
        """First sharded step: lay the gradient bucket out for the reduce-scatter, move the four parameter tensors
        into ONE flat buffer (they stay torch Parameters: their storage becomes a view of it) and keep only this
        rank's 1/G of the Adam moments."""
        if self.dp_mode not in ("rs_ag", "allreduce", "sparse"):
            raise ValueError(f"dp_mode must be 'rs_ag', 'allreduce' or 'sparse', not {self.dp_mode!r}")
        G, rank, N = dist.get_world_size(group), dist.get_rank(group), self.N
        if self._dp is not None:
            if self._dp["G"] != G or self._dp["rank"] != rank:
                raise RuntimeError("the process group changed under an engine whose optimiser state is sharded "
                                   f"({self._dp['G']} ranks -> {G}): build a new engine (or remap(reset_state=True))")
            return
        if self._dp_agreed != (G, rank):
            # every rank must run the SAME collective sequence: agree on the scheme once instead of trusting that
            # SLS_DP_MODE / N agree everywhere — MIN and MAX of a mode code (sparse 2, rs_ag 1, allreduce 0; a rank
            # that cannot shard asks for allreduce): sparse only if EVERY rank asks for it, a mix of sparse and a
            # dense scheme is an error on every rank (different collectives would hang), rs_ag needs all ranks able
            # (the sparse scheme's fused Adam and the reduce-scatter layout both need an even N: an odd model falls back
            #  to the all-reduce on every rank alike — N is the same everywhere)
            want = (2 if N % 2 == 0 else 0) if self.dp_mode == "sparse" else (1 if (self.dp_mode == "rs_ag" and N % 2 == 0 and 10 * N < 2 ** 32) else 0)
            t = torch.tensor([want, -want], dtype=torch.int32, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            lo, hi = int(t[0].item()), -int(t[1].item())
            if hi == 2 and lo != 2:
                raise RuntimeError("keyframe-parallel ranks disagree on the gradient exchange: some ask for dp_mode "
                                   "'sparse', others for a dense scheme (check SLS_DP_MODE on every rank)")
            self._dp_use_rs = lo == 1
            self._dp_agreed = (G, rank)
            self._dp_scheme = lo
        if self._dp_scheme == 2:
            if self._sx is not None and (self._sx["G"], self._sx["rank"]) != (G, rank):
                raise RuntimeError("the process group changed under an engine set up for the sparse exchange "
                                   f"({self._sx['G']} ranks -> {G}): build a new engine (or remap())")
            if self._sx is None:
                lib = _abi.lib()
                nw = int(lib.sls_grad_bitmap_words(N))
                self._sx = {"bitmap": torch.zeros((nw,), dtype=torch.int64, device=self.dev),       # the union (after the OR)
                            "mine": torch.zeros((nw,), dtype=torch.int64, device=self.dev),         # this rank's bitmap
                            "all": torch.zeros((G * nw,), dtype=torch.int64, device=self.dev),      # every rank's
                            "prefix": torch.zeros((nw,), dtype=torch.int32, device=self.dev),
                            "compact": torch.zeros((10 * N,), dtype=torch.float32, device=self.dev),
                            "index": torch.zeros((N,), dtype=torch.int32, device=self.dev),     # the surfel of every slot
                            "send": N,          # slots handed to the SUM collective: all of them until the union's size is known
                            "G": G, "rank": rank}
            return
        if not self._dp_use_rs:
            return
        C = dp_chunk(N, G)
        flat = torch.zeros((G * C,), dtype=torch.float32, device=self.dev)
        m = self.model
        for p, off, w in ((m._xyz, 0, 3), (m._opacity, 3 * N, 1), (m._scaling, 4 * N, 2), (m._rotation, 6 * N, 4)):
            view = flat[off:off + w * N].view(N, w)
            view.copy_(p.data)
            p.data = view
        lo, hi = min(rank * C, 10 * N), min((rank + 1) * C, 10 * N)
        ea, es = torch.zeros((C,), dtype=torch.float32, device=self.dev), torch.zeros((C,), dtype=torch.float32, device=self.dev)
        ea[:hi - lo].copy_(self.exp_avg[lo:hi])
        es[:hi - lo].copy_(self.exp_avg_sq[lo:hi])
        self.exp_avg, self.exp_avg_sq = ea, es                # sharded optimiser state
        self.grads = torch.zeros((G * (C + 4),), dtype=torch.float32, device=self.dev)
        self._dp = {"G": G, "rank": rank, "C": C, "flat": flat, "lo": lo, "hi": hi,
                    "gshard": torch.zeros((C + 4,), dtype=torch.float32, device=self.dev)}

    def _phase_for_exchange(self):
        """1 = the native step stops behind the tile backward (sparse scheme: the union of the touched sets is agreed on
        before the projection's backward runs, which then writes the rows where the collective reads them)"""
        return 1 if self._sx is not None else 0

    def _exchange_and_adam(self, group, status, mirror):
        ev = None
        if self.comm_events is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        if self._sx is not None:
            # touched set only: OR of the early bitmaps -> the projection's backward writes the union's rows into their
            # slots and updates everything else -> SUM of the first `send` slots -> Adam on the union
            lib, sx, N = _abi.lib(), self._sx, self.N
            st = torch.cuda.current_stream(self.dev).cuda_stream
            # (RCCL offers no bitwise-OR reduction: the bitmaps are gathered, the library ORs them)
            dist.all_gather_into_tensor(sx["all"], sx["mine"], group=group)
            G = int(sx["all"].numel() // sx["mine"].numel())
            send = int(sx["send"])
            _abi.check(lib.sls_grad_union(N, sx["all"].data_ptr(), G, sx["bitmap"].data_ptr(), send, sx["prefix"].data_ptr(),
                                          status.data_ptr(), st), "sls_grad_union")
            args, cfg = self._last_call
            cfg.phase, cfg.workspace_ready, cfg.apply_adam, cfg.keep_grads = 2, 1, 1, 0
            cfg.union_bitmap, cfg.union_prefix = sx["bitmap"].data_ptr(), sx["prefix"].data_ptr()
            cfg.grad_compact, cfg.grad_compact_index, cfg.grad_compact_capacity = sx["compact"].data_ptr(), sx["index"].data_ptr(), send
            _abi.check(lib.sls_mapping_step(*args), "sls_mapping_step (phase 2)")
            dist.all_reduce(sx["compact"][:10 * send], op=dist.ReduceOp.SUM, group=group)
            if ev:
                ev[1].record()
            xyz, scaling, rotation, opacity = self._params()
            _abi.check(lib.sls_adam_step_union(N, xyz.data_ptr(), opacity.data_ptr(), scaling.data_ptr(), rotation.data_ptr(),
                                               sx["index"].data_ptr(), sx["compact"].data_ptr(), send, self.exp_avg.data_ptr(),
                                               self.exp_avg_sq.data_ptr(), self.lrs[0], self.lrs[1], self.lrs[2], self.lrs[3],
                                               self.betas[0], self.betas[1], self.eps, self.t + 1, status.data_ptr(), mirror,
                                               st), "sls_adam_step_union")
            if ev:
                ev[2].record(); ev[3].record()
            self.exchanged_bytes = 8 * int(sx["mine"].numel()) + 40 * send
        elif self._dp is None:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=group)
            if ev:
                ev[1].record()
            self._adam_reduced(status, mirror)
            if ev:
                ev[2].record(); ev[3].record()
            self.exchanged_bytes = 4 * int(self.grads.numel())
        else:
            d = self._dp
            dist.reduce_scatter_tensor(d["gshard"], self.grads, op=dist.ReduceOp.SUM, group=group)
            if ev:
                ev[1].record()
            self._adam_shard(status, mirror)
            if ev:
                ev[2].record()
            C, r = d["C"], d["rank"]
            mine = d["flat"][r * C:(r + 1) * C]
            if d.get("ag_in_place", True):
                try:            # in place: the rank's shard already sits where the collective puts it
                    dist.all_gather_into_tensor(d["flat"], mine, group=group)
                except (RuntimeError, ValueError):     # a backend that refuses aliasing input / output
                    d["ag_in_place"] = False
            if not d.get("ag_in_place", True):
                dist.all_gather_into_tensor(d["flat"], mine.clone(), group=group)
            if ev:
                ev[3].record()
            self.exchanged_bytes = 4 * int(self.grads.numel()) + 4 * C        # reduce-scatter input + the gathered shard
        if ev:
            self.comm_events.append(ev)

    def _adam_shard(self, status, mirror):
        """Adam on the flat elements [lo, hi) this rank owns: up to four pieces, one per parameter group the range
        crosses (each has its own learning rate); gradients = the reduce-scattered chunk, moments = the local shard."""
        lib, d, N = _abi.lib(), self._dp, self.N
        lo, hi, C = d["lo"], d["hi"], d["C"]
        arr = (_abi.SlsAdamGroup * 4)()
        k = 0
        for a, b, lr in dp_pieces(N, lo, hi, self.lrs):
            arr[k].param = d["flat"].data_ptr() + 4 * a
            arr[k].grad = d["gshard"].data_ptr() + 4 * (a - lo)
            arr[k].exp_avg = self.exp_avg.data_ptr() + 4 * (a - lo)
            arr[k].exp_avg_sq = self.exp_avg_sq.data_ptr() + 4 * (a - lo)
            arr[k].numel = b - a
            arr[k].lr = lr
            k += 1
        if k == 0:       # (a rank whose chunk lies beyond 10 N: nothing to update, but it publishes the verdict)
            arr[0].param = arr[0].grad = arr[0].exp_avg = arr[0].exp_avg_sq = d["gshard"].data_ptr()
            arr[0].numel, arr[0].lr, k = 0, 0.0, 1
        _abi.check(lib.sls_adam_step_reduced(arr, k, self.betas[0], self.betas[1], self.eps, self.t + 1,
                                             d["gshard"].data_ptr() + 4 * C, status.data_ptr(), mirror,
                                             torch.cuda.current_stream(self.dev).cuda_stream),
                   "sls_adam_step_reduced")

    def _adam_reduced(self, status, mirror):
        lib = _abi.lib()
        N = self.N
        xyz, scaling, rotation, opacity = self._params()
        arr = (_abi.SlsAdamGroup * 4)()
        spec = ((xyz, 0, 3 * N, self.lrs[0]), (opacity, 3 * N, N, self.lrs[1]),
                (scaling, 4 * N, 2 * N, self.lrs[2]), (rotation, 6 * N, 4 * N, self.lrs[3]))
        for k, (p, off, n, lr) in enumerate(spec):
            arr[k].param = p.data_ptr()
            arr[k].grad = self.grads.data_ptr() + 4 * off
            arr[k].exp_avg = self.exp_avg.data_ptr() + 4 * off
            arr[k].exp_avg_sq = self.exp_avg_sq.data_ptr() + 4 * off
            arr[k].numel = n
            arr[k].lr = lr
        _abi.check(lib.sls_adam_step_reduced(arr, 4, self.betas[0], self.betas[1], self.eps, self.t + 1,
                                             self.grads.data_ptr() + 4 * 10 * N, status.data_ptr(), mirror,
                                             torch.cuda.current_stream(self.dev).cuda_stream),
                   "sls_adam_step_reduced")

    @torch.no_grad()
    def remap(self, keep=None, appended: int = 0, reset_state: bool = True):
        """The model's surfel set changed — call this AFTER replacing the model's parameter tensors, as
        Mapper.densify / the opacity pruning do through cat_tensors_to_optimizer / _prune_optimizer
        (scene/gaussian_model.py:223-316).  `keep`: bool (N_old,), the survivors in their old order (None: all);
        `appended`: number of new surfels that follow them.

        `reset_state=True` (default) is what the reference DOES: `_prune_optimizer` re-files the pruned parameter
        under `optimizer.state[group["name"]]` (scene/gaussian_model.py:237-256), so torch.optim.Adam finds no
        state for the new tensor, and since `update_model` runs densify -> optimize -> prune on every keyframe
        (slam/mapper.py:33-42), each keyframe's `optimize()` starts Adam at step 0 with zero moments.
        `reset_state=False` is what that code evidently intends (and what `cat_tensors_to_optimizer` does when it
        finds a state): the survivors keep their moments, new surfels start from zero, the step count goes on."""
        if self._lag_pending is not None or self._lag_ready:
            self.flush()
        n_old = self.N
        if keep is None:
            keep = torch.ones((n_old,), dtype=torch.bool, device=self.dev)
        keep = keep.to(self.dev).reshape(-1)
        if keep.dtype != torch.bool or keep.numel() != n_old or appended < 0:
            raise ValueError("keep must be a bool mask over the engine's current surfels, appended >= 0")
        n_keep = int(keep.sum().item())
        n_new = n_keep + int(appended)
        if int(self.model._xyz.shape[0]) != n_new:
            raise RuntimeError(f"the model holds {int(self.model._xyz.shape[0])} surfels, keep/appended describe {n_new}")
        if self._dp is not None and not reset_state:      # (before any state is touched: the engine stays consistent)
            raise RuntimeError("remap(reset_state=False) is not available once the optimiser state is sharded "
                               "(keyframe-parallel mode rs_ag): use reset_state=True or dp_mode='allreduce'")

        if reset_state:
            self.exp_avg = torch.zeros((10 * n_new,), dtype=torch.float32, device=self.dev)
            self.exp_avg_sq = torch.zeros((10 * n_new,), dtype=torch.float32, device=self.dev)
            self.t = 0
        else:
            self.exp_avg = carry_bucket(self.exp_avg, keep, n_new)
            self.exp_avg_sq = carry_bucket(self.exp_avg_sq, keep, n_new)
        self._dp_agreed = None
        self._sx = None
        self._dp = None                               # re-sharded at the next keyframe-parallel step
        self.N = n_new
        self._set_order_ages()
        self.grads = torch.zeros((10 * n_new + 2,), dtype=torch.float32, device=self.dev)
        self.workspace = None                        # sized by N: rebuilt at the next step
        self._orders.clear()                         # surfel indices changed: every kept depth order is void
        self._det_prev.clear()                       # ... and so is every predicted scale
        self._det_two_pass_next = True
        self._params()

    @torch.no_grad()
    def resize(self, n_new: int) -> None:
        """The surfel set changed size and the caller brings the optimiser state itself (fused_mapper.fused_optimize reads
        it from optimizer.state before every run, as the reference's densify / prune helpers leave it): everything that
        does not depend on N stays — the pinned status mirrors, the events, the keyframes' launch orders of the tile
        backward — and what does is re-used while it is large enough (buckets, workspace and depth-order buffers are
        allocated with a quarter of head room: a local model grows by < 1 % per keyframe).  Depth orders are void
        (surfel indices moved); moments and step count are whatever the caller writes next.  Single-GPU path only."""
        if self._lag_pending is not None or self._lag_ready:
            self.flush()
        if self._dp is not None or self._sx is not None:
            raise RuntimeError("resize() is for the single-GPU engine; keyframe-parallel state is laid out by N: use remap()")
        n_new = int(n_new)
        if int(self.model._xyz.shape[0]) != n_new:
            raise RuntimeError(f"the model holds {int(self.model._xyz.shape[0])} surfels, not {n_new}")
        n10 = 10 * n_new
        store = getattr(self, "_bucket_store", None)
        if store is None or store[0].numel() < n10 + 2:
            room = n10 + n10 // 4 + 2
            store = [torch.zeros((room,), dtype=torch.float32, device=self.dev) for _ in range(3)]
            self._bucket_store = store
        self.grads, self.exp_avg, self.exp_avg_sq = store[0][:n10 + 2], store[1][:n10], store[2][:n10]
        self.grads.zero_()
        self.N, self.t = n_new, 0
        self._set_order_ages()
        for ent in self._orders.values():
            ent[1] = None                                   # the order itself: from scratch at the keyframe's next visit
            if ent[0].numel() < n_new:
                ent[0] = torch.empty((n_new + n_new // 4,), dtype=torch.int32, device=self.dev)
        self._det_prev.clear()
        self._det_two_pass_next = True
        self._ws_ready = False                              # (the records' region moved with N: zeroed at the next step)
        self._params()

    def allmap(self, H, W) -> torch.Tensor:
        """Copy of the last iteration's allmap (7,H,W) out of the workspace.  With depth_ratio == 0 the mapper's
        loss neither reads nor differentiates the median / distortion planes (gaussian_renderer/__init__.py:79-86),
        and the native iteration does not track them: planes 5 and 6 are zeros then (render() through
        GaussianRasterizer always fills all seven)."""
        off = int(self.allmap_ptr.value) - self.workspace.data_ptr()
        return self.workspace[off:off + 7 * H * W * 4].view(torch.float32).view(7, H, W).clone()
