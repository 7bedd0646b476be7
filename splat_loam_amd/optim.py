"""Fused Adam for the surfel model: all parameter tensors in ONE HIP launch.

Drop-in for the `torch.optim.Adam(training_vars, lr=0.0, eps=1e-15)` the
reference builds in scene/gaussian_model.py:97-121 and steps at
slam/mapper.py:204 — same constructor arguments (list of param-group dicts with
"params", "lr", "name"), `zero_grad`, `step`, `param_groups`, `state`
(`exp_avg`, `exp_avg_sq`, `step`) so the reference's cat/prune helpers
(scene/gaussian_model.py:223-316) keep working on it.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-15):
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _abi.lib()
        # one launch per distinct (betas, eps, step, device); the reference has exactly one
        buckets: dict = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam needs ROCm device parameters; there is no CPU fallback")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous float32 parameters")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                key = (group["betas"], group["eps"], st["step"], p.device)
                buckets.setdefault(key, []).append((p, g, st, float(group["lr"])))
        for (betas, eps, step, dev), items in buckets.items():
            for i in range(0, len(items), 8):
                chunk = items[i:i + 8]
                arr = (_abi.SlsAdamGroup * len(chunk))()
                for k, (p, g, st, lr) in enumerate(chunk):
                    arr[k].param = p.data_ptr()
                    arr[k].grad = g.data_ptr()
                    arr[k].exp_avg = st["exp_avg"].data_ptr()
                    arr[k].exp_avg_sq = st["exp_avg_sq"].data_ptr()
                    arr[k].numel = p.numel()
                    arr[k].lr = lr
                _abi.check(lib.sls_adam_step(arr, len(chunk), float(betas[0]), float(betas[1]), float(eps),
                                             int(step), torch.cuda.current_stream(dev).cuda_stream),
                           "sls_adam_step")
        return loss
