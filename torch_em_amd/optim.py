"""FusedAdamW: torch.optim.AdamW semantics (as configured by the reference's
default_segmentation_trainer, segmentation.py:543) as ONE HIP launch over flat arenas.

state_dict() has the torch.optim.AdamW layout (per-parameter `step`, `exp_avg`, `exp_avg_sq`;
the tensors are views of the arenas), so optimizer checkpoints are interchangeable with the reference's.
"""
import torch

from . import ops
from .arena import ParamArena


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = grad_scale
        self._arena = None   # ParamArena, built lazily on the device the parameters live on
        self._m = self._v = None

    # -- arena management ---------------------------------------------------------------
    def _ensure_arena(self):
        params = [p for g in self.param_groups for p in g["params"]]
        if self._arena is not None and self._arena.is_current() and len(self._arena.params) == len(params):
            return
        old_state = {id(p): self.state.get(p) for p in params}

        class _Holder(torch.nn.Module):
            def __init__(self, ps):
                super().__init__()
                self.ps = torch.nn.ParameterList(ps)

        self._arena = ParamArena(_Holder(params))
        dev = self._arena.flat.device
        self._m = torch.zeros(self._arena.total, dtype=torch.float32, device=dev)
        self._v = torch.zeros(self._arena.total, dtype=torch.float32, device=dev)
        for p in params:
            o, n = self._arena.offsets[id(p)]
            st = old_state[id(p)]
            m, v = self._m[o:o + n].view(p.shape), self._v[o:o + n].view(p.shape)
            step = torch.tensor(0.0)
            if st:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                step = st["step"] if torch.is_tensor(st["step"]) else torch.tensor(float(st["step"]))
            self.state[p] = {"step": step.clone().float().cpu(), "exp_avg": m, "exp_avg_sq": v}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._arena = None  # re-home the loaded moments into arenas on the next step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure_arena()
        ar = self._arena
        group = self.param_groups[0]
        uniform = all(
            (g["lr"], g["betas"], g["eps"], g["weight_decay"]) ==
            (group["lr"], group["betas"], group["eps"], group["weight_decay"]) for g in self.param_groups)
        gflat = ar.grads_flat() if uniform else None
        steps = {int(self.state[p]["step"].item()) for p in ar.params}
        if gflat is not None and len(steps) == 1:
            step = steps.pop() + 1
            ops.adamw_step(ar.flat, gflat, self._m, self._v, group["lr"], group["betas"][0], group["betas"][1],
                           group["eps"], group["weight_decay"], step, self.grad_scale)
            for p in ar.params:
                self.state[p]["step"] += 1
            ops.bump_versions(ar.params)
            return loss
        # general case: one launch per parameter (still the HIP kernel)
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                st["step"] += 1
                grad = p.grad.contiguous()
                if p.data_ptr() % 16 or grad.data_ptr() % 16:
                    raise RuntimeError("FusedAdamW: parameter/gradient storage must be 16-byte aligned")
                ops.adamw_step(p.data.view(-1), grad.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1),
                               g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                               int(st["step"].item()), self.grad_scale)
                ops.bump_versions([p])
        return loss
