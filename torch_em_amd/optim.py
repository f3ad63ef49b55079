"""FusedAdamW: torch.optim.AdamW semantics (as configured by the reference's
default_segmentation_trainer, segmentation.py:543) as ONE HIP launch over flat arenas.

state_dict() has the torch.optim.AdamW layout (per-parameter `step`, `exp_avg`, `exp_avg_sq`;
the tensors are views of the arenas), so optimizer checkpoints are interchangeable with the reference's.
"""
from typing import Dict

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
        self._hyper = None   # device tensor [12] when the step is (being) captured in a HIP graph: see capturable()
        self._hyper_host = None
        self._table = self._sstate = None   # ... under dynamic loss scaling: see capturable_scaled()
        self._pre_state_dict = None         # the graph owner's hook: bring the host step counts up to date

    def state_dict(self):
        if self._pre_state_dict is not None:
            self._pre_state_dict()
        return super().state_dict()

    # -- HIP-graph capture ----------------------------------------------------------------
    def capturable(self, on: bool = True):
        """Switch to the step whose scalars (lr, bias corrections) live in a device buffer, so that a captured launch
        stays valid from step to step.  `refresh_hyper()` -- called by step() when it runs eagerly, and by
        `GraphedTrainStep` before every replay -- advances the step count and re-fills that buffer."""
        self._ensure_arena()
        if on and self._hyper is None:
            self._hyper = torch.zeros(12, dtype=torch.float32, device=self._arena.flat.device)
            # pinned staging slots, reused round-robin behind an event each: a copy from pageable memory would make the
            # host wait for everything queued before it (the previous replay), a single pinned buffer could be
            # overwritten before its copy ran
            self._hyper_host = [(torch.zeros(12, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
            self._hyper_slot = 0
        if not on:
            self._hyper = self._hyper_host = self._table = self._sstate = None
        return self

    TABLE_ROWS = 16

    def capturable_scaled(self, sstate: torch.Tensor):
        """capturable() under dynamic loss scaling: the device may skip a step (overflow) that the host has already
        counted, so the number of APPLIED steps lives in `sstate[3]` (the GradScaler's device state) and the host uploads
        the scalars of a window of step numbers (`refresh_table`); the kernel picks its row (tem_adamw_step_tab)."""
        self._ensure_arena()
        dev = self._arena.flat.device
        n = 4 + 12 * self.TABLE_ROWS
        self._sstate = sstate
        self._table = torch.zeros(n, dtype=torch.float32, device=dev)
        self._table_host = [(torch.zeros(n, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
        self._table_slot = 0
        self._hyper = None
        return self

    def refresh_table(self, applied: int):
        """Rows for steps applied + 1 .. applied + TABLE_ROWS (`applied` = a lower bound of the device's count that is at
        most TABLE_ROWS - 1 replays old)."""
        group = self.param_groups[0]
        if len(self.param_groups) != 1:
            raise RuntimeError("FusedAdamW.capturable: one parameter group is required")
        host, ev = self._table_host[self._table_slot]
        self._table_slot = (self._table_slot + 1) % len(self._table_host)
        ev.synchronize()
        host[0], host[1] = float(applied + 1), float(self.TABLE_ROWS)
        for j in range(self.TABLE_ROWS):
            ops.adamw_hyper(host[4 + 12 * j:4 + 12 * (j + 1)], group["lr"], group["betas"][0], group["betas"][1],
                            group["eps"], group["weight_decay"], applied + 1 + j, self.grad_scale)
        self._table.copy_(host, non_blocking=True)
        ev.record(torch.cuda.current_stream(self._table.device))

    def set_step_count(self, applied: int):
        for p in self._arena.params:
            self.state[p]["step"].fill_(float(applied))

    def refresh_hyper(self):
        """step += 1 for every parameter; hyper <- (lr, ..., bias corrections of the new step).  Runs OUTSIDE a graph: an
        ordinary H2D copy on the current stream, ordered before the replay that follows."""
        ar, group = self._arena, self.param_groups[0]
        steps = {int(self.state[p]["step"].item()) for p in ar.params}
        if len(steps) != 1 or len(self.param_groups) != 1:
            raise RuntimeError("FusedAdamW.capturable: one parameter group with a common step count is required")
        step = steps.pop() + 1
        host, ev = self._hyper_host[self._hyper_slot]
        self._hyper_slot = (self._hyper_slot + 1) % len(self._hyper_host)
        ev.synchronize()                      # the copy that last read this slot (4 steps ago) is long done
        ops.adamw_hyper(host, group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                        group["weight_decay"], step, self.grad_scale)
        self._hyper.copy_(host, non_blocking=True)
        ev.record(torch.cuda.current_stream(self._hyper.device))
        for p in ar.params:
            self.state[p]["step"] += 1

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
        """A captured step (torch_em_amd/graph.py) has the OLD arena / moment / scalar buffers baked in: loading a state
        re-homes them, so the capture-mode buffers are dropped here and every registered owner of a graph is told
        (GraphedTrainStep marks itself stale and the trainer captures a new one on the next step)."""
        super().load_state_dict(state_dict)
        self._arena = None  # re-home the loaded moments into arenas on the next step
        self._hyper = self._hyper_host = self._table = self._sstate = None
        self._pre_state_dict = None
        hooks, self._invalidate_hooks = getattr(self, "_invalidate_hooks", []), []
        for h in hooks:
            h()

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
        if self._table is not None:
            if gflat is None:
                raise RuntimeError("FusedAdamW.capturable: the gradients must be the engine's flat arena")
            if not torch.cuda.is_current_stream_capturing():   # eager use (warm-up): exact count, one host read
                applied = int(self._sstate[3].item())
                self.refresh_table(applied)
            ops.adamw_step_tab(ar.flat, gflat, self._m, self._v, self._table, self._sstate)
            ops.bump_versions(ar.params)
            return loss
        if self._hyper is not None:
            if gflat is None:
                raise RuntimeError("FusedAdamW.capturable: the gradients must be the engine's flat arena")
            if not torch.cuda.is_current_stream_capturing():
                self.refresh_hyper()     # under capture the owner of the graph refreshes before each replay
            ops.adamw_step_dev(ar.flat, gflat, self._m, self._v, self._hyper)
            ops.bump_versions(ar.params)
            return loss
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


class GradScaler:
    """Dynamic loss scaling for the mixed-precision path -- the interface and policy of `torch.amp.GradScaler` as the
    reference trainer drives it (trainer/default_trainer.py:134-142: created when `mixed_precision` and dtype float16;
    `_backprop_mixed` :789-794: `scale(loss).backward(); step(optimizer); update()`).

    Why it is needed here although activations and gradients are stored in fp32: in mixed-precision mode the MFMA
    convolutions round their OPERANDS to fp16 (tem_conv3d_fwd / _wgrad use_mfma = 5), so an incoming gradient below
    6e-8 would vanish and one above 65504 becomes inf.  The loss is multiplied by `scale` before backward; `step`
    divides the gradients by it again with one HIP launch over the flat gradient arena (`tem_amp_unscale`, which also
    raises the found-inf flag), reads that flag (the one host sync per step, as in torch) and skips the optimizer
    step on overflow; `update` halves the scale after an overflow and doubles it after `growth_interval` clean steps.
    """

    def __init__(self, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        if growth_factor <= 1.0:
            raise ValueError("The growth factor must be > 1.0.")
        if backoff_factor >= 1.0:
            raise ValueError("The backoff factor must be < 1.0.")
        self._enabled = enabled
        self._scale, self._growth_tracker = float(init_scale), 0
        self._growth_factor, self._backoff_factor, self._growth_interval = growth_factor, backoff_factor, growth_interval
        self._found_inf: Dict[torch.device, torch.Tensor] = {}
        self._unscaled = set()       # id(optimizer) already unscaled since the last update()
        self._overflow = False       # any optimizer saw inf/NaN since the last update()
        self._sstate = None          # device state [scale, growth_tracker, found_inf, applied_steps]: see capturable()

    # -- HIP-graph capture ------------------------------------------------------------------
    def capturable(self, device, applied_steps: int = 0):
        """Move the scaler's state to the device, as torch.amp.GradScaler keeps it: scale(), unscale_(), step() and
        update() then run without the host reading the overflow flag, so the whole step can be captured in a HIP graph
        (torch_em_amd/graph.py).  `applied_steps` seeds the optimizer step count that travels with it."""
        self._from_device()   # a re-capture: the live scale / growth tracker are on the device, not in the host fields
        self._sstate = torch.tensor([self._scale, float(self._growth_tracker), 0.0, float(applied_steps)],
                                    dtype=torch.float32, device=device)
        return self._sstate

    def _from_device(self):
        if self._sstate is not None:
            vals = self._sstate.tolist()     # host sync
            self._scale, self._growth_tracker = float(vals[0]), int(vals[1])

    # -- queries ---------------------------------------------------------------------------
    def is_enabled(self) -> bool:
        return self._enabled

    def get_scale(self) -> float:
        self._from_device()
        return self._scale if self._enabled else 1.0

    def get_growth_factor(self):
        return self._growth_factor

    def get_backoff_factor(self):
        return self._backoff_factor

    def get_growth_interval(self):
        return self._growth_interval

    # -- the three calls of the training loop ----------------------------------------------------
    def scale(self, outputs):
        if not self._enabled:
            return outputs
        if isinstance(outputs, (list, tuple)):
            return type(outputs)(self.scale(o) for o in outputs)
        if self._sstate is not None:
            return outputs * self._sstate[0]
        return outputs * self._scale

    def _flag(self, device):
        f = self._found_inf.get(device)
        if f is None:
            f = self._found_inf[device] = torch.zeros(1, dtype=torch.float32, device=device)
        return f

    def unscale_(self, optimizer):
        if not self._enabled:
            return
        if id(optimizer) in self._unscaled:
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().")
        self._unscaled.add(id(optimizer))
        inv = 1.0 / self._scale
        flat = None
        if isinstance(optimizer, FusedAdamW):
            optimizer._ensure_arena()
            flat = optimizer._arena.grads_flat()
        if self._sstate is not None:
            if flat is None:
                raise RuntimeError("GradScaler.capturable: needs FusedAdamW with the engine's flat gradient arena")
            ops.amp_unscale_dev(flat, self._sstate)
            return
        if flat is not None:
            ops.amp_unscale(flat, inv, self._flag(flat.device))
            return
        for group in optimizer.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                    raise RuntimeError("GradScaler: gradients must be contiguous fp32 CUDA tensors")
                ops.amp_unscale(g.view(-1), inv, self._flag(g.device))

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if id(optimizer) not in self._unscaled:
            self.unscale_(optimizer)
        if self._sstate is not None:
            if getattr(optimizer, "_table", None) is None:
                # the optimizer left its capture mode (FusedAdamW.load_state_dict drops the step table) while this scaler is
                # still on the device: its plain step would apply an overflowed gradient -- read the flag here
                if float(self._sstate[2].item()) != 0.0:
                    return None
            return optimizer.step(*args, **kwargs)      # tem_adamw_step_tab reads the overflow flag itself
        found = any(float(f.item()) != 0.0 for f in self._found_inf.values())   # host sync, like torch's _maybe_opt_step
        if found:
            self._overflow = True
            return None
        return optimizer.step(*args, **kwargs)

    def update(self, new_scale=None):
        if not self._enabled:
            return
        if self._sstate is not None:
            if new_scale is not None:
                raise NotImplementedError("GradScaler.capturable: update(new_scale)")
            ops.amp_update_dev(self._sstate, self._growth_factor, self._backoff_factor, self._growth_interval)
            self._unscaled.clear()
            return
        if new_scale is not None:
            self._scale = float(new_scale)
        elif self._overflow:
            self._scale *= self._backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self._growth_interval:
                self._scale *= self._growth_factor
                self._growth_tracker = 0
        for f in self._found_inf.values():
            f.zero_()
        self._unscaled.clear()
        self._overflow = False

    # -- checkpointing (torch.amp.GradScaler's keys) -------------------------------------------
    def state_dict(self):
        if not self._enabled:
            return {}
        self._from_device()
        return {"scale": self._scale, "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, state_dict):
        if not self._enabled:
            return
        if len(state_dict) == 0:
            raise RuntimeError("The source state dict is empty, possibly because it was saved from a disabled GradScaler.")
        self._scale = float(state_dict["scale"])
        self._growth_factor, self._backoff_factor = state_dict["growth_factor"], state_dict["backoff_factor"]
        self._growth_interval, self._growth_tracker = state_dict["growth_interval"], state_dict["_growth_tracker"]
        # a device-side copy of the state belongs to a captured step whose optimizer step count no longer matches what a
        # checkpoint restores: drop it and tell the owner of the graph (it re-captures with the loaded values)
        self._sstate = None
        hooks, self._invalidate_hooks = getattr(self, "_invalidate_hooks", []), []
        for h in hooks:
            h()
