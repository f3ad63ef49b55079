"""One training step -- zero_grad, forward, loss, backward, FusedAdamW -- captured once as a HIP graph and replayed.

Why: the cfg-2 step is ~220 kernel launches that the host enqueues through ctypes in 4-5 ms.  At 2x128^3 the GPU needs
18 ms and hides that; a 32^3 step, the mixed-precision step and anything faster are host-bound.  Every launch of this
library is capturable by construction (caller-owned buffers, no synchronisation, no host reads inside
forward / backward / optimizer), so the whole step becomes one `hipGraphLaunch`: ~0.1 ms of host time.

What stays outside the graph, per replay: the copy of the batch into the captured input buffers and the refresh of the
12-float optimizer buffer (step count -> bias corrections, learning rate: `FusedAdamW.refresh_hyper`).

With dynamic loss scaling (`scaler=`, the mixed-precision step) the overflow decision moves to the device as in
torch.amp.GradScaler: scale / unscale / skip-or-step / update all run inside the graph on a 4-float device state, and
because the host can no longer know how many steps were really applied when it enqueues the next replay, it uploads
the optimizer scalars for a WINDOW of step numbers and the kernel picks its row (`FusedAdamW.refresh_table`); the applied
count comes back through a ring of asynchronous 16-byte reads.

No reference counterpart (torch_em trains eagerly); results are bit-identical to the eager step
(tests/test_gpu_trainer.py::test_graphed_step_equals_eager)."""
import torch

from . import ops
from .optim import FusedAdamW


def cumulative_average_norm(model):
    """A norm layer with running statistics and momentum=None averages them over `num_batches_tracked`, which the engine
    reads on the HOST (engine._update_running): inside a capture that value would be frozen into the graph.  Returns
    the reason string for such a model, else None."""
    for name, mod in model.named_modules():
        if getattr(mod, "track_running_stats", False) and getattr(mod, "running_mean", None) is not None and \
                getattr(mod, "momentum", 0.1) is None:
            return (f"{name or type(mod).__name__}: running statistics with momentum=None (cumulative average) need the "
                    "batch counter on the host each step")
    return None


def multi_gpu_capture_blocker(model):
    """Data-parallel training (multi_gpu_training.DDP): the in-place all-reduces of the gradient arena are RCCL launches on
    ProcessGroupNCCL's stream, forked from and joined to the capturing stream by events -- they become nodes of the graph,
    so every rank replays forward, backward, the exchange and the optimizer with ONE launch (round 4; before, N ranks each
    enqueued ~220 launches per step from Python on one host).  What cannot be captured: a process group that is not
    RCCL (gloo: host-side collectives) and the per-forward buffer broadcast of nets with BatchNorm-style buffers.
    Returns the reason string, or None when the step can be captured."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return None
    sync = getattr(getattr(model, "module", model), "_tem_grad_sync", None)
    if sync is None:
        return None if dist.get_world_size() == 1 else "multi-GPU training without torch_em_amd.multi_gpu_training.DDP"
    if dist.get_backend(sync.pg) != "nccl":
        return f"the gradient all-reduce runs on the '{dist.get_backend(sync.pg)}' backend (only RCCL launches are capturable)"
    if getattr(model, "broadcast_buffers", False) and next(getattr(model, "module", model).buffers(), None) is not None:
        return "module buffers (running statistics) are broadcast from rank 0 before every forward pass"
    return None


class GraphedTrainStep:
    READBACK_SLOTS = 8   # < FusedAdamW.TABLE_ROWS - 1: the host runs at most this many replays ahead of what it knows

    def __init__(self, model, loss_fn, optimizer, x, y, warmup: int = 2, precision=None, scaler=None):
        if not (torch.is_tensor(x) and x.is_cuda):
            raise RuntimeError("GraphedTrainStep: HIP graphs need the batch on an MI355X (got a CPU tensor)")
        if not isinstance(optimizer, FusedAdamW):
            raise TypeError("GraphedTrainStep: the optimizer must be torch_em_amd.optim.FusedAdamW (its step reads the "
                            "learning rate and bias corrections from device memory; a captured torch optimizer would "
                            "replay the scalars of the captured step)")
        why = multi_gpu_capture_blocker(model)
        if why is not None:
            raise NotImplementedError("GraphedTrainStep: " + why)
        why = cumulative_average_norm(model)
        if why is not None:
            raise NotImplementedError("GraphedTrainStep: " + why)
        self.model, self.loss_fn, self.optimizer, self.precision = model, loss_fn, optimizer, precision
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.stale = False   # set when optimizer / scaler state is loaded: the captured pointers and counts are then wrong
        self.static_x, self.static_y = x.clone(), y.clone()
        self.scaler = scaler if (scaler is not None and scaler.is_enabled()) else None
        optimizer._ensure_arena()
        if self.scaler is not None:
            # the host may enqueue READBACK_SLOTS replays beyond the newest step count it has read back; the window of
            # optimizer rows it uploads must cover that lead (tem_adamw_step_tab refuses rows outside the window)
            assert self.READBACK_SLOTS < FusedAdamW.TABLE_ROWS - 1
            step0 = {int(optimizer.state[p]["step"].item()) for p in optimizer._arena.params}
            if len(step0) != 1:
                raise RuntimeError("GraphedTrainStep: the parameters must share one optimizer step count")
            self.sstate = self.scaler.capturable(x.device, applied_steps=step0.pop())
            optimizer.capturable_scaled(self.sstate)
            self._ring = [(torch.zeros(4, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in
                          range(self.READBACK_SLOTS)]
            self._pending = []          # (slot, replay index) of reads in flight, oldest first
            self._known = int(self.sstate[3].item())
            optimizer._pre_state_dict = self.sync_state
        else:
            optimizer.capturable(True)
        # Warm-up on a side stream (workspaces, gradient arena, packed-weight buffers and the allocator reach their
        # steady state), as torch's capture recipe asks -- but without consuming training steps: parameters, moments and
        # step counts are restored afterwards, so building the graph leaves the training state untouched
        ar = optimizer._arena
        saved = (ar.flat.clone(), optimizer._m.clone(), optimizer._v.clone(),
                 [optimizer.state[p]["step"].clone() for p in ar.params],
                 self.sstate.clone() if self.scaler is not None else None)
        # ... nor anything else a step advances: module buffers (BatchNorm / InstanceNormTrackStats running statistics and
        # their batch counters) and the device's random-number state
        buffers = [(b, b.detach().clone()) for b in model.buffers()]
        rng_state = torch.cuda.get_rng_state(x.device)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._eager_step()
            ar.flat.copy_(saved[0])
            optimizer._m.copy_(saved[1])
            optimizer._v.copy_(saved[2])
            if self.scaler is not None:
                self.sstate.copy_(saved[4])
        torch.cuda.current_stream(x.device).wait_stream(side)
        for p, st in zip(ar.params, saved[3]):
            optimizer.state[p]["step"].copy_(st)
        with torch.no_grad():
            for b, old in buffers:
                b.copy_(old)
        torch.cuda.set_rng_state(rng_state, x.device)
        ops.bump_versions(self.params)
        from .model import engine
        with self._scope():
            engine._repack_stale(prepare_only=True)   # the weight-packing job table: its upload cannot be captured
        self.graph = torch.cuda.CUDAGraph()
        sync = getattr(getattr(model, "module", model), "_tem_grad_sync", None)
        measure = None
        if sync is not None:
            measure, sync.measure = sync.measure, False   # its HIP events would become graph nodes: exposed time is an eager-step measurement
        optimizer.zero_grad(set_to_none=True)   # autograd must ASSIGN the captured gradients, not add to old ones
        try:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.pred, self.loss = self._forward_backward_step()
        finally:
            if sync is not None:
                sync.measure = measure   # later eager steps (validation, bench) keep their exposed-time measurement
        # Everything the captured launches point to that was allocated OUTSIDE the capture (and is therefore not owned by
        # the graph's memory pool) must outlive the graph even if the engine later replaces its own reference -- a new
        # pack-job table after an eager validation pass, new packed-weight buffers under another precision mode, a
        # re-homed gradient arena: hold them here.
        self._keepalive = [dict(engine._PACK_TABLES), [dict(c._tem_pack) for c in list(engine._PACKED_CONVS)
                                                        if getattr(c, "_tem_pack", None) is not None],
                           getattr(self.params[0], "_tem_grad_flat", None), ar.flat, optimizer._m, optimizer._v,
                           optimizer._hyper, optimizer._table]
        self.replays = 0
        # optimizer.load_state_dict / scaler.load_state_dict re-home the buffers this graph has baked in
        import weakref
        ref = weakref.ref(self)

        def _invalidate():
            me = ref()
            if me is not None:
                me.stale = True
        for owner in (optimizer, self.scaler):
            if owner is not None:
                if not hasattr(owner, "_invalidate_hooks"):
                    owner._invalidate_hooks = []
                owner._invalidate_hooks.append(_invalidate)

    def _scope(self):
        if self.precision is None:
            import contextlib
            return contextlib.nullcontext()
        from .model.engine import precision_scope
        return precision_scope(self.precision)

    def _forward_backward_step(self):
        from .model import engine
        with self._scope():
            # Gradients via fresh leaves + torch.autograd.grad + assignment instead of loss.backward(): see
            # engine.fresh_leaves (a parameter's AccumulateGrad node may belong to another stream; reaching it from the
            # capturing stream crashed hipStreamEndCapture when the caller still held a loss of an earlier eager step).
            with engine.fresh_leaves() as pairs:
                pred = self.model(self.static_x)
                loss = self.loss_fn(pred, self.static_y)
            if pairs:
                ps = [p for prm, lv in pairs for p, leaf in zip(prm, lv) if leaf.requires_grad]
                leaves = [leaf for prm, lv in pairs for leaf in lv if leaf.requires_grad]
            else:                      # not an engine model: plain parameters
                ps = leaves = [p for p in self.params if p.requires_grad]
            out = self.scaler.scale(loss) if self.scaler is not None else loss   # reference `_backprop_mixed` (:789-794)
            grads = torch.autograd.grad(out, leaves, allow_unused=True)
            for p, g in zip(ps, grads):
                p.grad = g
            if self.scaler is not None:
                self.scaler.step(self.optimizer)
                self.scaler.update()
            else:
                self.optimizer.step()
        return pred, loss

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        return self._forward_backward_step()

    def matches(self, x, y) -> bool:
        return (x.shape == self.static_x.shape and y.shape == self.static_y.shape and x.dtype == self.static_x.dtype and
                y.dtype == self.static_y.dtype and x.device == self.static_x.device)

    def __call__(self, x, y):
        """One optimisation step on (x, y).  Returns (prediction, loss): the graph's own output buffers, overwritten by
        the next call."""
        if self.stale:
            raise RuntimeError("GraphedTrainStep: optimizer or GradScaler state was loaded after the capture; the graph "
                               "still points at the old parameter / moment buffers -- capture a new one")
        if not self.matches(x, y):
            raise ValueError(f"GraphedTrainStep was captured for x{tuple(self.static_x.shape)} / "
                             f"y{tuple(self.static_y.shape)}; got x{tuple(x.shape)} / y{tuple(y.shape)}")
        self.static_x.copy_(x, non_blocking=True)
        self.static_y.copy_(y, non_blocking=True)
        if self.scaler is None:
            self.optimizer.refresh_hyper()
            self.graph.replay()
        else:
            self._poll(block_if_full=True)
            self.optimizer.refresh_table(self._known)
            self.graph.replay()
            slot = self.replays % self.READBACK_SLOTS
            buf, ev = self._ring[slot]
            buf.copy_(self.sstate, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.sstate.device))
            self._pending.append(slot)
        ops.bump_versions(self.params)   # eager code that runs next (validation) must re-pack the weights
        self.replays += 1
        return self.pred, self.loss

    # -- dynamic loss scaling: what the host knows about the device's step count ------------------------------------
    def _poll(self, block_if_full: bool = False):
        """Consume finished state reads (oldest first): `_known` = applied optimizer steps as of the newest one.  With every
        slot in flight, wait for the oldest -- that bounds how stale `_known` can be (READBACK_SLOTS replays), which is
        what the row window of `refresh_table` covers."""
        while self._pending:
            buf, ev = self._ring[self._pending[0]]
            if not ev.query():
                if not (block_if_full and len(self._pending) >= self.READBACK_SLOTS):
                    break
                ev.synchronize()
            self._known = int(buf[3])
            self._pending.pop(0)
        self.optimizer.set_step_count(self._known)

    def sync_state(self):
        """Blocking: bring the host-side mirrors (optimizer step counts, scaler scale) up to date, e.g. for a checkpoint."""
        if self.scaler is None:
            return
        self._known = int(self.sstate[3].item())
        self._pending.clear()
        self.optimizer.set_step_count(self._known)
        self.scaler._from_device()
