"""One training step -- zero_grad, forward, loss, backward, FusedAdamW -- captured once as a HIP graph and replayed.

Why: the cfg-2 step is ~220 kernel launches that the host enqueues through ctypes in 4-5 ms.  At 2x128^3 the GPU needs
18 ms and hides that; a 32^3 step, the mixed-precision step and anything faster are host-bound.  Every launch of this
library is capturable by construction (caller-owned buffers, no synchronisation, no host reads inside
forward / backward / optimizer), so the whole step becomes one `hipGraphLaunch`: ~0.1 ms of host time.

What stays outside the graph, per replay: the copy of the batch into the captured input buffers and the refresh of the
12-float optimizer buffer (step count -> bias corrections, learning rate: `FusedAdamW.refresh_hyper`).

No reference counterpart (torch_em trains eagerly); results are bit-identical to the eager step
(tests/test_gpu_trainer.py::test_graphed_step_equals_eager)."""
import torch

from . import ops
from .optim import FusedAdamW


class GraphedTrainStep:
    def __init__(self, model, loss_fn, optimizer, x, y, warmup: int = 2, precision=None):
        if not (torch.is_tensor(x) and x.is_cuda):
            raise RuntimeError("GraphedTrainStep: HIP graphs need the batch on an MI355X (got a CPU tensor)")
        if not isinstance(optimizer, FusedAdamW):
            raise TypeError("GraphedTrainStep: the optimizer must be torch_em_amd.optim.FusedAdamW (its step reads the "
                            "learning rate and bias corrections from device memory; a captured torch optimizer would "
                            "replay the scalars of the captured step)")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1:
            raise NotImplementedError("GraphedTrainStep: the gradient all-reduce of multi-GPU training is not captured")
        self.model, self.loss_fn, self.optimizer, self.precision = model, loss_fn, optimizer, precision
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.static_x, self.static_y = x.clone(), y.clone()
        optimizer.capturable(True)
        # Warm-up on a side stream (workspaces, gradient arena, packed-weight buffers and the allocator reach their
        # steady state), as torch's capture recipe asks -- but without consuming training steps: parameters, moments and
        # step counts are restored afterwards, so building the graph leaves the training state untouched.
        ar = optimizer._arena
        saved = (ar.flat.clone(), optimizer._m.clone(), optimizer._v.clone(),
                 [optimizer.state[p]["step"].clone() for p in ar.params])
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._eager_step()
            ar.flat.copy_(saved[0])
            optimizer._m.copy_(saved[1])
            optimizer._v.copy_(saved[2])
        torch.cuda.current_stream(x.device).wait_stream(side)
        for p, st in zip(ar.params, saved[3]):
            optimizer.state[p]["step"].copy_(st)
        ops.bump_versions(self.params)
        from .model import engine
        engine._repack_stale(prepare_only=True)   # the weight-packing job table: its upload cannot be captured
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)   # autograd must ASSIGN the captured gradients, not add to old ones
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.pred, self.loss = self._forward_backward_step()
        self.replays = 0

    def _scope(self):
        if self.precision is None:
            import contextlib
            return contextlib.nullcontext()
        from .model.engine import precision_scope
        return precision_scope(self.precision)

    def _forward_backward_step(self):
        with self._scope():
            pred = self.model(self.static_x)
            loss = self.loss_fn(pred, self.static_y)
            loss.backward()
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
        if not self.matches(x, y):
            raise ValueError(f"GraphedTrainStep was captured for x{tuple(self.static_x.shape)} / "
                             f"y{tuple(self.static_y.shape)}; got x{tuple(x.shape)} / y{tuple(y.shape)}")
        self.static_x.copy_(x, non_blocking=True)
        self.static_y.copy_(y, non_blocking=True)
        self.optimizer.refresh_hyper()
        self.graph.replay()
        ops.bump_versions(self.params)   # eager code that runs next (validation) must re-pack the weights
        self.replays += 1
        return self.pred, self.loss
