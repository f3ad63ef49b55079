"""SPOCOTrainer for the MI355X path (reference torch_em/trainer/spoco_trainer.py).

Student model + momentum ("EMA") teacher `model2` (`:36-38`); each iteration: student forward, no-grad
teacher forward, `loss((prediction, prediction2), y)`, backward/step, then the teacher update
theta' <- m*theta' + (1-m)*theta (`_momentum_update`, `:45-47`).  The update is ONE HIP launch over the flat
parameter arenas of both models (`tem_ema_update`) instead of 46 torch element-wise ops; checkpoints carry
`model2_state` exactly like the reference (`:49-63`).
"""
import time
from copy import deepcopy
from typing import Optional

import torch

from .. import ops
from ..arena import ParamArena
from .default_trainer import DefaultTrainer


class SPOCOTrainer(DefaultTrainer):
    def __init__(self, model: torch.nn.Module, momentum: float = 0.999,
                 semisupervised_loss: Optional[torch.nn.Module] = None,
                 semisupervised_loader: Optional[torch.utils.data.DataLoader] = None, logger=None, **kwargs):
        super().__init__(model=model, logger=logger, **kwargs)
        self.momentum = momentum
        self.model2 = deepcopy(self.model)  # the teacher never needs gradients
        for param in self.model2.parameters():
            param.requires_grad = False
        assert (semisupervised_loss is None) == (semisupervised_loader is None)
        self.semisupervised_loader, self.semisupervised_loss = semisupervised_loader, semisupervised_loss
        self._kwargs = kwargs
        self._arena1 = self._arena2 = None

    # ---- EMA teacher ------------------------------------------------------------------------
    def _momentum_update(self):
        p1 = [p for p in self.model.parameters()]
        if not p1 or not p1[0].is_cuda:
            raise RuntimeError("SPOCOTrainer._momentum_update runs on MI355X only (parameters are on the CPU)")
        if self._arena2 is None or not self._arena2.is_current():
            self._arena2 = ParamArena(self.model2)
        src = self._student_flat(p1)
        if src is not None:
            ops.ema_update(self._arena2.flat, src, float(self.momentum))
        else:  # no shared flat buffer (foreign optimizer layout): one launch per tensor, nothing is re-homed
            for q, p in zip(self._arena2.params, p1):
                ops.ema_update(q.data.view(-1), p.data.contiguous().view(-1), float(self.momentum))
        ops.bump_versions(self._arena2.params)

    def _student_flat(self, params):
        """The student's flat parameter buffer WITHOUT re-homing its parameters a second time: FusedAdamW already keeps
        them in one arena (optim.py); building another ParamArena over the same tensors would invalidate that one, and
        the two would rebuild each other (and every packed weight) on every step."""
        ar = getattr(self.optimizer, "_arena", None)
        if ar is not None and ar.is_current() and len(ar.params) == len(params) and \
                all(a is b for a, b in zip(ar.params, params)):
            return ar.flat
        if hasattr(self.optimizer, "_ensure_arena"):
            return None  # a FusedAdamW over a different parameter list: leave its arena alone
        if self._arena1 is None or not self._arena1.is_current():
            self._arena1 = ParamArena(self.model)
        return self._arena1.flat

    def save_checkpoint(self, name, current_metric, best_metric, **extra_save_dict):
        super().save_checkpoint(name, current_metric, best_metric, model2_state=self.model2.state_dict(),
                                **extra_save_dict)

    def load_checkpoint(self, checkpoint="best"):
        save_dict = super().load_checkpoint(checkpoint)
        self.model2.load_state_dict(save_dict["model2_state"])
        self.model2.to(self.device)
        return save_dict

    def _initialize(self, iterations, load_from_checkpoint, epochs=None):
        best_metric = super()._initialize(iterations, load_from_checkpoint, epochs)
        self.model2.to(self.device)
        return best_metric

    # ---- loops --------------------------------------------------------------------------------
    def _step(self, x, loss_fn, y=None):
        self.optimizer.zero_grad()
        with self._precision():
            prediction = self.model(x)
            with torch.no_grad():
                prediction2 = self.model2(x)
            loss = loss_fn(prediction, prediction2) if y is None else loss_fn((prediction, prediction2), y)
            self._backprop(loss)
        with torch.no_grad():
            self._momentum_update()
        return prediction, loss

    def _train_epoch(self, progress):
        self.model.train()
        self.model2.train()
        n_iter, t0 = 0, time.time()
        for x, y in self._batches(self.train_loader, train=True):
            prediction, loss = self._step(x, self.loss, y)
            if self.logger is not None:
                lr = [pm["lr"] for pm in self.optimizer.param_groups][0]
                self.logger.log_train(self._iteration, loss, lr, x, y, prediction, log_gradients=True)
            self._iteration += 1
            n_iter += 1
            if self._iteration >= self.max_iteration:
                break
            progress.update(1)
        if self.semisupervised_loader is not None:
            progress.set_description(
                f"Run semi-supervised training for {len(self.semisupervised_loader)} iterations", refresh=True)
            for x in self.semisupervised_loader:
                self._step(x.to(self.device, non_blocking=True), self.semisupervised_loss)
        return (time.time() - t0) / max(n_iter, 1)

    def _validate(self):
        self.model.eval()
        self.model2.eval()
        metric = loss = None
        with torch.no_grad(), self._precision():
            for x, y in self._batches(self.val_loader, train=False):
                prediction, prediction2 = self.model(x), self.model2(x)
                lv = self.loss((prediction, prediction2), y).detach()
                mv = self.metric(prediction, y).detach()
                loss = lv if loss is None else loss + lv
                metric = mv if metric is None else metric + mv
        n = len(self.val_loader)
        metric, loss = float(metric) / n, float(loss) / n
        if self.logger is not None:
            self.logger.log_validation(self._iteration, metric, loss, x, y, prediction)
        return metric
