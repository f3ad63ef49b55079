"""DefaultTrainer for the MI355X path.

Same constructor, `fit()` contract, counters and checkpoint schema as the reference
(torch_em/trainer/default_trainer.py: ctor :86-146, `_initialize` :503-575, `save_checkpoint` :577-602,
`load_checkpoint` :604-642, `fit` :651-775, hot loop `_train_epoch_impl` :805-831, `_validate_impl` :841-861),
so scripts written against torch-em's trainer run unchanged and reference tools can read the
checkpoints (`iteration, epoch, best_epoch, best_metric, current_metric, model_state, optimizer_state,
init, train_time, timestamp[, scheduler_state]`).

Differences that follow from the MI355X-first design:
  * `mixed_precision=True` -- the reference's DEFAULT -- means on a GPU what it means in the reference (:132-140):
    float16 autocast + GradScaler.  Since round 6 this trainer does the same: the step runs inside
    `engine.precision_scope("amp")` (fp16 tensors between the kernels, conv operands fp16, one MFMA per product, fp32
    accumulation: the counterpart of what `torch.autocast(float16)` does to this network, reference :134-142, :800-803) with
    `optim.GradScaler` doing `scale(loss).backward(); step(optimizer); update()` exactly as `_backprop_mixed`
    (:789-794), `scaler_state` saved / restored like the reference (:595-596, :636-638).  Rounds 1-5 kept the fp32-class
    arithmetic for the bare flag (and warned): the mixed mode had nothing reference-held under it.  It has now -- G10
    (tests/golden/gen_golden_amp_step.py, tests/test_gpu_amp_reference.py): one step of the reference under its own
    autocast sits 4.8e-2 (fp16) / 1.6e-1 (bf16) from the float64 gradient, this mode 4.4e-2 / 1.4e-1, tensor by tensor the
    same class -- so an unchanged torch-em script trains the numbers the reference would, at twice the speed of the
    fp32-class path.  The parity-grade arithmetic (fp32-class products, the 1e-3 contract of the north star) is
    `mixed_precision=False`, or TEM_MIXED_PRECISION=0 in the environment for scripts that cannot be edited (the test
    suite pins it that way, tests/conftest.py).
    `mixed_precision_dtype="bfloat16"` runs `precision_scope("amp_bf16")`: operands rounded to bf16, one MFMA per product,
    and -- as in the reference, which creates a GradScaler for float16 only -- no loss scaling;
  * `compile_model` is accepted and ignored: the model is already one hand-scheduled autograd node,
    there is no tracing compiler in this path;
  * the hot loop never synchronises the host: loss values are kept on device and only read at
    validation / logging time.
"""
import contextlib
import os
import time
import warnings
from collections import OrderedDict
from datetime import datetime
from importlib import import_module
from typing import Any, Callable, Dict, Optional, Union

import numpy as np
import torch

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = None


def _class_path(obj):
    cls = obj if isinstance(obj, type) else obj.__class__
    return f"{cls.__module__}.{cls.__name__}"


def _import_class(path):
    mod, name = path.rsplit(".", 1)
    return getattr(import_module(mod), name)


def _init_kwargs(obj):
    """Constructor arguments of torch_em-style objects (they store `init_kwargs`; reference util/util.py:299-304)."""
    if hasattr(obj, "init_kwargs"):
        return dict(obj.init_kwargs)
    if isinstance(obj, torch.optim.Optimizer):
        return {k: v for k, v in obj.defaults.items() if k in ("lr", "betas", "eps", "weight_decay")}
    if isinstance(obj, torch.optim.lr_scheduler.ReduceLROnPlateau):
        return {"mode": obj.mode, "factor": obj.factor, "patience": obj.patience}
    return {}


_FROM_CHECKPOINT = object()   # sentinel: "take it from the checkpoint"


class _NullProgress:
    total = 0

    def update(self, n):
        pass

    def set_description(self, *a, **k):
        pass


class DefaultTrainer:
    """Trainer with the reference's interface; see the module docstring."""

    def __init__(self, name: Optional[str], train_loader, val_loader, model: torch.nn.Module, loss, optimizer,
                 metric: Callable, device: Union[str, torch.device, int], lr_scheduler=None,
                 log_image_interval: int = 100, mixed_precision: bool = True, early_stopping: Optional[int] = None,
                 logger=None, logger_kwargs: Optional[Dict[str, Any]] = None, id_: Optional[str] = None,
                 save_root: Optional[str] = None, compile_model: Optional[Union[bool, str]] = None,
                 rank: Optional[int] = None, mixed_precision_dtype: Optional[str] = None,
                 target_transform: Optional[Callable] = None, augmentation: Optional[Callable] = None,
                 raw_transform: Optional[Callable] = None, prefetch: bool = True, hip_graph: Optional[bool] = None):
        if name is None:
            raise TypeError("Name cannot be None if not using the WandbLogger")
        self.name, self.id_ = name, id_ or name
        self.train_loader, self.val_loader = train_loader, val_loader
        self.model, self.loss, self.optimizer, self.metric = model, loss, optimizer, metric
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.lr_scheduler = lr_scheduler
        self.log_image_interval = log_image_interval
        self.save_root, self.compile_model, self.rank = save_root, compile_model, rank
        self._iteration = self._epoch = self._best_epoch = 0
        self.mixed_precision = mixed_precision
        self.mixed_precision_dtype = mixed_precision_dtype or "float16"
        # see the module docstring: mixed_precision=True on a GPU => float16 "amp" arithmetic + GradScaler, as in the reference
        # (:132-140); TEM_MIXED_PRECISION=0 keeps the fp32-class path for the bare flag (an explicit dtype always wins)
        asked = mixed_precision_dtype == "float16" or (mixed_precision_dtype is None and
                                                        os.environ.get("TEM_MIXED_PRECISION", "1") != "0")
        self._amp = bool(mixed_precision) and asked and self.device.type == "cuda"
        # "bfloat16": one bf16 MFMA per product, fp32 exponent range => no scaler (reference :134-142)
        self._amp_bf16 = bool(mixed_precision) and mixed_precision_dtype == "bfloat16" and self.device.type == "cuda"
        self._mixed_precision_explicit = mixed_precision_dtype is not None
        if self._amp:
            from ..optim import GradScaler
            self.scaler = GradScaler()
        elif self._amp_bf16:
            from ..optim import GradScaler
            self.scaler = GradScaler(enabled=False)   # as the reference: created, but disabled for bfloat16 (:138-140)
        else:
            self.scaler = None  # fp32-class path: no loss scaling
        self.early_stopping = early_stopping
        self.train_time = 0.0
        # hip_graph: replay the training step (zero_grad .. optimizer step) as one captured HIP graph instead of ~220
        # launches enqueued from Python (torch_em_amd/graph.py).  Same results bit for bit; pays off where the host is
        # the bottleneck (small patches).  None: TEM_HIP_GRAPH=1 in the environment.  Needs FusedAdamW and a single GPU;
        # otherwise the step runs eagerly (self._graph_why says why).  With mixed precision the GradScaler's state moves
        # to the device for good (as torch.amp.GradScaler keeps it).
        self.hip_graph = (os.environ.get("TEM_HIP_GRAPH", "0") == "1") if hip_graph is None else bool(hip_graph)
        self._graphed, self._graph_why = None, None
        self.logger_class, self.logger_kwargs = logger, logger_kwargs
        self.logger = None
        # not in the reference: computes the training target from the label batch ON DEVICE (e.g.
        # transform.label.BatchTargets(AffinityTransform(...))) instead of in the CPU data-loader workers
        self.target_transform = target_transform
        # not in the reference either: an on-device augmentation pipeline (transform.augmentation.get_augmentations)
        # applied to the (x, y) TRAINING batch already in HBM; the reference augments in the CPU data-loader workers
        self.augmentation = augmentation
        # not in the reference: a raw transform applied to the device batch (e.g. functools.partial(standardize,
        # per_sample=True): the reference standardises each sample in the loader workers, transform/raw.py:40-65), and
        # `prefetch`: copy + pre-pass of batch k+1 run on a side stream during step k (trainer/input_pipeline.py)
        self.raw_transform = raw_transform
        self.prefetch = prefetch

    # ---- bookkeeping ------------------------------------------------------------------
    @property
    def checkpoint_folder(self):
        root = getattr(self, "save_root", None)
        return os.path.join("./checkpoints", self.id_) if root is None else os.path.join(root, "./checkpoints", self.id_)

    iteration = property(lambda self: self._iteration)
    epoch = property(lambda self: self._epoch)

    def _build_init(self):
        """What `from_checkpoint` needs to rebuild the trainer (class paths + kwargs; reference :332-501)."""
        inner = getattr(self.model, "module", self.model)
        return {
            "name": self.name, "id_": self.id_, "device": str(self.device), "rank": self.rank,
            "save_root": self.save_root, "compile_model": self.compile_model,
            "mixed_precision": self.mixed_precision, "mixed_precision_dtype": self.mixed_precision_dtype,
            "mixed_precision_explicit": self._mixed_precision_explicit,
            "early_stopping": self.early_stopping, "log_image_interval": self.log_image_interval,
            "hip_graph": self.hip_graph,
            "logger_class": None if self.logger_class is None else _class_path(self.logger_class),
            "logger_kwargs": self.logger_kwargs,
            "model_class": _class_path(inner), "model_kwargs": _init_kwargs(inner),
            "loss_class": _class_path(self.loss), "loss_kwargs": _init_kwargs(self.loss),
            "metric_class": _class_path(self.metric), "metric_kwargs": _init_kwargs(self.metric),
            "optimizer_class": _class_path(self.optimizer), "optimizer_kwargs": _init_kwargs(self.optimizer),
            "lr_scheduler_class": None if self.lr_scheduler is None else _class_path(self.lr_scheduler),
            "lr_scheduler_kwargs": None if self.lr_scheduler is None else _init_kwargs(self.lr_scheduler),
            "train_dataset": getattr(self.train_loader, "dataset", None),
            "val_dataset": getattr(self.val_loader, "dataset", None),
            "train_loader_kwargs": {"batch_size": getattr(self.train_loader, "batch_size", None)},
            "val_loader_kwargs": {"batch_size": getattr(self.val_loader, "batch_size", None)},
            # the on-device pre-pass (standardisation, target generation, augmentation) replaces transforms that the
            # reference keeps inside its pickled datasets -- so it has to survive a resume the same way
            **self._device_transforms_record(),
        }

    def _device_transforms_record(self):
        import pickle
        rec, lost = {"prefetch": self.prefetch}, []
        for key in ("raw_transform", "target_transform", "augmentation"):
            obj = getattr(self, key)
            try:
                pickle.dumps(obj)
            except Exception:  # a lambda / local function: cannot travel in the checkpoint
                obj = None
                lost.append(key)
            rec[key] = obj
        rec["unpicklable_transforms"] = lost
        if lost:
            warnings.warn(f"DefaultTrainer: {lost} cannot be pickled into the checkpoint (lambda / local function?); "
                          "from_checkpoint() will ask for them again")
        return rec

    def _plan(self, iterations, epochs):
        """(iterations, epochs) to run from here: the caller names exactly one of them, the other follows from the
        number of batches per epoch (same contract and message as the reference's `_initialize`, :512-528)."""
        named = [v for v in (iterations, epochs) if v is not None]
        if len(named) != 1:
            raise ValueError(
                "Exactly one of 'iterations' or 'epochs' has to be specified to initialize the trainer."
                f"You have passed 'iterations'={iterations} and 'epochs'={epochs}"
            )
        per_epoch = len(self.train_loader)
        if iterations is not None:
            return iterations, -(-int(iterations) // per_epoch)      # ceil: a partial last epoch still counts as one
        return epochs * per_epoch, epochs

    def _initialize(self, iterations, load_from_checkpoint, epochs=None):
        missing = [a for a in ("train_loader", "val_loader", "model", "loss", "optimizer", "metric", "device")
                   if getattr(self, a) is None]
        assert not missing, f"DefaultTrainer: {missing} must be set before fit()"
        if load_from_checkpoint is not None:
            self.load_checkpoint(load_from_checkpoint)
        iterations, epochs = self._plan(iterations, epochs)
        self.max_iteration, self.max_epoch = self._iteration + iterations, self._epoch + epochs
        if not getattr(self, "_is_initialized", False):
            self.model.to(self.device)
            if isinstance(self.loss, torch.nn.Module):
                self.loss.to(self.device)
            self.init_data = self._build_init()
            if self.logger_class is not None:
                self.logger = self.logger_class(self, self.save_root, **(self.logger_kwargs or {}))
            try:
                os.makedirs(self.checkpoint_folder, exist_ok=True)
            except PermissionError:
                warnings.warn(f"The checkpoint folder at {self.checkpoint_folder} could not be created.")
            self._is_initialized = True
        return np.inf

    # ---- checkpoints --------------------------------------------------------------------
    def save_checkpoint(self, name, current_metric, best_metric, train_time=0.0, **extra_save_dict):
        extra_init = extra_save_dict.pop("init", {})
        save_dict = {
            "iteration": self._iteration, "epoch": self._epoch, "best_epoch": self._best_epoch,
            "best_metric": best_metric, "current_metric": current_metric,
            "model_state": self.model.state_dict(), "optimizer_state": self.optimizer.state_dict(),
            "init": {**self.init_data, **extra_init}, "train_time": train_time,
            "timestamp": datetime.now().strftime("%d-%m-%Y (%H:%M:%S)"),
        }
        save_dict.update(**extra_save_dict)
        if self.lr_scheduler is not None:
            save_dict["scheduler_state"] = self.lr_scheduler.state_dict()
        if self.scaler is not None:
            save_dict["scaler_state"] = self.scaler.state_dict()
        if self.rank is None or self.rank == 0:  # rank-0 only, as the reference (:600-602)
            torch.save(save_dict, os.path.join(self.checkpoint_folder, f"{name}.pt"))

    def load_checkpoint(self, checkpoint="best"):
        if isinstance(checkpoint, str):
            path = os.path.join(self.checkpoint_folder, f"{checkpoint}.pt")
            if not os.path.exists(path):
                warnings.warn(f"Cannot load checkpoint. {path} does not exist.")
                return
            save_dict = torch.load(path, weights_only=False)
        elif isinstance(checkpoint, dict):
            save_dict = checkpoint
        else:
            raise RuntimeError
        self._iteration, self._epoch = save_dict["iteration"], save_dict["epoch"]
        self._best_epoch = save_dict["best_epoch"]
        self.best_metric, self.current_metric = save_dict["best_metric"], save_dict["current_metric"]
        self.train_time = save_dict.get("train_time", 0.0)
        prefix = "_orig_mod."  # checkpoints written from torch.compile'd reference models (:626-630)
        state = OrderedDict((k[len(prefix):] if k.startswith(prefix) else k, v)
                            for k, v in save_dict["model_state"].items())
        self.model.load_state_dict(state)
        self.model.to(self.device)
        self._graphed = None   # a captured step points at the buffers the next line re-homes (it also marks itself stale)
        self.optimizer.load_state_dict(save_dict["optimizer_state"])
        if self.lr_scheduler is not None and "scheduler_state" in save_dict:
            self.lr_scheduler.load_state_dict(save_dict["scheduler_state"])
        scaler_state = save_dict.get("scaler_state")
        if self.scaler is not None and scaler_state:
            self.scaler.load_state_dict(scaler_state)
        return save_dict

    @classmethod
    def from_checkpoint(cls, checkpoint_folder, name="best", device=None, train_loader=None, val_loader=None,
                        raw_transform=_FROM_CHECKPOINT, target_transform=_FROM_CHECKPOINT, augmentation=_FROM_CHECKPOINT):
        """Rebuild a trainer from `<checkpoint_folder>/<name>.pt` (reference :288-330).  Loaders are rebuilt
        from the pickled datasets unless given.  The on-device `raw_transform` / `target_transform` / `augmentation` of
        the saved trainer come back from the checkpoint; one that could not be pickled must be passed again (RuntimeError
        otherwise: training on un-standardised inputs or raw label ids must not happen silently)."""
        save_dict = torch.load(os.path.join(checkpoint_folder, f"{name}.pt"), weights_only=False)
        init = save_dict["init"]
        given = {"raw_transform": raw_transform, "target_transform": target_transform, "augmentation": augmentation}
        missing = [k for k in init.get("unpicklable_transforms", []) if given[k] is _FROM_CHECKPOINT]
        if missing:
            raise RuntimeError(f"from_checkpoint: the saved trainer used {missing}, which could not be stored in the "
                               "checkpoint; pass them to from_checkpoint()")
        transforms = {k: (init.get(k) if v is _FROM_CHECKPOINT else v) for k, v in given.items()}
        model = _import_class(init["model_class"])(**init["model_kwargs"])
        loss = _import_class(init["loss_class"])(**init["loss_kwargs"])
        metric = _import_class(init["metric_class"])(**init["metric_kwargs"])
        optimizer = _import_class(init["optimizer_class"])(model.parameters(), **init["optimizer_kwargs"])
        sched = None
        if init.get("lr_scheduler_class"):
            sched = _import_class(init["lr_scheduler_class"])(optimizer, **init["lr_scheduler_kwargs"])
        if train_loader is None:
            train_loader = torch.utils.data.DataLoader(init["train_dataset"], **init["train_loader_kwargs"])
        if val_loader is None:
            val_loader = torch.utils.data.DataLoader(init["val_dataset"], **init["val_loader_kwargs"])
        trainer = cls(name=init["name"], train_loader=train_loader, val_loader=val_loader, model=model, loss=loss,
                      optimizer=optimizer, metric=metric, device=device or init["device"], lr_scheduler=sched,
                      log_image_interval=init["log_image_interval"], mixed_precision=init["mixed_precision"],
                      early_stopping=init["early_stopping"], logger=None, id_=init["id_"],
                      save_root=init["save_root"], rank=init.get("rank"),
                      mixed_precision_dtype=init["mixed_precision_dtype"]
                      if init.get("mixed_precision_explicit", False) else None,
                      prefetch=init.get("prefetch", True), hip_graph=init.get("hip_graph"), **transforms)
        trainer._initialize(0, save_dict)
        trainer._is_initialized = True
        return trainer

    def _verify_if_training_completed(self, checkpoint="latest"):
        path = os.path.join(self.checkpoint_folder, f"{checkpoint}.pt")
        save_dict = torch.load(path, weights_only=False) if os.path.exists(path) else None
        return bool(save_dict and self.max_iteration == save_dict.get("iteration"))

    # ---- training ------------------------------------------------------------------------
    def fit(self, iterations: Optional[int] = None, load_from_checkpoint=None, epochs: Optional[int] = None,
            save_every_kth_epoch: Optional[int] = None, progress=None, overwrite_training: bool = True):
        """Run training; exactly one of `iterations` / `epochs` (reference :651-775)."""
        best_metric = self._initialize(iterations, load_from_checkpoint, epochs)
        if not overwrite_training:
            if load_from_checkpoint is not None:
                raise ValueError(
                    "We do not support 'overwrite_training=False' and 'load_from_checkpoint' at the same time."
                )
            if self._verify_if_training_completed():
                print(f"The model is trained for {self.max_iteration} iterations / {self.max_epoch} epochs "
                      "and 'overwrite_training' is set to 'False'.")
                return
        print("Start fitting for", self.max_iteration - self._iteration, "iterations / ",
              self.max_epoch - self._epoch, "epochs")
        print("with", len(self.train_loader), "iterations per epoch")
        if getattr(self, "_amp", False):      # the reference prints one of two such lines (:712-716)
            print("Training with mixed precision (fp16 operands on the MI355X matrix cores, dynamic loss scaling)")
        elif getattr(self, "_amp_bf16", False):
            print("Training with mixed precision (bf16 operands on the MI355X matrix cores)")
        else:
            print("Training with single precision (fp32-class products on the MI355X matrix cores)")
        total = epochs * len(self.train_loader) if iterations is None else iterations
        if progress is None:
            progress = tqdm(total=total, desc=f"Epoch {self._epoch}", leave=True) if tqdm else _NullProgress()
        else:
            progress.total = total
            progress.set_description(f"Epoch {self._epoch}")
        msg = "Epoch %i: average [s/it]: %f, current metric: %f, best metric: %f"
        t_start = time.time()
        total_train_time = self.train_time
        for epoch in range(self.max_epoch - self._epoch):
            try:
                self.train_loader.sampler.set_epoch(epoch)  # DistributedSampler reshuffle
            except AttributeError:
                pass
            t_per_iter = self._train_epoch(progress)
            current_metric = self._validate()
            if self.lr_scheduler is not None:
                self.lr_scheduler.step(current_metric)
            total_train_time = (time.time() - t_start) + self.train_time
            if current_metric < best_metric:
                best_metric, self._best_epoch = current_metric, self._epoch
                self.save_checkpoint("best", current_metric, best_metric, train_time=total_train_time)
            self.save_checkpoint("latest", current_metric, best_metric, train_time=total_train_time)
            if save_every_kth_epoch is not None and (self._epoch + 1) % save_every_kth_epoch == 0:
                self.save_checkpoint(f"epoch-{self._epoch + 1}", current_metric, best_metric,
                                     train_time=total_train_time)
            if self.early_stopping is not None and self._epoch - self._best_epoch > self.early_stopping:
                print("Stopping training because there has been no improvement for", self.early_stopping, "epochs")
                break
            self._epoch += 1
            progress.set_description(msg % (self._epoch, t_per_iter, current_metric, best_metric), refresh=True)
        print(f"Finished training after {self._epoch} epochs / {self._iteration} iterations.")
        print(f"The best epoch is number {self._best_epoch}.")
        self.train_time = total_train_time

    def _targets(self, y):
        """On-device target generation (transform/label.py `BatchTargets`): applied ONCE per batch, so that the loss, the
        metric and the logger all see the same transformed targets (as they do when the reference's datasets apply
        `label_transform` in the loader workers, data/segmentation_dataset.py:226-249)."""
        tt = getattr(self, "target_transform", None)
        return y if tt is None else tt(y)

    def _prepass(self, train: bool):
        """(x, y) -> (x, y) on the device: raw transform, augmentation (training only), target transform."""
        def run(x, y):
            rt = getattr(self, "raw_transform", None)
            if rt is not None:
                x = rt(x)
            if train:
                x, y = self._augment(x, y)
            return x, self._targets(y)
        return run

    def _batches(self, loader, train: bool):
        from .input_pipeline import DevicePrefetcher
        on_gpu = self.device.type == "cuda"
        return DevicePrefetcher(loader, self.device, self._prepass(train), enabled=on_gpu and getattr(self, "prefetch", True))

    def _forward_and_loss(self, x, y):
        pred = self.model(x)
        return pred, self.loss(pred, y)

    def _augment(self, x, y):
        aug = getattr(self, "augmentation", None)
        if aug is None:
            return x, y
        xa, ya = aug(x, y)
        return xa.to(x.dtype), ya.to(y.dtype)

    def _precision(self):
        """The counterpart of the reference's autocast context (:800-803)."""
        if getattr(self, "_amp", False) or getattr(self, "_amp_bf16", False):
            from ..model.engine import precision_scope
            return precision_scope("amp" if getattr(self, "_amp", False) else "amp_bf16")
        return contextlib.nullcontext()

    def _backprop(self, loss):
        if self.scaler is not None:  # reference `_backprop_mixed` (:789-794)
            self.scaler.scale(loss).backward()
            self.scaler.step(self.optimizer)
            self.scaler.update()
            return
        loss.backward()
        self.optimizer.step()

    def _graphed_step(self, x, y):
        """The captured step for this batch shape, or None (with the reason in self._graph_why) when it must run eagerly."""
        g = self._graphed
        if g is not None and g.stale:   # load_checkpoint / load_state_dict since the capture: capture again
            g = self._graphed = None
        if g is not None and g.matches(x, y):
            return g
        from ..graph import GraphedTrainStep
        from ..optim import FusedAdamW
        why = None
        if not (torch.is_tensor(x) and torch.is_tensor(y) and x.is_cuda):
            why = "batch is not a pair of device tensors"
        elif not isinstance(self.optimizer, FusedAdamW):
            why = "optimizer is not FusedAdamW"
        elif g is not None:
            why = "batch shape changed (the graph is captured for one shape)"
        else:
            from ..graph import cumulative_average_norm, multi_gpu_capture_blocker
            why = cumulative_average_norm(self.model) or multi_gpu_capture_blocker(self.model)
        self._graph_why = why
        if why is not None:
            return None
        self._graphed = GraphedTrainStep(self.model, self.loss, self.optimizer, x, y, scaler=self.scaler,
                                         precision="amp" if getattr(self, "_amp", False) else
                                         "amp_bf16" if getattr(self, "_amp_bf16", False) else None)
        return self._graphed

    def _train_epoch(self, progress):
        """The hot loop (reference :805-831): H2D copy, zero_grad, forward, loss, backward, step."""
        self.model.train()
        n_iter, t0 = 0, time.time()
        for x, y in self._batches(self.train_loader, train=True):
            step = self._graphed_step(x, y) if self.hip_graph else None
            if step is not None:
                pred, loss = step(x, y)
            else:
                self.optimizer.zero_grad()
                with self._precision():
                    pred, loss = self._forward_and_loss(x, y)
                    self._backprop(loss)
            if self.logger is not None:
                lr = [pm["lr"] for pm in self.optimizer.param_groups][0]
                self.logger.log_train(self._iteration, loss, lr, x, y, pred, log_gradients=True)
            self._iteration += 1
            n_iter += 1
            if self._iteration >= self.max_iteration:
                break
            progress.update(1)
        return (time.time() - t0) / max(n_iter, 1)

    def _validate(self):
        """Mean metric over the validation loader (reference :841-861)."""
        self.model.eval()
        metric_val = loss_val = None
        with torch.no_grad(), self._precision():
            for x, y in self._batches(self.val_loader, train=False):
                pred, loss = self._forward_and_loss(x, y)
                metric = self.metric(pred, y)
                loss_val = loss.detach() if loss_val is None else loss_val + loss.detach()
                metric_val = metric.detach() if metric_val is None else metric_val + metric.detach()
        n = len(self.val_loader)
        metric_val, loss_val = float(metric_val) / n, float(loss_val) / n  # the only host sync of the epoch
        if self.logger is not None:
            self.logger.log_validation(self._iteration, metric_val, loss_val, x, y, pred)
        return metric_val
