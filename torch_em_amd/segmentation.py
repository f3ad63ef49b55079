"""Convenience factories of the MI355X path (reference torch_em/segmentation.py).

`default_segmentation_trainer` keeps the reference's signature and defaults (:466-577):
AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2) (:543), ReduceLROnPlateau(mode="min",
factor=0.5, patience=5) (:544, :19), DiceLoss as loss and metric (:546-547) -- with the optimizer
being the single-launch FusedAdamW.  The loader factory covers the in-memory (`TensorDataset`) branch
(:143-148); file-backed datasets are plumbing outside the hot path.
"""
from typing import Any, Callable, Dict, Optional, Union

import numpy as np
import torch

from .data import TensorDataset
from .loss import DiceLoss
from .optim import FusedAdamW
from .trainer import DefaultTrainer

DEFAULT_SCHEDULER_KWARGS = {"mode": "min", "factor": 0.5, "patience": 5}


def standardize_raw(raw, eps: float = 1e-7):
    """Default raw transform (reference transform/raw.py:40-65, whole-array mean/std)."""
    raw = np.asarray(raw, dtype="float32")
    return (raw - raw.mean()) / (raw.std() + eps)


def default_segmentation_dataset(raw_paths, raw_key, label_paths, label_key, patch_shape, label_transform=None,
                                 label_transform2=None, raw_transform=None, transform=None, dtype=torch.float32,
                                 label_dtype=torch.float32, n_samples=None, sampler=None, with_channels=False,
                                 with_padding=True, **unused):
    if not (isinstance(raw_paths, (list, tuple)) and len(raw_paths) and
            isinstance(raw_paths[0], (np.ndarray, torch.Tensor))):
        raise NotImplementedError(
            "torch_em_amd builds in-memory datasets (lists of arrays/tensors, the reference's TensorDataset branch); "
            "file-backed datasets are outside the MI355X hot path -- load the arrays and pass them in"
        )
    assert raw_key is None and label_key is None
    return TensorDataset(list(raw_paths), list(label_paths), patch_shape=patch_shape,
                         raw_transform=standardize_raw if raw_transform is None else raw_transform,
                         label_transform=label_transform, label_transform2=label_transform2, transform=transform,
                         dtype=dtype, label_dtype=label_dtype, n_samples=n_samples, sampler=sampler,
                         with_padding=with_padding, with_channels=with_channels)


def default_segmentation_loader(raw_paths, raw_key, label_paths, label_key, batch_size, patch_shape,
                                **kwargs) -> torch.utils.data.DataLoader:
    """Dataset + DataLoader (reference :222-330); loader kwargs are split off like the reference does."""
    loader_keys = ("shuffle", "num_workers", "pin_memory", "drop_last", "sampler", "collate_fn", "persistent_workers")
    loader_kwargs = {k: kwargs.pop(k) for k in list(kwargs) if k in loader_keys and k != "sampler"}
    ds = default_segmentation_dataset(raw_paths, raw_key, label_paths, label_key, patch_shape, **kwargs)
    loader = torch.utils.data.DataLoader(ds, batch_size=batch_size, **loader_kwargs)
    loader.shuffle = loader_kwargs.get("shuffle", False)
    return loader


def default_segmentation_trainer(name: str, model: torch.nn.Module, train_loader, val_loader,
                                 loss: Optional[torch.nn.Module] = None, metric: Optional[Callable] = None,
                                 learning_rate: float = 1e-3, device: Optional[Union[str, torch.device, int]] = None,
                                 log_image_interval: int = 100, mixed_precision: bool = True,
                                 early_stopping: Optional[int] = None, logger=None,
                                 logger_kwargs: Optional[Dict[str, Any]] = None,
                                 scheduler_kwargs: Dict[str, Any] = DEFAULT_SCHEDULER_KWARGS,
                                 optimizer_kwargs: Dict[str, Any] = {}, trainer_class=DefaultTrainer,
                                 id_: Optional[str] = None, save_root: Optional[str] = None,
                                 compile_model: Optional[Union[bool, str]] = None, rank: Optional[int] = None,
                                 mixed_precision_dtype: Optional[str] = None, optimizer=None, lr_scheduler=None,
                                 target_transform: Optional[Callable] = None, augmentation: Optional[Callable] = None,
                                 raw_transform: Optional[Callable] = None, prefetch: bool = True,
                                 hip_graph: Optional[bool] = None):
    """Trainer with the reference's defaults (reference :466-577).  Not in the reference: the on-device `target_transform` /
    `augmentation` / `raw_transform`, `prefetch` and `hip_graph` arguments of this path's DefaultTrainer, handed through."""
    if optimizer is None:
        optimizer = FusedAdamW(model.parameters(), lr=learning_rate, **optimizer_kwargs)
    if lr_scheduler is None:
        lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, **scheduler_kwargs)
    loss = DiceLoss() if loss is None else loss
    metric = DiceLoss() if metric is None else metric
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("torch_em_amd trains on MI355X only: no GPU is visible and there is no CPU fallback")
        device = torch.device("cuda")
    return trainer_class(name=name, model=model, train_loader=train_loader, val_loader=val_loader, loss=loss,
                         metric=metric, optimizer=optimizer, device=device, lr_scheduler=lr_scheduler,
                         mixed_precision=mixed_precision, early_stopping=early_stopping,
                         log_image_interval=log_image_interval, logger=logger, logger_kwargs=logger_kwargs, id_=id_,
                         save_root=save_root, compile_model=compile_model, rank=rank,
                         mixed_precision_dtype=mixed_precision_dtype, target_transform=target_transform,
                         augmentation=augmentation, raw_transform=raw_transform, prefetch=prefetch, hip_graph=hip_graph)
