"""Loss wrappers (reference loss/wrapper.py): LossWrapper (:7-65), ApplyMask (:90-126),
ApplyAndRemoveMask (:129-152), MaskIgnoreLabel (:155-183).

The affinity loss of the reference is `LossWrapper(DiceLoss(), ApplyAndRemoveMask("multiply"))`
(cli.py:263-267).  When the wrapped loss is this package's DiceLoss and the transform masks by
multiplication, the mask is handed to the Dice kernel, which multiplies on the fly -- the
`prediction * mask` / `target * mask` temporaries of `_multiply` (:84-87) and their backward
never touch HBM.  Gradients are exactly zero outside the mask, as the reference tests require
(test/loss/test_loss_wrapper.py:6-87).
"""
from typing import Callable

import torch
import torch.nn as nn

from .dice import DiceLoss


class _Masked:
    """prediction/target pair plus a multiplicative mask that the loss kernel applies itself."""

    def __init__(self, prediction, target, mask):
        self.prediction, self.target, self.mask = prediction, target, mask


def _crop(prediction, target, mask, channel_dim):
    if mask.shape[channel_dim] != 1:
        raise ValueError(
            "_crop only supports a mask with a singleton channel axis. Please consider using masking_method=multiply."
        )
    mask = mask.type(torch.bool).squeeze(channel_dim)
    prediction = prediction.moveaxis(channel_dim, -1)
    target = target.moveaxis(channel_dim, -1)
    return prediction[mask], target[mask]


def _multiply(prediction, target, mask, channel_dim):
    return _Masked(prediction, target, mask), None


class ApplyMask:
    """Mask prediction and target ('crop' or 'multiply'), reference loss/wrapper.py:90-126."""
    MASKING_FUNCS = {"crop": _crop, "multiply": _multiply}

    def __init__(self, masking_method: str = "crop", channel_dim: int = 1):
        if masking_method not in self.MASKING_FUNCS:
            raise ValueError(f"{masking_method} is not available, please use one of {list(self.MASKING_FUNCS.keys())}.")
        self.masking_method = masking_method
        self.masking_func = self.MASKING_FUNCS[masking_method]
        self.channel_dim = channel_dim
        self.init_kwargs = {"masking_method": masking_method, "channel_dim": channel_dim}

    def __call__(self, prediction, target, mask):
        mask.requires_grad = False
        return self.masking_func(prediction, target, mask, self.channel_dim)


class ApplyAndRemoveMask(ApplyMask):
    """The target carries the mask in its second half of channels (reference :129-152)."""

    def __call__(self, prediction, target):
        assert target.dim() == prediction.dim(), f"{target.dim()}, {prediction.dim()}"
        assert target.size(1) == 2 * prediction.size(1), f"{target.size(1)}, {prediction.size(1)}"
        assert target.shape[2:] == prediction.shape[2:], f"{str(target.shape)}, {str(prediction.shape)}"
        sep = target.size(1) // 2
        return super().__call__(prediction, target[:, :sep], target[:, sep:])


class MaskIgnoreLabel(ApplyMask):
    """Mask where target == ignore_label (reference :155-183)."""

    def __init__(self, ignore_label: int = -1, masking_method: str = "crop", channel_dim: int = 1):
        super().__init__(masking_method, channel_dim)
        self.ignore_label = ignore_label
        self.init_kwargs["ignore_label"] = ignore_label

    def __call__(self, prediction, target):
        mask = (target != self.ignore_label)
        return super().__call__(prediction, target, mask)


class LossWrapper(nn.Module):
    """Apply `transform(prediction, target, **kwargs)` and then `loss` (reference :7-65)."""

    def __init__(self, loss: nn.Module, transform: Callable):
        super().__init__()
        self.loss = loss
        if not callable(transform):
            raise ValueError("transform has to be callable.")
        self.transform = transform
        self.init_kwargs = {"loss": loss, "transform": transform}

    def _call_loss(self, prediction, target):
        if isinstance(prediction, _Masked):
            m = prediction
            if isinstance(self.loss, DiceLoss):
                return self.loss(m.prediction, m.target, mask=m.mask.to(m.prediction.dtype))
            # any other loss: materialise the masked tensors like the reference does
            mask = m.mask.to(m.prediction.dtype)
            return self.loss(m.prediction * mask, m.target * mask)
        return self.loss(prediction, target)

    def apply_transform(self, prediction, target, **kwargs):
        if isinstance(prediction, (list, tuple)):
            assert isinstance(target, (list, tuple))
            out = [self.transform(p, t, **kwargs) for p, t in zip(prediction, target)]
            return [o[0] for o in out], [o[1] for o in out]
        return self.transform(prediction, target, **kwargs)

    def forward(self, prediction, target, **kwargs):
        prediction, target = self.apply_transform(prediction, target, **kwargs)
        if isinstance(prediction, list):
            raise NotImplementedError("list-valued predictions need a loss that accepts lists (as in the reference)")
        return self._call_loss(prediction, target)
