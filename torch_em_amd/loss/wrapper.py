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
    prediction = prediction * mask
    target = target * mask
    return prediction, target


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

    def split(self, prediction, target, mask=None):
        """(prediction, target, mask) BEFORE masking: what LossWrapper hands to a loss kernel that multiplies on the fly."""
        if mask is None:
            raise TypeError("ApplyMask needs a mask: loss(prediction, target, mask=...)")
        return prediction, target, mask

    def __call__(self, prediction, target, mask=None):
        prediction, target, mask = self.split(prediction, target, mask)
        mask.requires_grad = False
        return self.masking_func(prediction, target, mask, self.channel_dim)


class ApplyAndRemoveMask(ApplyMask):
    """The target carries the mask in its second half of channels (reference :129-152)."""

    def split(self, prediction, target, mask=None):
        assert target.dim() == prediction.dim(), f"{target.dim()}, {prediction.dim()}"
        assert target.size(1) == 2 * prediction.size(1), f"{target.size(1)}, {prediction.size(1)}"
        assert target.shape[2:] == prediction.shape[2:], f"{str(target.shape)}, {str(prediction.shape)}"
        sep = target.size(1) // 2
        return prediction, target[:, :sep], target[:, sep:]


class MaskIgnoreLabel(ApplyMask):
    """Mask where target == ignore_label (reference :155-183)."""

    def __init__(self, ignore_label: int = -1, masking_method: str = "crop", channel_dim: int = 1):
        super().__init__(masking_method, channel_dim)
        self.ignore_label = ignore_label
        self.init_kwargs["ignore_label"] = ignore_label

    def split(self, prediction, target, mask=None):
        return prediction, target, (target != self.ignore_label)


class LossWrapper(nn.Module):
    """Apply `transform(prediction, target, **kwargs)` and then `loss` (reference :7-65)."""

    def __init__(self, loss: nn.Module, transform: Callable):
        super().__init__()
        self.loss = loss
        if not callable(transform):
            raise ValueError("transform has to be callable.")
        self.transform = transform
        self.init_kwargs = {"loss": loss, "transform": transform}

    def _fused(self, prediction, target, kwargs):
        """DiceLoss behind a multiply-mask transform: the Dice kernel applies the mask itself (no masked temporaries)."""
        if not (isinstance(self.loss, DiceLoss) and isinstance(self.transform, ApplyMask) and
                self.transform.masking_method == "multiply" and torch.is_tensor(prediction) and prediction.is_cuda):
            return None
        prediction, target, mask = self.transform.split(prediction, target, **kwargs)
        mask = mask.to(prediction.dtype)
        if mask.shape != target.shape or mask.stride() != target.stride():
            # a broadcast mask (singleton channel) or a differently laid out one: the kernel reads target and mask with one
            # set of strides, so both become dense tensors of the same layout
            target, mask = (t.contiguous() for t in torch.broadcast_tensors(target, mask))
        return self.loss(prediction, target, mask=mask)

    def apply_transform(self, prediction, target, **kwargs):
        if isinstance(prediction, (list, tuple)):
            assert isinstance(target, (list, tuple))
            out = [self.transform(p, t, **kwargs) for p, t in zip(prediction, target)]
            return [o[0] for o in out], [o[1] for o in out]
        return self.transform(prediction, target, **kwargs)

    def forward(self, prediction, target, **kwargs):
        fused = self._fused(prediction, target, kwargs)
        if fused is not None:
            return fused
        prediction, target = self.apply_transform(prediction, target, **kwargs)
        return self.loss(prediction, target)
