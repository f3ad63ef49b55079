"""Dice score / loss for the MI355X path.

Drop-in for `torch_em.loss.dice` (reference loss/dice.py): `flatten_samples` (:7-31), `dice_score`
(:34-93), `DiceLoss` (:96-133), `DiceLossWithLogits` (:136-173), `BCEDiceLoss` (:176-214),
`BCEDiceLossWithLogits` (:217-253).  Same signatures, `init_kwargs`, and `ValueError`s.  The arithmetic runs in libtem_hip.so: prediction and target are
read once, in place, through their strides (NDHWC predictions of the engine and NCDHW targets of
the data loader alike) -- the reference's two `permute().contiguous()` copies never happen.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import ops


def flatten_samples(input_: torch.Tensor) -> torch.Tensor:
    """(N, C, ...) -> (C, N * ...), as reference loss/dice.py:7-31 (utility; the loss does not need it)."""
    num_channels = input_.size(1)
    axes = list(range(input_.dim()))
    axes[0], axes[1] = axes[1], axes[0]
    return input_.permute(*axes).contiguous().view(num_channels, -1)


def _is_channels_last(t: torch.Tensor) -> bool:
    return t.dim() >= 3 and t.stride(1) == 1 and t.shape[1] > 1


class _DiceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_, target, mask, invert, channelwise, reduce_channel, eps):
        sums, p, t = ops.dice_sums(input_, target, mask)
        out, ca, cb = ops.dice_finalize(sums, eps, channelwise, invert, reduce_channel)
        ctx.save_for_backward(p, t, ca, cb)
        ctx.mask = mask
        ctx.per_channel = channelwise and reduce_channel is None
        ctx.cl = _is_channels_last(input_)
        return out if ctx.per_channel else out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        p, t, ca, cb = ctx.saved_tensors
        gout = gout.contiguous().float()
        gp = ops.dice_grad(p, t, ctx.mask, ca, cb, gout, ctx.per_channel, ctx.cl)
        return gp, None, None, None, None, None, None


def _check_inputs(input_, target):
    if input_.shape != target.shape:
        raise ValueError(f"Expect input and target of same shape, got: {input_.shape}, {target.shape}.")
    if input_.dim() < 2:
        raise ValueError("dice: the input must be at least 2d")


def dice_score(input_: torch.Tensor, target: torch.Tensor, invert: bool = False, channelwise: bool = True,
               reduce_channel: Optional[str] = "sum", eps: float = 1e-7, mask: Optional[torch.Tensor] = None):
    """Dice score between input and target (reference loss/dice.py:34-93).

    `mask` (not in the reference signature) multiplies prediction and target first: it is the fused
    form of LossWrapper + ApplyAndRemoveMask(masking_method="multiply") (loss/wrapper.py:84-87)."""
    _check_inputs(input_, target)
    if channelwise and reduce_channel not in ("sum", "mean", "max", "min", None):
        raise ValueError(f"Unsupported channel reduction {reduce_channel}")
    if not input_.is_cuda:
        raise RuntimeError("torch_em_amd.loss runs on MI355X only (got CPU tensors); there is no CPU fallback")
    target = target.to(torch.float32)
    input_ = input_.to(torch.float32)
    if mask is not None:
        mask = mask.to(torch.float32)
    return _DiceFunction.apply(input_, target, mask, invert, channelwise, reduce_channel, eps)


class DiceLoss(nn.Module):
    """Dice error 1 - 2<x,t>/(|x|^2+|t|^2) per channel, reduced over channels (reference loss/dice.py:96-133)."""

    def __init__(self, channelwise: bool = True, eps: float = 1e-7, reduce_channel: Optional[str] = "sum"):
        if reduce_channel not in ("sum", "mean", "max", "min", None):
            raise ValueError(f"Unsupported channel reduction {reduce_channel}")
        super().__init__()
        self.channelwise, self.eps, self.reduce_channel = channelwise, eps, reduce_channel
        self.init_kwargs = {"channelwise": channelwise, "eps": self.eps, "reduce_channel": self.reduce_channel}

    def forward(self, input_: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor] = None):
        return dice_score(input_, target, invert=True, channelwise=self.channelwise, eps=self.eps,
                          reduce_channel=self.reduce_channel, mask=mask)


class _DiceFamilyFunction(torch.autograd.Function):
    """alpha * dice + beta * mean binary cross entropy on probabilities or logits: one pass for the sums (the sigmoid and the
    cross entropy are evaluated while the prediction streams by), one for the gradient (tem_dice_sums2 / tem_dice_grad2)."""

    @staticmethod
    def forward(ctx, input_, target, flags, alpha, beta, channelwise, reduce_channel, eps):
        sums, p, t = ops.dice_sums2(input_, target, flags)
        out, ca, cb = ops.dice_finalize2(sums, eps, channelwise, True, reduce_channel)
        ctx.per_channel = channelwise and reduce_channel is None
        if flags & ops.DICE_BCE:   # O(channels) scalars: combined with torch, on the device, no sync
            out = alpha * out + beta * (sums[:, 3].sum() / p.numel()).to(torch.float32)
        ctx.save_for_backward(p, t, ca, cb)
        ctx.flags, ctx.w = flags, (alpha, beta / p.numel() if flags & ops.DICE_BCE else 0.0)
        ctx.cl = _is_channels_last(input_)
        return out if ctx.per_channel else out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        p, t, ca, cb = ctx.saved_tensors
        gp = ops.dice_grad2(p, t, ca, cb, gout.contiguous().float(), ctx.per_channel, ctx.cl, ctx.flags, ctx.w[0], ctx.w[1])
        return gp, None, None, None, None, None, None, None


def _family(input_, target, flags, alpha, beta, channelwise, reduce_channel, eps):
    _check_inputs(input_, target)
    if not input_.is_cuda:
        raise RuntimeError("torch_em_amd.loss runs on MI355X only (got CPU tensors); there is no CPU fallback")
    return _DiceFamilyFunction.apply(input_.to(torch.float32), target.to(torch.float32), flags, alpha, beta, channelwise,
                                     reduce_channel, eps)


class DiceLossWithLogits(nn.Module):
    """DiceLoss of sigmoid(input) (reference loss/dice.py:136-173); the sigmoid is never materialised."""

    def __init__(self, channelwise: bool = True, eps: float = 1e-7, reduce_channel: Optional[str] = "sum"):
        if reduce_channel not in ("sum", "mean", "max", "min", None):
            raise ValueError(f"Unsupported channel reduction {reduce_channel}")
        super().__init__()
        self.channelwise, self.eps, self.reduce_channel = channelwise, eps, reduce_channel
        self.init_kwargs = {"channelwise": channelwise, "eps": self.eps, "reduce_channel": self.reduce_channel}

    def forward(self, input_: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _family(input_, target, ops.DICE_LOGITS, 1.0, 0.0, self.channelwise, self.reduce_channel, self.eps)


class BCEDiceLoss(nn.Module):
    """alpha * DiceLoss + beta * binary_cross_entropy on probabilities (reference loss/dice.py:176-214)."""

    def __init__(self, alpha: float = 1.0, beta: float = 1.0, channelwise: bool = True, eps: float = 1e-7):
        super().__init__()
        self.alpha, self.beta, self.channelwise, self.eps = alpha, beta, channelwise, eps
        self.init_kwargs = {"alpha": alpha, "beta": beta, "channelwise": channelwise, "eps": self.eps}

    def forward(self, input_: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _family(input_, target, ops.DICE_BCE, self.alpha, self.beta, self.channelwise, "sum", self.eps)


class BCEDiceLossWithLogits(nn.Module):
    """alpha * DiceLoss(sigmoid(x)) + beta * binary_cross_entropy_with_logits(x) (reference loss/dice.py:217-253)."""

    def __init__(self, alpha: float = 1.0, beta: float = 1.0, channelwise: bool = True, eps: float = 1e-7):
        super().__init__()
        self.alpha, self.beta, self.channelwise, self.eps = alpha, beta, channelwise, eps
        self.init_kwargs = {"alpha": alpha, "beta": beta, "channelwise": channelwise, "eps": self.eps}

    def forward(self, input_, target):
        return _family(input_, target, ops.DICE_LOGITS | ops.DICE_BCE, self.alpha, self.beta, self.channelwise, "sum",
                       self.eps)
