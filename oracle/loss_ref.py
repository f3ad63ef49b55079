"""CPU restatement of the reference Dice losses (TEST INFRASTRUCTURE, see oracle/__init__.py).

  flatten_samples / dice_score / DiceLoss   /root/reference/torch_em/loss/dice.py:7-133
  LossWrapper + ApplyAndRemoveMask(multiply) /root/reference/torch_em/loss/wrapper.py:44-65,84-87,129-152
"""
import torch


def _flatten(t):
    c = t.shape[1]
    return t.transpose(0, 1).reshape(c, -1)


def dice_score(p, t, invert=False, channelwise=True, reduce_channel="sum", eps=1e-7):
    if p.shape != t.shape:
        raise ValueError(f"Expect input and target of same shape, got: {p.shape}, {t.shape}.")
    if channelwise:
        pf, tf = _flatten(p), _flatten(t)
        num = (pf * tf).sum(-1)
        den = (pf * pf).sum(-1) + (tf * tf).sum(-1)
        score = 2 * (num / den.clamp(min=eps))
        if invert:
            score = 1.0 - score
        if reduce_channel is None:
            return score
        return getattr(score, reduce_channel)()
    num = (p * t).sum()
    den = (p * p).sum() + (t * t).sum()
    score = 2.0 * (num / den.clamp(min=eps))
    return 1.0 - score if invert else score


def dice_loss(p, t, channelwise=True, eps=1e-7, reduce_channel="sum"):
    return dice_score(p, t, invert=True, channelwise=channelwise, reduce_channel=reduce_channel, eps=eps)


def masked_dice_loss(p, target_with_mask, **kw):
    """LossWrapper(DiceLoss(), ApplyAndRemoveMask(masking_method="multiply")) (reference cli.py:263-267)."""
    assert target_with_mask.shape[1] == 2 * p.shape[1]
    c = p.shape[1]
    t, m = target_with_mask[:, :c], target_with_mask[:, c:]
    return dice_loss(p * m, t * m, **kw)
