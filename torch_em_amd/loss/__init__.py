"""Losses of the MI355X path (reference torch_em/loss/__init__.py:4-11 lists the public names)."""
from .dice import BCEDiceLoss, BCEDiceLossWithLogits, DiceLoss, DiceLossWithLogits, dice_score, flatten_samples
from .wrapper import ApplyAndRemoveMask, ApplyMask, LossWrapper, MaskIgnoreLabel
from .affinity_side_loss import AffinitySideLoss
from .spoco_loss import (ExtendedContrastiveLoss, GaussianKernel, SPOCOConsistencyLoss, SPOCOLoss,
                         compute_cluster_means)
from .contrastive import ContrastiveLoss
