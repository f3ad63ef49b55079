"""Raw-data transforms on the hot path (reference torch_em/transform/raw.py)."""
import numpy as np
import torch

from .. import ops


def standardize(raw, mean=None, std=None, axis=None, eps: float = 1e-7, per_sample: bool = False):
    """(raw - mean) / (std + eps), the default raw transform (reference `:40-65`, `segmentation.py:394-395`).
    A CUDA tensor runs through one HIP pass pair (`tem_standardize`) with the reference's semantics: `axis=None` means
    statistics over the WHOLE array.  `per_sample=True` (not in the reference: for a device batch [N, ...] that the
    reference would have standardised sample by sample in its loader workers) or `axis` = all axes but the first give
    one mean / std per entry of the first axis.  numpy input, or explicit mean / std: the reference's numpy expression."""
    if torch.is_tensor(raw) and raw.is_cuda and mean is None and std is None:
        rest = tuple(range(1, raw.dim()))
        if axis is not None and tuple(a % raw.dim() for a in np.atleast_1d(axis)) == rest:
            per_sample, axis = True, None
        if axis is None:
            x = raw.float()
            return ops.standardize(x, eps) if per_sample else ops.standardize(x.reshape(1, -1), eps).reshape(x.shape)
        raise NotImplementedError("standardize on the device: axis must be None or all axes but the first")
    raw = np.asarray(raw, dtype="float32")
    mean = raw.mean(axis=axis, keepdims=True) if mean is None else mean
    std = raw.std(axis=axis, keepdims=True) if std is None else std
    return (raw - mean) / (std + eps)
