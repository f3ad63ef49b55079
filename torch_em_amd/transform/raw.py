"""Raw-data transforms on the hot path (reference torch_em/transform/raw.py)."""
import numpy as np
import torch

from .. import ops


def standardize(raw, mean=None, std=None, axis=None, eps: float = 1e-7):
    """(raw - mean) / (std + eps), the default raw transform (reference `:40-65`, `segmentation.py:394-395`).
    CUDA tensor input: per-sample (first axis) statistics in one HIP pass pair (`tem_standardize`);
    numpy input with explicit arguments: the reference's numpy expression (host-side data loading)."""
    if torch.is_tensor(raw) and raw.is_cuda and mean is None and std is None and axis is None:
        return ops.standardize(raw.float(), eps)
    raw = np.asarray(raw, dtype="float32")
    mean = raw.mean(axis=axis, keepdims=True) if mean is None else mean
    std = raw.std(axis=axis, keepdims=True) if std is None else std
    return (raw - mean) / (std + eps)
