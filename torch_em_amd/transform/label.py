"""Label-to-target transforms on device (reference torch_em/transform/label.py).

`BoundaryTransform` (`:100-129`) and `AffinityTransform` (`:248-327`) keep the reference's constructor
arguments and output conventions (channel order, float 0/1 values, "disaffinity" = 1 - affinity, mask channels
appended), but run as integer-compare HIP kernels (`tem_boundary_target`, `tem_affinity_target`), bit-exact
with the reference semantics (scikit-image `find_boundaries(mode="thick")`; the brute-force affinity
definitions of the reference's test/transform/test_label_transforms.py:5-55).

MI355X-first placement: in the reference these run on the CPU inside `Dataset.__getitem__`
(`data/segmentation_dataset.py:233-245`); here they are meant to run on the training device on the label
batch right after the H2D copy (`DefaultTrainer(target_transform=...)`), so the loader ships int labels
(8 B/voxel) instead of float targets (up to 96 B/voxel for 12 affinity channels + masks).  CUDA tensors in ->
CUDA tensors out; numpy in -> numpy out (via the device; main process only).
"""
from typing import List, Optional

import numpy as np
import torch

from .. import ops


def labels_to_binary(labels, background_label: int = 0):
    """(labels != background) in the labels' dtype (reference `:34-44`)."""
    if torch.is_tensor(labels):
        return (labels != background_label).to(labels.dtype)
    return (labels != background_label).astype(labels.dtype)


def _to_device(labels, ndim):
    is_np = not torch.is_tensor(labels)
    t = torch.as_tensor(np.ascontiguousarray(labels)) if is_np else labels
    if t.dtype in (torch.uint64,):
        t = t.to(torch.int64)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("torch_em_amd.transform.label runs on MI355X only; there is no CPU fallback")
        t = t.cuda()
    while ndim is not None and t.dim() > ndim and t.shape[0] == 1:  # ensure_spatial_array: drop singleton axes
        t = t[0]
    return t.to(torch.int64).contiguous(), is_np


class BoundaryTransform:
    """Instance labels -> boundary map [1(+1), *spatial] (reference `:100-129`)."""

    def __init__(self, mode: str = "thick", add_binary_target: bool = False, ndim: Optional[int] = None):
        if mode not in ops.BOUNDARY_MODES:
            raise NotImplementedError(f"BoundaryTransform mode '{mode}': the MI355X kernel has "
                                      f"{sorted(ops.BOUNDARY_MODES)}; 'subpixel' returns a 2n-1 grid, which cannot "
                                      "be a training target")
        self.mode, self.add_binary_target, self.ndim = mode, add_binary_target, ndim

    def __call__(self, labels):
        t, is_np = _to_device(labels, self.ndim)
        if t.dim() not in (2, 3):
            raise ValueError(f"expected 2-D or 3-D labels, got shape {tuple(t.shape)}")
        out = ops.boundary_target(t, self.add_binary_target, self.mode)
        return out.cpu().numpy() if is_np else out


class AffinityTransform:
    """Instance labels -> (dis)affinities [+binary] [+masks] (reference `:248-327`)."""

    def __init__(self, offsets: List[List[int]], ignore_label: Optional[int] = None, add_binary_target: bool = False,
                 add_mask: bool = False, include_ignore_transitions: bool = False):
        self.offsets = offsets
        self.ndim = len(self.offsets[0])
        assert self.ndim in (2, 3)
        self.ignore_label = ignore_label
        self.add_binary_target = add_binary_target
        self.add_mask = add_mask
        self.include_ignore_transitions = include_ignore_transitions

    def __call__(self, labels):
        t, is_np = _to_device(labels, self.ndim)
        if t.dim() != self.ndim:
            raise ValueError(f"expected {self.ndim}-D labels, got shape {tuple(t.shape)}")
        out = ops.affinity_target(t, self.offsets, ignore_label=self.ignore_label,
                                  add_binary_target=self.add_binary_target, add_mask=self.add_mask,
                                  include_ignore_transitions=self.include_ignore_transitions)
        return out.cpu().numpy() if is_np else out


class BatchTargets:
    """Apply a label transform to every sample of an int label batch [N, 1, *spatial] on device."""

    def __init__(self, transform):
        self.transform = transform

    def __call__(self, y: torch.Tensor) -> torch.Tensor:
        return torch.stack([self.transform(s) for s in y])
