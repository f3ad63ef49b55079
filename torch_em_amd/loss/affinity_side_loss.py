"""Affinity side loss for embedding training on the MI355X path.

Drop-in for `torch_em.loss.affinity_side_loss.AffinitySideLoss` (reference loss/affinity_side_loss.py:114-172): the
offsets are drawn from the global numpy RNG exactly like the reference (:158-159); embeddings -> affinities
(`embeddings_to_affinities`, :92-111), labels -> affinities with replication padding (`shift_tensor` :9-61,
`segmentation_to_affinities` :70-89) and the Dice score between them are fused into two streaming kernels
(csrc/spoco.hip: k_aff_sums / k_aff_grad) -- no [K, E, *S] shifted copies.  Batch size 1 (how the SPOCO losses call it).
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import spoco_loss as _sp


class _AffinityFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, target, offsets, delta):
        e = _sp._prep(emb.detach())
        if e.shape[0] != 1:
            raise NotImplementedError("AffinitySideLoss on MI355X handles one sample per call (as the SPOCO losses use it)")
        E = e.shape[1]
        spatial = tuple(e.shape[2:])
        nz, D, H, W = _sp._geom(spatial)
        V = D * H * W
        lbl = target.to(torch.int64).contiguous()
        c = _sp._Ctx(e, 1, E, V, nz, 0, len(offsets))
        need = emb.requires_grad
        grad = torch.zeros_like(e) if need else None
        val = torch.zeros(1, dtype=torch.float32, device=e.device)
        _sp._affinity_side(c, e[0], V, lbl, D, H, W, E, offsets, len(spatial), delta, val, 1.0,
                           grad[0] if need else None, V)
        ctx.grad = grad
        return val.reshape(())

    @staticmethod
    def backward(ctx, gout):
        if ctx.grad is None:
            return None, None, None, None
        return ctx.grad * gout, None, None, None


class AffinitySideLoss(nn.Module):
    """Loss between affinities derived from predicted embeddings and a target segmentation; random offsets per call."""

    def __init__(self, offset_ranges: List[Tuple[int, int]], n_samples: int, delta: float):
        assert all(len(orange) == 2 for orange in offset_ranges)
        super().__init__()
        if n_samples > 32:
            raise ValueError("torch_em_amd supports at most 32 offsets per call")
        self.ndim = len(offset_ranges)
        self.offset_ranges = offset_ranges
        self.n_samples = n_samples
        self.delta = delta

    def __call__(self, input_: torch.Tensor, target: torch.Tensor, ignore_labels: Optional[List[int]] = None,
                 ignore_in_variance_term: Optional[List[int]] = None,
                 ignore_in_distance_term: Optional[List[int]] = None) -> torch.Tensor:
        assert input_.dim() == target.dim(), f"{input_.dim()}, {target.dim()}"
        assert input_.shape[2:] == target.shape[2:]
        assert input_.dim() - 2 == self.ndim
        offsets = _sp._draw_offsets(self.offset_ranges, self.n_samples)
        return _AffinityFunction.apply(input_, target, offsets, self.delta)
