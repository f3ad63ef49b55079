"""Discriminative ("contrastive") embedding loss for the MI355X path.

Drop-in for `torch_em.loss.contrastive.ContrastiveLoss` (reference loss/contrastive.py:25-169; "Semantic Instance
Segmentation with a Discriminative Loss Function"): per sample  alpha * variance + beta * distance + gamma *
regulariser, averaged over the batch.  The reference's two implementations ("expand": one-hot expansion, memory
hungry; "scatter": torch_scatter) compute the same numbers (its tests assert that, test/loss/test_contrastive.py);
here both names select the same HIP kernels (csrc/spoco.hip: fixed-point segment means, pull term, pairwise push
term), so `impl` is accepted and ignored.  No CPU fallback.
"""
import ctypes
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib, ops
from . import spoco_loss as _sp


def check_consecutive(labels: torch.Tensor) -> bool:
    """Labels (sorted unique ids) are consecutive and start at zero (reference :9-19)."""
    diff = labels[1:] - labels[:-1]
    return bool((labels[0] == 0) and (diff == 1).all())


class _ContrastiveFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, target, delta_var, delta_dist, alpha, beta, gamma):
        # one stream handle serves every launch below: the embeddings' device stays current throughout
        with torch.cuda.device(emb.device if emb.is_cuda else None):
            return _ContrastiveFunction._forward(ctx, emb, target, delta_var, delta_dist, alpha, beta, gamma)

    @staticmethod
    def _forward(ctx, emb, target, delta_var, delta_dist, alpha, beta, gamma):
        e_all = _sp._prep(emb.detach())
        N, E = e_all.shape[0], e_all.shape[1]
        spatial = tuple(e_all.shape[2:])
        nz, D, H, W = _sp._geom(spatial)
        V = D * H * W
        if not target.is_cuda:
            raise RuntimeError("torch_em_amd.loss.contrastive runs on MI355X only (target on CPU); no CPU fallback")
        target = target.to(torch.int64).contiguous()
        dev = e_all.device
        lib = _lib.load()
        stream = ops._stream(e_all)
        boot = ops._workspace(1 << 20, dev)
        stats = torch.empty(2 * N, dtype=torch.int64, device=dev)
        for b in range(N):
            _lib.check(lib.tem_label_range(ops._p(target[b, 0]), V, ctypes.c_void_p(stats.data_ptr() + 16 * b), ops._p(boot),
                                           boot.numel(), stream), "tem_label_range")
        host = stats.tolist()  # the one host sync: instance counts set the shapes below
        need = emb.requires_grad
        grad = torch.zeros_like(e_all) if need else None
        vals = torch.zeros(3 * N, dtype=torch.float32, device=dev)
        total = None
        for b in range(N):
            assert host[2 * b] == 0, "labels must start at zero and be consecutive"
            C = host[2 * b + 1] + 1
            c = _sp._Ctx(e_all, C, E, V, nz, 0, 0)
            e, lbl = e_all[b], target[b, 0]
            means, counts = _sp._cluster_means(c, e, V, lbl, V, E, C)
            p, n = c.wsp()
            S = torch.empty(C, E, dtype=torch.float32, device=dev)
            ddist = torch.empty(C, E, dtype=torch.float32, device=dev)
            dreg = torch.empty(C, E, dtype=torch.float32, device=dev)

            def vp(i):
                return ctypes.c_void_p(vals.data_ptr() + 4 * (3 * b + i))
            _lib.check(lib.tem_spoco_pull(ops._p(e), V, ops._p(lbl), V, E, C, ops._p(means), ops._p(counts),
                                          float(delta_var), vp(0), ops._p(S), p, n, stream), "tem_spoco_pull")
            _lib.check(lib.tem_spoco_means_terms(ops._p(means), C, E, float(delta_dist), 0, vp(1), ops._p(ddist),
                                                 ops._p(dreg), p, n, stream), "tem_spoco_means_terms")
            if need:
                s = 1.0 / N
                _lib.check(lib.tem_spoco_embed_grad(
                    ops._p(e), V, ops._p(lbl), V, E, C, ops._p(means), ops._p(counts), ops._p(S), ops._p(ddist),
                    ops._p(dreg), None, float(delta_var), float(C), s * alpha, s * beta, s * gamma, 0.0, ops._p(grad[b]),
                    V, 0, p, n, stream), "tem_spoco_embed_grad")
            lb = alpha * vals[3 * b:3 * b + 1] / C + beta * vals[3 * b + 1:3 * b + 2] + gamma * vals[3 * b + 2:3 * b + 3]
            total = lb if total is None else total + lb
        ctx.grad = grad
        return total / N

    @staticmethod
    def backward(ctx, gout):
        if ctx.grad is None:
            return (None,) * 7
        return (ctx.grad * gout.reshape(()).to(ctx.grad.dtype),) + (None,) * 6


class ContrastiveLoss(nn.Module):
    """alpha * variance + beta * distance + gamma * regularisation of pixel embeddings w.r.t. instance labels."""
    implementations = (None, "scatter", "expand")

    def __init__(self, delta_var: float, delta_dist: float, norm: str = "fro", alpha: float = 1.0, beta: float = 1.0,
                 gamma: float = 0.001, ignore_label: Optional[int] = None, impl: Optional[str] = None):
        assert ignore_label is None, "Not implemented"
        super().__init__()
        if norm != "fro":
            raise ValueError("torch_em_amd implements the default Frobenius (L2) norm only")
        assert impl in self.implementations
        self.delta_var, self.delta_dist, self.norm = delta_var, delta_dist, norm
        self.alpha, self.beta, self.gamma, self.ignore_label = alpha, beta, gamma, ignore_label
        self.init_kwargs = {"delta_var": delta_var, "delta_dist": delta_dist, "norm": norm, "alpha": alpha, "beta": beta,
                            "gamma": gamma, "ignore_label": ignore_label, "impl": impl}

    @staticmethod
    def has_torch_scatter():
        return False  # not needed: the segment means are a HIP kernel

    def forward(self, input_: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        assert target.dim() == input_.dim()
        assert target.shape[1] == 1
        assert input_.shape[0] == target.shape[0]
        assert input_.size()[2:] == target.size()[2:]
        assert input_.dim() - 2 in (2, 3)
        return _ContrastiveFunction.apply(input_, target, self.delta_var, self.delta_dist, self.alpha, self.beta,
                                          self.gamma)
