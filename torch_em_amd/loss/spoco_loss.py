"""SPOCO / contrastive embedding losses for the MI355X path.

Drop-in for `torch_em.loss.spoco_loss` (reference loss/spoco_loss.py): `compute_cluster_means` (:16-33),
`GaussianKernel` (:85-95), `ExtendedContrastiveLoss` (:301-430), `SPOCOLoss` (:433-566),
`SPOCOConsistencyLoss` (:569-643) -- same constructor arguments, `init_kwargs`, call signatures
(`SPOCOLoss((emb_q, emb_k), target)`), the same [1]-shaped result, and the same use of the GLOBAL numpy RNG
(`np.random.randint`, in the reference's call order) for the consistency anchors and the affinity offsets, so a
seeded reference run and a seeded run of this module draw identical anchors.

Reference semantics that are kept on purpose:
  * `loss = ...; loss += loss` inside the batch loop (:289-298): only the LAST sample's contrastive loss survives,
    doubled, divided by N -- so only the last sample is evaluated here (the affinity offsets of the earlier
    samples are still drawn, to keep the RNG stream aligned);
  * the instance Dice term is detached (`torch.tensor(list)`, :422): value only;
  * the variance-term ignore mask is a no-op but still decrements the instance count (contrastive_impl.py:115-118).
The one deliberate difference: with `unlabeled_push_weight > 0` the reference's in-place `variance *= mask`
(contrastive_impl.py:116) makes autograd raise on current torch; here that configuration backpropagates.

All arithmetic runs in libtem_hip.so (csrc/spoco.hip): each term is one or two streaming passes over the embeddings,
read in place; the reference's per-instance Python loop, `torch.nonzero` per anchor and torch_scatter dependency
are gone.  There is no CPU fallback.
"""
import ctypes
import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops
from .dice import DiceLoss

_ZCH = 1024  # chunk size of tem_zero_count


def _geom(spatial):
    """(nz, D, H, W) for 2-D or 3-D spatial shapes; nz is the first spatial axis (DiceLoss()'s channel axis)."""
    if len(spatial) == 2:
        return spatial[0], 1, spatial[0], spatial[1]
    if len(spatial) == 3:
        return spatial[0], spatial[0], spatial[1], spatial[2]
    raise AssertionError(f"expected 2 or 3 spatial dimensions, got {len(spatial)}")


def _prep(emb, what="embeddings"):
    if not emb.is_cuda:
        raise RuntimeError(f"torch_em_amd.loss.spoco_loss runs on MI355X only ({what} on CPU); there is no CPU fallback")
    emb = emb.to(torch.float32)
    return emb if emb.is_contiguous() else emb.contiguous()


class _Ctx:
    """Per-call scratch: library handle, stream, workspace."""

    def __init__(self, ref: torch.Tensor, C, E, V, nz, A, K):
        self.lib = _lib.load()
        self.stream = ops._stream(ref)
        n = int(self.lib.tem_spoco_ws(int(C), int(E), int(V), int(nz), int(A), int(K)))
        self.ws = ops._workspace(n, ref.device)
        self.nws = self.ws.numel()
        self.dev = ref.device

    def wsp(self):
        return ops._p(self.ws), self.nws


def _label_stats(target_b: torch.Tensor, lib, stream, ws):
    out = torch.empty(2, dtype=torch.int64, device=target_b.device)
    _lib.check(lib.tem_label_range(ops._p(target_b), target_b.numel(), ops._p(out), ops._p(ws), ws.numel(), stream),
               "tem_label_range")
    return out


def _cluster_means(ctx, e, cs, lbl, V, E, C):
    means = torch.empty(C, E, dtype=torch.float32, device=ctx.dev)
    counts = torch.empty(C, dtype=torch.float32, device=ctx.dev)
    p, n = ctx.wsp()
    _lib.check(ctx.lib.tem_spoco_cluster_means(ops._p(e), cs, ops._p(lbl), V, E, C, ops._p(means), ops._p(counts), p, n,
                                               ctx.stream), "tem_spoco_cluster_means")
    return means, counts


def compute_cluster_means(embeddings: torch.Tensor, target: torch.Tensor, n_instances: int) -> torch.Tensor:
    """Mean embedding per instance: embeddings [E, *S], target [*S] -> [n_instances, E] (reference :16-33; no autograd)."""
    e = _prep(embeddings.detach())
    lbl = target.to(torch.int64).contiguous()
    E, V = e.shape[0], lbl.numel()
    with torch.cuda.device(e.device):   # the stream handle of _Ctx is used for two launches
        ctx = _Ctx(e, n_instances, E, V, 1, 0, 0)
        rng = _label_stats(lbl, ctx.lib, ctx.stream, ctx.ws).tolist()
        assert rng[0] == 0, "The target min value has to be zero, otherwise this will lead to errors in scatter."
        if rng[1] >= n_instances:
            raise ValueError(f"target contains label {rng[1]} but n_instances is {n_instances}")
        return _cluster_means(ctx, e, V, lbl, V, E, n_instances)[0]


class GaussianKernel(nn.Module):
    """dist -> exp(-dist^2 / two_sigma), two_sigma = delta_var^2 / -ln(pmaps_threshold) (reference :85-95)."""

    def __init__(self, delta_var, pmaps_threshold):
        super().__init__()
        self.delta_var = delta_var
        self.two_sigma = delta_var * delta_var / (-math.log(pmaps_threshold))

    def forward(self, dist_map):
        return torch.exp(-dist_map * dist_map / self.two_sigma)


def _draw_offsets(offset_ranges, n_samples):
    # same call order as affinity_side_loss.py:158-159
    return [[int(np.random.randint(r[0], r[1])) for r in offset_ranges] for _ in range(n_samples)]


def _offsets_zyx(offsets, ndim):
    arr = (ctypes.c_int * (3 * len(offsets)))()
    for k, off in enumerate(offsets):
        zyx = [0] + list(off) if ndim == 2 else list(off)
        arr[3 * k], arr[3 * k + 1], arr[3 * k + 2] = zyx
    return arr


def _affinity_side(ctx, e, cs, lbl, D, H, W, E, offsets, ndim, delta, value, grad_scale, grad, gcs, eps=1e-7):
    arr = _offsets_zyx(offsets, ndim)
    p, n = ctx.wsp()
    _lib.check(ctx.lib.tem_affinity_side(ops._p(e), cs, ops._p(lbl), D, H, W, E, arr, len(offsets), float(delta), eps,
                                         ops._p(value), float(grad_scale), ops._p(grad), gcs, p, n, ctx.stream),
               "tem_affinity_side")


class _SpocoFunction(torch.autograd.Function):
    """value [1] and d value / d emb_q of the (extended contrastive [+ consistency]) loss; the gradient is produced
    during the forward sweep (the terms are recomputed nowhere) and scaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, emb_q, emb_k, target, cfg):
        need_grad = emb_q.requires_grad
        # one stream handle serves all launches of the evaluation: the embeddings' device stays current for its whole
        # duration (ops._stream() alone holds it only until the first _lib.check())
        with torch.cuda.device(emb_q.device):
            value, grad = _evaluate(emb_q.detach(), None if emb_k is None else emb_k.detach(), target, cfg, need_grad)
        ctx.grad = grad
        return value

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad
        if g is None:
            return None, None, None, None
        return g * gout.reshape(()).to(g.dtype), None, None, None


def _evaluate(emb_q, emb_k, target, cfg, need_grad):
    emb_q = _prep(emb_q)
    N, E = emb_q.shape[0], emb_q.shape[1]
    spatial = tuple(emb_q.shape[2:])
    nz, D, H, W = _geom(spatial)
    V = D * H * W
    ndim = len(spatial)
    assert tuple(target.shape[2:]) == spatial, f"{tuple(emb_q.shape)}, {tuple(target.shape)}"
    assert target.shape[1] == 1
    if not target.is_cuda:
        raise RuntimeError("torch_em_amd.loss.spoco_loss runs on MI355X only (target on CPU); there is no CPU fallback")
    target = target.to(torch.int64).contiguous()
    dev = emb_q.device
    aux = cfg["aux_loss"]
    with_aff = aux in ("affinity", "dice_aff")
    with_dice = aux in ("dice", "dice_aff")
    consistency = emb_k is not None
    A = cfg["max_anchors"] if consistency else 0
    K = cfg["n_samples"] if with_aff else 0
    lib = _lib.load()
    stream = ops._stream(emb_q)

    # ---- host-visible integers: label range of the last sample, unlabeled counts of every sample (ONE sync) ----
    boot = ops._workspace(1 << 20, dev)
    nchunk = (V + _ZCH - 1) // _ZCH
    stats = torch.empty(2 + N, dtype=torch.int64, device=dev)
    last = target[N - 1, 0]
    _lib.check(lib.tem_label_range(ops._p(last), V, ops._p(stats), ops._p(boot), boot.numel(), stream), "tem_label_range")
    chunk_counts = None
    if consistency:
        chunk_counts = torch.empty(N, nchunk, dtype=torch.int32, device=dev)
        for b in range(N):
            _lib.check(lib.tem_zero_count(ops._p(target[b, 0]), V, ops._p(chunk_counts[b]),
                                          ctypes.c_void_p(stats.data_ptr() + 8 * (2 + b)), stream), "tem_zero_count")
    host = stats.tolist()
    lmin, lmax = host[0], host[1]
    assert lmin == 0, "The target min value has to be zero, otherwise this will lead to errors in scatter."
    C = lmax + 1
    c = _Ctx(emb_q, C, E, V, nz, A, K)

    # ---- RNG draws, in the reference's order: affinity offsets for every sample, then anchors per sample ----
    offsets = None
    if with_aff:
        for b in range(N):
            offsets = _draw_offsets(cfg["offset_ranges"], cfg["n_samples"])  # only the last sample's are used
        assert len(cfg["offset_ranges"]) == ndim, "offset_ranges must have one range per spatial axis"
    ranks = []
    if consistency:
        for b in range(N):
            n_unl = host[2 + b]
            if n_unl < cfg["volume_threshold"] * V:
                ranks.append(None)
            else:
                ranks.append([int(np.random.randint(n_unl)) for _ in range(A)])

    grad = torch.zeros_like(emb_q) if need_grad else None
    vals = torch.zeros(8 + N, dtype=torch.float32, device=dev)  # var_sum, dist, reg, dice, push, aff | consistency[N]

    def vptr(i):
        return ctypes.c_void_p(vals.data_ptr() + 4 * i)

    # ---- extended contrastive loss of the LAST sample (reference :249-298) ----
    b = N - 1
    e = emb_q[b]
    gb = grad[b] if need_grad else None
    means, counts = _cluster_means(c, e, V, last, V, E, C)
    ignore_zero = cfg["unlabeled_push_weight"] > 0  # contains_bg holds (min label is 0)
    n_inst = C - 1 if ignore_zero else C
    s = 2.0 / N  # `loss += loss` then `/ n_batches`
    p, n = c.wsp()
    S = torch.empty(C, E, dtype=torch.float32, device=dev)
    ddist = torch.empty(C, E, dtype=torch.float32, device=dev)
    dreg = torch.empty(C, E, dtype=torch.float32, device=dev)
    dpush = None
    if n_inst > 0:
        _lib.check(lib.tem_spoco_pull(ops._p(e), V, ops._p(last), V, E, C, ops._p(means), ops._p(counts),
                                      float(cfg["delta_var"]), vptr(0), ops._p(S), p, n, stream), "tem_spoco_pull")
    else:
        S.zero_()
    _lib.check(lib.tem_spoco_means_terms(ops._p(means), C, E, float(cfg["delta_dist"]), int(ignore_zero), vptr(1),
                                         ops._p(ddist), ops._p(dreg), p, n, stream), "tem_spoco_means_terms")
    if with_dice:
        _lib.check(lib.tem_spoco_instance_dice(ops._p(e), V, ops._p(last), V, nz, E, C, ops._p(means),
                                               float(cfg["two_sigma"]), 1e-7, vptr(3), p, n, stream),
                   "tem_spoco_instance_dice")
    use_push = ignore_zero and C > 1
    if use_push:
        dpush = torch.empty(C, E, dtype=torch.float32, device=dev)
        _lib.check(lib.tem_spoco_push(ops._p(e), V, ops._p(last), V, E, C, ops._p(means), ops._p(counts),
                                      float(cfg["delta_dist"]), vptr(4), s * cfg["unlabeled_push_weight"], ops._p(gb), V,
                                      ops._p(dpush), p, n, stream), "tem_spoco_push")
    inst_w_aff = 0.0
    if with_aff:
        inst_w_aff = cfg["instance_term_weight"] * (cfg["aff_weight"] if aux == "dice_aff" else 1.0)
        _affinity_side(c, e, V, last, D, H, W, E, offsets, ndim, cfg["delta_dist"], vals[5:6], s * inst_w_aff, gb, V)
    if need_grad:
        _lib.check(lib.tem_spoco_embed_grad(
            ops._p(e), V, ops._p(last), V, E, C, ops._p(means), ops._p(counts), ops._p(S), ops._p(ddist), ops._p(dreg),
            ops._p(dpush), float(cfg["delta_var"]), float(max(n_inst, 1)), s * cfg["alpha"] if n_inst > 0 else 0.0,
            s * cfg["beta"], s * cfg["gamma"], s * cfg["unlabeled_push_weight"] if use_push else 0.0, ops._p(gb), V, 1,
            p, n, stream), "tem_spoco_embed_grad")

    # ---- consistency term for every sample with enough unlabeled voxels (reference :559-564) ----
    cw = []
    if consistency:
        emb_k = _prep(emb_k, "teacher embeddings")
        assert emb_k.shape == emb_q.shape
        for b in range(N):
            if ranks[b] is None:
                continue
            rk = torch.tensor(ranks[b], dtype=torch.int64, device=dev)
            idx = torch.empty(A, dtype=torch.int64, device=dev)
            _lib.check(lib.tem_zero_select(ops._p(target[b, 0]), V, ops._p(chunk_counts[b]), ops._p(rk), A, ops._p(idx),
                                           stream), "tem_zero_select")
            _lib.check(lib.tem_spoco_consistency(
                ops._p(emb_q[b]), ops._p(emb_k[b]), V, V, nz, E, ops._p(idx), A, float(cfg["two_sigma"]), 1e-7,
                vptr(8 + b), float(cfg["consistency_term_weight"]), ops._p(grad[b]) if need_grad else None, V, p, n,
                stream), "tem_spoco_consistency")
            cw.append(8 + b)

    # ---- combine the device scalars (tiny elementwise ops) ----
    var = vals[0:1] / n_inst if n_inst > 0 else vals[0:1] * 0.0
    if aux == "dice":
        inst = vals[3:4]
    elif aux == "affinity":
        inst = vals[5:6]
    else:
        inst = cfg["dice_weight"] * vals[3:4] + cfg["aff_weight"] * vals[5:6]
    loss = (cfg["alpha"] * var + cfg["beta"] * vals[1:2] + cfg["gamma"] * vals[2:3]
            + cfg["instance_term_weight"] * inst + cfg["unlabeled_push_weight"] * vals[4:5]) * s
    for i in cw:
        loss = loss + cfg["consistency_term_weight"] * vals[i:i + 1]
    return loss, grad


class ExtendedContrastiveLoss(nn.Module):
    """Contrastive loss extended with the instance-based term and the background push term (reference :301-430).

    Same arguments as the reference.  `forward(input_, target)`: input_ [N,E,(D,)H,W] float (or the SPOCO trainer's
    (student, teacher) tuple, of which the student is used), target [N,1,(D,)H,W] with consecutive integer ids from 0.
    """

    def __init__(self, delta_var: float, delta_dist: float, norm: str = "fro", alpha: float = 1.0, beta: float = 1.0,
                 gamma: float = 0.001, unlabeled_push_weight: float = 1.0, instance_term_weight: float = 1.0,
                 aux_loss: str = "dice", pmaps_threshold: float = 0.9, **kwargs):
        super().__init__()
        assert aux_loss in ["dice", "affinity", "dice_aff"]
        if norm != "fro":
            raise ValueError("torch_em_amd implements the default Frobenius (L2) norm only")
        self.delta_var, self.delta_dist, self.norm = delta_var, delta_dist, norm
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        self.unlabeled_push_weight = unlabeled_push_weight
        self.unlabeled_push = unlabeled_push_weight > 0
        self.instance_term_weight = instance_term_weight
        self.aux_loss = aux_loss
        self.dice_weight = kwargs.get("dice_weight", 1.0)
        self.aff_weight = kwargs.get("aff_weight", 1.0)
        self.dice_loss = DiceLoss() if aux_loss in ("dice", "dice_aff") else None
        self.aff_loss = None
        if aux_loss in ("affinity", "dice_aff"):
            from .affinity_side_loss import AffinitySideLoss
            self.aff_loss = AffinitySideLoss(delta=delta_dist,
                                             offset_ranges=kwargs.get("offset_ranges", [(-18, 18), (-18, 18)]),
                                             n_samples=kwargs.get("n_samples", 9))
        self.dist_to_mask = GaussianKernel(delta_var=delta_var, pmaps_threshold=pmaps_threshold)
        self.init_kwargs = {
            "delta_var": delta_var, "delta_dist": delta_dist, "norm": norm, "alpha": alpha, "beta": beta,
            "gamma": gamma, "unlabeled_push_weight": unlabeled_push_weight,
            "instance_term_weight": instance_term_weight, "aux_loss": aux_loss, "pmaps_threshold": pmaps_threshold
        }
        self.init_kwargs.update(kwargs)

    def __str__(self):
        return super().__str__() + f"\ndelta_var: {self.delta_var}\ndelta_dist: {self.delta_dist}" \
                                   f"\nalpha: {self.alpha}\nbeta: {self.beta}\ngamma: {self.gamma}" \
                                   f"\nunlabeled_push_weight: {self.unlabeled_push_weight}" \
                                   f"\ninstance_term_weight: {self.instance_term_weight}"

    def _cfg(self):
        cfg = dict(delta_var=self.delta_var, delta_dist=self.delta_dist, alpha=self.alpha, beta=self.beta,
                   gamma=self.gamma, unlabeled_push_weight=self.unlabeled_push_weight,
                   instance_term_weight=self.instance_term_weight, aux_loss=self.aux_loss,
                   two_sigma=self.dist_to_mask.two_sigma, dice_weight=self.dice_weight, aff_weight=self.aff_weight,
                   max_anchors=0, volume_threshold=0.0, consistency_term_weight=0.0, n_samples=0, offset_ranges=None)
        if self.aff_loss is not None:
            cfg["n_samples"] = self.aff_loss.n_samples
            cfg["offset_ranges"] = self.aff_loss.offset_ranges
        return cfg

    def forward(self, input_, target):
        if isinstance(input_, tuple):
            assert len(input_) == 2
            input_ = input_[0]
        return _SpocoFunction.apply(input_, None, target, self._cfg())


class SPOCOLoss(ExtendedContrastiveLoss):
    """The full SPOCO loss: extended contrastive loss on the student embeddings + embedding-consistency term between
    student and (no-grad) teacher embeddings (reference :433-566).  `forward((emb_q, emb_k), target)`."""

    def __init__(self, delta_var: float, delta_dist: float, norm: str = "fro", alpha: float = 1.0, beta: float = 1.0,
                 gamma: float = 0.001, unlabeled_push_weight: float = 0.0, instance_term_weight: float = 1.0,
                 consistency_term_weight: float = 1.0, aux_loss: str = "dice", pmaps_threshold: float = 0.9,
                 max_anchors: int = 20, volume_threshold: float = 0.05, **kwargs):
        super().__init__(delta_var, delta_dist, norm=norm, alpha=alpha, beta=beta, gamma=gamma,
                         unlabeled_push_weight=unlabeled_push_weight, instance_term_weight=instance_term_weight,
                         aux_loss=aux_loss, pmaps_threshold=pmaps_threshold, **kwargs)
        if max_anchors > 64:
            raise ValueError("torch_em_amd supports at most 64 consistency anchors")
        self.consistency_term_weight = consistency_term_weight
        self.max_anchors = max_anchors
        self.volume_threshold = volume_threshold
        self.consistency_loss = DiceLoss()
        self.init_kwargs = {
            "delta_var": delta_var, "delta_dist": delta_dist, "norm": norm, "alpha": alpha, "beta": beta,
            "gamma": gamma, "unlabeled_push_weight": unlabeled_push_weight,
            "instance_term_weight": instance_term_weight, "aux_loss": aux_loss, "pmaps_threshold": pmaps_threshold,
            "max_anchors": max_anchors, "volume_threshold": volume_threshold
        }
        self.init_kwargs.update(kwargs)

    def __str__(self):
        return super().__str__() + f"\nconsistency_term_weight: {self.consistency_term_weight}"

    def forward(self, input_, target):
        assert len(input_) == 2
        emb_q, emb_k = input_
        cfg = self._cfg()
        cfg.update(max_anchors=self.max_anchors, volume_threshold=self.volume_threshold,
                   consistency_term_weight=self.consistency_term_weight)
        return _SpocoFunction.apply(emb_q, emb_k, target, cfg)


class _ConsistencyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb_q, emb_k, two_sigma, max_anchors):
        q, k = _prep(emb_q.detach()), _prep(emb_k.detach(), "teacher embeddings")
        N, E = q.shape[0], q.shape[1]
        nz, D, H, W = _geom(tuple(q.shape[2:]))
        V = D * H * W
        need = emb_q.requires_grad
        grad = torch.zeros_like(q) if need else None
        vals = torch.zeros(N, dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):   # c.stream serves N launches: keep q's device current for all of them
            c = _Ctx(q, 1, E, V, nz, max_anchors, 0)
            p, n = c.wsp()
            for b in range(N):
                # the reference's mask is all ones (:598), so the rank IS the flat voxel index
                idx = torch.tensor([int(np.random.randint(V)) for _ in range(max_anchors)], dtype=torch.int64,
                                   device=q.device)
                _lib.check(c.lib.tem_spoco_consistency(
                    ops._p(q[b]), ops._p(k[b]), V, V, nz, E, ops._p(idx), max_anchors, float(two_sigma), 1e-7,
                    ctypes.c_void_p(vals.data_ptr() + 4 * b), 1.0, ops._p(grad[b]) if need else None, V, p, n, c.stream),
                    "tem_spoco_consistency")
        ctx.grad = grad
        return vals.sum()

    @staticmethod
    def backward(ctx, gout):
        if ctx.grad is None:
            return None, None, None, None
        return ctx.grad * gout, None, None, None


class SPOCOConsistencyLoss(nn.Module):
    """Unsupervised consistency term between two embedding predictions (reference :569-643)."""

    def __init__(self, delta_var: float, pmaps_threshold: float, max_anchors: int = 30, norm: str = "fro"):
        super().__init__()
        if norm != "fro":
            raise ValueError("torch_em_amd implements the default Frobenius (L2) norm only")
        if max_anchors > 64:
            raise ValueError("torch_em_amd supports at most 64 consistency anchors")
        self.max_anchors = max_anchors
        self.consistency_loss = DiceLoss()
        self.norm = norm
        self.dist_to_mask = GaussianKernel(delta_var=delta_var, pmaps_threshold=pmaps_threshold)
        self.init_kwargs = {"delta_var": delta_var, "pmaps_threshold": pmaps_threshold,
                            "max_anchors": max_anchors, "norm": norm}

    def forward(self, emb_q: torch.Tensor, emb_k: torch.Tensor) -> torch.Tensor:
        return _ConsistencyFunction.apply(emb_q, emb_k, self.dist_to_mask.two_sigma, self.max_anchors)
