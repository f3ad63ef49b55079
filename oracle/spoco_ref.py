"""CPU restatement of the reference SPOCO / contrastive embedding losses.

TEST INFRASTRUCTURE (see oracle/__init__.py): imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  torch-CPU autograd provides the gradients; nothing here is on the product path.

Reference lines this follows (all under /root/reference/torch_em/loss/):
  segment mean of embeddings         spoco_loss.py:16-33, contrastive_impl.py:14-25 (torch_scatter.scatter_mean)
  variance (pull) term               contrastive_impl.py:86-129   (the ignore mask at :115 is a no-op, C is still
                                                                   decremented at :118)
  distance (push) term               contrastive_impl.py:28-80    (bg row/col scaled past the hinge, C_norm = C-1)
  regulariser                        spoco_loss.py:205-213
  unlabeled (background) push        spoco_loss.py:162-190
  per-sample combination             spoco_loss.py:229-298        (`loss = ...; loss += loss` => only the LAST
                                                                   sample survives, doubled, then / N)
  instance term                      spoco_loss.py:386-430        (Dice of Gaussian pmaps, detached by
                                                                   torch.tensor(list) at :422)
  Gaussian kernel                    spoco_loss.py:85-95
  consistency term                   spoco_loss.py:497-566        (<= max_anchors anchors drawn with
                                                                   np.random.randint at :514)
  affinity side loss                 affinity_side_loss.py:9-172  (offsets drawn with np.random.randint at :158)

Host randomness: like the reference, anchors / offsets come from the GLOBAL numpy RNG, in the reference's call
order, unless they are injected through `anchors=` / `offsets=` (lists per sample).
"""
import math

import numpy as np
import torch

from .loss_ref import dice_score


def cluster_means(emb, lbl, n):
    """emb [E, *S] float, lbl [*S] int64 in [0, n) -> [n, E] mean embedding per label (empty label -> 0)."""
    e = emb.flatten(1)
    l = lbl.flatten()
    sums = torch.zeros(e.shape[0], n, dtype=e.dtype).index_add(1, l, e)
    cnt = torch.zeros(n, dtype=e.dtype).index_add(0, l, torch.ones_like(l, dtype=e.dtype))
    return (sums / cnt.clamp(min=1)).transpose(0, 1)


def variance_term(means, emb, lbl, sizes, delta_var, ignore_zero):
    n = means.shape[0]
    mu = means[lbl]                                   # [*S, E]
    mu = mu.movedim(-1, 0)
    d = torch.norm(emb - mu, dim=0)
    if ignore_zero:
        n -= 1
        if n == 0:
            return 0.0
    h = torch.clamp(d - delta_var, min=0) ** 2
    return (h / sizes[lbl]).sum().reshape(1) / n


def distance_term(means, delta_dist, ignore_zero):
    c = means.shape[0]
    if c == 1:
        return 0.0
    dist = torch.norm(means[None, :, :] - means[:, None, :], dim=2)
    c_norm = c
    if ignore_zero:
        if c == 2:
            return 0.0
        d_min = torch.min(dist[dist > 0]).item()
        mult = 2 * delta_dist / d_min + 1e-3
        mask = torch.ones_like(dist)
        mask[0, 1:] = mult
        mask[1:, 0] = mult
        dist = dist * mask
        c_norm -= 1
    rep = 2 * delta_dist * (1 - torch.eye(c, dtype=means.dtype))
    return (torch.clamp(rep - dist, min=0) ** 2).sum() / (c_norm * (c_norm - 1))


def regularizer_term(means):
    return torch.norm(means, dim=1).sum() / means.shape[0]


def unlabeled_push(means, emb, lbl, delta_dist):
    n = means.shape[0] - 1
    if n == 0:
        return 0.0
    e = emb.movedim(0, -1)
    bg = lbl == 0
    n_bg = bg.sum()
    out = 0.0
    for mu in means[1:]:
        d = torch.norm(e - mu, dim=-1)
        out = out + (torch.clamp((delta_dist - d) * bg, min=0) ** 2).sum() / n_bg
    return out / n


def two_sigma(delta_var, pmaps_threshold):
    return delta_var * delta_var / (-math.log(pmaps_threshold))


def pmap(emb_last, anchor, ts):
    d = torch.norm(emb_last - anchor, dim=-1)
    return torch.exp(-d * d / ts)


def instance_dice_term(emb, means, lbl, ts):
    """Value only (the reference detaches it)."""
    e = emb.detach().movedim(0, -1)
    vals = []
    for i in torch.unique(lbl):
        if i == 0:
            continue
        p = pmap(e, means[i].detach(), ts)[None]
        m = (lbl == i).float()[None]
        vals.append(dice_score(p, m, invert=True))
    return torch.tensor(vals).mean() if vals else 0.0


def shift_replicate(t, off):
    """t [..., *S]; result[x] = t[clamp(x - off)] per spatial axis (affinity_side_loss.py:9-61)."""
    nd = len(off)
    for ax, o in enumerate(off):
        dim = t.dim() - nd + ax
        n = t.shape[dim]
        idx = (torch.arange(n) - o).clamp(0, n - 1)
        t = t.index_select(dim, idx)
    return t


def affinity_side_loss(emb, lbl, offsets, delta):
    """emb [1,E,*S], lbl [1,1,*S]; offsets: list of per-axis ints (already drawn)."""
    inv = [[-o for o in off] for off in offsets]
    sh = torch.stack([shift_replicate(emb, o) for o in inv], dim=1)           # [1,K,E,*S]
    affs = (2 * delta - torch.norm(emb.unsqueeze(1) - sh, dim=2)) / (2 * delta)
    affs = 1.0 - torch.clamp(affs, min=0) ** 2
    seg = lbl.float()
    shs = torch.cat([shift_replicate(seg, o) for o in inv], dim=1)
    taffs = 1.0 - (seg - shs).eq(0.0).float()
    return dice_score(affs, taffs, invert=True)


def draw_offsets(offset_ranges, n_samples):
    return [[int(np.random.randint(r[0], r[1])) for r in offset_ranges] for _ in range(n_samples)]


def contrastive_forward(emb, target, delta_var, delta_dist, alpha=1.0, beta=1.0, gamma=0.001,
                        unlabeled_push_weight=1.0, instance_term_weight=1.0, aux_loss="dice", pmaps_threshold=0.9,
                        offset_ranges=((-18, 18), (-18, 18)), n_samples=9, dice_weight=1.0, aff_weight=1.0,
                        offsets=None):
    """ExtendedContrastiveLoss.forward; emb [N,E,*S], target [N,1,*S] int64.  Returns a [1] tensor."""
    n_b = emb.shape[0]
    ts = two_sigma(delta_var, pmaps_threshold)
    loss = 0.0
    for b in range(n_b):
        e, t = emb[b], target[b, 0]
        contains_bg = bool((t == 0).any())
        ignore_zero = unlabeled_push_weight > 0 and contains_bg
        ids, sizes = torch.unique(t, return_counts=True)
        c = ids.shape[0]
        means = cluster_means(e, t, c)
        var = variance_term(means, e, t, sizes, delta_var, ignore_zero)
        push = unlabeled_push(means, e, t, delta_dist) if ignore_zero else 0.0
        aff = None
        if aux_loss in ("affinity", "dice_aff"):
            offs = offsets[b] if offsets is not None else draw_offsets(offset_ranges, n_samples)
            aff = affinity_side_loss(e[None], t[None, None], offs, delta_dist)
        dice = instance_dice_term(e, means, t, ts) if aux_loss in ("dice", "dice_aff") else None
        if aux_loss == "dice":
            inst = dice
        elif aux_loss == "affinity":
            inst = aff
        else:
            inst = dice_weight * dice + aff_weight * aff
        dist = distance_term(means, delta_dist, ignore_zero)
        reg = regularizer_term(means)
        loss = alpha * var + beta * dist + gamma * reg + instance_term_weight * inst + unlabeled_push_weight * push
        loss = loss + loss
    return loss / n_b


def kth_unlabeled(mask, k):
    """Index tuple of the k-th non-zero voxel of mask in row-major order (torch.nonzero order)."""
    flat = torch.nonzero(mask.flatten())[k, 0].item()
    return np.unravel_index(flat, tuple(mask.shape))


def consistency_term(e_q, e_k, mask, ts, max_anchors, volume_threshold, anchors=None):
    """SPOCOLoss.emb_consistency for one sample; anchors = list of flat ranks into nonzero(mask)."""
    q_l, k_l = [], []
    n_mask = int(mask.sum())
    for a in range(max_anchors):
        if n_mask < volume_threshold * mask.numel():
            break
        ind = anchors[a] if anchors is not None else int(np.random.randint(n_mask))
        pos = kth_unlabeled(mask, ind)
        sl = (slice(None),) + tuple(int(p) for p in pos)
        q_l.append(pmap(e_q.movedim(0, -1), e_q[sl], ts))
        k_l.append(pmap(e_k.movedim(0, -1), e_k[sl], ts))
    return dice_score(torch.stack(q_l), torch.stack(k_l), invert=True)


def spoco_forward(emb_q, emb_k, target, delta_var, delta_dist, alpha=1.0, beta=1.0, gamma=0.001,
                  unlabeled_push_weight=0.0, instance_term_weight=1.0, consistency_term_weight=1.0, aux_loss="dice",
                  pmaps_threshold=0.9, max_anchors=20, volume_threshold=0.05, anchors=None, offsets=None, **kw):
    """SPOCOLoss.forward((emb_q, emb_k), target)."""
    loss = contrastive_forward(emb_q, target, delta_var, delta_dist, alpha, beta, gamma, unlabeled_push_weight,
                               instance_term_weight, aux_loss, pmaps_threshold, offsets=offsets, **kw)
    ts = two_sigma(delta_var, pmaps_threshold)
    for b in range(emb_q.shape[0]):
        mask = (target[b, 0] == 0).int()
        if mask.sum() < volume_threshold * mask.numel():
            continue
        a = anchors[b] if anchors is not None else None
        loss = loss + consistency_term_weight * consistency_term(emb_q[b], emb_k[b], mask, ts, max_anchors,
                                                                 volume_threshold, a)
    return loss


def contrastive_loss(emb, target, delta_var, delta_dist, alpha=1.0, beta=1.0, gamma=0.001):
    """ContrastiveLoss.forward with the scatter implementation (loss/contrastive.py:121-169): every sample counts."""
    total = 0.0
    for b in range(emb.shape[0]):
        e, t = emb[b], target[b, 0]
        ids, sizes = torch.unique(t, return_counts=True)
        means = cluster_means(e, t, ids.shape[0])
        total = total + alpha * variance_term(means, e, t, sizes, delta_var, False) \
            + beta * distance_term(means, delta_dist, False) + gamma * regularizer_term(means)
    return total / emb.shape[0]
