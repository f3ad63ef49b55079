"""GPU parity of the SPOCO / contrastive / affinity-side losses (csrc/spoco.hip through the C-ABI) against
(a) the reference's golden vectors and (b) the CPU oracle in float64 on larger seeded inputs.

Tolerances (floating point; stated per check): loss values rtol 2e-5 (fp32 sums in a different order than torch's),
gradients relative L2 error <= 2e-4 vs the float64 oracle / 1e-4 vs the fp32 golden reference gradients.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mods():
    from oracle import spoco_ref
    from torch_em_amd.loss import affinity_side_loss, spoco_loss
    return spoco_loss, affinity_side_loss, spoco_ref


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


from test_oracle_spoco import CASES  # noqa: E402  (same constructor arguments as the golden generator)


def _build(kind, kw):
    sp = _mods()[0]
    return sp.SPOCOLoss(**kw) if kind == "spoco" else sp.ExtendedContrastiveLoss(**kw)


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden(name):
    kind, kw = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    loss = _build(kind, kw)
    q = torch.from_numpy(g["emb_q"]).cuda().requires_grad_(True)
    k = torch.from_numpy(g["emb_k"]).cuda()
    t = torch.from_numpy(g["target"]).cuda()
    np.random.seed(int(g["np_seed"]))
    val = loss((q, k), t) if kind == "spoco" else loss(q, t)
    assert val.shape == (1,)
    val.sum().backward()
    np.testing.assert_allclose(val.detach().cpu().numpy(), g["loss"], rtol=2e-5)
    if g["grad_q"].size:
        ref = g["grad_q"]
    else:  # the reference cannot backpropagate this configuration (see gen_golden_spoco.py): use the oracle
        from test_oracle_spoco import run_oracle
        ref = run_oracle(kind, kw, g, torch.float64)[1].numpy()
    err = _rel(q.grad.cpu().numpy(), ref)
    assert err < 1e-4, err
    n = g["emb_q"].shape[0]
    if kind == "ext" and n > 1:
        assert float(q.grad[: n - 1].abs().max()) == 0.0  # only the last sample receives contrastive gradient


def _labels(shape, n_ids, seed, block=4):
    g = torch.Generator().manual_seed(seed)
    small = [(s + block - 1) // block for s in shape]
    lbl = torch.randint(0, n_ids, small, generator=g)
    for ax in range(len(shape)):
        lbl = lbl.repeat_interleave(block, ax)
    lbl = lbl[tuple(slice(0, s) for s in shape)].contiguous()
    ids = torch.unique(lbl)
    remap = torch.zeros(int(ids.max()) + 1, dtype=torch.int64)
    remap[ids] = torch.arange(len(ids))
    return remap[lbl]


@pytest.mark.parametrize("cfg", [
    dict(shape=(2, 8, 16, 40, 48), n_ids=14, kw=dict(delta_var=0.75, delta_dist=2.0)),
    dict(shape=(1, 16, 12, 32, 32), n_ids=9, kw=dict(delta_var=0.5, delta_dist=1.5, unlabeled_push_weight=0.7,
                                                     max_anchors=12)),
    dict(shape=(2, 5, 70, 90), n_ids=20, kw=dict(delta_var=0.75, delta_dist=2.0, aux_loss="dice_aff",
                                                 offset_ranges=[(-9, 9), (-9, 9)], n_samples=6)),
    dict(shape=(1, 20, 8, 24, 24), n_ids=300, kw=dict(delta_var=0.75, delta_dist=2.0, aux_loss="affinity",
                                                      offset_ranges=[(-3, 3), (-6, 6), (-6, 6)], n_samples=7), block=2),
])
def test_spoco_vs_float64_oracle(cfg):
    sp, _, ref = _mods()
    shape = cfg["shape"]
    g = torch.Generator().manual_seed(5)
    q = torch.randn(shape, generator=g) * 1.2
    k = q + 0.25 * torch.randn(shape, generator=g)
    t = torch.stack([_labels(shape[2:], cfg["n_ids"], 100 + b, cfg.get("block", 4)) for b in range(shape[0])])[:, None]
    qd = q.double().requires_grad_(True)
    np.random.seed(3)
    want = ref.spoco_forward(qd, k.double(), t, **cfg["kw"])
    want.sum().backward()
    qg = q.cuda().requires_grad_(True)
    np.random.seed(3)
    got = sp.SPOCOLoss(**cfg["kw"])((qg, k.cuda()), t.cuda())
    got.sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5)
    err = _rel(qg.grad.cpu().numpy(), qd.grad.numpy())
    assert err < 2e-4, err


def test_cluster_means_and_determinism():
    sp, _, ref = _mods()
    g = torch.Generator().manual_seed(0)
    e = torch.randn(8, 24, 64, 64, generator=g)
    lbl = _labels((24, 64, 64), 40, 1)
    c = int(lbl.max()) + 1
    want = ref.cluster_means(e.double(), lbl, c).numpy()
    a = sp.compute_cluster_means(e.cuda(), lbl.cuda(), c)
    b = sp.compute_cluster_means(e.cuda(), lbl.cuda(), c)
    assert torch.equal(a, b)  # fixed-point segment sums: bit-reproducible
    np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    with pytest.raises(AssertionError):
        sp.compute_cluster_means(e.cuda(), (lbl + 1).cuda(), c + 1)


def test_kth_unlabeled_matches_nonzero():
    import ctypes
    from torch_em_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    for V in (1000, 1024, 70001):
        lbl = (torch.rand(V, generator=g) > 0.3).long().cuda()
        nchunk = (V + 1023) // 1024
        cc = torch.empty(nchunk, dtype=torch.int32, device="cuda")
        tot = torch.empty(1, dtype=torch.int64, device="cuda")
        st = ops._stream(lbl)
        _lib.check(lib.tem_zero_count(ops._p(lbl), V, ops._p(cc), ops._p(tot), st))
        nz = torch.nonzero(lbl == 0)[:, 0]
        assert int(tot.item()) == nz.numel()
        ranks = torch.tensor([0, 1, nz.numel() // 2, nz.numel() - 1, 7], dtype=torch.int64, device="cuda")
        idx = torch.empty_like(ranks)
        _lib.check(lib.tem_zero_select(ops._p(lbl), V, ops._p(cc), ops._p(ranks), ranks.numel(), ops._p(idx), st))
        assert torch.equal(idx, nz[ranks])


@pytest.mark.parametrize("shape,ranges", [((1, 6, 40, 36), [(-7, 7), (-7, 7)]), ((1, 4, 6, 20, 18), [(-2, 3), (-5, 5), (-5, 5)]),
                                          ((1, 3, 1, 9, 9), [(-3, 3), (-12, 12), (-12, 12)])])
def test_affinity_side_loss(shape, ranges):
    _, aff, ref = _mods()
    g = torch.Generator().manual_seed(4)
    e = torch.randn(shape, generator=g)
    t = _labels(shape[2:], 6, 9, 3)[None, None]
    np.random.seed(8)
    offs = ref.draw_offsets(ranges, 8)
    ed = e.double().requires_grad_(True)
    want = ref.affinity_side_loss(ed, t, offs, 1.5)
    want.backward()
    eg = e.cuda().requires_grad_(True)
    np.random.seed(8)
    got = aff.AffinitySideLoss(ranges, 8, 1.5)(eg, t.cuda())
    got.backward()
    np.testing.assert_allclose(got.item(), want.item(), rtol=2e-5)
    assert _rel(eg.grad.cpu().numpy(), ed.grad.numpy()) < 2e-4


def test_consistency_loss_module():
    sp, _, ref = _mods()
    g = torch.Generator().manual_seed(6)
    q = torch.randn(2, 6, 10, 20, 20, generator=g)
    k = q + 0.2 * torch.randn(q.shape, generator=g)
    loss = sp.SPOCOConsistencyLoss(delta_var=0.75, pmaps_threshold=0.9, max_anchors=9)
    ts = loss.dist_to_mask.two_sigma
    qd = q.double().requires_grad_(True)
    np.random.seed(1)
    want = 0.0
    for b in range(2):
        mask = torch.ones(q.shape[2:])
        want = want + ref.consistency_term(qd[b], k[b].double(), mask, ts, 9, 0.0)
    want.backward()
    qg = q.cuda().requires_grad_(True)
    np.random.seed(1)
    got = loss(qg, k.cuda())
    got.backward()
    np.testing.assert_allclose(got.item(), want.item(), rtol=2e-5)
    assert _rel(qg.grad.cpu().numpy(), qd.grad.numpy()) < 2e-4


@pytest.mark.parametrize("name", ["g6g_contrastive_2d", "g6h_contrastive_3d"])
def test_contrastive_loss_golden(name):
    """ContrastiveLoss (reference loss/contrastive.py; both of its implementations agree on these vectors)."""
    from torch_em_amd.loss import ContrastiveLoss
    g = np.load(os.path.join(GOLD, name + ".npz"))
    emb = torch.from_numpy(g["emb"]).cuda().requires_grad_(True)
    for impl in (None, "scatter", "expand"):
        emb.grad = None
        val = ContrastiveLoss(delta_var=0.5, delta_dist=1.5, alpha=1.0, beta=0.7, gamma=0.01, impl=impl)(
            emb, torch.from_numpy(g["target"]).cuda())
        val.sum().backward()
        np.testing.assert_allclose(val.detach().cpu().numpy().reshape(-1), g["loss"], rtol=2e-5)
        assert _rel(emb.grad.cpu().numpy(), g["grad"]) < 1e-4
    assert float(emb.grad[0].abs().max()) > 0  # every sample contributes (unlike the SPOCO base class)
