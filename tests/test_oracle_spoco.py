"""Pins oracle/spoco_ref.py against golden vectors produced by the reference itself (tests/golden/gen_golden_spoco.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import spoco_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")

CASES = {
    "g6a_spoco_3d": ("spoco", dict(delta_var=0.75, delta_dist=2.0)),
    "g6b_spoco_2d": ("spoco", dict(delta_var=0.5, delta_dist=1.5, max_anchors=7)),
    "g6c_extcontrastive_2d": ("ext", dict(delta_var=0.5, delta_dist=2.0)),
    "g6d_extcontrastive_3d": ("ext", dict(delta_var=0.75, delta_dist=2.0, unlabeled_push_weight=0.5)),
    "g6e_spoco_affinity_2d": ("spoco", dict(delta_var=0.75, delta_dist=2.0, aux_loss="affinity",
                                            offset_ranges=[(-6, 6), (-6, 6)], n_samples=5)),
    "g6f_spoco_diceaff_3d": ("spoco", dict(delta_var=0.75, delta_dist=2.0, aux_loss="dice_aff", aff_weight=0.5,
                                           offset_ranges=[(-2, 3), (-5, 5), (-5, 5)], n_samples=4)),
}


def run_oracle(kind, kw, g, dtype=torch.float32):
    q = torch.from_numpy(g["emb_q"]).to(dtype).requires_grad_(True)
    k = torch.from_numpy(g["emb_k"]).to(dtype)
    t = torch.from_numpy(g["target"])
    np.random.seed(int(g["np_seed"]))
    if kind == "spoco":
        val = spoco_ref.spoco_forward(q, k, t, **kw)
    else:
        val = spoco_ref.contrastive_forward(q, t, **kw)
    val.sum().backward()
    return val.detach(), q.grad


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference(name):
    kind, kw = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    val, grad = run_oracle(kind, kw, g)
    assert val.shape == (1,)  # the reference returns a 1-element tensor (variance term keeps its batch axis)
    np.testing.assert_allclose(val.numpy(), g["loss"], rtol=2e-6, atol=1e-6)
    if g["grad_q"].size:
        ref = g["grad_q"]
        err = np.linalg.norm(grad.numpy() - ref) / np.linalg.norm(ref)
        assert err < 2e-6, err
        assert np.array_equal(grad.numpy() == 0, ref == 0) or err < 1e-7  # only the last sample gets contrastive gradient


def test_draw_order_matches_reference():
    """Injecting the logged draws reproduces the seeded run => the oracle consumes the global RNG like the reference."""
    g = np.load(os.path.join(GOLD, "g6f_spoco_diceaff_3d.npz"))
    draws = g["draws"].tolist()
    kw = CASES["g6f_spoco_diceaff_3d"][1]
    n, nd, ns = g["emb_q"].shape[0], 3, kw["n_samples"]
    offsets = [np.asarray(draws[b * ns * nd:(b + 1) * ns * nd]).reshape(ns, nd).tolist() for b in range(n)]
    rest = draws[n * ns * nd:]
    anchors = [rest[b * 20:(b + 1) * 20] for b in range(n)]
    q = torch.from_numpy(g["emb_q"])
    val = spoco_ref.spoco_forward(q, torch.from_numpy(g["emb_k"]), torch.from_numpy(g["target"]), anchors=anchors,
                                  offsets=offsets, **kw)
    np.testing.assert_allclose(val.numpy(), g["loss"], rtol=2e-6)


@pytest.mark.parametrize("name", ["g6g_contrastive_2d", "g6h_contrastive_3d"])
def test_contrastive_oracle_matches_reference(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    emb = torch.from_numpy(g["emb"]).requires_grad_(True)
    val = spoco_ref.contrastive_loss(emb, torch.from_numpy(g["target"]), 0.5, 1.5, 1.0, 0.7, 0.01)
    val.sum().backward()
    np.testing.assert_allclose(val.detach().numpy().reshape(-1), g["loss"], rtol=2e-6)
    assert np.linalg.norm(emb.grad.numpy() - g["grad"]) / np.linalg.norm(g["grad"]) < 2e-6
