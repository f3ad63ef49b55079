"""Golden vectors for the SPOCO / contrastive losses, made by IMPORTING THE REFERENCE (build container only).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_spoco.py

torch_scatter is not installed here; the reference only needs `scatter_mean(src, index, dim=-1, dim_size=None)`, which
is provided by a shim module (index_add sum / clamped count -- torch_scatter's documented semantics).  Every
np.random.randint call the reference makes is logged so the drawn anchors / offsets travel with the fixture.
The fixtures are data (inputs, reference outputs, drawn integers); no reference source is copied.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/torch_em"
OUT = os.path.dirname(os.path.abspath(__file__))


def scatter_mean(src, index, dim=-1, dim_size=None):
    dim = dim % src.dim()
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[dim] = n
    out = torch.zeros(shape, dtype=src.dtype).index_add(dim, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add(0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
    view = [1] * src.dim()
    view[dim] = n
    return out / cnt.view(view)


def load_reference():
    shim = types.ModuleType("torch_scatter")
    shim.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = shim
    for name, path in (("torch_em", REF), ("torch_em.loss", REF + "/loss")):
        mod = types.ModuleType(name)
        mod.__path__ = [path]
        sys.modules[name] = mod
    out = {}
    for name, rel in (("torch_em.loss.dice", "loss/dice.py"), ("torch_em.loss.contrastive_impl", "loss/contrastive_impl.py"),
                      ("torch_em.loss.affinity_side_loss", "loss/affinity_side_loss.py"),
                      ("torch_em.loss.spoco_loss", "loss/spoco_loss.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        out[name.split(".")[-1]] = mod
    return out


class DrawLog:
    def __init__(self):
        self.draws = []
        self._orig = np.random.randint

    def __enter__(self):
        def logged(*a, **k):
            v = self._orig(*a, **k)
            self.draws.append(int(v))
            return v
        np.random.randint = logged
        return self

    def __exit__(self, *exc):
        np.random.randint = self._orig


def labels(shape, n_ids, seed, zero_frac=0.3):
    g = torch.Generator().manual_seed(seed)
    lbl = torch.randint(1, n_ids, shape, generator=g)
    # blocky instances so that segment statistics are non-trivial, with a background region
    lbl = lbl[..., ::4, ::4].repeat_interleave(4, -1).repeat_interleave(4, -2)[..., :shape[-2], :shape[-1]].contiguous()
    bg = torch.rand(shape, generator=g) < zero_frac
    lbl[bg] = 0
    for b in range(shape[0]):  # consecutive ids per sample (the reference's scatter requires it)
        ids = torch.unique(lbl[b])
        remap = torch.zeros(int(ids.max()) + 1, dtype=torch.int64)
        remap[ids] = torch.arange(len(ids))
        lbl[b] = remap[lbl[b]]
    return lbl


def case(name, loss, emb_shape, n_ids, seed, with_k):
    g = torch.Generator().manual_seed(seed)
    emb_q = (torch.randn(emb_shape, generator=g) * 1.5).requires_grad_(True)
    emb_k = emb_q.detach() + 0.3 * torch.randn(emb_shape, generator=g)
    tgt = labels((emb_shape[0], 1) + tuple(emb_shape[2:]), n_ids, seed + 1)
    np.random.seed(seed)
    with DrawLog() as log:
        val = loss((emb_q, emb_k), tgt) if with_k else loss(emb_q, tgt)
    try:
        val.sum().backward()
        grad = emb_q.grad.numpy()
    except RuntimeError as err:
        # torch 2.10: the in-place `variance *= mask` (contrastive_impl.py:116) invalidates the norm's saved output, so
        # the reference itself cannot backpropagate when unlabeled_push_weight > 0 and background is present.
        print(name, "reference backward failed:", str(err)[:90])
        grad = np.zeros(0, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), emb_q=emb_q.detach().numpy(), emb_k=emb_k.numpy(),
                        target=tgt.numpy(), loss=val.detach().numpy().reshape(-1), grad_q=grad,
                        draws=np.asarray(log.draws, dtype=np.int64), np_seed=np.int64(seed))
    print(name, float(val.sum()), "draws", len(log.draws), "grad L2", float(np.linalg.norm(grad)))


def main():
    torch.set_num_threads(4)
    sp = load_reference()["spoco_loss"]
    case("g6a_spoco_3d", sp.SPOCOLoss(delta_var=0.75, delta_dist=2.0), (2, 8, 8, 16, 16), 4, 11, True)
    case("g6b_spoco_2d", sp.SPOCOLoss(delta_var=0.5, delta_dist=1.5, max_anchors=7), (3, 6, 24, 20), 6, 12, True)
    case("g6c_extcontrastive_2d", sp.ExtendedContrastiveLoss(delta_var=0.5, delta_dist=2.0), (2, 6, 24, 24), 5, 13, False)
    case("g6d_extcontrastive_3d", sp.ExtendedContrastiveLoss(delta_var=0.75, delta_dist=2.0, unlabeled_push_weight=0.5),
         (1, 4, 6, 12, 12), 4, 14, False)
    case("g6e_spoco_affinity_2d", sp.SPOCOLoss(delta_var=0.75, delta_dist=2.0, aux_loss="affinity",
                                               offset_ranges=[(-6, 6), (-6, 6)], n_samples=5), (2, 4, 32, 32), 5, 15, True)
    case("g6f_spoco_diceaff_3d", sp.SPOCOLoss(delta_var=0.75, delta_dist=2.0, aux_loss="dice_aff", aff_weight=0.5,
                                              offset_ranges=[(-2, 3), (-5, 5), (-5, 5)], n_samples=4),
         (2, 4, 6, 16, 16), 4, 16, True)


def contrastive_cases():
    """ContrastiveLoss (loss/contrastive.py): both reference implementations must agree; store the expand result."""
    spec = importlib.util.spec_from_file_location("torch_em.loss.contrastive", os.path.join(REF, "loss/contrastive.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["torch_em.loss.contrastive"] = mod
    spec.loader.exec_module(mod)
    for name, shape, n_ids, seed in (("g6g_contrastive_2d", (3, 6, 24, 20), 6, 21), ("g6h_contrastive_3d", (2, 8, 8, 16, 16), 5, 22)):
        g = torch.Generator().manual_seed(seed)
        emb = (torch.randn(shape, generator=g) * 1.5).requires_grad_(True)
        tgt = labels((shape[0], 1) + tuple(shape[2:]), n_ids, seed + 1)
        vals = {}
        for impl in ("expand", "scatter"):
            emb.grad = None
            val = mod.ContrastiveLoss(delta_var=0.5, delta_dist=1.5, alpha=1.0, beta=0.7, gamma=0.01, impl=impl)(emb, tgt)
            val.sum().backward()
            vals[impl] = (val.detach().numpy().reshape(-1), emb.grad.numpy().copy())
        assert np.allclose(vals["expand"][0], vals["scatter"][0], rtol=1e-5), (vals["expand"][0], vals["scatter"][0])
        assert np.allclose(vals["expand"][1], vals["scatter"][1], rtol=1e-3, atol=1e-7)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), emb=emb.detach().numpy(), target=tgt.numpy(),
                            loss=vals["scatter"][0], grad=vals["scatter"][1])
        print(name, vals["scatter"][0])


if __name__ == "__main__":
    load_reference()
    contrastive_cases()
    main()
