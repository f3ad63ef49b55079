"""Where does the gradient error of the depth-4 benchmark network come from?  ("show the flip")

For a seed of scripts/depth4_error_survey.py: the float64 oracle against this library's forward pass (default arithmetic
`split16`, and the exact-fp32 build `fp32`), layer by layer in forward order:

  * every ReLU: entries whose MASK differs (float64 pre-activation > 0 vs stored activation > 0) and their margin --
    |float64 pre-activation| / rms of that layer's pre-activations;
  * every max-pool: windows whose ARG-MAX differs and their margin -- (largest - second largest float64 candidate) / rms.

A layer's disagreements include the consequences of flips further up (a flipped entry perturbs everything behind it), so the
decisive experiment comes last: the float64 network is re-run with every ReLU mask and every pooling arg-max FORCED to the
library's decisions.  If the gradient of that network agrees with the library's gradient to rounding level, the whole
distance to the float64 gradient is those decisions -- near-ties that an arithmetic with a different rounding resolves the
other way -- and not an error of the kernels.

    python scripts/flip_census.py [seed ...] > profiles/r05_flip_census.txt          (GPU box)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref  # noqa: E402
from torch_em_amd.loss import DiceLoss  # noqa: E402
from torch_em_amd.model import UNet3d, engine  # noqa: E402

DEPTH = 4


def make(seed):
    torch.manual_seed(seed)
    model = UNet3d(1, 2, depth=DEPTH, initial_features=32)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(1, 1, 64, 64, 64, generator=g)
    y = (torch.rand(1, 2, 64, 64, 64, generator=g) > 0.5).float()
    return model, x, y


def oracle_grads(sd, x, y, force=None):
    sd64 = {k: v.double().clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    with unet_ref.DecisionTap(force) as tap:
        pred = unet_ref.unet_forward(sd64, x.double(), [2] * DEPTH, norm="InstanceNorm")
    from oracle import loss_ref
    loss = loss_ref.dice_loss(pred, y.double())
    loss.backward()
    return tap, float(loss.detach()), {k: v.grad.detach().numpy() for k, v in sd64.items() if v.grad is not None}


def hip_run(model, x, y, mode):
    """-> (ReLU outputs in the oracle's call order as NCDHW CPU tensors, pooling inputs, loss, gradients)"""
    model.zero_grad(set_to_none=True)
    with engine.precision_scope(mode):
        _, st = engine._forward_impl(model, x.cuda(), keep=True)
        blocks = [lv["bs"] for lv in st["levels"]] + [st["base"]] + [d["bs"] for d in st["dec"]]
        relus = []
        for bs in blocks:
            relus += [bs["a1"].permute(0, 4, 1, 2, 3).float().cpu(), bs["out"].permute(0, 4, 1, 2, 3).float().cpu()]
        pools = [lv["skip"].permute(0, 4, 1, 2, 3).float().cpu() for lv in st["levels"]]
        del st
        pred = model(x.cuda())
        loss = DiceLoss()(pred, y.cuda())
        loss.backward()
    return relus, pools, float(loss.detach()), {k: p.grad.double().cpu().numpy() for k, p in model.named_parameters()}


def l2(a, b, keys):
    cat = lambda d: np.concatenate([d[k].ravel() for k in keys])  # noqa: E731
    return float(np.linalg.norm(cat(a) - cat(b)) / np.linalg.norm(cat(b)))


def names():
    out = []
    for l in range(DEPTH):
        out += [f"enc{l}.conv1 ({64 >> l}^3)", f"enc{l}.conv2 ({64 >> l}^3)"]
    out += [f"base.conv1 ({64 >> DEPTH}^3)", f"base.conv2 ({64 >> DEPTH}^3)"]
    for i in range(DEPTH):
        r = 64 >> (DEPTH - 1 - i)
        out += [f"dec{i}.conv1 ({r}^3)", f"dec{i}.conv2 ({r}^3)"]
    return out


def main():
    seeds = [int(s) for s in sys.argv[1:]] or [0, 5]
    for seed in seeds:
        model, x, y = make(seed)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        tap, loss64, g64 = oracle_grads(sd, x, y)
        keys = [k for k in g64 if np.abs(g64[k]).max() > 1e-4 * max(np.abs(v).max() for v in g64.values())]
        model.to("cuda")
        print(f"==== seed {seed}: float64 oracle loss {loss64:.10f}")
        for mode in ("split16", "fp32"):
            relus, pools, loss_h, gh = hip_run(model, x, y, mode)
            print(f"\n-- arithmetic {mode}: loss {loss_h:.10f}, global gradient L2 vs float64 {l2(gh, g64, keys):.3e}")
            print("   layer                    entries   mask differs   smallest / median / largest margin of the differing entries")
            masks, first = [], None
            for name, pre, act in zip(names(), tap.pre, relus):
                m64, mh = pre > 0, act > 0
                masks.append(mh)
                bad = m64 != mh
                nb = int(bad.sum())
                rms = float(pre.pow(2).mean().sqrt())
                if nb:
                    mar = (pre[bad].abs() / rms).numpy()
                    first = first or name
                    print(f"   {name:24s} {pre.numel():9d}   {nb:8d}       {mar.min():.1e} / {np.median(mar):.1e} / {mar.max():.1e}")
                else:
                    print(f"   {name:24s} {pre.numel():9d}   {nb:8d}")
            idxs = []
            for l, (p64, ph) in enumerate(zip(tap.pool_in, pools)):
                _, i64 = F.max_pool3d_with_indices(p64, 2)
                _, ih = F.max_pool3d_with_indices(ph.double(), 2)
                idxs.append(ih)
                bad = i64 != ih
                nb = int(bad.sum())
                line = f"   pool{l} ({64 >> l}^3 -> {32 >> l}^3)      {i64.numel():9d}   {nb:8d}"
                if nb:
                    w = p64.unfold(2, 2, 2).unfold(3, 2, 2).unfold(4, 2, 2).reshape(*i64.shape, 8)
                    top = w.topk(2, dim=-1).values
                    gap = ((top[..., 0] - top[..., 1])[bad] / float(p64.pow(2).mean().sqrt())).numpy()
                    line += f"       {gap.min():.1e} / {np.median(gap):.1e} / {gap.max():.1e}"
                print(line)
            print(f"   first layer with a differing decision: {first}")
            _, loss_f, gf = oracle_grads(sd, x, y, force={"relu": masks, "pool": idxs})
            print(f"   float64 network with the library's ReLU masks and pooling arg-maxes FORCED: loss {loss_f:.10f}")
            print(f"     gradient of the library vs that network:   global L2 {l2(gh, gf, keys):.3e}   "
                  f"worst tensor {max(l2(gh, gf, [k]) for k in keys):.3e}")
            print(f"     that network vs the free float64 network:  global L2 {l2(gf, g64, keys):.3e}")
        model.cpu()


if __name__ == "__main__":
    main()
