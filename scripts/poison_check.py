"""Does any kernel read memory it (or a producer) did not write?  Every torch.empty / empty_like allocation is filled
with NaN (or with a large finite pattern) before use; results must stay finite and bit-identical to the unpoisoned run.
usage: python scripts/poison_check.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
from torch_em_amd.model import AnisotropicUNet, UNet3d, UNet2d

_empty, _empty_like = torch.empty, torch.empty_like
FILL = [None]
def empty(*a, **k):
    t = _empty(*a, **k)
    if FILL[0] is not None and t.is_cuda and t.is_floating_point():
        t.fill_(FILL[0])
    return t
def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if FILL[0] is not None and t.is_cuda and t.is_floating_point():
        t.fill_(FILL[0])
    return t
torch.empty, torch.empty_like = empty, empty_like

def case(name, model, x, y, loss_fn):
    outs = []
    for fill in (None, float("nan"), 3.0e38, None):
        FILL[0] = fill
        model.zero_grad()
        pred = model(x)
        loss = loss_fn(pred, y)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((float(loss), pred.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    FILL[0] = None
    ref = outs[0]
    for tag, o in zip(("nan", "3e38", "plain2"), outs[1:]):
        ok = o[0] == ref[0] and torch.equal(o[1], ref[1]) and all(torch.equal(a, b) for a, b in zip(o[2], ref[2]))
        bad = [k for (k, _), a, b in zip(model.named_parameters(), o[2], ref[2]) if not torch.equal(a, b)]
        print(f"{name:28s} fill={tag:6s} loss {o[0]:.6f} identical={ok} {bad[:4]}")

dev = "cuda"
torch.manual_seed(0)
g = torch.Generator().manual_seed(0)
sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
m = AnisotropicUNet(1, 12, scale_factors=sf, initial_features=32, final_activation="Sigmoid").to(dev)
x = torch.randn(2, 1, 32, 128, 128, generator=g).to(dev)
y = (torch.rand(2, 24, 32, 128, 128, generator=g) > 0.5).float().to(dev)
case("cfg3 aniso 2x32x128x128", m, x, y, LossWrapper(DiceLoss(), ApplyAndRemoveMask(masking_method="multiply")))
m = UNet3d(1, 2, initial_features=32, depth=4).to(dev)
x = torch.randn(2, 1, 64, 64, 64, generator=g).to(dev)
y = (torch.rand(2, 2, 64, 64, 64, generator=g) > 0.5).float().to(dev)
case("cfg2 unet3d 2x64^3", m, x, y, DiceLoss())
m = UNet3d(1, 2, initial_features=32, depth=3, norm="GroupNorm").to(dev)
x = torch.randn(1, 1, 40, 48, 56, generator=g).to(dev)
y = (torch.rand(1, 2, 40, 48, 56, generator=g) > 0.5).float().to(dev)
case("groupnorm ragged 40x48x56", m, x, y, DiceLoss())
m = UNet2d(1, 2).to(dev)
x = torch.randn(4, 1, 256, 256, generator=g).to(dev)
y = (torch.rand(4, 2, 256, 256, generator=g) > 0.5).float().to(dev)
case("cfg1 unet2d 4x256^2", m, x, y, DiceLoss())
