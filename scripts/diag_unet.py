"""Diagnostic: per-parameter gradient errors of the engine vs a golden fixture."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_em_amd import ops  # noqa: E402
from torch_em_amd.loss import DiceLoss  # noqa: E402
from torch_em_amd.model import UNet3d  # noqa: E402

for norm in ("InstanceNorm", "None"):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"g1_unet3d_{norm}.npz")))
    model = UNet3d(1, 2, depth=2, initial_features=4, norm=None if norm == "None" else norm)
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")})
    model.cuda()
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    pred = model(x)
    loss = DiceLoss()(pred, y)
    loss.backward()
    print(norm, "loss", float(loss), float(g["loss"]))
    for k, p in model.named_parameters():
        ref = g[f"grad.{k}"]
        err = np.abs(p.grad.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-5)
        print(f"  {k:45s} {err:.3e}  |ref|={np.abs(ref).max():.3e}")

# in-place vs out-of-place norm backward
torch.manual_seed(0)
x5 = torch.randn(2, 8, 8, 8, 32, device="cuda")
g5 = torch.randn(2, 8, 8, 8, 32, device="cuda")
mean, rstd, scale, shift = ops.norm_stats(x5, 32)
out = torch.empty_like(g5)
ops.norm_bwd(g5, x5, 32, None, mean, rstd, True, out)
g5b = g5.clone()
ops.norm_bwd(g5b, x5, 32, None, mean, rstd, True, g5b)
print("inplace diff", float((out - g5b).abs().max()))

# MFMA-size case vs oracle
from oracle import unet_ref  # noqa: E402
torch.manual_seed(0)
model = UNet3d(1, 2, depth=2, initial_features=32)
gen = torch.Generator().manual_seed(4)
x = torch.randn(2, 1, 16, 24, 32, generator=gen)
y = (torch.rand(2, 2, 16, 24, 32, generator=gen) > 0.5).float()
pred_o, loss_o, grads_o = unet_ref.unet_loss_and_grads({k: v.detach().clone() for k, v in model.state_dict().items()},
                                                       x, y, [2, 2])
model.cuda()
pred = model(x.cuda())
loss = DiceLoss()(pred, y.cuda())
loss.backward()
print("mfma case loss", float(loss), float(loss_o))
for k, p in model.named_parameters():
    ref = grads_o[k].numpy()
    d = np.abs(p.grad.cpu().numpy() - ref).max()
    print(f"  {k:45s} diff {d:.3e}  |ref|={np.abs(ref).max():.3e} rel {d / np.abs(ref).max():.2e}")
