"""Is the fp32 gradient discrepancy conditioning or a bug?  Compare engine(fp32, GPU) and
oracle(fp32, CPU) against the oracle in float64."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref  # noqa: E402
from torch_em_amd.loss import DiceLoss  # noqa: E402
from torch_em_amd.model import UNet3d  # noqa: E402

torch.manual_seed(0)
model = UNet3d(1, 2, depth=2, initial_features=32)
gen = torch.Generator().manual_seed(4)
x = torch.randn(2, 1, 16, 24, 32, generator=gen)
y = (torch.rand(2, 2, 16, 24, 32, generator=gen) > 0.5).float()
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
_, l32, g32 = unet_ref.unet_loss_and_grads(sd, x, y, [2, 2])
_, l64, g64 = unet_ref.unet_loss_and_grads({k: v.double() for k, v in sd.items()}, x.double(), y.double(), [2, 2])
model.cuda()
loss = DiceLoss()(model(x.cuda()), y.cuda())
loss.backward()
print("loss hip/cpu32/cpu64", float(loss), float(l32), float(l64))
print(f"{'param':45s} {'hip-vs-64':>10s} {'cpu32-vs-64':>11s} {'hip-vs-cpu32':>12s}")
for k, p in model.named_parameters():
    r = g64[k].numpy()
    s = np.abs(r).max()
    a = np.abs(p.grad.cpu().numpy() - r).max() / s
    b = np.abs(g32[k].numpy() - r).max() / s
    c = np.abs(p.grad.cpu().numpy() - g32[k].numpy()).max() / s
    print(f"{k:45s} {a:10.2e} {b:11.2e} {c:12.2e}")
