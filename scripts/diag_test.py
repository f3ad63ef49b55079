import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_unet as T
from torch_em_amd.loss import DiceLoss
from torch_em_amd.model import UNet3d
torch.manual_seed(0)
model = UNet3d(1, 2, depth=2, initial_features=32, norm="InstanceNorm")
g = torch.Generator().manual_seed(4)
x = torch.randn(2, 1, 16, 24, 32, generator=g)
y = (torch.rand(2, 2, 16, 24, 32, generator=g) > 0.5).float()
case = (model, [2, 2], x, y, "InstanceNorm")
model.to("cuda")
pred = model(x.cuda()); loss = DiceLoss()(pred, y.cuda()); loss.backward()
_, _, g64 = T._oracle_case(*case, dtype=torch.float64)
_, _, g32 = T._oracle_case(*case, dtype=torch.float32)
k = "decoder.blocks.0.block.1.bias"
p = dict(model.named_parameters())[k]
print("hip ", p.grad.cpu().numpy()[:4])
print("g32 ", g32[k].numpy()[:4])
print("g64 ", g64[k].numpy()[:4], g64[k].dtype)
