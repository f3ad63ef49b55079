import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_ref
from torch_em_amd.model import UNet3d
torch.manual_seed(0)
sd = {k: v.detach().clone() for k, v in UNet3d(1, 2, initial_features=32, depth=4).state_dict().items()}
x = torch.randn(1, 1, 32, 32, 32); y = (torch.rand(1, 2, 32, 32, 32) > 0.5).float()
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    unet_ref.unet_loss_and_grads(sd, x, y, [2, 2, 2, 2])
    t0 = time.perf_counter(); unet_ref.unet_loss_and_grads(sd, x, y, [2, 2, 2, 2]); dt = time.perf_counter() - t0
    print(th, f"{dt:.3f}s", f"{32**3/dt:.3e} vox/s", flush=True)
