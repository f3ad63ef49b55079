"""How long does the HOST need to enqueue one training step (cfg 2)?  If it is close to the GPU time of a step the GPU
starves on a slower host.  usage: python scripts/host_overhead.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd.loss import DiceLoss
from torch_em_amd.model import UNet3d
from torch_em_amd.optim import FusedAdamW
torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = UNet3d(1, 2, initial_features=32, depth=4).cuda()
opt = FusedAdamW(net.parameters(), lr=1e-3)
loss_fn = DiceLoss()
x = torch.randn(2, 1, S, S, S, device="cuda")
y = (torch.rand(2, 2, S, S, S, device="cuda") > 0.5).float()
def step():
    opt.zero_grad()
    loss = loss_fn(net(x), y)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"size {S}: host enqueue {(t1 - t0) / 10 * 1e3:.2f} ms/step, total {(t2 - t0) / 10 * 1e3:.2f} ms/step")
