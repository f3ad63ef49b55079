"""cProfile of the HOST side of one training step (32^3: the GPU is idle most of the time, so wall = host work).
usage: python scripts/host_profile.py [n_steps]"""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd.loss import DiceLoss
from torch_em_amd.model import UNet3d
from torch_em_amd.optim import FusedAdamW
torch.manual_seed(0)
S = 32
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pr = cProfile.Profile()
pr.enable()
for _ in range(n): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
