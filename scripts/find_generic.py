"""Which layers of a configuration fall back to the generic VALU kernels?  Prints the slowest conv launches of one
training step (live HIP events).  usage: python scripts/find_generic.py [cfg3|spoco|rgb|side|feat64]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import ops
from torch_em_amd.loss import DiceLoss
from torch_em_amd.model import AnisotropicUNet, UNet3d
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dev = "cuda"
torch.manual_seed(0)
if which == "cfg3":
    sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    model = AnisotropicUNet(1, 12, scale_factors=sf, initial_features=32, final_activation="Sigmoid").to(dev)
    x = torch.randn(1, 1, 32, 128, 128, device=dev)
elif which == "spoco":
    model = UNet3d(1, 8, initial_features=32, depth=4).to(dev)
    x = torch.randn(1, 1, 64, 128, 128, device=dev)
elif which == "rgb":
    model = UNet3d(3, 2, initial_features=32, depth=4).to(dev)
    x = torch.randn(1, 3, 64, 128, 128, device=dev)
elif which == "side":
    model = UNet3d(1, 2, initial_features=32, depth=4, return_side_outputs=True).to(dev)
    x = torch.randn(1, 1, 64, 128, 128, device=dev)
else:
    model = UNet3d(1, 2, initial_features=64, depth=3).to(dev)
    x = torch.randn(1, 1, 64, 128, 128, device=dev)
def step():
    model.zero_grad()
    out = model(x)
    out = out if isinstance(out, list) else [out]
    sum(o.square().mean() for o in out).backward()
step()
ops.PROFILER = []
step()
torch.cuda.synchronize()
rows = sorted(((e0.elapsed_time(e1), tag, shape, fl) for (tag, shape), fl, e0, e1 in ops.PROFILER), reverse=True)
tot = sum(r[0] for r in rows)
print(which, "conv ms/step", round(tot, 2))
for ms, tag, shape, fl in rows[:8]:
    print(f"  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF  {tag}  {shape}")
