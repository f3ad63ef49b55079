"""Mixed-precision ("amp") arithmetic vs the default fp32-class path on BASELINE cfg 2: output / gradient distance and
step time.  usage: python scripts/amp_error.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd.loss import DiceLoss
from torch_em_amd.model import UNet3d, engine
torch.manual_seed(0)
dev = "cuda"
model = UNet3d(1, 2, initial_features=32, depth=4).to(dev)
x = torch.randn(2, 1, 128, 128, 128, device=dev)
y = (torch.rand(2, 2, 128, 128, 128, device=dev) > 0.5).float()
loss_fn = DiceLoss()
def run(mode, scale):
    model.zero_grad()
    with engine.precision_scope(mode):
        out = model(x)
        l = loss_fn(out, y)
        (l * scale).backward()
    return out.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()]).clone() / scale, float(l)
o32, g32, l32 = run("split16", 1.0)
for mode, s in (("amp", 1.0), ("amp", 1024.0), ("amp", 65536.0), ("amp_bf16", 1.0)):
    o16, g16, l16 = run(mode, s)
    print(f"{mode:8s} storage {engine.act_dtype() if False else ('16-bit' if engine._AMP_STORAGE16 else 'fp32')} scale {s:8.0f}: out rel L2 {float((o16 - o32).norm() / o32.norm()):.2e}  loss {l16:.6f} vs {l32:.6f}  "
          f"grad rel L2 {float((g16 - g32).norm() / g32.norm()):.2e}  finite {bool(torch.isfinite(g16).all())}")
for mode in ("split16", "amp"):
    for _ in range(3):
        run(mode, 1024.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        run(mode, 1024.0)
    torch.cuda.synchronize()
    print(mode, f"{(time.perf_counter() - t0) * 100:.2f} ms fwd+bwd")
