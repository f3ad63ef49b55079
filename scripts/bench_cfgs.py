"""Step times of the other BASELINE configs (parity-test cases, not bench lines): cfg 1 (UNet2d boundaries), cfg 3
(AnisotropicUNet affinities + masked Dice), cfg 5 (SPOCO, see scripts/bench_spoco.py).  usage: python scripts/bench_cfgs.py [1|3]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper  # noqa: E402
from torch_em_amd.model import AnisotropicUNet, UNet2d  # noqa: E402
from torch_em_amd.optim import FusedAdamW  # noqa: E402
from torch_em_amd.transform.label import AffinityTransform, BatchTargets, BoundaryTransform  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda"
torch.manual_seed(0)
if cfg == 3:
    sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    model = AnisotropicUNet(1, 12, scale_factors=sf, initial_features=32, final_activation="Sigmoid").to(dev)
    x = torch.randn(2, 1, 64, 256, 256, device=dev)
    lbl = torch.randint(0, 200, (2, 1, 8, 16, 16), device=dev).repeat_interleave(8, 2).repeat_interleave(16, 3) \
        .repeat_interleave(16, 4)
    offsets = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [-3, 0, 0], [0, -9, 0], [0, 0, -9],
               [-4, 0, 0], [0, -27, 0], [0, 0, -27]]
    target = BatchTargets(AffinityTransform(offsets=offsets, add_mask=True))
    loss = LossWrapper(DiceLoss(), ApplyAndRemoveMask(masking_method="multiply"))
else:
    model = UNet2d(1, 2).to(dev)
    x = torch.randn(8, 1, 256, 256, device=dev)
    lbl = torch.randint(0, 32, (8, 1, 16, 16), device=dev).repeat_interleave(16, 2).repeat_interleave(16, 3)
    target = BatchTargets(BoundaryTransform(add_binary_target=True, ndim=2))
    loss = DiceLoss()
opt = FusedAdamW(model.parameters(), lr=1e-4)


def step():
    opt.zero_grad()
    y = target(lbl)
    val = loss(model(x), y)
    val.backward()
    opt.step()
    return val


for _ in range(3):
    v = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    v = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"cfg {cfg}: {ms:.2f} ms/step  {x.numel() / ms * 1e3:.3e} voxels/s  (target transform on device, loss {float(v):.4f})")

if os.environ.get("TEM_TABLE"):
    from torch_em_amd import ops
    ops.PROFILER = []
    step()
    torch.cuda.synchronize()
    tab = {}
    for (tag, shape), flops, e0, e1 in ops.PROFILER:
        d = tab.setdefault((tag, shape), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += flops
    for (tag, shape), d in sorted(tab.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{tag + '  ' + shape:80s} {d[0]:3d} {d[1] / d[0]:8.4f} ms {d[2] / d[1] / 1e9:8.1f} TF")
    print("conv total", sum(d[1] for d in tab.values()))
