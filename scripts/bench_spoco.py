"""Times the SPOCO loss (and optionally the whole SPOCO training step) at BASELINE cfg 5's per-GPU size:
x [1,1,96,192,192], UNet3d(1->8), ~30 instances, >=5 % unlabeled.   usage: python scripts/bench_spoco.py [loss|step] [iters]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd.loss import SPOCOLoss  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "loss"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
D, H, W, E = 96, 192, 192, 8
dev = "cuda"
torch.manual_seed(0)
np.random.seed(0)
# ~30 blocky instances + background
small = torch.randint(0, 34, (4, 6, 6))
small[small > 30] = 0  # ~10 % background
lbl = small.repeat_interleave(24, 0).repeat_interleave(32, 1).repeat_interleave(32, 2)[None, None].contiguous().to(dev)
ids = torch.unique(lbl)
remap = torch.zeros(int(ids.max()) + 1, dtype=torch.int64, device=dev)
remap[ids] = torch.arange(len(ids), device=dev)
lbl = remap[lbl]
print("instances", int(lbl.max()) + 1, "unlabeled frac", float((lbl == 0).float().mean()))
loss = SPOCOLoss(delta_var=0.75, delta_dist=2.0, aux_loss="dice")


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


if kind == "loss":
    q = torch.randn(1, E, D, H, W, device=dev, requires_grad=True)
    k = (q.detach() + 0.1 * torch.randn_like(q))

    def run():
        q.grad = None
        loss((q, k), lbl).sum().backward()
    print(f"SPOCOLoss fwd+bwd {timed(run):.3f} ms")
else:
    from torch_em_amd.model import UNet3d
    from torch_em_amd.optim import FusedAdamW
    from torch_em_amd.trainer import SPOCOTrainer
    model = UNet3d(1, E, initial_features=32, depth=4).to(dev)
    x = torch.randn(1, 1, D, H, W, device=dev)
    ds = torch.utils.data.TensorDataset(x.cpu(), lbl[0].cpu()[None])
    dl = torch.utils.data.DataLoader(ds, batch_size=1)
    tr = SPOCOTrainer(model=model, momentum=0.999, name="b", train_loader=dl, val_loader=dl, loss=loss,
                      optimizer=FusedAdamW(model.parameters(), lr=1e-4), metric=loss, device=dev, save_root="/tmp/spoco_b",
                      logger=None, mixed_precision=False)   # the fp32-class step (the bare flag means fp16 autocast since round 6)
    tr._initialize(1, None)

    def run():
        tr._step(x, tr.loss, lbl)
    ms = timed(run)
    print(f"SPOCO step (student fwd+bwd, teacher fwd, loss, AdamW, EMA) {ms:.2f} ms  {D * H * W / ms * 1e3:.3e} voxels/s")
