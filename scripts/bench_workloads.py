"""Step times of the BASELINE configs that are parity-test cases rather than bench lines (cfg 1, 3, 5), as functions:
bench.py reports them as extra keys next to the cfg-2 headline; `python scripts/bench_workloads.py [1|3|5]` prints one.
Every workload computes its label targets on the device inside the step (the loader ships int labels)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _time(step, steps, warmup):
    for _ in range(warmup):
        v = step().detach()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        v = step().detach()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, v


def cfg1(steps=5, warmup=3, dev="cuda"):
    """UNet2d(1->2) + BoundaryTransform on 8x1x256x256."""
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet2d
    from torch_em_amd.optim import FusedAdamW
    from torch_em_amd.transform.label import BatchTargets, BoundaryTransform
    torch.manual_seed(0)
    model = UNet2d(1, 2).to(dev)
    x = torch.randn(8, 1, 256, 256, device=dev)
    lbl = torch.randint(0, 32, (8, 1, 16, 16), device=dev).repeat_interleave(16, 2).repeat_interleave(16, 3)
    target, loss, opt = BatchTargets(BoundaryTransform(add_binary_target=True, ndim=2)), DiceLoss(), FusedAdamW(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        val = loss(model(x), target(lbl))
        val.backward()
        opt.step()
        return val
    ms, v = _time(step, steps, warmup)
    out = {"ms_per_step": ms, "voxels_per_s": x.numel() / ms * 1e3, "loss": float(v)}
    # the same step replayed as one HIP graph (torch_em_amd/graph.py): this small 2-D workload is host-bound when its ~150
    # launches are enqueued one by one; the label targets stay outside the graph (they are the input pre-pass)
    from torch_em_amd.graph import GraphedTrainStep
    y0 = target(lbl)
    gstep = GraphedTrainStep(model, loss, opt, x, y0)
    gms, _ = _time(lambda: gstep(x, target(lbl))[1], steps, warmup)
    out["hip_graph_ms_per_step"] = gms
    return out


def cfg3(steps=3, warmup=2, dev="cuda"):
    """AnisotropicUNet(1->12) + 12-offset affinities + masked Dice on 2x1x64x256x256."""
    from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
    from torch_em_amd.model import AnisotropicUNet
    from torch_em_amd.optim import FusedAdamW
    from torch_em_amd.transform.label import AffinityTransform, BatchTargets
    torch.manual_seed(0)
    sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    model = AnisotropicUNet(1, 12, scale_factors=sf, initial_features=32, final_activation="Sigmoid").to(dev)
    x = torch.randn(2, 1, 64, 256, 256, device=dev)
    lbl = torch.randint(0, 200, (2, 1, 8, 16, 16), device=dev).repeat_interleave(8, 2).repeat_interleave(16, 3) \
        .repeat_interleave(16, 4)
    offsets = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [-3, 0, 0], [0, -9, 0], [0, 0, -9],
               [-4, 0, 0], [0, -27, 0], [0, 0, -27]]
    target = BatchTargets(AffinityTransform(offsets=offsets, add_mask=True))
    loss = LossWrapper(DiceLoss(), ApplyAndRemoveMask(masking_method="multiply"))
    opt = FusedAdamW(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        val = loss(model(x), target(lbl))
        val.backward()
        opt.step()
        return val
    ms, v = _time(step, steps, warmup)
    return {"ms_per_step": ms, "voxels_per_s": x.numel() / ms * 1e3, "loss": float(v)}


def cfg5(steps=3, warmup=2, dev="cuda"):
    """Per-GPU SPOCO step of cfg 5: UNet3d(1->8) student fwd+bwd, EMA teacher fwd, SPOCOLoss, AdamW, EMA, on 1x1x96x192x192."""
    import numpy as np
    from torch_em_amd.loss import SPOCOLoss
    from torch_em_amd.model import UNet3d
    from torch_em_amd.optim import FusedAdamW
    from torch_em_amd.trainer import SPOCOTrainer
    D, H, W, E = 96, 192, 192, 8
    torch.manual_seed(0)
    np.random.seed(0)
    small = torch.randint(0, 34, (4, 6, 6))
    small[small > 30] = 0
    lbl = small.repeat_interleave(24, 0).repeat_interleave(32, 1).repeat_interleave(32, 2)[None, None].contiguous().to(dev)
    ids = torch.unique(lbl)
    remap = torch.zeros(int(ids.max()) + 1, dtype=torch.int64, device=dev)
    remap[ids] = torch.arange(len(ids), device=dev)
    lbl = remap[lbl]
    loss = SPOCOLoss(delta_var=0.75, delta_dist=2.0, aux_loss="dice")
    model = UNet3d(1, E, initial_features=32, depth=4).to(dev)
    x = torch.randn(1, 1, D, H, W, device=dev)
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x.cpu(), lbl[0].cpu()[None]), batch_size=1)
    tr = SPOCOTrainer(model=model, momentum=0.999, name="b", train_loader=dl, val_loader=dl, loss=loss,
                      optimizer=FusedAdamW(model.parameters(), lr=1e-4), metric=loss, device=dev, save_root="/tmp/spoco_b",
                      logger=None, mixed_precision=False)   # the fp32-class step (the bare flag means fp16 autocast since round 6)
    tr._initialize(1, None)
    ms, v = _time(lambda: tr._step(x, tr.loss, lbl)[1], steps, warmup)
    return {"ms_per_step": ms, "voxels_per_s": D * H * W / ms * 1e3, "loss": float(v)}


if __name__ == "__main__":
    which = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    print(which, {1: cfg1, 3: cfg3, 5: cfg5}[which]())
