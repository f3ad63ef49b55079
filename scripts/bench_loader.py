"""cfg 2 with the loader in the loop: ms/step of the DefaultTrainer hot loop when every batch comes
out of a torch DataLoader as a RAW uint-like volume + INTEGER instance labels (pinned, 16 MB + 16 MB per 2x128^3 batch)
and is standardised, flipped and turned into boundary targets on the device by the trainer's pre-pass, versus the same
step on inputs that already sit in HBM.   usage: python scripts/bench_loader.py [steps] [workers]"""
import functools
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(steps=12, workers=2, size=128, batch=2, dev="cuda"):
    import torch_em_amd
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    from torch_em_amd.optim import FusedAdamW
    from torch_em_amd.transform import BoundaryTransform, get_augmentations, standardize
    from torch_em_amd.transform.label import BatchTargets
    torch.manual_seed(0)
    nb = 4
    raw = torch.rand(nb * batch, 1, size, size, size) * 255.0
    lab = torch.randint(1, 40, (nb * batch, 1, size // 16, size // 16, size // 16)).repeat_interleave(16, 2) \
        .repeat_interleave(16, 3).repeat_interleave(16, 4).to(torch.int32)

    class Repeat(torch.utils.data.Dataset):
        def __len__(self):
            return steps * batch

        def __getitem__(self, i):
            return raw[i % raw.shape[0]], lab[i % lab.shape[0]]
    out = {}
    for mode in ("resident", "loader_prefetch", "loader_serial"):
        model = UNet3d(1, 2, initial_features=32, depth=4).to(dev)
        loader = torch.utils.data.DataLoader(Repeat(), batch_size=batch, shuffle=False, num_workers=workers,
                                             pin_memory=True, persistent_workers=workers > 0)
        trainer = torch_em_amd.default_segmentation_trainer(
            "bl", model, loader, loader, device=dev, logger=None, save_root="/tmp/bench_loader", mixed_precision=False,
            raw_transform=functools.partial(standardize, per_sample=True), augmentation=get_augmentations(3),
            target_transform=BatchTargets(BoundaryTransform(add_binary_target=True, ndim=3)), prefetch=(mode != "loader_serial"))
        trainer._initialize(steps, None)
        trainer.model.train()
        if mode == "resident":
            x = trainer._prepass(True)(raw[:batch].to(dev), lab[:batch].to(dev))
            batches = [x] * steps
        else:
            batches = trainer._batches(loader, train=True)
        n, t0 = 0, None
        for x, y in batches:
            if n == 3:   # warm-up: packs, arenas, worker start-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            trainer.optimizer.zero_grad()
            pred, loss = trainer._forward_and_loss(x, y)
            trainer._backprop(loss)
            n += 1
        torch.cuda.synchronize()
        out[mode + "_ms_per_step"] = (time.perf_counter() - t0) / (n - 3) * 1e3
        del loader, trainer, model
    return out


if __name__ == "__main__":
    st = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    wk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    print(measure(st, wk))
