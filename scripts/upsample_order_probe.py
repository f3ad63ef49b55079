"""is the ORDER of the Upsampler (reference: interpolate, then 1x1x1 conv, model/unet.py:455-458; this
library: 1x1x1 conv at the low resolution, then interpolate -- identical in exact arithmetic, 8x fewer FLOPs, no 1 GiB
intermediate) what puts the exact-fp32 build further from float64 than the reference's fp32 path on the depth-4 survey?
The fp32 CPU oracle runs both orders against the float64 oracle (reference order), seeds as scripts/depth4_error_survey.py.
CPU only.    python scripts/upsample_order_probe.py 0 5 > profiles/r04_upsample_order_probe.txt"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref  # noqa: E402
from torch_em_amd.model import UNet3d  # noqa: E402

_interp = F.interpolate


def run(sd, x, y, dt, conv_first):
    """oracle step in dtype dt; conv_first swaps the Upsampler's order by intercepting interpolate / the sampler conv"""
    orig_conv = unet_ref._conv
    pending = {}
    if conv_first:
        def interp(t, **kw):          # postpone: remember the arguments, hand the low-resolution tensor on
            pending["kw"] = kw
            return t

        def conv(t, w, b):
            out = orig_conv(t, w, b)
            if "kw" in pending and w.shape[2:] == (1, 1, 1):
                out = _interp(out, **pending.pop("kw"))
            return out
        F.interpolate, unet_ref._conv = interp, conv
    try:
        _, _, gr = unet_ref.unet_loss_and_grads({k: v.to(dt) for k, v in sd.items()}, x.to(dt), y.to(dt), [2, 2, 2, 2],
                                                norm="InstanceNorm")
    finally:
        F.interpolate, unet_ref._conv = _interp, orig_conv
    return {k: v.double().numpy() for k, v in gr.items()}


def thread_spread(seeds):
    """the fp32 reference path against float64 for several oneDNN thread counts: the same arithmetic CLASS, another blocking"""
    print("# seed | global L2 error of the fp32 reference path vs float64 with torch.set_num_threads(1 / 2 / 4 / 8)")
    for seed in seeds:
        torch.manual_seed(seed)
        model = UNet3d(1, 2, depth=4, initial_features=32)
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(1, 1, 64, 64, 64, generator=g)
        y = (torch.rand(1, 2, 64, 64, 64, generator=g) > 0.5).float()
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        ref = run(sd, x, y, torch.float64, False)
        keys = [k for k in ref if np.abs(ref[k]).max() > 1e-4 * max(np.abs(v).max() for v in ref.values())]
        cat = lambda d: np.concatenate([d[k].ravel() for k in keys])  # noqa: E731
        r = cat(ref)
        row = []
        for nt in (1, 2, 4, 8):
            torch.set_num_threads(nt)
            row.append(np.linalg.norm(cat(run(sd, x, y, torch.float32, False)) - r) / np.linalg.norm(r))
        torch.set_num_threads(os.cpu_count())
        print(f"{seed:4d} | " + " | ".join(f"{v:.2e}" for v in row), flush=True)


def main():
    if "--threads" in sys.argv:
        return thread_spread([int(s) for s in sys.argv[1:] if s != "--threads"] or [0, 5])
    seeds = [int(s) for s in sys.argv[1:]] or [0, 5]
    print("# seed | global L2 error vs float64 (reference order): fp32 reference order | fp32 conv-first order | float64 conv-first order")
    for seed in seeds:
        torch.manual_seed(seed)
        model = UNet3d(1, 2, depth=4, initial_features=32)
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(1, 1, 64, 64, 64, generator=g)
        y = (torch.rand(1, 2, 64, 64, 64, generator=g) > 0.5).float()
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        ref = run(sd, x, y, torch.float64, False)
        keys = [k for k in ref if np.abs(ref[k]).max() > 1e-4 * max(np.abs(v).max() for v in ref.values())]
        cat = lambda d: np.concatenate([d[k].ravel() for k in keys])  # noqa: E731
        r = cat(ref)
        row = []
        for dt, cf in ((torch.float32, False), (torch.float32, True), (torch.float64, True)):
            row.append(np.linalg.norm(cat(run(sd, x, y, dt, cf)) - r) / np.linalg.norm(r))
        print(f"{seed:4d} | " + " | ".join(f"{v:.2e}" for v in row), flush=True)


if __name__ == "__main__":
    main()
