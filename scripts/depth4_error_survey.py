"""Gradient error of the benchmark network (UNet3d(1, 2, initial_features=32, depth=4), one 64^3 volume) against the float64
oracle for several seeds: this library in its default arithmetic (TEM_PRECISION=split16) / with TEM_PRECISION from the
environment, and the fp32 reference path (oracle in float32).  Run once per precision mode:
    TEM_PRECISION=split16 python scripts/depth4_error_survey.py > profiles/r03_depth4_error_split16.txt
The point: at these widths the global error is dominated by a handful of near-tie decisions
(ReLU masks / pooling arg-maxes of the 4^3 and 8^3 levels), so it scatters by 3x between seeds and between arithmetics
that agree to 1e-7 in the forward pass."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref  # noqa: E402
from torch_em_amd.loss import DiceLoss  # noqa: E402
from torch_em_amd.model import UNet3d  # noqa: E402


def case(seed, cfg3):
    """(model, scale factors, x, y, oracle kwargs, loss on the device)"""
    if not cfg3:
        torch.manual_seed(seed)
        model = UNet3d(1, 2, depth=4, initial_features=32)
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(1, 1, 64, 64, 64, generator=g)
        y = (torch.rand(1, 2, 64, 64, 64, generator=g) > 0.5).float()
        return model, [2, 2, 2, 2], x, y, dict(norm="InstanceNorm"), DiceLoss()
    # cfg 3 of BASELINE.json (tests/test_gpu_unet.py::test_anisotropic_cfg3_factors_match_fp64_oracle): seed 0 is that test
    from oracle import loss_ref
    from torch_em_amd.loss import ApplyAndRemoveMask, LossWrapper
    from torch_em_amd.model import AnisotropicUNet
    torch.manual_seed(seed)
    sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    model = AnisotropicUNet(1, 12, sf, initial_features=32, final_activation="Sigmoid", anisotropic_kernel=True)
    g = torch.Generator().manual_seed(7 + seed)
    x = torch.randn(1, 1, 16, 64, 64, generator=g)
    y = torch.cat([(torch.rand(1, 12, 16, 64, 64, generator=g) > 0.5).float(),
                   (torch.rand(1, 12, 16, 64, 64, generator=g) > 0.3).float()], dim=1)
    return model, sf, x, y, dict(norm="InstanceNorm", final_activation="Sigmoid", loss_fn=loss_ref.masked_dice_loss), \
        LossWrapper(DiceLoss(), ApplyAndRemoveMask("multiply"))


def main():
    args = sys.argv[1:]
    cfg3 = "--cfg3" in args
    seeds = [int(s) for s in ([a for a in args if a != "--cfg3"] or ["0", "1", "2", "3", "4", "5"])]
    print(f"# TEM_PRECISION={os.environ.get('TEM_PRECISION', 'split16')}  TEM_DGRAD16={os.environ.get('TEM_DGRAD16', 'default')}")
    print("# seed  global_L2(hip)  global_L2(fp32 ref)  worst tensor (hip)  its L2 hip / ref   entries > 1e-2 of max (hip / ref)")
    for seed in seeds:
        model, sf, x, y, okw, loss_fn = case(seed, cfg3)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        res = {}
        for dt in (torch.float64, torch.float32):
            _, _, gr = unet_ref.unet_loss_and_grads({k: v.to(dt) for k, v in sd.items()}, x.to(dt), y.to(dt), sf, **okw)
            res[dt] = {k: v.double().numpy() for k, v in gr.items()}
        model.to("cuda")
        pred = model(x.cuda())
        loss_fn(pred, y.cuda()).backward()
        hip = {k: p.grad.double().cpu().numpy() for k, p in model.named_parameters()}
        g64, g32 = res[torch.float64], res[torch.float32]
        keys = [k for k in hip if np.abs(g64[k]).max() > 1e-4 * max(np.abs(v).max() for v in g64.values())]
        cat = lambda d: np.concatenate([d[k].ravel() for k in keys])  # noqa: E731
        r = cat(g64)
        e_h = np.linalg.norm(cat(hip) - r) / np.linalg.norm(r)
        e_c = np.linalg.norm(cat(g32) - r) / np.linalg.norm(r)
        per = {k: (np.linalg.norm(hip[k] - g64[k]) / np.linalg.norm(g64[k]), np.linalg.norm(g32[k] - g64[k]) / np.linalg.norm(g64[k]))
               for k in keys}
        worst = max(per, key=lambda k: per[k][0])
        nh = sum(int((np.abs(hip[k] - g64[k]) > 1e-2 * np.abs(g64[k]).max()).sum()) for k in keys)
        nc = sum(int((np.abs(g32[k] - g64[k]) > 1e-2 * np.abs(g64[k]).max()).sum()) for k in keys)
        print(f"{seed:5d}  {e_h:.2e}  {e_c:.2e}  {worst:38s}  {per[worst][0]:.2e} / {per[worst][1]:.2e}   {nh} / {nc}", flush=True)


if __name__ == "__main__":
    main()
