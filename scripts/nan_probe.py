"""Does a NaN in g reach the weight gradient in every arithmetic / kernel?  (developer probe)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import ops, _lib
gen = torch.Generator().manual_seed(1)
x5 = torch.randn(1, 16, 8, 8, 32, generator=gen).cuda()
for val in (float("nan"), float("inf")):
    g5 = torch.randn(1, 16, 8, 8, 32, generator=gen).cuda()
    g5[0, 3, 3, 3, 7] = val
    for zs in (1, 3):
        _lib.set_option("wgrad_zs", zs)
        for mode in (1, 2, 5, 7, 8):
            dw = torch.empty(32 * 32 * 27, device="cuda")
            if mode == 8:
                ops.conv_wgrad_gscaled(x5, g5, (3, 3, 3), 32, 32, dw, None, ops.absmax(g5))
            else:
                ops.conv_wgrad(x5, g5, (3, 3, 3), 32, 32, dw, None, mfma=mode)
            d = dw.view(32, 32, 27)
            print(f"g has {val}: wgrad_zs={zs} mode={mode}: non-finite in dw[co=7]: {int((~torch.isfinite(d[7])).sum())} of {d[7].numel()}, elsewhere {int((~torch.isfinite(d)).sum() - (~torch.isfinite(d[7])).sum())}")
# forward convolutions: a NaN / inf in the input
w = (torch.randn(32, 32, 3, 3, 3, generator=gen) * 0.1).cuda()
one, zero = torch.ones(1, 32, device="cuda"), torch.zeros(1, 32, device="cuda")
for val in (float("nan"), float("inf")):
    xx = torch.randn(1, 32, 64, 64, 32, generator=gen).cuda()
    xx[0, 5, 5, 5, 3] = val
    for mode in (1, 2, 3, 4, 5, 7):
        y = ops.new_act(1, 32, 64, 64, 32, "cuda")
        ops.conv_fwd(xx, ops.pack_weights(w, transpose=False, mfma=mode), None, y, (3, 3, 3), 32, 32, scale=one, shift=zero, mfma=mode)
        print(f"x has {val}: conv_fwd mode={mode} (kernel family {ops.conv_fwd_family(xx, (3, 3, 3), 32, 32, mode)}): non-finite outputs {int((~torch.isfinite(y)).sum())}, max |y| {float(y[torch.isfinite(y)].abs().max()):.3g}")
