"""Repeat one data-gradient launch of the z-reuse kernel with the fused norm backward (tem_conv3d_fwd_refnorm, MODE 3 of
k_conv_zr) and compare every result bit for bit with the first one.  This is the launch that exposed the store-data hazard
described at zr_store4 (csrc/conv_zr.hip): 1-99 % of the repeats differed in a handful of elements before the fix.
usage: python scripts/race_zr_store.py [repeats]      (TEM_LIB=<variant .so> for A/B builds)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import _lib, ops  # noqa: E402

DEV = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500


def to5(t):
    return t.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


total = 0
for wide, blocks in [(1, 1), (1, 0), (0, 1)]:
    _lib.set_option("zr_wide", wide)
    _lib.set_option("zr_tile_blocks", blocks)
    for case in [(2, 18, 61, 67, 64, 32), (2, 32, 64, 64, 32, 32)]:
        for mode in (5, 7, 2, 4):
            N, D, H, W, Cin, Cout = case
            k = (3, 3, 3)
            g = torch.Generator().manual_seed(23)
            w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
            g5 = to5(torch.randn(N, Cout, D, H, W, generator=g))
            a1 = to5(torch.relu(torch.randn(N, Cin, D, H, W, generator=g) + 0.2))
            coef = torch.randn(N, Cin, 4, generator=g).to(DEV)
            wp = ops.pack_weights(w, transpose=True, mfma=mode)
            first, bad = None, 0
            junk = torch.empty(64 << 20, device=DEV)
            for i in range(reps):
                got = torch.full((N, D, H, W, Cin), float("nan"), device=DEV)
                ops.conv_fwd_refnorm(g5, wp, got, k, Cout, Cin, a1, coef, mode)
                if i % 3 == 0:
                    junk.normal_()   # perturb timing / caches
                if first is None:
                    first = got.clone()
                elif not torch.equal(got, first):
                    bad += 1
            total += bad
            print(f"zr_wide {wide} zr_tile_blocks {blocks} case {case} mode {mode}: {bad} mismatching repeats of {reps}", flush=True)
print("TOTAL mismatches", total)
