"""What does the BACKWARD arithmetic of the MFMA convolutions cost in gradient accuracy, with the forward pass held fixed?

The six-seed survey (scripts/depth4_error_survey.py) compares whole training steps against float64 and is dominated by
near-tie ReLU / arg-max flips of the forward pass.  This script removes that noise: the forward pass and
every non-conv backward op run in float64 on the CPU; only the two gradient products of the 3x3x3 convolutions
(data gradient g * w, weight gradient x^ * g) see their OPERANDS rounded the way a kernel variant would round them
(products and sums exact, i.e. fp32 accumulation is not modelled: it is the same in every variant).  Output: the error
of the whole parameter gradient against the exact float64 backward, per variant -- the part of the error budget that a
choice of backward arithmetic is responsible for.  CPU only (this is an analysis of arithmetic, not a test of kernels).

    python scripts/backward_arith_sim.py [--size 64] [--seeds 0 1] > profiles/r04_backward_arith_sim.txt

Operand formats (value kept after rounding):
    b1   one bf16 term (8 bits)                    b2   two bf16 terms hi + lo (16 bits)
    h1   one fp16 term (11 bits, power-of-two prescale from max |.| so that nothing leaves fp16's normal range ... values
         below 2^-24 of the maximum flush)          h2   two fp16 terms hi + lo' 2^-12 (22 bits)
A product of an n-term and an m-term operand written "AxB-k" issues k MFMAs: the k largest of the n*m term products."""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref  # noqa: E402
from oracle.loss_ref import dice_loss  # noqa: E402


def _bf16(v):
    return v.float().bfloat16().double()


def _fp16_scaled(v, amax):
    """round to fp16 after the power-of-two prescale that puts max |v| at 2^14 (what tem_conv3d_fwd_gscaled does)"""
    if amax == 0:
        return v.clone(), 1.0
    s = 2.0 ** (14 - int(np.floor(np.log2(amax))))
    return (v * s).float().half().double() / s, s


def terms(v, fmt):
    """-> list of float64 tensors whose sum is the operand the MFMAs see (largest term first)"""
    if fmt == "exact":
        return [v]
    if fmt == "b1":
        return [_bf16(v)]
    if fmt == "b2":
        hi = _bf16(v)
        return [hi, _bf16(v - hi)]
    amax = float(v.abs().max())
    if fmt == "h1":
        return [_fp16_scaled(v, amax)[0]]
    if fmt == "h2":
        hi, s = _fp16_scaled(v, amax)
        lo = ((v - hi) * s * 4096.0).float().half().double() / (s * 4096.0)
        return [hi, lo]
    raise ValueError(fmt)


def bilinear(op, a_terms, b_terms, k):
    """sum of the k largest term products op(a_i, b_j) (order: i + j ascending, a-major)"""
    pairs = sorted(((i + j, i, j) for i in range(len(a_terms)) for j in range(len(b_terms))))[:k]
    # group by the a term: op is linear in b
    out = None
    for i in sorted({p[1] for p in pairs}):
        b = sum(b_terms[p[2]] for p in pairs if p[1] == i)
        r = op(a_terms[i], b)
        out = r if out is None else out + r
    return out


VARIANT = {"dgrad": ("exact", "exact", 1), "wgrad": ("exact", "exact", 1), "min_cin": 32, "wgrad_min_voxels": 0}


class ConvSim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        pad = tuple(k // 2 for k in w.shape[2:])
        return F.conv3d(x, w, b, padding=pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        pad = tuple(k // 2 for k in w.shape[2:])
        sim = (w.shape[2:] == (3, 3, 3) or (VARIANT.get("sim_1x1") and w.shape[2:] == (1, 1, 1) and w.shape[0] % 32 == 0)) and \
            w.shape[1] >= VARIANT["min_cin"]   # sim_1x1: the 1x1x1 convs of the Upsampler ride on the same bf16x3 data gradient
        gf, wf, kd = VARIANT["dgrad"] if sim else ("exact", "exact", 1)
        xf, gf2, kw = VARIANT["wgrad"] if sim else ("exact", "exact", 1)
        if sim and x.shape[2] * x.shape[3] * x.shape[4] * x.shape[0] < VARIANT["wgrad_min_voxels"]:
            xf, gf2, kw = VARIANT["wgrad_small"]
        if sim and xf == "h1s":
            # one fp16 term of the SCALED RAW activation s*x (the ReLU zeros of the previous layer stay exact zeros, so
            # no rounding error is shared by many voxels) + the shift part t * sum_v g handled exactly
            t = getattr(x, "_tem_shift", None)
            assert t is not None
            t = t.expand_as(x)
            gw = bilinear(lambda a, b_: torch.nn.grad.conv3d_weight(a, w.shape, b_, padding=pad), terms(x - t, "h1"),
                          terms(g, gf2), kw) + torch.nn.grad.conv3d_weight(t.contiguous(), w.shape, g, padding=pad)
            gx = bilinear(lambda a, b_: torch.nn.grad.conv3d_input(x.shape, b_, a, padding=pad), terms(g, gf), terms(w, wf), kd) \
                if ctx.needs_input_grad[0] else None
            return gx, gw, g.sum(dim=(0, 2, 3, 4))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = bilinear(lambda a, b_: torch.nn.grad.conv3d_input(x.shape, b_, a, padding=pad), terms(g, gf), terms(w, wf), kd)
        gw = bilinear(lambda a, b_: torch.nn.grad.conv3d_weight(a, w.shape, b_, padding=pad), terms(x, xf), terms(g, gf2), kw)
        return gx, gw, g.sum(dim=(0, 2, 3, 4))


def _norm_tagged(x, norm, gamma, beta, n_groups=32, buffers=None, training=True):
    assert norm == "InstanceNorm"
    mean = x.mean((2, 3, 4), keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var((2, 3, 4), unbiased=False, keepdim=True) + 1e-5)
    out = (x - mean) * rstd
    out._tem_shift = (-mean * rstd).detach()
    return out


def run(sd, x, y, sf):
    orig, orig_norm = unet_ref._conv, unet_ref._norm
    unet_ref._conv = lambda x_, w_, b_: ConvSim.apply(x_, w_, b_) if w_.dim() == 5 else orig(x_, w_, b_)
    unet_ref._norm = _norm_tagged
    try:
        p = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        pred = unet_ref.unet_forward(p, x, sf, "InstanceNorm", None)
        dice_loss(pred, y).backward()
        return {k: v.grad.numpy().copy() for k, v in p.items()}
    finally:
        unet_ref._conv, unet_ref._norm = orig, orig_norm


VARIANTS = [
    # name, dgrad (g fmt, w fmt, MFMAs), wgrad (x fmt, g fmt, MFMAs)
    ("today: dgrad b2xb2-3, wgrad b2xb2-3", ("b2", "b2", 3), ("b2", "b2", 3)),
    ("wgrad b2xb1-2 (x two terms, g one)", ("b2", "b2", 3), ("b2", "b1", 2)),
    ("wgrad b1xb2-2 (x one term, g two)", ("b2", "b2", 3), ("b1", "b2", 2)),
    ("wgrad b1xb1-1", ("b2", "b2", 3), ("b1", "b1", 1)),
    ("wgrad h1xh1-1 (g prescaled)", ("b2", "b2", 3), ("h1", "h1", 1)),
    ("wgrad h2xh1-2", ("b2", "b2", 3), ("h2", "h1", 2)),
    ("wgrad h1xh2-2", ("b2", "b2", 3), ("h1", "h2", 2)),
    ("dgrad h2xh1-2 (g two terms prescaled, w one)", ("h2", "h1", 2), ("b2", "b2", 3)),
    ("dgrad h1xh2-2 (g one term, w two)", ("h1", "h2", 2), ("b2", "b2", 3)),
    ("dgrad h1xh1-1", ("h1", "h1", 1), ("b2", "b2", 3)),
    ("dgrad b2xb1-2", ("b2", "b1", 2), ("b2", "b2", 3)),
    ("dgrad h2xh2-3 (TEM_DGRAD16)", ("h2", "h2", 3), ("b2", "b2", 3)),
    ("dgrad h2xh1-2 + wgrad h1xh1-1", ("h2", "h1", 2), ("h1", "h1", 1)),
    ("dgrad b2xb2-3 + wgrad h1xh1-1 >= 2^17 voxels else b2xb2-3", ("b2", "b2", 3), ("h1", "h1", 1)),
    ("wgrad h1(s*x)xh1-1 + exact shift term", ("b2", "b2", 3), ("h1s", "h1", 1)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--seeds", type=int, nargs="*", default=[0, 1])
    ap.add_argument("--only", type=int, nargs="*", default=None)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    print(f"# UNet3d(1, 2, initial_features={args.features}, depth=4), 1 x {args.size}^3, float64 forward held fixed; error of the parameter")
    print("# gradient caused by the operand rounding of the 3x3x3 gradient convolutions alone (Cin >= 32), relative L2:")
    print("# variant | seed | global | worst tensor (its error) | median tensor error")
    from torch_em_amd.model import UNet3d
    for seed in args.seeds:
        torch.manual_seed(seed)
        model = UNet3d(1, 2, depth=4, initial_features=args.features)
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(1, 1, args.size, args.size, args.size, generator=g).double()
        y = (torch.rand(1, 2, args.size, args.size, args.size, generator=g) > 0.5).double()
        sd = {k: v.detach().double() for k, v in model.state_dict().items()}
        VARIANT.update(dgrad=("exact", "exact", 1), wgrad=("exact", "exact", 1), wgrad_min_voxels=0)
        t0 = time.time()
        ref = run(sd, x, y, [2, 2, 2, 2])
        print(f"# seed {seed}: exact backward {time.time() - t0:.1f} s", flush=True)
        keys = [k for k in ref if np.abs(ref[k]).max() > 1e-4 * max(np.abs(v).max() for v in ref.values())]
        cat = lambda d: np.concatenate([d[k].ravel() for k in keys])  # noqa: E731
        for vi, (name, dg, wg) in enumerate(VARIANTS):
            if args.only is not None and vi not in args.only:
                continue
            VARIANT.update(dgrad=dg, wgrad=wg, wgrad_min_voxels=0)
            if ">=" in name:
                VARIANT.update(wgrad_min_voxels=2 ** 17, wgrad_small=("b2", "b2", 3))
            got = run(sd, x, y, [2, 2, 2, 2])
            e = np.linalg.norm(cat(got) - cat(ref)) / np.linalg.norm(cat(ref))
            per = {k: np.linalg.norm(got[k] - ref[k]) / np.linalg.norm(ref[k]) for k in keys}
            worst = max(per, key=per.get)
            print(f"{name:62s} | {seed} | {e:.2e} | {worst} ({per[worst]:.2e}) | {np.median(list(per.values())):.2e}", flush=True)


if __name__ == "__main__":
    main()
