"""A/B of the forward / dgrad convolution kernel variants (tem_set_option("conv_fwd_variant", v)) on one GPU:
correctness of every variant against a float64 torch convolution on ragged shapes, then interleaved timing rounds on
the layer shapes of cfg 2.   usage: python scripts/pp_ab.py [check|bench|all] [rounds] [iters]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import _lib, ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda"
VARIANTS = [int(v) for v in os.environ.get("PP_VARIANTS", "0,1,2").split(",")]


def to5(t):  # NCDHW -> NDHWC contiguous
    return t.permute(0, 2, 3, 4, 1).contiguous()


def run_case(N, D, H, W, cin, cout, k, mode, variant, use_norm, use_ref, want_stats, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(N, cin, D, H, W, generator=g)
    w = torch.randn(cout, cin, *k, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    scale = (torch.rand(N, cin, generator=g) + 0.5) if use_norm else None
    shift = torch.randn(N, cin, generator=g) if use_norm else None
    refm = torch.randn(N, cout, D, H, W, generator=g) if use_ref else None
    xn = x.double()
    if use_norm:
        xn = xn * scale.double()[:, :, None, None, None] + shift.double()[:, :, None, None, None]
    yref = F.conv3d(xn.to(dev), w.double().to(dev), b.double().to(dev), padding=tuple(v // 2 for v in k))
    yref = torch.relu(yref)
    if use_ref:
        yref = yref * (refm.to(dev) > 0)
    _lib.set_option("conv_fwd_variant", variant)
    x5 = to5(x).to(dev)
    y5 = torch.full((N, D, H, W, cout), float("nan"), device=dev)
    wp = ops.pack_weights(w.to(dev), False, mode)
    r5 = to5(refm).to(dev) if use_ref else None
    out = ops.conv_fwd(x5, wp, b.to(dev), y5, k, cin, cout, scale=scale.to(dev) if use_norm else None,
                       shift=shift.to(dev) if use_norm else None, act="relu", ref=r5, mfma=mode, want_stats=want_stats)
    torch.cuda.synchronize()
    got = y5.permute(0, 4, 1, 2, 3).double()
    err = float((got - yref).abs().max() / yref.abs().max())
    serr = None
    if want_stats and out is not None:
        part, nblk = out
        s = part.double().sum(1)  # [N, cout, 2]
        s_ref = torch.stack([got.sum((2, 3, 4)), (got * got).sum((2, 3, 4))], -1)
        serr = float((s - s_ref).abs().max() / s_ref.abs().max())
    return err, serr


def check():
    ok = True
    shapes = [(1, 20, 24, 40, 32, 32), (2, 9, 17, 33, 32, 64), (1, 16, 16, 16, 64, 32), (1, 8, 32, 32, 16, 96)]
    for (N, D, H, W, cin, cout) in shapes:
        for k in ((3, 3, 3), (1, 3, 3)):
            for mode, tol in ((4, 3e-6), (6, 3e-6), (2, 1e-4)):
                for variant in VARIANTS:
                    if mode == 6 and variant != 0:
                        continue
                    for (use_norm, use_ref, want_stats) in ((True, False, True), (False, True, False)):
                        err, serr = run_case(N, D, H, W, cin, cout, k, mode, variant, use_norm, use_ref, want_stats)
                        good = err < tol and (serr is None or serr < 1e-5)
                        ok &= good
                        print(f"check {N}x{D}x{H}x{W} {cin}->{cout} k{k} mode {mode} variant {variant} norm {int(use_norm)} "
                              f"ref {int(use_ref)}: err {err:.2e} stats {serr if serr is None else f'{serr:.1e}'} "
                              f"{'ok' if good else 'FAIL'}", flush=True)
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    return ok


def bench():
    shapes = [(2, 128, 128, 128, 32, 32), (2, 128, 128, 128, 64, 32), (2, 128, 128, 128, 32, 64), (2, 64, 64, 64, 64, 64),
              (2, 64, 64, 64, 128, 64), (2, 64, 64, 64, 64, 128), (2, 32, 32, 32, 128, 128), (2, 32, 32, 32, 256, 128),
              (2, 16, 16, 16, 256, 256)]
    if os.environ.get("PP_SHAPES") == "small":
        shapes = [(2, 16, 16, 16, 256, 256), (2, 16, 16, 16, 512, 256), (2, 16, 16, 16, 128, 256), (2, 8, 8, 8, 512, 512),
                  (2, 8, 8, 8, 256, 512), (2, 32, 32, 32, 128, 64)]
    k = (3, 3, 3)
    for (N, D, H, W, cin, cout) in shapes:
        torch.manual_seed(0)
        x = torch.randn(N, D, H, W, cin, device=dev)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        scale = torch.rand(N, cin, device=dev) + 0.5
        shift = torch.randn(N, cin, device=dev)
        refm = torch.randn(N, D, H, W, cout, device=dev)
        y = torch.empty(N, D, H, W, cout, device=dev)
        fl = 2.0 * N * D * H * W * cin * cout * 27
        arms = []
        for mode in (4, 2):
            wp = ops.pack_weights(w, False, mode)
            for variant in VARIANTS:
                if mode == 6 and variant != 0:
                    continue
                arms.append((mode, variant, wp))
        res = {(m, v): [] for m, v, _ in arms}

        def run(mode, variant, wp):
            _lib.set_option("conv_fwd_variant", variant)
            if mode == 2:   # dgrad-like launch: no norm, ReLU mask
                ops.conv_fwd(x, wp, None, y, k, cin, cout, ref=refm, mfma=mode)
            else:           # forward-like launch: fused norm, bias, ReLU, statistics
                ops.conv_fwd(x, wp, b, y, k, cin, cout, scale=scale, shift=shift, act="relu", mfma=mode, want_stats=True)
        for a in arms:
            run(*a)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for a in arms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run(*a)
                e1.record()
                torch.cuda.synchronize()
                res[(a[0], a[1])].append(e0.elapsed_time(e1) / iters)
        for (m, v), ts in res.items():
            t = min(ts)
            print(f"bench {N}x{D}x{H}x{W} {cin}->{cout} mode {m} variant {v}: min {t:.4f} ms med {sorted(ts)[len(ts) // 2]:.4f} ms  "
                  f"{fl / t / 1e9:.0f} TF alg  mfma_frac {fl * 3 / t / 1e9 / 2500:.3f}", flush=True)


if __name__ == "__main__":
    ok = True
    if what in ("check", "all"):
        ok = check()
    if what in ("bench", "all"):
        bench()
    sys.exit(0 if ok else 1)
