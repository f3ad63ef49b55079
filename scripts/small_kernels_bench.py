"""Times the HBM-bound kernels of the cfg-2 step one by one (HIP events, 20 launches each) and prints achieved GB/s over
their algorithmic bytes: the first conv (Cin = 1) forward / weight gradient, factor-2 upsampling forward (+ statistics) /
backward (+ deferred norm), max-pool, the 1x1x1 out_conv.  Developer tool; run on the GPU box:
    python scripts/small_kernels_bench.py [name-filter]
TEM_LIB=<path> selects an alternative build of libtem_hip.so for A/B runs."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch_em_amd import ops  # noqa: E402

DEV = "cuda"
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(name, nbytes, fn, iters=20):
    if flt and flt not in name:
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    print(f"{name:44s} {us:9.1f} us   {nbytes / us / 1e3:8.0f} GB/s   ({nbytes / 1e6:.0f} MB)", flush=True)


def main():
    torch.manual_seed(0)
    N, S = 2, 128
    T = N * S ** 3 * 32 * 4
    # first conv: 1 -> 32
    x1 = torch.randn(N, S, S, S, 1, device=DEV)
    w = torch.randn(32, 1, 3, 3, 3, device=DEV) * 0.1
    b = torch.randn(32, device=DEV)
    y = ops.new_act(N, S, S, S, 32, DEV)
    wp = ops.pack_weights(w, False, False)
    sc, sf = torch.ones(N, 1, device=DEV), torch.zeros(N, 1, device=DEV)
    timeit("cin1 fwd 1->32 @128^3 (+stats)", T, lambda: ops.conv_fwd(x1, wp, b, y, (3, 3, 3), 1, 32, scale=sc, shift=sf, act="relu",
                                                                      want_stats=True))
    g = torch.randn(N, S, S, S, 32, device=DEV)
    dw, db = torch.empty_like(w), torch.empty(32, device=DEV)
    timeit("cin1 wgrad 1->32 @128^3", T, lambda: ops.conv_wgrad(x1, g, (3, 3, 3), 1, 32, dw, db, scale=sc, shift=sf))
    # upsampling, level by level (coarse size, channels of the upsampled half)
    for lvl, (cs, C) in enumerate([(64, 32), (32, 64), (16, 128), (8, 256)]):
        u = torch.randn(N, cs, cs, cs, C, device=DEV)
        cat = torch.empty(N, 2 * cs, 2 * cs, 2 * cs, 2 * C, device=DEV)
        fine = N * (2 * cs) ** 3 * C * 4
        timeit(f"upsample fwd+stats L{lvl} {cs}^3x{C}", fine + fine // 8, lambda: ops.upsample_fwd(u, cat[..., :C], (2, 2, 2), stats=True))
        gc = torch.randn_like(cat)
        gu = torch.empty_like(u)
        timeit(f"upsample bwd L{lvl}", fine + fine // 8, lambda: ops.upsample_bwd(gc[..., :C], gu, (2, 2, 2)))
        coef = torch.randn(N, 2 * C, 4, device=DEV)
        timeit(f"upsample bwd+norm L{lvl}", fine + fine // 4, lambda: ops.upsample_bwd(gc[..., :C], gu, (2, 2, 2), norm=(u, coef[:, :C])))
        if lvl == 0:
            sk = torch.randn(N, 2 * cs, 2 * cs, 2 * cs, C, device=DEV)
            po = torch.empty(N, cs, cs, cs, C, device=DEV)
            timeit("maxpool fwd L0", fine + fine // 8, lambda: ops.maxpool_fwd(sk, po, (2, 2, 2)))
            gs = torch.empty_like(sk)
            gp = torch.randn_like(po)
            timeit("maxpool bwd L0 (+gskip, relu mask)", 3 * fine + fine // 8,
                   lambda: ops.maxpool_bwd(gp, sk, gs, (2, 2, 2), gskip=gc[..., C:], relu_mask=True))
    # out_conv 32 -> 2
    xo = torch.randn(N, S, S, S, 32, device=DEV)
    wo = torch.randn(2, 32, 1, 1, 1, device=DEV) * 0.1
    bo = torch.randn(2, device=DEV)
    yo = ops.new_act(N, S, S, S, 2, DEV)
    wpo = ops.pack_weights(wo, False, False)
    timeit("out_conv fwd 32->2 (sigmoid)", T, lambda: ops.conv_fwd(xo, wpo, bo, yo, (1, 1, 1), 32, 2, act="sigmoid"))
    go = torch.randn_like(yo)
    gxo = torch.empty_like(xo)
    wpt = ops.pack_weights(wo, True, False)
    timeit("out_conv dgrad 2->32", T, lambda: ops.conv_fwd(go, wpt, None, gxo, (1, 1, 1), 2, 32))
    dwo, dbo = torch.empty_like(wo), torch.empty(2, device=DEV)
    timeit("out_conv wgrad", T, lambda: ops.conv_wgrad(xo, go, (1, 1, 1), 32, 2, dwo, dbo))


if __name__ == "__main__":
    main()
