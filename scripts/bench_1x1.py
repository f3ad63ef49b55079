"""1x1x1 convolution / data gradient: streaming GEMM (tem_set_option("conv1x1_stream", 1)) vs the patch kernel (0) on the
upsampler shapes of cfg 2, with a float64 check.  usage: python scripts/bench_1x1.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import _lib, ops  # noqa: E402

dev = "cuda"
shapes = [(2, 64, 64, 64, 64, 32), (2, 32, 32, 32, 128, 64), (2, 16, 16, 16, 256, 128), (2, 8, 8, 8, 512, 256)]
for (N, D, H, W, cin, cout) in shapes:
    for what, mode in (("fwd", 3), ("dgrad", 2)):
        ci, co = (cin, cout) if what == "fwd" else (cout, cin)
        torch.manual_seed(0)
        x = torch.randn(N, D, H, W, ci, device=dev)
        w = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.1
        b = torch.randn(co, device=dev) if what == "fwd" else None
        refm = torch.randn(N, D, H, W, co, device=dev) if what == "dgrad" else None
        wp = ops.pack_weights(w, what == "dgrad", mode)
        wm = w.view(cout, cin).double()
        yref = x.double().view(-1, ci) @ (wm.t() if what == "fwd" else wm)
        if b is not None:
            yref = yref + b.double()
        if refm is not None:
            yref = yref * (refm.view(-1, co) > 0)
        res = {}
        for opt in (0, 1):
            _lib.set_option("conv1x1_stream", opt)
            y = torch.full((N, D, H, W, co), float("nan"), device=dev)
            run = lambda: ops.conv_fwd(x, wp, b, y, (1, 1, 1), ci, co, ref=refm, mfma=mode)  # noqa: E731
            run()
            torch.cuda.synchronize()
            err = float((y.double().view(-1, co) - yref).abs().max() / yref.abs().max())
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            res[opt] = (best, err)
        mb = N * D * H * W * (ci + co + (co if refm is not None else 0)) * 4 / 1e6
        print(f"{what} {N}x{D}x{H}x{W} {ci}->{co} mode {mode}: patch {res[0][0]*1e3:.1f} us (err {res[0][1]:.1e})  stream "
              f"{res[1][0]*1e3:.1f} us (err {res[1][1]:.1e})  {mb / res[1][0] / 1e3:.2f} TB/s algorithmic", flush=True)
