"""Micro-benchmark of single convolution launches (for rocprofv3 PMC passes and kernel tuning).
usage: python scripts/bench_conv.py [fwd|wgrad] N D H W Cin Cout [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import _lib, ops  # noqa: E402
if os.environ.get('TEM_LIB'):
    _lib.LIB_PATH = os.environ['TEM_LIB']

kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
N, D, H, W, Cin, Cout = [int(v) for v in (sys.argv[2:8] if len(sys.argv) > 7 else (2, 128, 128, 128, 32, 32))]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
k = (3, 3, 3)
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(N, D, H, W, Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(Cout, device=dev)
scale = torch.rand(N, Cin, device=dev) + 0.5
shift = torch.randn(N, Cin, device=dev)
if os.environ.get("TEM_NOSCALE"):  # dgrad-like launch: no fused pre-norm
    scale = shift = None
y = torch.empty(N, D, H, W, Cout, device=dev)
g = torch.randn(N, D, H, W, Cout, device=dev)
MODE = int(os.environ.get("TEM_MODE", "1"))
wp = ops.pack_weights(w, False, MODE)
dw = torch.empty(w.numel(), device=dev)
db = torch.empty(Cout, device=dev)


def run():
    if kind == "fwd":
        ops.conv_fwd(x, wp, b, y, k, Cin, Cout, scale=scale, shift=shift, act="relu", mfma=MODE)
    else:
        ops.conv_wgrad(x, g, k, Cin, Cout, dw, db, scale=scale, shift=shift, mfma=MODE)


for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fl = 2.0 * N * D * H * W * Cin * Cout * 27
print(f"{kind} {N}x{D}x{H}x{W} {Cin}->{Cout}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TFLOP/s")
