"""Same launch as bench_conv.py fwd, with selectable operand data (power / DVFS sensitivity probe).
usage: python scripts/bench_conv_zero.py {rand|zero|small} MODE [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import ops  # noqa: E402
kind, MODE = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N, D, H, W, Cin, Cout = 2, 128, 128, 128, 32, 32
torch.manual_seed(0)
x = torch.randn(N, D, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
if kind == "zero":
    x.zero_(); w.zero_()
elif kind == "xzero":
    x.zero_()
elif kind == "wzero":
    w.zero_()
b = torch.zeros(Cout, device="cuda")
y = torch.empty(N, D, H, W, Cout, device="cuda")
wp = ops.pack_weights(w, False, MODE)
def run():
    ops.conv_fwd(x, wp, b, y, (3, 3, 3), Cin, Cout, act="relu", mfma=MODE)
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(iters): run()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / iters * 1e3)
print(kind, "MODE", MODE, "ms per launch:", " ".join(f"{t:.3f}" for t in ts))
