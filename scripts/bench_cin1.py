"""First-layer (Cin = 1) kernels: time fwd and wgrad at 2x128^3, 1 -> 32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd import ops
N, D, C = 2, 128, 32
x = torch.randn(N, D, D, D, 1, device="cuda")
w = torch.randn(C, 1, 3, 3, 3, device="cuda") * 0.1
b = torch.randn(C, device="cuda")
y = torch.empty(N, D, D, D, C, device="cuda")
g = torch.randn(N, D, D, D, C, device="cuda")
wp = ops.pack_weights(w, False, 0)
dw = torch.empty(w.numel(), device="cuda"); db = torch.empty(C, device="cuda")
sc = torch.rand(N, 1, device="cuda") + 0.5; sh = torch.randn(N, 1, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("fwd  cin1 %.3f ms" % t(lambda: ops.conv_fwd(x, wp, b, y, (3, 3, 3), 1, C, scale=sc, shift=sh, act="relu", mfma=0)))
print("wgrad cin1 %.3f ms" % t(lambda: ops.conv_wgrad(x, g, (3, 3, 3), 1, C, dw, db, scale=sc, shift=sh, mfma=0)))
