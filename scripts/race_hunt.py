"""Repeat the cfg-3 step many times and compare every result bit for bit with the first one (races / uninitialised reads
show up as rare mismatches).  usage: python scripts/race_hunt.py [repeats] [poison]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
from torch_em_amd.model import AnisotropicUNet
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
poison = len(sys.argv) > 2
_empty, _empty_like = torch.empty, torch.empty_like
FILL = [None]
def _p(t):
    if FILL[0] is not None and t.is_cuda and t.is_floating_point():
        t.fill_(FILL[0])
    return t
if poison:
    torch.empty = lambda *a, **k: _p(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _p(_empty_like(*a, **k))
dev = "cuda"
torch.manual_seed(0)
g = torch.Generator().manual_seed(0)
sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
m = AnisotropicUNet(1, 12, scale_factors=sf, initial_features=32, final_activation="Sigmoid").to(dev)
x = torch.randn(2, 1, 64, 256, 256, generator=g).to(dev)
y = (torch.rand(2, 24, 64, 256, 256, generator=g) > 0.5).float().to(dev)
loss_fn = LossWrapper(DiceLoss(), ApplyAndRemoveMask(masking_method="multiply"))
ref = None
names = [k for k, _ in m.named_parameters()]
bad_total = 0
for i in range(reps):
    FILL[0] = None if not poison else (float("nan") if i % 2 else 3e38)
    m.zero_grad()
    pred = m(x)
    loss = loss_fn(pred, y)
    loss.backward()
    cur = (float(loss.detach()), pred.detach().clone(), [p.grad.clone() for p in m.parameters()])
    if ref is None:
        ref = cur
        continue
    bad = [k for k, a, b in zip(names, cur[2], ref[2]) if not torch.equal(a, b)]
    if cur[0] != ref[0] or not torch.equal(cur[1], ref[1]) or bad:
        bad_total += 1
        print(f"rep {i}: MISMATCH loss {cur[0]!r} vs {ref[0]!r} pred_equal={torch.equal(cur[1], ref[1])} grads={bad[:6]}")
print(f"{reps} repeats, {bad_total} mismatches, poison={poison}")
