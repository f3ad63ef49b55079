"""SHA-256 digests of one training step in the DEFAULT (fp32-class) arithmetic: prediction, loss, flat parameter gradient.

Every kernel of the library is deterministic (no floating-point atomics, fixed reduction orders), so the digests are a
property of the source tree and the chip, not of the run.  tests/golden/default_step_digest.json holds the digests of the
tree at the end of round 4; tests/test_gpu_unet.py::test_default_arithmetic_is_bit_identical_to_round4 recomputes them --
the 16-bit activation storage of round 5 must not move a single bit of the fp32-class path.
usage: python scripts/digest_default.py [--write tests/golden/default_step_digest.json]   (TEM_LIB=<older .so> for an A/B)"""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

CASES = {
    # name: (model kwargs, input shape, pooling note)
    "cfg2_2x128": (dict(in_channels=1, out_channels=2, depth=4, initial_features=32), (2, 1, 128, 128, 128)),
    "gn_depth3_1x32x48x40": (dict(in_channels=1, out_channels=2, depth=3, initial_features=32, norm="GroupNorm"), (1, 1, 32, 48, 40)),
    "depth2_2x16": (dict(in_channels=1, out_channels=3, depth=2, initial_features=16), (2, 1, 16, 16, 16)),
}


def digest(t):
    return hashlib.sha256(t.detach().float().contiguous().cpu().numpy().tobytes()).hexdigest()


def run_case(kw, shape):
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    torch.manual_seed(11)
    model = UNet3d(**kw).to("cuda")
    g = torch.Generator().manual_seed(12)
    x = torch.randn(*shape, generator=g).to("cuda")
    y = (torch.rand(shape[0], kw["out_channels"], *shape[2:], generator=g) > 0.5).float().to("cuda")
    out = {}
    for rep in range(2):   # the second step runs on the batched weight re-pack: both must agree
        model.zero_grad(set_to_none=True)
        pred = model(x)
        loss = DiceLoss()(pred, y)
        loss.backward()
        torch.cuda.synchronize()
        grads = torch.cat([p.grad.flatten() for p in model.parameters()])
        cur = {"pred": digest(pred), "loss": float(loss).hex(), "grads": digest(grads)}
        assert rep == 0 or cur == out, "two identical steps differ"
        out = cur
    return out


def main():
    res = {name: run_case(kw, shape) for name, (kw, shape) in CASES.items()}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 2 and sys.argv[1] == "--write":
        with open(sys.argv[2], "w") as f:
            json.dump(res, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
