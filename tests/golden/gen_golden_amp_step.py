"""G10: ONE training step of the reference under its own autocast, at MFMA widths (runs only in the build container).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_amp_step.py

The reference trains under `torch.autocast(device_type, dtype=float16 | bfloat16)` with a GradScaler for float16
(torch_em/trainer/default_trainer.py:134-142, 789-803).  This script imports the reference's UNet3d and DiceLoss
(/root/reference, read-only; loader recipe of gen_golden.py) and runs ONE zero_grad + forward + loss + backward of
`UNet3d(1, 2, depth=2, initial_features=32)` on a 1x1x16x24x32 volume four ways:

    f64   the same module in float64 (the yardstick both 16-bit paths are measured against)
    f32   plain float32 (the reference's mixed_precision=False path)
    f16   torch.autocast("cpu", float16), loss scaled before backward and the gradients unscaled afterwards (what
          `scaler.scale(loss).backward(); scaler.step(...)` does, :789-794); the scale starts at GradScaler's 2^16 and is
          halved while any gradient is non-finite -- the scaler's own backoff over its first (skipped) steps
    bf16  torch.autocast("cpu", bfloat16), no loss scaling (the reference disables the scaler for bfloat16, :138-140)

and stores the inputs, a SHA-256 of the weights (torch.manual_seed(0) construction: the GPU test rebuilds them with this
repo's UNet3d, whose initialisation is pinned to the reference's by g1b_init_unet3d.npz, and checks the digest), the four
predictions and losses, the float64 gradients (as float32: 6e-8, far below any error measured here) and, per parameter
tensor, the L2 distance of the f32 / f16 / bf16 gradient from the float64 one (1.27 M parameters: the gradients themselves
would be 5 MB per run).  The fixture is data: it lets the GPU test ask "is the library's amp / amp_bf16 step as far from
float64 as the reference's autocast step is?" without the reference on the GPU box.  The CPU autocast op lists differ a little from the CUDA ones (max_pool3d / upsample run in fp32 on the CPU),
so this is the reference's autocast as THIS container can run it -- the convolutions, which carry the 16-bit products, and the
16-bit activation tensors between conv / ReLU / InstanceNorm are the same."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
LOSS_SCALE = 65536.0


def one_step(model, x, y, loss_fn, autocast_dtype=None, scale=1.0):
    model.zero_grad()
    if autocast_dtype is None:
        pred = model(x)
        loss = loss_fn(pred, y)
    else:
        with torch.autocast("cpu", dtype=autocast_dtype):
            pred = model(x)
            loss = loss_fn(pred, y)
    (loss * scale).backward()
    grads = {k: (p.grad.detach().double() / scale).numpy() for k, p in model.named_parameters()}
    return pred.detach(), float(loss.detach().double()), grads


def main():
    torch.set_num_threads(8)
    ref = load_reference()
    unet, dice = ref["unet"], ref["dice"]
    torch.manual_seed(0)
    model = unet.UNet3d(1, 2, depth=2, initial_features=32)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(1, 1, 16, 24, 32, generator=g)
    y = (torch.rand(1, 2, 16, 24, 32, generator=g) > 0.5).float()
    loss_fn = dice.DiceLoss()
    import hashlib
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    out = {"x": x.numpy(), "y": y.numpy(), "sd_sha256": np.array(h.hexdigest())}
    names = [k for k, _ in model.named_parameters()]
    out["param_names"] = np.array(names)

    m64 = unet.UNet3d(1, 2, depth=2, initial_features=32).double()
    m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    pred, loss, g64 = one_step(m64, x.double(), y.double(), loss_fn)
    out["f64.pred"], out["f64.loss"] = pred.numpy(), np.float64(loss)
    out.update({f"f64.grad.{k}": v.astype("float32") for k, v in g64.items()})
    out["f64.grad_norm"] = np.array([np.linalg.norm(g64[k]) for k in names])

    def rel(a, b):
        return float(np.linalg.norm(a.astype("float64") - b) / max(np.linalg.norm(b), 1e-300))
    for tag, dt in (("f32", None), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        scale = LOSS_SCALE if tag == "f16" else 1.0
        while True:
            pred, loss, grads = one_step(model, x, y, loss_fn, dt, scale)
            if all(np.isfinite(v).all() for v in grads.values()):
                break
            scale /= 2          # GradScaler: skip the step, halve the scale
        out[f"{tag}.pred"] = pred.float().numpy()
        out[f"{tag}.pred_dtype"] = np.array(str(pred.dtype))
        out[f"{tag}.loss"], out[f"{tag}.loss_scale"] = np.float64(loss), np.float64(scale)
        out[f"{tag}.grad_err"] = np.array([np.linalg.norm(grads[k] - g64[k]) for k in names])
        ga = np.concatenate([grads[k].ravel() for k in names])
        gb = np.concatenate([g64[k].ravel() for k in names])
        out[f"{tag}.grad_err_global"] = np.float64(rel(ga, gb))
        print(f"{tag:5s} pred rel L2 {rel(out[f'{tag}.pred'], out['f64.pred']):.3e}  loss {out[f'{tag}.loss']:.8f} "
              f"(f64 {out['f64.loss']:.8f})  gradient rel L2 {rel(ga, gb):.3e}  loss scale {scale:g}  pred dtype {out[f'{tag}.pred_dtype']}")
        for k, e, n in zip(names, out[f"{tag}.grad_err"], out["f64.grad_norm"]):
            print(f"      {k:40s} {e / n:.3e}")
    np.savez_compressed(os.path.join(OUT, "g10_amp_step.npz"), **out)


if __name__ == "__main__":
    main()
