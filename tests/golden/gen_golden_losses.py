"""Golden vectors for the remaining Dice-family losses by IMPORTING THE REFERENCE (build container only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_losses.py

DiceLossWithLogits, BCEDiceLoss, BCEDiceLossWithLogits (reference loss/dice.py:136-253): inputs, loss values and the
gradient w.r.t. the prediction for several constructor arguments -> g5b_dice_variants.npz (data only)."""
import os

import numpy as np
import torch

from gen_golden import OUT, load_reference


def main():
    torch.set_num_threads(4)
    dice = load_reference()["dice"]
    g = torch.Generator().manual_seed(31)
    logits = torch.randn(2, 3, 5, 6, 7, generator=g) * 2.0
    logits[0, 0, 0, 0, :3] = torch.tensor([40.0, -40.0, 0.0])       # saturated sigmoids / BCE log clamps
    probs = torch.sigmoid(torch.randn(2, 3, 5, 6, 7, generator=g) * 2.0)
    probs[0, 1, 0, 0, :2] = torch.tensor([0.0, 1.0])                # log(0) clamp of binary_cross_entropy
    target = (torch.rand(2, 3, 5, 6, 7, generator=g) > 0.5).float()
    out = dict(logits=logits.numpy(), probs=probs.numpy(), target=target.numpy())
    cases = {
        "dwl_default": (dice.DiceLossWithLogits(), logits),
        "dwl_mean": (dice.DiceLossWithLogits(reduce_channel="mean"), logits),
        "dwl_flat": (dice.DiceLossWithLogits(channelwise=False), logits),
        "bce_default": (dice.BCEDiceLoss(), probs),
        "bce_weighted": (dice.BCEDiceLoss(alpha=0.7, beta=1.3), probs),
        "bce_flat": (dice.BCEDiceLoss(alpha=0.5, beta=2.0, channelwise=False), probs),
        "bcel_default": (dice.BCEDiceLossWithLogits(), logits),
        "bcel_weighted": (dice.BCEDiceLossWithLogits(alpha=1.5, beta=0.25, channelwise=False), logits),
    }
    for name, (loss, inp) in cases.items():
        x = inp.clone().requires_grad_(True)
        val = loss(x, target)
        val.backward()
        out[f"{name}.loss"] = np.float32(val.item())
        out[f"{name}.grad"] = x.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g5b_dice_variants.npz"), **out)
    print("wrote g5b_dice_variants.npz:", {k: float(v) for k, v in out.items() if k.endswith(".loss")})


if __name__ == "__main__":
    main()
