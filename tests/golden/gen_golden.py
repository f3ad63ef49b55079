"""Generate golden vectors by IMPORTING THE REFERENCE (runs only in the build container).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden.py

Reads /root/reference (read-only), writes small .npz fixtures next to this file.  The fixtures are
data (inputs + the reference's outputs); no reference source is copied.  Loader recipe: register
empty namespace modules `torch_em`, `torch_em.model`, `torch_em.loss` and load the reference files
by path (route A) -- `import torch_em` itself needs packages this image lacks.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/torch_em"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    for name, path in (("torch_em", REF), ("torch_em.model", REF + "/model"), ("torch_em.loss", REF + "/loss")):
        mod = types.ModuleType(name)
        mod.__path__ = [path]
        sys.modules[name] = mod
    out = {}
    for name, rel in (("torch_em.model.unet", "model/unet.py"), ("torch_em.loss.dice", "loss/dice.py"),
                      ("torch_em.loss.wrapper", "loss/wrapper.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        out[name.split(".")[-1]] = mod
    return out


def run_model(model, x, y, loss):
    model.zero_grad()
    pred = model(x)
    val = loss(pred, y)
    val.backward()
    sd = {f"sd.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
    grads = {f"grad.{k}": p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
    return dict(x=x.numpy(), y=y.numpy(), pred=pred.detach().numpy(), loss=np.float32(val.item()), **sd, **grads)


def main():
    torch.set_num_threads(4)
    ref = load_reference()
    unet, dice, wrapper = ref["unet"], ref["dice"], ref["wrapper"]

    # G1: UNet3d, three norms
    for norm in ("InstanceNorm", "GroupNorm", None):
        torch.manual_seed(0)
        model = unet.UNet3d(1, 2, depth=2, initial_features=4, norm=norm)
        if norm == "GroupNorm":  # make the affine parameters non-trivial
            g = torch.Generator().manual_seed(7)
            for m in model.modules():
                if isinstance(m, torch.nn.GroupNorm):
                    m.weight.data = 1.0 + 0.2 * torch.randn(m.weight.shape, generator=g)
                    m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(1, 1, 16, 16, 16, generator=g)
        y = (torch.rand(1, 2, 16, 16, 16, generator=g) > 0.5).float()
        np.savez_compressed(os.path.join(OUT, f"g1_unet3d_{norm}.npz"), **run_model(model, x, y, dice.DiceLoss()))

    # G1b: init parity -- state_dict of the default-seed construction only (no forward)
    torch.manual_seed(0)
    model = unet.UNet3d(1, 2, depth=2, initial_features=4)
    np.savez_compressed(os.path.join(OUT, "g1b_init_unet3d.npz"),
                        **{k: v.numpy() for k, v in model.state_dict().items()})

    # G2: AnisotropicUNet + Sigmoid + masked Dice (the reference's affinity loss, cli.py:263-267)
    for aniso in (False, True):
        torch.manual_seed(0)
        model = unet.AnisotropicUNet(1, 12, [[1, 2, 2], [2, 2, 2]], initial_features=4, final_activation="Sigmoid",
                                     anisotropic_kernel=aniso)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(1, 1, 8, 16, 16, generator=g)
        t = (torch.rand(1, 12, 8, 16, 16, generator=g) > 0.5).float()
        m = (torch.rand(1, 12, 8, 16, 16, generator=g) > 0.3).float()
        y = torch.cat([t, m], dim=1)
        loss = wrapper.LossWrapper(dice.DiceLoss(), transform=wrapper.ApplyAndRemoveMask(masking_method="multiply"))
        np.savez_compressed(os.path.join(OUT, f"g2_aniso_{int(aniso)}.npz"), **run_model(model, x, y, loss))

    # G3: UNet2d
    torch.manual_seed(0)
    model = unet.UNet2d(1, 2, depth=2, initial_features=4)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1, 32, 32, generator=g)
    y = (torch.rand(2, 2, 32, 32, generator=g) > 0.5).float()
    np.savez_compressed(os.path.join(OUT, "g3_unet2d.npz"), **run_model(model, x, y, dice.DiceLoss()))

    # G5: Dice value + gradient, all reductions, plus the masked variant and the reference KATs
    g = torch.Generator().manual_seed(5)
    out = {}
    p = torch.rand(2, 3, 8, 8, 8, generator=g)
    t = (torch.rand(2, 3, 8, 8, 8, generator=g) > 0.5).float()
    out["p"], out["t"] = p.numpy(), t.numpy()
    for cw in (True, False):
        for red in ("sum", "mean", "max", "min"):
            pp = p.clone().requires_grad_(True)
            val = dice.DiceLoss(channelwise=cw, reduce_channel=red)(pp, t)
            val.backward()
            out[f"loss_{int(cw)}_{red}"] = np.float32(val.item())
            out[f"grad_{int(cw)}_{red}"] = pp.grad.numpy()
    out["score_none"] = dice.dice_score(p, t, reduce_channel=None).numpy()
    pm = torch.rand(1, 12, 4, 8, 8, generator=g).requires_grad_(True)
    tm = torch.cat([(torch.rand(1, 12, 4, 8, 8, generator=g) > 0.5).float(),
                    (torch.rand(1, 12, 4, 8, 8, generator=g) > 0.4).float()], dim=1)
    loss = wrapper.LossWrapper(dice.DiceLoss(), transform=wrapper.ApplyAndRemoveMask(masking_method="multiply"))
    val = loss(pm, tm)
    val.backward()
    out["pm"], out["tm"], out["loss_masked"], out["grad_masked"] = \
        pm.detach().numpy(), tm.numpy(), np.float32(val.item()), pm.grad.numpy()
    ones, zeros = torch.ones(1, 1, 32, 32), torch.zeros(1, 1, 32, 32)
    out["kat_ones_ones"] = np.float32(dice.DiceLoss()(ones, ones).item())    # test/loss/test_dice.py:25-31
    out["kat_ones_zeros"] = np.float32(dice.DiceLoss()(ones, zeros).item())  # test/loss/test_dice.py:33-38
    np.savez_compressed(os.path.join(OUT, "g5_dice.npz"), **out)

    # G4: side outputs (UNetBase._apply_with_side_outputs): list of per-level outputs, full resolution first
    torch.manual_seed(0)
    model = unet.UNet3d(1, 2, depth=2, initial_features=4, return_side_outputs=True, final_activation="Sigmoid")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 1, 8, 16, 16, generator=g)
    ys = [(torch.rand(1, 2, 8 // f, 16 // f, 16 // f, generator=g) > 0.5).float() for f in (1, 2)]
    model.zero_grad()
    outs = model(x)
    val = sum(dice.DiceLoss()(o, y) for o, y in zip(outs, ys))
    val.backward()
    res = dict(x=x.numpy(), loss=np.float32(val.item()))
    for i, (o, y) in enumerate(zip(outs, ys)):
        res[f"out{i}"], res[f"y{i}"] = o.detach().numpy(), y.numpy()
    res.update({f"sd.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()})
    res.update({f"grad.{k}": p.grad.detach().numpy().copy() for k, p in model.named_parameters()})
    np.savez_compressed(os.path.join(OUT, "g4_side_outputs.npz"), **res)

    # G1c: stateful norms (BatchNorm, InstanceNormTrackStats): training step (outputs, gradients, updated running
    # statistics), then the eval-mode forward that uses them
    for norm in ("BatchNorm", "InstanceNormTrackStats"):
        torch.manual_seed(0)
        model = unet.UNet3d(1, 2, depth=2, initial_features=4, norm=norm)
        g = torch.Generator().manual_seed(11)
        for m in model.modules():  # non-trivial affine parameters
            if isinstance(m, (torch.nn.modules.batchnorm._NormBase,)) and m.weight is not None:
                m.weight.data = 1.0 + 0.3 * torch.randn(m.weight.shape, generator=g)
                m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=g)
        x = torch.randn(2, 1, 8, 16, 16, generator=g) * 1.5 + 0.3
        y = (torch.rand(2, 2, 8, 16, 16, generator=g) > 0.5).float()
        sd0 = {f"sd.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
        model.train()
        res = run_model(model, x, y, dice.DiceLoss())
        res = {k: v for k, v in res.items() if not k.startswith("sd.")}
        res.update(sd0)
        res.update({f"after.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()
                    if "running" in k or "num_batches" in k})
        model.eval()
        with torch.no_grad():
            res["pred_eval"] = model(x).numpy()
        np.savez_compressed(os.path.join(OUT, f"g1c_unet3d_{norm}.npz"), **res)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
