"""The mixed modes anchored to the REFERENCE's autocast (G10, tests/golden/gen_golden_amp_step.py).

The reference trains under torch.autocast(float16 | bfloat16) (trainer/default_trainer.py:134-142, 789-803).  G10 holds ONE
step of the reference's UNet3d(1, 2, depth=2, initial_features=32) on 1x1x16x24x32 in float64, float32 and under both
autocasts: predictions, losses, the float64 gradients and the reference-autocast gradients' own L2 distance from float64 per
parameter tensor.  The library's `amp` / `amp_bf16` step (16-bit storage + one 16-bit MFMA per product) is held to the SAME
class as the reference's autocast, the criterion the fp32-class path already passes against the fp32 reference path:

  * gradient error vs float64 <= 2 x the reference-autocast's own error vs float64 -- globally and per parameter tensor
    (tensors whose float64 gradient is numerically zero -- a bias in front of an InstanceNorm -- are compared on the scale of
    the global gradient instead);
  * prediction within the 16-bit rounding bound of the reference's autocast prediction: |pred - pred_ref_autocast| <=
    2 ulp(16-bit) * max|pred|  +  the distance both have from float64;
  * loss within the reference-autocast's own distance from the float64 loss (x2).
"""
import hashlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    from torch_em_amd.model import UNet3d
    g = dict(np.load(os.path.join(GOLDEN, "g10_amp_step.npz")))
    torch.manual_seed(0)
    model = UNet3d(1, 2, depth=2, initial_features=32)
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    assert h.hexdigest() == str(g["sd_sha256"]), "the seed-0 weights differ from the ones the reference ran G10 with"
    names = [k for k, _ in model.named_parameters()]
    assert names == [str(n) for n in g["param_names"]]
    return g, model.to(DEV), names


def _library_step(model, x, y, mode, scale):
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import engine
    model.zero_grad(set_to_none=True)
    with engine.precision_scope(mode):
        assert engine.act_dtype() == (torch.float16 if mode == "amp" else torch.bfloat16)   # 16-bit STORAGE is what ships
        pred = model(x)
        loss = DiceLoss()(pred, y)
        (loss * scale).backward()
    torch.cuda.synchronize()
    grads = {k: (p.grad.detach().double() / scale).cpu().numpy() for k, p in model.named_parameters()}
    return pred.detach().double().cpu().numpy(), float(loss.detach().double()), grads


@pytest.mark.parametrize("mode,tag", [("amp", "f16"), ("amp_bf16", "bf16")])
def test_mixed_mode_step_is_in_the_class_of_the_reference_autocast(mode, tag):
    g, model, names = _load()
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    scale = float(g[f"{tag}.loss_scale"])        # the scale the reference's GradScaler settles on for this step (fp16); 1 for bf16
    pred, loss, grads = _library_step(model, x, y, mode, scale)
    assert all(np.isfinite(v).all() for v in grads.values())

    g64 = {k: g[f"f64.grad.{k}"].astype("float64") for k in names}
    norm64 = g["f64.grad_norm"]
    ref_err = g[f"{tag}.grad_err"]                        # ||g_ref_autocast - g_f64|| per tensor
    glob64 = float(np.sqrt((norm64 ** 2).sum()))
    lib_err = np.array([np.linalg.norm(grads[k] - g64[k]) for k in names])
    lib_glob, ref_glob = float(np.sqrt((lib_err ** 2).sum())) / glob64, float(g[f"{tag}.grad_err_global"])
    rows = []
    worst = 0.0
    for k, le, re, n in zip(names, lib_err, ref_err, norm64):
        # a numerically-zero float64 gradient (norm below 1e-6 of the whole): both paths produce rounding noise there;
        # the bound is the noise the reference's autocast itself shows, floored at 1e-6 of the global gradient
        bound = 2.0 * max(re, 1e-6 * glob64)
        worst = max(worst, le / bound)
        rows.append(f"  {k:40s} library {le / max(n, 1e-300):9.3e}   reference autocast {re / max(n, 1e-300):9.3e}   ratio {le / max(re, 1e-300):6.2f}")
    print(f"\n{mode}: gradient rel L2 vs float64: library {lib_glob:.3e}, reference autocast {ref_glob:.3e}\n" + "\n".join(rows))
    assert lib_glob <= 2.0 * ref_glob, (lib_glob, ref_glob)
    assert worst <= 1.0, "\n".join(rows)

    p64, pref = g["f64.pred"], g[f"{tag}.pred"].astype("float64")
    ulp = 2.0 ** -10 if tag == "f16" else 2.0 ** -7
    d_ref64 = float(np.abs(pref - p64).max())
    d_lib64 = float(np.abs(pred - p64).max())
    d_libref = float(np.abs(pred - pref).max())
    print(f"{mode}: prediction max-abs: library-f64 {d_lib64:.3e}, reference autocast-f64 {d_ref64:.3e}, library-reference "
          f"{d_libref:.3e}; rel L2 library-f64 {np.linalg.norm(pred - p64) / np.linalg.norm(p64):.3e}, "
          f"reference-f64 {np.linalg.norm(pref - p64) / np.linalg.norm(p64):.3e}")
    assert d_libref <= 2 * ulp * float(np.abs(p64).max()) + d_ref64 + d_lib64
    assert np.linalg.norm(pred - p64) <= 2.0 * np.linalg.norm(pref - p64)
    dl_ref = abs(float(g[f"{tag}.loss"]) - float(g["f64.loss"]))
    print(f"{mode}: loss library {loss:.8f}, reference autocast {float(g[f'{tag}.loss']):.8f}, float64 {float(g['f64.loss']):.8f}")
    assert abs(loss - float(g["f64.loss"])) <= 2.0 * dl_ref + 1e-6


def test_default_mode_step_against_the_reference_fp32_step():
    """the same fixture for the default (fp32-class) arithmetic: its gradient error vs float64 is within 2x the reference's own
    fp32 path's, per tensor and globally -- the criterion of tests/test_gpu_unet.py on reference-held numbers"""
    g, model, names = _load()
    from torch_em_amd.loss import DiceLoss
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    model.zero_grad(set_to_none=True)
    pred = model(x)
    loss = DiceLoss()(pred, y)
    loss.backward()
    torch.cuda.synchronize()
    norm64, ref_err = g["f64.grad_norm"], g["f32.grad_err"]
    glob64 = float(np.sqrt((norm64 ** 2).sum()))
    lib_err = np.array([np.linalg.norm(p.grad.double().cpu().numpy() - g[f"f64.grad.{k}"].astype("float64"))
                        for k, p in model.named_parameters()])
    lib_glob = float(np.sqrt((lib_err ** 2).sum())) / glob64
    print(f"\ndefault: gradient rel L2 vs float64: library {lib_glob:.3e}, reference fp32 {float(g['f32.grad_err_global']):.3e}")
    # float32 storage of the f64 gradients in the fixture: 6e-8 relative, added to the bound
    assert lib_glob <= 2.0 * float(g["f32.grad_err_global"]) + 1e-6
    for k, le, re, n in zip(names, lib_err, ref_err, norm64):
        assert le <= 2.0 * max(re, 2e-4 * n, 1e-6 * glob64), (k, le / max(n, 1e-300), re / max(n, 1e-300))
    assert abs(float(loss) - float(g["f64.loss"])) < 1e-5
    assert float(np.abs(pred.detach().double().cpu().numpy() - g["f64.pred"]).max()) < 1e-4 * float(np.abs(g["f64.pred"]).max())


@pytest.mark.parametrize("hip_graph", [False, True])
def test_fp16_storage_trainer_survives_an_overflow_backoff_cycle(hip_graph, tmp_path):
    """fp16-STORED data gradients under the GradScaler: the scale starts so high that the first steps overflow in
    the stored gradients; every such step must be skipped (parameters bit-identical), the scale must halve each time, and
    training must continue with finite parameters once it fits -- eagerly and as a replayed HIP graph."""
    import torch_em_amd
    from torch_em_amd.model import UNet3d
    from torch_em_amd.optim import GradScaler
    torch.manual_seed(0)
    model = UNet3d(1, 2, depth=2, initial_features=32)
    g = torch.Generator().manual_seed(10)
    xs = torch.randn(6, 1, 16, 24, 32, generator=g)
    ys = (torch.rand(6, 2, 16, 24, 32, generator=g) > 0.5).float()
    ds = torch.utils.data.TensorDataset(xs, ys)
    train = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    val = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xs[:1], ys[:1]), batch_size=1)
    trainer = torch_em_amd.default_segmentation_trainer(
        "overflow", model, train, val, learning_rate=1e-4, device=DEV, mixed_precision=True, mixed_precision_dtype="float16",
        save_root=str(tmp_path), hip_graph=hip_graph)
    # (the reference's CPU autocast overflows this net at 2^16 already, G10: its fp16 tensors start at the loss; here the Dice
    #  gradient and the out_conv backward are fp32 and the first STORED fp16 gradient is 1e-5-sized: 2^22 still fits, measured)
    start = 2.0 ** 36
    trainer.scaler = GradScaler(init_scale=start)
    before = torch.cat([p.detach().flatten().clone() for p in model.parameters()])
    scales, moved = [], []
    for it in range(1, 31):
        trainer.fit(iterations=1)
        torch.cuda.synchronize()
        now = torch.cat([p.detach().flatten() for p in trainer.model.parameters()])
        assert bool(torch.isfinite(now).all()), it
        scales.append(trainer.scaler.get_scale())
        moved.append(not torch.equal(now.cpu(), before.cpu()))
        before = now.clone()
    print(f"\nhip_graph={hip_graph}: scales {[int(np.log2(s)) for s in scales]}, parameter moved {moved}")
    n_skipped = moved.index(True)
    assert 1 <= n_skipped <= 26, (scales, moved)                 # at least the first step overflowed, and it recovered
    assert all(moved[n_skipped:])                                   # once the scale fits every step is applied
    assert scales[:n_skipped] == [start / 2 ** (i + 1) for i in range(n_skipped)]   # halved once per skipped step
    assert scales[-1] == start / 2 ** n_skipped                     # and untouched afterwards (growth interval 2000)
    if hip_graph:
        assert trainer._graphed is not None, trainer._graph_why
