"""CPU: pin the oracle (oracle/*.py) to the reference through the golden vectors that
tests/golden/gen_golden.py produced by importing the reference itself."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import label_ref, loss_ref, optim_ref, unet_ref


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def _sd(g):
    return {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}


@pytest.mark.parametrize("norm", ["InstanceNorm", "GroupNorm", None])
def test_unet3d_oracle_matches_reference(norm):
    g = _load(f"g1_unet3d_{norm}.npz")
    pred, loss, grads = unet_ref.unet_loss_and_grads(_sd(g), torch.from_numpy(g["x"]), torch.from_numpy(g["y"]),
                                                     [2, 2], norm=norm)
    assert rel_err(pred, g["pred"]) < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for k, v in grads.items():
        assert rel_err(v, g[f"grad.{k}"]) < 1e-4, k


@pytest.mark.parametrize("aniso", [0, 1])
def test_aniso_oracle_matches_reference(aniso):
    g = _load(f"g2_aniso_{aniso}.npz")
    pred, loss, grads = unet_ref.unet_loss_and_grads(
        _sd(g), torch.from_numpy(g["x"]), torch.from_numpy(g["y"]), [[1, 2, 2], [2, 2, 2]],
        final_activation="Sigmoid", loss_fn=loss_ref.masked_dice_loss)
    assert rel_err(pred, g["pred"]) < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for k, v in grads.items():
        assert rel_err(v, g[f"grad.{k}"]) < 1e-4, k


def test_crop_oracle_matches_reference():
    """Decoder._crop (reference model/unet.py:363-373): 12^3 upsampled against a 14^3 skip tensor (check_shape = False)."""
    g = _load("g2c_crop.npz")
    pred, loss, grads = unet_ref.unet_loss_and_grads(_sd(g), torch.from_numpy(g["x"]), torch.from_numpy(g["y"]),
                                                     [[3, 3, 3], [2, 2, 2]])
    assert tuple(pred.shape) == (2, 2, 12, 12, 12) and rel_err(pred, g["pred"]) < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for k, v in grads.items():
        assert rel_err(v, g[f"grad.{k}"]) < 1e-4, k


def test_unet2d_oracle_matches_reference():
    g = _load("g3_unet2d.npz")
    pred, loss, grads = unet_ref.unet_loss_and_grads(_sd(g), torch.from_numpy(g["x"]), torch.from_numpy(g["y"]),
                                                     [2, 2])
    assert rel_err(pred, g["pred"]) < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for k, v in grads.items():
        assert rel_err(v, g[f"grad.{k}"]) < 1e-4, k


def test_dice_oracle_matches_reference():
    g = _load("g5_dice.npz")
    p, t = torch.from_numpy(g["p"]), torch.from_numpy(g["t"])
    for cw in (True, False):
        for red in ("sum", "mean", "max", "min"):
            pp = p.clone().requires_grad_(True)
            val = loss_ref.dice_loss(pp, t, channelwise=cw, reduce_channel=red)
            val.backward()
            assert abs(float(val) - float(g[f"loss_{int(cw)}_{red}"])) < 1e-6
            assert rel_err(pp.grad, g[f"grad_{int(cw)}_{red}"]) < 1e-5
    assert rel_err(loss_ref.dice_score(p, t, reduce_channel=None), g["score_none"]) < 1e-6
    pm = torch.from_numpy(g["pm"]).requires_grad_(True)
    val = loss_ref.masked_dice_loss(pm, torch.from_numpy(g["tm"]))
    val.backward()
    assert abs(float(val) - float(g["loss_masked"])) < 1e-6
    assert rel_err(pm.grad, g["grad_masked"]) < 1e-5
    # the reference's own known answers (test/loss/test_dice.py:25-38)
    ones, zeros = torch.ones(1, 1, 32, 32), torch.zeros(1, 1, 32, 32)
    assert abs(float(loss_ref.dice_loss(ones, ones)) - float(g["kat_ones_ones"])) < 1e-7
    assert abs(float(loss_ref.dice_loss(ones, zeros)) - float(g["kat_ones_zeros"])) < 1e-7
    assert float(g["kat_ones_ones"]) == 0.0 and float(g["kat_ones_zeros"]) == 1.0


# --- label targets: the reference's own brute-force definitions are the vectors ----------
OFFSETS_2D = [[-1, 0], [0, -1], [-3, 0], [0, -3], [4, 5], [-3, 2]]  # test_label_transforms.py:70-72


def _labels(shape, with_zero, seed):
    rng = np.random.RandomState(seed)
    lab = rng.randint(1, 6, size=shape).astype("int64")
    if with_zero:
        lab[rng.rand(*shape) < 0.25] = 0
    return lab


def test_affinities_vectorised_equals_brute_force_2d():
    seg = _labels((64, 64), False, 0)
    affs = label_ref.affinities(seg, OFFSETS_2D)
    exp, _ = label_ref.affinities_brute_force(seg, OFFSETS_2D)
    assert np.array_equal(affs, exp)
    seg = _labels((64, 64), True, 1)
    for incl in (False, True):
        out = label_ref.affinities(seg, OFFSETS_2D, ignore_label=0, add_mask=True, include_ignore_transitions=incl)
        ea, em = label_ref.affinities_brute_force(seg, OFFSETS_2D, ignore_label=0, include_ignore_transitions=incl)
        assert np.array_equal(out[:6], ea) and np.array_equal(out[6:], em)


def test_affinities_3d_and_channels():
    seg = _labels((6, 9, 10), True, 2)
    offs = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [1, 2, -3]]
    out = label_ref.affinities(seg, offs, ignore_label=0, add_binary_target=True, add_mask=True)
    ea, em = label_ref.affinities_brute_force(seg, offs, ignore_label=0)
    n = len(offs)
    assert out.shape == (2 * (n + 1),) + seg.shape
    assert np.array_equal(out[0], (seg != 0).astype("float32"))
    assert np.array_equal(out[1:n + 1], ea)
    assert np.array_equal(out[n + 1], (seg != 0).astype("float32"))
    assert np.array_equal(out[n + 2:], em)


def test_boundaries_kat_and_morphology():
    lab = np.zeros((5, 6), "int64")
    lab[:, 3:] = 2
    b = label_ref.boundaries(lab)[0]
    exp = np.zeros((5, 6), "float32")
    exp[:, 2:4] = 1
    assert np.array_equal(b, exp)  # 'thick': both sides of the edge, nothing at the image border
    for shape, seed in (((32, 32), 3), ((8, 12, 10), 4)):
        lab = _labels(shape, True, seed)
        assert np.array_equal(label_ref.boundaries(lab), label_ref.boundaries_morphology(lab))
    out = label_ref.boundaries(lab, add_binary_target=True)
    assert out.shape == (2,) + lab.shape and np.array_equal(out[0], (lab != 0).astype("float32"))


def test_boundary_modes_known_answers():
    """The three same-shape modes on the example of scikit-image's find_boundaries docstring (its published outputs)."""
    lab = np.zeros((9, 10), "int64")
    lab[2:7, 5:8] = 5
    lab[3:6, 2:5] = 1
    rows = {"thick": ["0000000000", "0000011100", "0011111110", "0111110110", "0110110110", "0111110110", "0011111110",
                      "0000011100", "0000000000"],
            "inner": ["0000000000", "0000000000", "0000011100", "0011110100", "0010110100", "0011110100", "0000011100",
                      "0000000000", "0000000000"],
            "outer": ["0000000000", "0000011100", "0011110010", "0100110010", "0100110010", "0100110010", "0011110010",
                      "0000011100", "0000000000"]}
    for mode, r in rows.items():
        want = np.array([[int(c) for c in line] for line in r], "float32")
        assert np.array_equal(label_ref.boundaries_mode(lab, mode)[0], want), mode
        assert np.array_equal(label_ref.boundaries_morphology(lab, mode)[0], want), mode
    for shape, seed in (((32, 32), 3), ((8, 12, 10), 4), ((1, 9), 5)):
        lab = _labels(shape, True, seed)
        lab[lab == lab.max()] = -1                       # an ignore label below the background
        for mode in rows:
            assert np.array_equal(label_ref.boundaries_mode(lab, mode), label_ref.boundaries_morphology(lab, mode))


def test_adamw_oracle_matches_torch():
    rng = np.random.RandomState(0)
    p0 = rng.randn(1000).astype("float32")
    p_t = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([p_t], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = rng.randn(1000).astype("float32")
        p_t.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m, v = optim_ref.adamw_step(p, g, m, v, step)
        assert rel_err(p, p_t.detach().numpy()) < 1e-6


def test_side_outputs_golden():
    """return_side_outputs=True (reference model/unet.py:211-228): outputs, loss and every gradient."""
    import torch
    from oracle import loss_ref, unet_ref
    g = np.load(os.path.join(GOLDEN, "g4_side_outputs.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sd.")}
    outs = unet_ref.unet_forward(sd, torch.from_numpy(g["x"]), [2, 2], final_activation="Sigmoid")
    assert [tuple(o.shape) for o in outs] == [(1, 2, 8, 16, 16), (1, 2, 4, 8, 8)]  # full resolution first
    val = sum(loss_ref.dice_loss(o, torch.from_numpy(g[f"y{i}"])) for i, o in enumerate(outs))
    val.backward()
    assert abs(float(val) - float(g["loss"])) < 1e-6
    for i, o in enumerate(outs):
        assert float((o.detach() - torch.from_numpy(g[f"out{i}"])).abs().max()) < 1e-6
    for k in g.files:
        if k.startswith("grad."):
            assert float((sd[k[5:]].grad - torch.from_numpy(g[k])).abs().max()) < 1e-6, k


@pytest.mark.parametrize("norm", ["BatchNorm", "InstanceNormTrackStats"])
def test_stateful_norm_golden(norm):
    """BatchNorm / InstanceNormTrackStats (reference get_norm_layer, model/unet.py:391-406): training forward,
    gradients, the running statistics after the step, and the eval-mode forward that uses them."""
    from oracle import loss_ref, unet_ref
    g = np.load(os.path.join(GOLDEN, f"g1c_unet3d_{norm}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd.")}
    for k, v in sd.items():
        if v.dtype == torch.float32 and "running" not in k:
            v.requires_grad_(True)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    pred = unet_ref.unet_forward(sd, x, [2, 2], norm=norm)
    val = loss_ref.dice_loss(pred, y)
    val.backward()
    assert float((pred.detach() - torch.from_numpy(g["pred"])).abs().max()) < 1e-6
    for k in g.files:
        if k.startswith("grad."):
            assert float((sd[k[5:]].grad - torch.from_numpy(g[k])).abs().max()) < 1e-6, k
        if k.startswith("after.") and "running" in k:
            assert float((sd[k[6:]] - torch.from_numpy(g[k])).abs().max()) < 1e-6, k
    with torch.no_grad():
        pe = unet_ref.unet_forward(sd, x, [2, 2], norm=norm, training=False)
    assert float((pe - torch.from_numpy(g["pred_eval"])).abs().max()) < 1e-6


def _g7_oracle_run(g):
    """The oracle's training loop (what tests/test_gpu_trainer.py runs beside the HIP trainer): oracle/unet_ref.py +
    oracle/loss_ref.py + torch.optim.AdamW / ReduceLROnPlateau as default_segmentation_trainer configures them
    (reference segmentation.py:543-544), over G7's batches in G7's order."""
    params = {k[4:]: torch.from_numpy(v).clone().requires_grad_(v.dtype.kind == "f") for k, v in g.items()
              if k.startswith("sd0.")}
    opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=float(g["learning_rate"]))
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=5)
    xt, yt, xv, yv = (torch.from_numpy(g[k]) for k in ("xt", "yt", "xv", "yv"))
    losses, lrs, metrics = [], [], []
    for _epoch in range(2):
        for i in range(0, xt.shape[0], 2):
            opt.zero_grad()
            loss = loss_ref.dice_loss(unet_ref.unet_forward(params, xt[i:i + 2], [2, 2]), yt[i:i + 2])
            loss.backward()
            opt.step()
            losses.append(float(loss))
            lrs.append(opt.param_groups[0]["lr"])
        with torch.no_grad():
            m = np.mean([float(loss_ref.dice_loss(unet_ref.unet_forward(params, xv[i:i + 2], [2, 2], training=False),
                                                  yv[i:i + 2])) for i in range(0, xv.shape[0], 2)])
        metrics.append(m)
        sched.step(m)
    return losses, lrs, metrics, params


def test_oracle_training_loop_matches_reference_trainer():
    """G7: the reference's DefaultTrainer.fit(8) run by tests/golden/gen_golden_trainer.py (route-B import of torch_em):
    loss of every iteration, learning rate, validation metric of every epoch, final parameters."""
    g = _load("g7_trainer_unet2d.npz")
    losses, lrs, metrics, params = _g7_oracle_run(g)
    assert np.allclose(losses, g["train_loss"], rtol=0, atol=2e-5), (losses, g["train_loss"])
    assert np.allclose(lrs, g["lr"])
    assert np.allclose(metrics, g["val_metric"], atol=2e-5)
    for k, v in params.items():
        if "samplers" in k and k.endswith("bias"):
            continue  # mathematically zero gradient in front of an InstanceNorm: Adam turns round-off into +-lr steps
        a, b = v.detach().double(), torch.from_numpy(g[f"sd1.{k}"]).double()
        assert float((a - b).norm() / b.norm()) < 2e-3, k
    assert int(g["iteration"]) == 8 and int(g["epoch"]) == 1  # the checkpoint is written before the epoch counter moves


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_affinities_match_the_reference_definitions(tag):
    """G8: outputs of the reference's own affs_brute_force / affs_brute_force_with_mask
    (test/transform/test_label_transforms.py:5-55), which its tests require AffinityTransform to reproduce."""
    g = _load("g8_affinities_bruteforce.npz")
    seg, offs = g[f"{tag}_seg"], [list(map(int, o)) for o in g[f"{tag}_offsets"]]
    n = len(offs)
    assert np.array_equal(label_ref.affinities(seg, offs), g[f"{tag}_affs"])
    out = label_ref.affinities(seg, offs, ignore_label=0, add_mask=True)
    assert np.array_equal(out[:n], g[f"{tag}_affs_ignore0"]) and np.array_equal(out[n:], g[f"{tag}_mask_ignore0"])
    out = label_ref.affinities(seg, offs, ignore_label=0, add_mask=True, include_ignore_transitions=True)
    assert np.array_equal(out[:n], g[f"{tag}_affs_trans"]) and np.array_equal(out[n:], g[f"{tag}_mask_trans"])


def test_predict_oracle_matches_the_reference_helpers():
    """G9 = outputs of the reference's own `_load_block`, `_pad_for_shift_left` / `_crop_after_shift_left`,
    `_prepare_block_input` and `_write_prediction` (util/prediction.py:79-142, 388-447; tests/golden/gen_golden_predict.py):
    the halo / reflect / inner-crop / mask / channel-split semantics of the tiled prediction are pinned; only the block grid
    (`bioimage_cpp.utils.Blocking`, not installed) stays restated."""
    from oracle import predict_ref
    g = dict(np.load(os.path.join(GOLDEN, "g9_predict_helpers.npz")))
    for case in g["load_block_cases"]:
        name, key, wc = str(case).split("|")
        off, bs, ha = (list(map(int, r)) for r in g[name + ".args"])
        if key == "vol2":
            off, bs, ha = off[:2], bs[:2], ha[:2]
        data, bb = predict_ref.load_block_bb(g[key], off, bs, ha, with_channels=bool(int(wc)))
        assert np.array_equal(data, g[name + ".data"]), name
        assert [list(b) for b in bb] == g[name + ".bb"].tolist(), name
    padded, pad_left = predict_ref.pad_for_shift_left(g["vol3"], g["shift.pad_left"], False)
    assert np.array_equal(padded, g["shift.pad"])
    assert np.array_equal(predict_ref.crop_after_shift_left(padded, pad_left, False, g["vol3"].shape), g["shift.crop"])
    padded_c, pad_left_c = predict_ref.pad_for_shift_left(g["vol3c"], (2, 1, 0), True)
    assert np.array_equal(padded_c, g["shift.pad_c"])
    assert np.array_equal(predict_ref.crop_after_shift_left(padded_c, pad_left_c, True, g["vol3c"].shape[1:]), g["shift.crop_c"])
    for case in g["prepare_cases"]:
        name, use_mask = str(case).split("|")
        begin, end, bs, ha = (list(map(int, r)) for r in g[name + ".args"])
        res = predict_ref.prepare_block_input(g["vol3"], g["mask3"] if int(use_mask) else None, begin, end, bs, ha, False, None,
                                              predict_ref.standardize)
        assert (res is None) == bool(g[name + ".skipped"]), name
        if res is not None:
            assert np.allclose(res[0], g[name + ".tensor"], rtol=1e-6, atol=1e-6), name
            assert [list(b) for b in res[2]] == g[name + ".inner_bb"].tolist()
            if int(use_mask):
                assert np.array_equal(res[1], g[name + ".mask_block"])
    res = predict_ref.prepare_block_input(g["vol3c"], None, [0, 3, 0], [3, 6, 6], (3, 3, 6), (1, 1, 2), True, lambda a: False, None)
    assert np.array_equal(res[0], g["pb_channels.tensor"])
    assert predict_ref.prepare_block_input(g["vol3c"], None, [0, 3, 0], [3, 6, 6], (3, 3, 6), (1, 1, 2), True, lambda a: True, None) is None
    begin, end, halo = (list(map(int, r)) for r in g["wp.args"])
    inner = tuple(slice(h, h + e - b) for h, b, e in zip(halo, begin, end))
    mb = g["wp.mask_block"]
    o = np.full((3, 7, 9, 11), -7.0, dtype="float32")
    predict_ref.write_prediction(g["wp.pred"].copy(), begin, end, o, 3, None, inner, None)
    assert np.array_equal(o, g["wp.out_channels"])
    o = np.full((3, 7, 9, 11), -7.0, dtype="float32")
    predict_ref.write_prediction(g["wp.pred"].copy(), begin, end, o, 3, mb, inner, None)
    assert np.array_equal(o, g["wp.out_masked"])
    oa, ob = np.full((7, 9, 11), -7.0, dtype="float32"), np.full((2, 7, 9, 11), -7.0, dtype="float32")
    predict_ref.write_prediction(g["wp.pred"].copy(), begin, end, [(oa, 0), (ob, slice(1, 3))], 3, mb, inner, None)
    assert np.array_equal(oa, g["wp.out_list_a"]) and np.array_equal(ob, g["wp.out_list_b"])
    o = np.full((7, 9, 11), -7.0, dtype="float32")
    predict_ref.write_prediction(g["wp.pred"].copy(), begin, end, o, 3, mb, inner, lambda p: p[1] * 2.0)
    assert np.array_equal(o, g["wp.out_post"])


def test_oracle_at_mfma_widths_matches_the_reference_float64_and_fp32_steps():
    """G10 (tests/golden/gen_golden_amp_step.py): ONE step of the reference's UNet3d(1, 2, depth=2, initial_features=32) on
    1x1x16x24x32 in float64 and in float32.  Pins the oracle at the widths the MFMA kernels run at: its float64 gradients equal the
    reference's (stored as float32: 1e-6), its float32 step sits at the reference's own fp32 distance from float64."""
    import hashlib
    from torch_em_amd.model import UNet3d
    g = _load("g10_amp_step.npz")
    torch.manual_seed(0)
    model = UNet3d(1, 2, depth=2, initial_features=32)
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    assert h.hexdigest() == str(g["sd_sha256"])     # this repo's seed-0 initialisation IS the reference's
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    names = [str(n) for n in g["param_names"]]
    pred, loss, grads = unet_ref.unet_loss_and_grads({k: v.double() for k, v in sd.items()}, x.double(), y.double(), [2, 2])
    assert rel_err(pred, g["f64.pred"]) < 1e-12 and abs(float(loss) - float(g["f64.loss"])) < 1e-12
    gscale = float(np.sqrt((g["f64.grad_norm"] ** 2).sum()))
    for k, n64 in zip(names, g["f64.grad_norm"]):
        ref = g[f"f64.grad.{k}"].astype("float64")
        err = float(np.linalg.norm(grads[k].numpy() - ref))
        assert err <= 2e-7 * max(float(n64), 1e-6 * gscale), (k, err, n64)      # float32 storage of the fixture
    # the oracle's fp32 step: the distance the reference's fp32 step has from float64, tensor by tensor (same ATen kernels)
    pred32, loss32, g32 = unet_ref.unet_loss_and_grads(sd, x, y, [2, 2])
    assert abs(float(loss32) - float(g["f32.loss"])) < 1e-6
    for k, n64, e_ref in zip(names, g["f64.grad_norm"], g["f32.grad_err"]):
        err = float(np.linalg.norm(g32[k].double().numpy() - g[f"f64.grad.{k}"].astype("float64")))
        assert err <= 3.0 * max(float(e_ref), 1e-6 * gscale), (k, err, e_ref)
