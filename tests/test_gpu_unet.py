"""GPU: the whole drop-in model (+ loss) against the reference's golden vectors and, at larger
MFMA-eligible sizes, against the oracle on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_grads, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3  # north-star tolerance (relative fp32); measured errors are ~1e-5


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def _run(model, g, loss):
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    model.load_state_dict(sd)
    model.to(DEV)
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    pred = model(x)
    val = loss(pred, y)
    val.backward()
    assert tuple(pred.shape) == tuple(g["pred"].shape)
    assert rel_err(pred.detach().cpu(), g["pred"]) < TOL
    assert abs(float(val) - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    grads = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
    return check_grads(grads, {k: g[f"grad.{k}"] for k in grads}, TOL)


@pytest.fixture(params=["sums_on_every_layer", "product_default"])
def dispatch_mix(request):
    """tests/conftest.py lowers `wgrad_sums_min_mb` to 0 so that every qualifying layer derives its norm-backward sums from
    the weight gradient (csrc/wgrad_sums.hip); the PRODUCT default is 128 (only tensors >= 128 MB, the rest run the
    reduction pass).  Tests that take this fixture run under both kernel mixes."""
    from torch_em_amd import _lib
    old = _lib.get_option("wgrad_sums_min_mb")
    _lib.set_option("wgrad_sums_min_mb", 0 if request.param == "sums_on_every_layer" else 128)
    yield request.param
    _lib.set_option("wgrad_sums_min_mb", old)


@pytest.mark.parametrize("norm", ["InstanceNorm", "GroupNorm", None])
def test_unet3d_matches_reference_golden(norm, dispatch_mix):
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    worst = _run(UNet3d(1, 2, depth=2, initial_features=4, norm=norm), _load(f"g1_unet3d_{norm}.npz"), DiceLoss())
    print("worst grad rel err", worst)


@pytest.mark.parametrize("aniso", [0, 1])
def test_anisotropic_unet_masked_dice_matches_reference_golden(aniso):
    from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
    from torch_em_amd.model import AnisotropicUNet
    model = AnisotropicUNet(1, 12, [[1, 2, 2], [2, 2, 2]], initial_features=4, final_activation="Sigmoid",
                            anisotropic_kernel=bool(aniso))
    _run(model, _load(f"g2_aniso_{aniso}.npz"), LossWrapper(DiceLoss(), ApplyAndRemoveMask("multiply")))


def test_decoder_crop_matches_reference_golden():
    """Decoder._crop (reference model/unet.py:363-373): scale factors [[3,3,3],[2,2,2]] on 14^3 -- the level-0 skip tensor
    (14^3) is centre-cropped to the 12^3 of the upsampled tensor.  Reachable only with `model.check_shape = False`
    (UNetBase.forward :248), which the golden run set; with the check on, both raise the same ValueError."""
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import AnisotropicUNet
    model = AnisotropicUNet(1, 2, [[3, 3, 3], [2, 2, 2]], initial_features=4)
    with pytest.raises(ValueError, match="is not divisible by"):
        model.to(DEV)(torch.zeros(1, 1, 14, 14, 14, device=DEV))
    model.check_shape = False
    _run(model, _load("g2c_crop.npz"), DiceLoss())
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):   # an odd difference cannot be cropped away
        model(torch.zeros(1, 1, 13, 14, 14, device=DEV))


def test_unet2d_matches_reference_golden():
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet2d
    _run(UNet2d(1, 2, depth=2, initial_features=4), _load("g3_unet2d.npz"), DiceLoss())


def _oracle_case(model, scale_factors, x, y, norm, final_activation=None, loss_fn=None, dtype=torch.float32):
    from oracle import unet_ref
    sd = {k: v.detach().cpu().clone().to(dtype) for k, v in model.state_dict().items()}
    return unet_ref.unet_loss_and_grads(sd, x.to(dtype), y.to(dtype), scale_factors, norm=norm,
                                        final_activation=final_activation, loss_fn=loss_fn)


def _check_against_fp64(model, pred, loss, case, l2_factor=4.0, global_factor=2.0, known=None):
    """At MFMA-eligible widths fp32 gradients through InstanceNorm / ReLU / max-pool are ill-conditioned:
    a 1e-7 forward difference flips a ReLU mask or a pooling arg-max of a near-tie, which moves single
    gradient entries by up to ~1e-2 -- the reference's OWN fp32 arithmetic deviates from the exact
    (float64) gradient by that much.  Parity is therefore judged against the float64 oracle:
      * forward and loss within TOL (max norm),
      * the whole gradient (all parameters concatenated) in the relative L2 norm within
        max(TOL, 2 x the error of the fp32 reference path against float64),
      * every gradient tensor in the relative L2 norm (robust to isolated flips) within
        max(TOL, 4 x the reference path's error) and never above 1e-2 (which side of a near-tie an
        implementation falls on is random, so single tensors scatter by a small factor),
      * and in the max norm within 5e-2 (isolated flips only).
    Kernel-level arithmetic is pinned separately to 2e-5 (tests/test_gpu_ops.py) and the whole model to
    TOL in the max norm on the reference's golden vectors (narrow nets, no near-ties).
    known: {tensor name: explicit L2 bound} for DOCUMENTED deviations (the caller says why); replaces the relative bound."""
    pred64, loss64, g64 = _oracle_case(*case, dtype=torch.float64)
    _, loss32, g32 = _oracle_case(*case, dtype=torch.float32)
    assert rel_err(pred.detach().cpu(), pred64) < TOL
    assert abs(float(loss) - float(loss64)) < TOL and abs(float(loss) - float(loss32)) < TOL
    gscale = max(float(v.abs().max()) for v in g64.values())
    report = {}
    for k, p in model.named_parameters():
        r = g64[k].numpy()
        if float(np.abs(r).max()) < 1e-4 * gscale:  # mathematically-zero gradient (see conftest.check_grads)
            assert float(p.grad.abs().max()) < 1e-4 * gscale, k
            continue
        a, b = p.grad.cpu().numpy().astype(np.float64), g32[k].numpy().astype(np.float64)
        l2 = float(np.linalg.norm(r))
        e_hip, e_ref = float(np.linalg.norm(a - r)) / l2, float(np.linalg.norm(b - r)) / l2
        m_hip = float(np.abs(a - r).max() / np.abs(r).max())
        m_ref = float(np.abs(b - r).max() / np.abs(r).max())
        report[k] = (e_hip, e_ref, m_hip)
        if known and k in known:
            assert e_hip <= known[k], (k, "L2 (documented deviation)", e_hip, known[k])
        else:
            assert e_hip <= min(1e-2, max(TOL, l2_factor * e_ref)), (k, "L2", e_hip, e_ref)
        # isolated flips only -- or, where the fp32 reference path itself is that far from float64 in single entries
        # (instance statistics over the 64 voxels of a 4^3 level), the same class as the reference
        # isolated flips only.  At a level with few voxels (4^3 = 64 per channel under the depth-4 net) ONE flipped ReLU
        # mask entry is 1/64 of every sum it enters, so single entries may move further -- but only a handful of them
        frac = float((np.abs(a - r) > 1e-2 * np.abs(r).max()).mean())
        if os.environ.get("TEM_TEST_VERBOSE"):
            print(f"{k}: L2 hip {e_hip:.2e} ref {e_ref:.2e}  max hip {m_hip:.2e} ref {m_ref:.2e}  frac>1e-2 {frac:.2e}")
        # "a handful" in units of flips: ONE flipped mask entry of output channel co moves the whole row dw[co] (cin x taps
        # entries) of that layer's weight gradient, so up to two rows' worth of entries may exceed the threshold
        few = max(2.0, 1e-3 * a.size, 2.0 * a.size / a.shape[0] if a.ndim > 1 else 0.0)
        assert m_hip <= 5e-2 or (m_hip <= 0.25 and frac * a.size <= few), (k, "max", m_hip, m_ref, frac)
    keys = sorted(report)
    cat = lambda d: np.concatenate([np.asarray(d[k], dtype=np.float64).ravel() for k in keys])  # noqa: E731
    r_all = cat({k: g64[k].numpy() for k in keys})
    h_all = cat({k: dict(model.named_parameters())[k].grad.cpu().numpy() for k in keys})
    c_all = cat({k: g32[k].numpy() for k in keys})
    e_h, e_c = np.linalg.norm(h_all - r_all) / np.linalg.norm(r_all), np.linalg.norm(c_all - r_all) / np.linalg.norm(r_all)
    print(f"global gradient L2 rel err vs float64: hip {e_h:.2e}, fp32 reference path {e_c:.2e}")
    assert e_h <= max(TOL, global_factor * e_c), ("global L2", e_h, e_c)
    worst = max(report.items(), key=lambda kv: kv[1][0])
    print(f"worst L2 rel err vs float64: {worst[0]} hip {worst[1][0]:.2e} (fp32 reference path {worst[1][1]:.2e}), "
          f"max-norm {max(v[2] for v in report.values()):.2e}")
    return report


@pytest.mark.parametrize("norm", ["InstanceNorm", "GroupNorm"])
def test_unet3d_mfma_sizes_match_oracle(norm, dispatch_mix):
    """initial_features=32 => every 3x3x3 conv but the first runs on the MFMA kernels."""
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    torch.manual_seed(0)
    model = UNet3d(1, 2, depth=2, initial_features=32, norm=norm)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 1, 16, 24, 32, generator=g)
    y = (torch.rand(2, 2, 16, 24, 32, generator=g) > 0.5).float()
    case = (model, [2, 2], x, y, norm)
    _oracle_case(*case)  # state_dict snapshot happens inside; run before moving the model
    model.to(DEV)
    pred = model(x.to(DEV))
    loss = DiceLoss()(pred, y.to(DEV))
    loss.backward()
    _check_against_fp64(model, pred, loss, case)


@pytest.mark.parametrize("sums_min_mb", [128, 0])
def test_exact_fp32_mode_groupnorm_gradients_with_forced_decisions(sums_min_mb):
    """(sums_min_mb = 0: every layer that can takes the norm-backward sums from its weight gradient and the norm backward in the
    data gradient's epilogue -- round 6: the exact mode runs on the z-reuse kernels and their fused epilogues, too; 128 = the
    product threshold, which leaves these small layers on the two-pass norm backward.)
    The exact-fp32 build (TEM_PRECISION=fp32, not the default) on the GroupNorm case of test_unet3d_mfma_sizes_match_oracle.
    Rounds 3-4 carried a "known deviation" of this mode: the bias gradient of the first norm 5e-3 from float64 (the fp32
    reference path: 5e-5), explained as a cancelling sum.  The per-tensor table (round 5) says otherwise: EVERY encoder tensor
    is 2-3e-3 off while base and decoder agree to 1e-6 -- the signature of one decision between them that the two arithmetics
    take differently (a ReLU mask / pooling arg-max of a near-tie in the encoder), not of a summation.  So this test runs the
    float64 oracle with the library's decisions forced (oracle.unet_ref.DecisionTap) and asks for the plain 1e-3 max-norm bound
    on every tensor, the first norm's bias included."""
    from oracle import loss_ref, unet_ref
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d, engine
    torch.manual_seed(0)
    model = UNet3d(1, 2, depth=2, initial_features=32, norm="GroupNorm")
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 1, 16, 24, 32, generator=g)
    y = (torch.rand(2, 2, 16, 24, 32, generator=g) > 0.5).float()
    sd = {k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    model.to(DEV)
    from torch_em_amd import _lib
    old_mb = _lib.get_option("wgrad_sums_min_mb")
    _lib.set_option("wgrad_sums_min_mb", sums_min_mb)
    try:
        with engine.precision_scope("fp32"):
            force = _library_decisions(model, x)
            pred = model(x.to(DEV))
            loss = DiceLoss()(pred, y.to(DEV))
            loss.backward()
    finally:
        _lib.set_option("wgrad_sums_min_mb", old_mb)
    worst = {}
    for forced in (False, True):
        for v in sd.values():
            v.grad = None
        with unet_ref.DecisionTap(force if forced else None):
            pred64 = unet_ref.unet_forward(sd, x.double(), [2, 2], norm="GroupNorm")
        loss64 = loss_ref.dice_loss(pred64, y.double())
        loss64.backward()
        gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
        for k, p in model.named_parameters():
            ref = sd[k].grad
            if float(ref.abs().max()) < 1e-4 * gscale:
                continue
            err = float((p.grad.double().cpu() - ref).abs().max() / ref.abs().max())
            if err >= worst.get(forced, (0.0, ""))[0]:
                worst[forced] = (err, k)
            if forced:
                assert err < TOL, (k, err)
    print(f"exact-fp32 mode, GroupNorm: worst max-norm gradient error free {worst[False][0]:.2e} ({worst[False][1]}), "
          f"with the library's decisions forced {worst[True][0]:.2e} ({worst[True][1]})")


def test_unet3d_benchmark_widths_depth4_match_fp64_oracle(dispatch_mix):
    """The benchmark network itself -- UNet3d(1, 2, initial_features=32, depth=4): 32 ... 512 features, every kernel family
    of cfg 2 incl. the split-K convolutions of the 8^3 / 4^3 levels -- on 64^3 volumes against the float64 oracle, with
    the STANDARD bounds of _check_against_fp64 (whole gradient within 2x, every tensor within 4x the fp32 reference
    path's own error) applied to the MEDIAN over four seeds.
    Why a median: at these widths the gradient error against float64 is decided by a handful of near-tie decisions (ReLU
    masks / pooling arg-maxes of the 4^3 and 8^3 levels: one flipped entry is 1/64 of every sum it enters), so it scatters
    by 3-10x from seed to seed for ANY arithmetic that differs from float64 at the 1e-7 level -- the reference's fp32 CPU
    path included.  Measured over six seeds (profiles/r03_depth4_error_*.txt, scripts/depth4_error_survey.py), global
    relative L2 error: fp32 reference path 1.4e-3 ... 3.9e-3 (median 1.8e-3), this library (default arithmetic) 1.3e-3 ...
    3.0e-3 with one 1.5e-2 outlier (median 1.9e-3), its exact-fp32 build 2.3e-3 ... 6.2e-3.  Round 2 looked at ONE seed
    (4.5e-3 vs 1.3e-3), widened the bound to 6x / 5x and blamed the 16-bit backward products; fp32-class data gradients
    (TEM_DGRAD16=1) leave that number unchanged."""
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d

    def make(seed):
        torch.manual_seed(seed)
        model = UNet3d(1, 2, depth=4, initial_features=32)
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(1, 1, 64, 64, 64, generator=g)
        y = (torch.rand(1, 2, 64, 64, 64, generator=g) > 0.5).float()
        return (model, [2, 2, 2, 2], x, y, "InstanceNorm"), DiceLoss()
    _median_over_seeds(make, (0, 1, 2, 3))


def _library_decisions(model, x):
    """ReLU masks (forward order of oracle.unet_ref.unet_forward) and pooling arg-maxes of this library's forward pass"""
    import torch.nn.functional as F
    from torch_em_amd.model import engine
    _, st = engine._forward_impl(model, x.to(DEV), keep=True)
    nchw = lambda t: t.permute(0, 4, 1, 2, 3).float().cpu()  # noqa: E731
    blocks = [lv["bs"] for lv in st["levels"]] + [st["base"]] + [d["bs"] for d in st["dec"]]
    masks = [nchw(bs[k]) > 0 for bs in blocks for k in ("a1", "out")]
    pools = [F.max_pool3d_with_indices(nchw(lv["skip"]), tuple(lv["f"]))[1] for lv in st["levels"]]
    return {"relu": masks, "pool": pools}


@pytest.mark.parametrize("seed,norm,size", [(0, "InstanceNorm", 64), (5, "InstanceNorm", 64), (0, "GroupNorm", 64),
                                            (1, "InstanceNorm", 128)])
def test_depth4_gradient_max_norm_with_the_library_decisions_forced(seed, norm, size):
    """The plain north-star bound -- 1e-3, MAX norm, every parameter tensor -- for the 32 ... 512-feature benchmark network,
    on a comparison without near-ties: the float64 oracle runs with every ReLU mask and every pooling arg-max forced to the
    decisions this library took (oracle.unet_ref.DecisionTap), so what is compared is the arithmetic of the kernels and not
    which way a pre-activation of 1e-9 x rms was rounded.  profiles/r05_flip_census.txt (scripts/flip_census.py) has the
    census behind this: seed 0 is the 1.5e-2 outlier of the free comparison, seed 5 an ordinary one; both must pass here.
    size 128 (round 6): ONE sample of the benchmark volume itself, 1 x 1 x 128^3 -- the level where `k_conv_wgrad_tr<3>` sums
    2.1 M voxels per weight-gradient entry with an 11-bit g, and where the statistics / slab merges have their longest sums
    (the float64 oracle at this size is about a minute of host time)."""
    from oracle import loss_ref, unet_ref
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    torch.manual_seed(seed)
    model = UNet3d(1, 2, depth=4, initial_features=32, norm=norm)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(1, 1, size, size, size, generator=g)
    y = (torch.rand(1, 2, size, size, size, generator=g) > 0.5).float()
    if norm == "GroupNorm":   # affine norms with gains up to 3 (the two-MFMA weight gradient under large gamma)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if p.dim() == 1 and k.endswith("weight"):
                    p.copy_(1.0 + 2.0 * torch.rand(p.shape, generator=g))
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model.to(DEV)
    force = _library_decisions(model, x)
    with unet_ref.DecisionTap(force):
        pred64 = unet_ref.unet_forward(sd, x.double(), [2, 2, 2, 2], norm=norm)
    loss64 = loss_ref.dice_loss(pred64, y.double())
    loss64.backward()
    pred = model(x.to(DEV))
    loss = DiceLoss()(pred, y.to(DEV))
    loss.backward()
    assert rel_err(pred.detach().cpu(), pred64.detach()) < TOL and abs(float(loss.detach()) - float(loss64.detach())) < TOL
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    worst = 0.0
    first_norm = {}
    for k, p in model.named_parameters():
        ref = sd[k].grad
        if float(ref.abs().max()) < 1e-4 * gscale:
            continue                       # conv biases in front of an InstanceNorm: analytically zero
        # a tensor whose whole gradient is below 1 % of the network's largest entry (the gain / bias of the FIRST norm: the
        # norm behind the next conv makes the loss nearly invariant to them, their gradient is a cancelling sum over all
        # voxels) is measured against that 1 %, not against itself
        denom = max(float(ref.abs().max()), 1e-2 * gscale)
        err = float((p.grad.double().cpu() - ref).abs().max() / denom)
        if os.environ.get("TEM_TEST_VERBOSE"):
            print(f"  {k}: {err:.2e}  (|grad|max / net max {float(ref.abs().max()) / gscale:.1e})")
        if k.startswith("encoder.blocks.0.block.0."):
            # gain / bias of the FIRST norm, GroupNorm(1, 1) on the 1-channel input: ONE number each, the sum of gz * xhat
            # (gz) over all 262144 voxels behind a norm that makes the loss almost invariant to it -- the summands cancel
            # to 3e-3 of the network's gradient scale and the 16-bit products of the data-gradient chain (1e-5 each) leave
            # 7e-3 of THAT (2e-5 of the scale).  Bounded on its own, not hidden -- and PREDICTED below (first_norm):
            assert err < 1e-2, (k, err)
            first_norm[k] = (p.grad.double().cpu(), ref)
            continue
        worst = max(worst, err)
        assert err < TOL, (k, err)
    print(f"seed {seed} {norm} {size}^3: worst per-tensor max-norm gradient error with forced decisions {worst:.2e}")
    if first_norm:
        # The first norm's two numbers as a PREDICTION instead of a tolerance: the same float64 backward
        # with the same forced decisions, in which every data-gradient convolution that runs on the matrix cores sees its
        # operands as the kernels round them (g and w in two bf16 terms, the three largest term products: "b2xb2-3" of
        # scripts/backward_arith_sim.py; sums exact).  If the 16-bit products of that chain are what moves these cancelling
        # sums, the library must agree with the SIMULATED value to the plain 1e-3 of the tensor's own magnitude.
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
        import backward_arith_sim as bas
        saved = dict(bas.VARIANT)
        bas.VARIANT.update(dgrad=("b2", "b2", 3), wgrad=("exact", "exact", 1), min_cin=16, sim_1x1=True, wgrad_min_voxels=0)
        orig_conv = unet_ref._conv
        unet_ref._conv = lambda x_, w_, b_: bas.ConvSim.apply(x_, w_, b_) if w_.dim() == 5 else orig_conv(x_, w_, b_)
        try:
            sd2 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
            with unet_ref.DecisionTap(force):
                pred_s = unet_ref.unet_forward(sd2, x.double(), [2, 2, 2, 2], norm=norm)
            loss_ref.dice_loss(pred_s, y.double()).backward()
        finally:
            unet_ref._conv = orig_conv
            bas.VARIANT.clear()
            bas.VARIANT.update(saved)
        # ... and the reference's own arithmetic on the same decisions: the oracle in fp32 (what torch-em's CPU path computes)
        sd3 = {k: v.detach().float().clone().requires_grad_(True) for k, v in sd.items()}
        with unet_ref.DecisionTap(force):
            pred32 = unet_ref.unet_forward(sd3, x.float(), [2, 2, 2, 2], norm=norm)
        loss_ref.dice_loss(pred32, y.float()).backward()
        for k, (got, ref) in first_norm.items():
            sim, g32 = sd2[k].grad, sd3[k].grad.double()
            e_exact = float((got - ref).abs().max() / ref.abs().max())
            e_chain = float((sim - ref).abs().max() / ref.abs().max())
            e_ref32 = float((g32 - ref).abs().max() / ref.abs().max())
            print(f"  {k}: of its own magnitude -- library vs float64 {e_exact:.2e}; float64 with the simulated 16-bit data-gradient "
                  f"chain vs float64 {e_chain:.2e}; the fp32 oracle (reference arithmetic, same decisions) vs float64 {e_ref32:.2e}")
            # measured (seed 0, GroupNorm, 64^3): gain 6.7e-3 / 6.0e-6 / 5.1e-5, bias 2.7e-4 / 5.2e-5 / 4.9e-6 -- the operand
            # rounding of the data-gradient chain explains NOTHING of the library's distance.  What does: the 16-bit MFMAs align
            # every product to the accumulator and TRUNCATE (csrc/conv_split.h: a one-sided error of ~1e-5 per output, measured
            # in round 2), a bias that is coherent over the volume and therefore survives a sum whose terms cancel to 3e-3.
            assert e_chain < 1e-4, (k, e_chain)
        # The counter-test: the exact-fp32 mode, whose v_mfma_f32_32x32x2_f32 IS a round-to-nearest fmaf chain
        # (test_conv_exact_fp32_zreuse_is_an_fmaf_chain: bit for bit), with ITS decisions forced into the float64 oracle, must
        # put the same two numbers inside the plain 1e-3 of their own magnitude -- the class of the fp32 oracle, not of the
        # default arithmetic.
        from torch_em_amd.model import engine
        with engine.precision_scope("fp32"):
            force_x = _library_decisions(model, x)
            model.zero_grad()
            DiceLoss()(model(x.to(DEV)), y.to(DEV)).backward()
        sd4 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        with unet_ref.DecisionTap(force_x):
            pred_x = unet_ref.unet_forward(sd4, x.double(), [2, 2, 2, 2], norm=norm)
        loss_ref.dice_loss(pred_x, y.double()).backward()
        named = dict(model.named_parameters())
        for k in first_norm:
            ref_x = sd4[k].grad
            e_x = float((named[k].grad.double().cpu() - ref_x).abs().max() / ref_x.abs().max())
            print(f"  {k}: exact-fp32 mode vs float64 (its own decisions forced) {e_x:.2e} of its own magnitude")
            assert e_x < TOL, (k, e_x)


def test_default_arithmetic_is_bit_identical_to_round4():
    """The 16-bit activation storage of round 5 templates every kernel of the step on its element type; the fp32-class
    default must not have moved by a bit.  tests/golden/default_step_digest.json holds SHA-256 digests of prediction / loss
    / flat gradient of three networks written by scripts/digest_default.py under TEM_LIB=<the round-4 library>; every
    kernel is deterministic, so the digests are a property of the source tree and the chip.  (A deliberate change of a
    summation order in a default-path kernel moves them: regenerate the fixture in the same commit and say so there.)"""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
    import digest_default
    with open(os.path.join(GOLDEN, "default_step_digest.json")) as f:
        want = json.load(f)
    from torch_em_amd import _lib
    old = _lib.get_option("wgrad_sums_min_mb")
    _lib.set_option("wgrad_sums_min_mb", 128)     # the fixture is of the PRODUCT kernel mix (tests/conftest.py lowers this to 0)
    try:
        for name, (kw, shape) in digest_default.CASES.items():
            assert digest_default.run_case(kw, shape) == want[name], name
    finally:
        _lib.set_option("wgrad_sums_min_mb", old)


def _median_over_seeds(make, seeds):
    """The bounds of _check_against_fp64 (whole gradient within 2x, every tensor within 4x the fp32 reference path's own
    error against float64, never above 1e-2) on the MEDIAN over `seeds`; make(seed) -> (oracle case, loss on the device)."""
    per_seed, per_tensor = [], {}
    for seed in seeds:
        case, loss_fn = make(seed)
        model, x, y = case[0], case[2], case[3]
        pred64, loss64, g64 = _oracle_case(*case, dtype=torch.float64)
        _, _, g32 = _oracle_case(*case, dtype=torch.float32)
        model.to(DEV)
        pred = model(x.to(DEV))
        loss = loss_fn(pred, y.to(DEV))
        loss.backward()
        assert rel_err(pred.detach().cpu(), pred64) < TOL and abs(float(loss) - float(loss64)) < TOL
        gscale = max(float(v.abs().max()) for v in g64.values())
        keys = [k for k, _ in model.named_parameters() if float(g64[k].abs().max()) >= 1e-4 * gscale]
        hip = {k: p.grad.double().cpu().numpy() for k, p in model.named_parameters()}
        cat = lambda d: np.concatenate([np.asarray(d[k], dtype=np.float64).ravel() for k in keys])  # noqa: E731
        r_all = cat({k: g64[k].numpy() for k in keys})
        e_h = float(np.linalg.norm(cat(hip) - r_all) / np.linalg.norm(r_all))
        e_c = float(np.linalg.norm(cat({k: g32[k].numpy() for k in keys}) - r_all) / np.linalg.norm(r_all))
        per_seed.append((e_h, e_c))
        for k in keys:
            r = g64[k].numpy().astype(np.float64)
            per_tensor.setdefault(k, []).append((float(np.linalg.norm(hip[k] - r) / np.linalg.norm(r)),
                                                 float(np.linalg.norm(g32[k].numpy() - r) / np.linalg.norm(r))))
        print(f"seed {seed}: global gradient L2 rel err vs float64: hip {e_h:.2e}, fp32 reference path {e_c:.2e}")
        assert e_h < 2e-2, (seed, e_h)              # a single seed may lose the near-tie lottery (profiles/r05_flip_census.txt), but not by more than this
        del model
    med_h, med_c = float(np.median([a for a, _ in per_seed])), float(np.median([b for _, b in per_seed]))
    print(f"median over seeds: hip {med_h:.2e}, fp32 reference path {med_c:.2e}")
    assert med_h <= max(TOL, 2.0 * med_c), ("global L2 (median over seeds)", med_h, med_c)
    for k, v in per_tensor.items():
        mh, mc = float(np.median([a for a, _ in v])), float(np.median([b for _, b in v]))
        assert mh <= min(1e-2, max(TOL, 4.0 * mc)), (k, "L2 (median over seeds)", mh, mc)


def test_anisotropic_cfg3_factors_match_fp64_oracle():
    """AnisotropicUNet with the scale factors of cfg 3 ([[1,2,2],[1,2,2],[2,2,2],[2,2,2]], 12 affinity channels, Sigmoid)
    at initial_features=32 on 1x1x16x64x64: the 1x3x3 (2-D) MFMA kernels, anisotropic pooling / upsampling.  Median over
    three seeds, as for the depth-4 network above: here even the fp32 reference path is 1.4e-3 ... 5.2e-3 from float64
    depending on the seed (scripts/depth4_error_survey.py --cfg3: this library 1.6e-3 ... 5.1e-3, below the reference path
    in 5 of 6 seeds), and which single tensor is hit by a near-tie flip changes with every 1e-7 change of the arithmetic."""
    from oracle import loss_ref
    from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
    from torch_em_amd.model import AnisotropicUNet
    sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]

    def make(seed):
        torch.manual_seed(seed)
        model = AnisotropicUNet(1, 12, sf, initial_features=32, final_activation="Sigmoid", anisotropic_kernel=True)
        g = torch.Generator().manual_seed(7 + seed)
        x = torch.randn(1, 1, 16, 64, 64, generator=g)
        y = torch.cat([(torch.rand(1, 12, 16, 64, 64, generator=g) > 0.5).float(),
                       (torch.rand(1, 12, 16, 64, 64, generator=g) > 0.3).float()], dim=1)
        return (model, sf, x, y, "InstanceNorm", "Sigmoid", loss_ref.masked_dice_loss), \
            LossWrapper(DiceLoss(), ApplyAndRemoveMask("multiply"))
    _median_over_seeds(make, (0, 1, 2))


def test_anisotropic_mfma_sizes_match_oracle():
    from oracle import loss_ref
    from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
    from torch_em_amd.model import AnisotropicUNet
    torch.manual_seed(0)
    sf = [[1, 2, 2], [2, 2, 2]]
    model = AnisotropicUNet(1, 12, sf, initial_features=32, final_activation="Sigmoid", anisotropic_kernel=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 1, 8, 32, 32, generator=g)
    y = torch.cat([(torch.rand(1, 12, 8, 32, 32, generator=g) > 0.5).float(),
                   (torch.rand(1, 12, 8, 32, 32, generator=g) > 0.3).float()], dim=1)
    case = (model, sf, x, y, "InstanceNorm", "Sigmoid", loss_ref.masked_dice_loss)
    model.to(DEV)
    pred = model(x.to(DEV))
    loss = LossWrapper(DiceLoss(), ApplyAndRemoveMask("multiply"))(pred, y.to(DEV))
    loss.backward()
    _check_against_fp64(model, pred, loss, case)


def test_inference_mode_and_determinism():
    from torch_em_amd.model import UNet3d
    torch.manual_seed(0)
    model = UNet3d(1, 2, depth=2, initial_features=32).to(DEV)
    x = torch.randn(1, 1, 16, 16, 16, device=DEV)
    with torch.no_grad():
        a, a2 = model(x), model(x)
    b, b2 = model(x), model(x)
    assert torch.equal(a, a2) and torch.equal(b.detach(), b2.detach())  # no atomics anywhere: bitwise reproducible
    assert not a.requires_grad and b.requires_grad
    # the no-grad forward runs the bf16x3 kernels (engine._INFER_BF16X3), the training forward bf16x6: same function
    # to ~1e-5, far inside the 1e-3 tolerance
    assert 0.0 < rel_err(a.cpu(), b.detach().cpu()) < 1e-4


def test_weight_gradient_arithmetic_and_producer_amax(monkeypatch):
    """The default weight gradients of the pre-normalised 3x3x3 layers (D >= 16) run the fp16 2x1 arithmetic with a
    prescale from max |g| (engine._wgrad_f16x2_ok).  (1) max |g| delivered by the kernel that PRODUCED g and max |g| from
    a separate pass (TEM_FUSE_AMAX=0) are the same bits, so every gradient is bit-identical and most absmax passes
    disappear; (2) against the three-product bf16x3 weight gradients (TEM_WGRAD_ARITH=bf16x3) every weight tensor moves by
    the rounding of an 11-bit g: < 1e-3 of its norm (measured ~2e-4); the other tensors keep their arithmetic."""
    from torch_em_amd import ops
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d, engine
    torch.manual_seed(0)
    model = UNet3d(1, 2, depth=2, initial_features=32).to(DEV)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 1, 32, 64, 64, generator=g).to(DEV)
    y = (torch.rand(1, 2, 32, 64, 64, generator=g) > 0.5).float().to(DEV)
    calls = []
    real = ops.absmax
    monkeypatch.setattr(ops, "absmax", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run():
        calls.clear()
        model.zero_grad()
        DiceLoss()(model(x), y).backward()
        return {k: p.grad.clone() for k, p in model.named_parameters()}, len(calls)

    assert engine._WGRAD_F16X2 and engine._FUSE_AMAX
    fused, n_fused = run()
    monkeypatch.setattr(engine, "_FUSE_AMAX", False)
    plain, n_plain = run()
    assert n_plain >= 4 and n_fused < n_plain, (n_fused, n_plain)
    for k in fused:
        assert torch.equal(fused[k], plain[k]), k
    monkeypatch.setattr(engine, "_WGRAD_F16X2", False)
    three, n_three = run()
    assert n_three == 0
    moved = 0
    # (gradients that are mathematically zero -- the bias of a conv in front of an InstanceNorm -- are pure round-off:
    # every tensor is judged against at least 1e-3 of the largest gradient norm)
    floor = 1e-3 * max(float(v.norm()) for v in three.values())
    for k in fused:
        d = float((fused[k] - three[k]).norm() / three[k].norm().clamp_min(floor))
        if k.endswith("weight") and fused[k].dim() == 5 and fused[k].shape[2:] == (3, 3, 3) and fused[k].shape[1] >= 32:
            assert d < 1e-3, (k, d)
            moved += d > 3e-5
        else:
            # biases, first layer, 1x1x1 convs: their own arithmetic is unchanged; they see the weight gradients only
            # through the norm-backward sums that tem_conv3d_wgrad_sums derives from them
            assert d < 2e-4, (k, d)
    assert moved >= 4


def test_benchmark_config_full_size_properties(dispatch_mix):
    """cfg 2 at full size (2x1x128^3): size-independent checks -- finite, deterministic
    (bitwise), and the CPU oracle (fp32, a few seconds on 16 threads; ATen-on-GPU would spend minutes in MIOpen's
    kernel search on a fresh box) agrees on prediction, loss and every parameter gradient."""
    from oracle import loss_ref, unet_ref
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    torch.manual_seed(0)
    model = UNet3d(1, 2, initial_features=32, depth=4).to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 128, 128, 128, generator=g).to(DEV)
    y = (torch.rand(2, 2, 128, 128, 128, generator=g) > 0.5).float().to(DEV)
    vals, grads = [], []
    for _ in range(2):
        model.zero_grad()
        loss = DiceLoss()(model(x), y)
        loss.backward()
        vals.append(float(loss))
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    assert np.isfinite(vals[0]) and vals[0] == vals[1]
    for k in grads[0]:   # no atomics anywhere on the path: EVERY gradient is bitwise reproducible
        assert torch.equal(grads[0][k], grads[1][k]), k
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    nthreads = torch.get_num_threads()
    torch.set_num_threads(16)  # the oracle's sweet spot on the 2x64-core host (scripts/cpu_threads_probe.py)
    try:
        pred = unet_ref.unet_forward(sd, x.cpu(), [2, 2, 2, 2])
        lo = loss_ref.dice_loss(pred, y.cpu())
        lo.backward()
    finally:
        torch.set_num_threads(nthreads)
    assert rel_err(model(x).detach().cpu(), pred.detach().cpu()) < TOL
    assert abs(vals[0] - float(lo)) < 1e-4
    # Gradients: two fp32-class implementations of an ill-conditioned gradient (see _check_against_fp64; at this
    # size float64 is not affordable on the CPU and ATen has no fast float64 conv here).  Robust norms only:
    # every tensor within 5e-2 and the whole gradient within 1e-2 in relative L2.
    named = dict(model.named_parameters())
    num = den = 0.0
    for k, v in sd.items():
        a = named[k].grad.double().cpu().numpy().ravel()
        r = v.grad.double().cpu().numpy().ravel()
        gmax = max(float(np.abs(r).max()), 1e-30)
        if gmax < 1e-4 * 2.0:  # mathematically-zero gradients (sampler bias before InstanceNorm): noise only
            assert float(np.abs(a).max()) < 1e-3, k
            continue
        e = float(np.linalg.norm(a - r) / np.linalg.norm(r))
        assert e < 5e-2, (k, e)
        num += float(np.sum((a - r) ** 2))
        den += float(np.sum(r ** 2))
    glob = (num / den) ** 0.5
    # measured value on record (stdout with -s, and gpurun_out/ -> profiles/r03_fullsize_grad_error.json): the bound below
    # is loose because both sides are fp32-class implementations of an ill-conditioned gradient (see the depth-4 test)
    print(f"cfg 2 full size: global gradient L2 rel err vs the fp32 CPU oracle {glob:.3e}, loss {vals[0]:.7f} vs {float(lo):.7f}")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "fullsize_grad_error.json"), "w") as f:
            json.dump({"config": "UNet3d(1,2,initial_features=32,depth=4) 2x1x128^3 DiceLoss, default arithmetic",
                       "global_grad_l2_rel_err_vs_fp32_cpu_oracle": glob, "loss_hip": vals[0], "loss_oracle": float(lo)}, f)
    assert glob < 1e-2, glob


def test_benchmark_config_full_size_exact_fp32_on_the_team_kernels():
    """cfg 2 at full size in the exact-fp32 mode (engine precision "fp32": every convolution output one fp32 fmaf chain, the
    arithmetic of the reference's CPU path, model/unet.py:417-438).  Since round 6 its 3x3x3 layers run on the z-reuse team
    kernel with the fused statistics / norm-backward epilogues (k_conv_zr<..., X32>, option fp32_zr).  Size-independent
    checks: the step is bitwise reproducible, and it agrees with the same step on the one-patch-per-workgroup kernels of
    rounds 1-5 (fp32_zr = 0: another summation order, two-pass norms) -- prediction 1e-4 (measured 3.4e-5 through the 18
    layers), loss 1e-5, whole gradient 1e-2 in
    relative L2 (two fp32 implementations of an ill-conditioned gradient: a handful of near-tie decisions differ)."""
    from torch_em_amd import _lib
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d, engine
    torch.manual_seed(0)
    model = UNet3d(1, 2, initial_features=32, depth=4).to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 128, 128, 128, generator=g).to(DEV)
    y = (torch.rand(2, 2, 128, 128, 128, generator=g) > 0.5).float().to(DEV)

    def step():
        model.zero_grad()
        pred = model(x)
        loss = DiceLoss()(pred, y)
        loss.backward()
        return pred.detach().clone(), float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()}

    assert _lib.get_option("fp32_zr") == 1
    with engine.precision_scope("fp32"):
        p1, l1, g1 = step()
        p1b, l1b, g1b = step()
        _lib.set_option("fp32_zr", 0)
        try:
            p0, l0, g0 = step()
        finally:
            _lib.set_option("fp32_zr", 1)
    assert np.isfinite(l1) and l1 == l1b and torch.equal(p1, p1b)
    for k in g1:
        assert torch.equal(g1[k], g1b[k]), k
    assert rel_err(p1.cpu(), p0.cpu()) < 1e-4 and abs(l1 - l0) < 1e-5
    num = sum(float(((g1[k].double() - g0[k].double()) ** 2).sum()) for k in g1)
    den = sum(float((g0[k].double() ** 2).sum()) for k in g1)
    glob = (num / den) ** 0.5
    print(f"cfg 2 full size, exact fp32: team kernels vs patch kernels: prediction {rel_err(p1.cpu(), p0.cpu()):.2e}, "
          f"loss {l1:.7f} / {l0:.7f}, global gradient L2 {glob:.2e}")
    assert glob < 1e-2, glob


def test_affinity_config_full_size_properties():
    """BASELINE cfg 3 at its full size (AnisotropicUNet 1->12 + Sigmoid, 2x1x64x256x256, masked Dice on 12 affinity
    channels): the CPU oracle needs minutes there, so the size-independent properties of the path are checked instead
    -- bitwise determinism; samples are independent (InstanceNorm: sample 0 of the batch == sample 0 alone); the
    backward pass is linear in the incoming gradient (scaling the loss by 2 scales every gradient by exactly 2); the
    masked loss sends no gradient into masked-out voxels; the device-side affinity targets equal the numpy oracle."""
    from oracle import label_ref
    from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
    from torch_em_amd.model import AnisotropicUNet
    from torch_em_amd.transform.label import AffinityTransform, BatchTargets
    torch.manual_seed(0)
    sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    model = AnisotropicUNet(1, 12, scale_factors=sf, initial_features=32, final_activation="Sigmoid").to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 64, 256, 256, generator=g).to(DEV)
    lbl = torch.randint(0, 50, (2, 1, 8, 16, 16), generator=g).repeat_interleave(8, 2).repeat_interleave(16, 3) \
        .repeat_interleave(16, 4)
    offsets = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [-3, 0, 0], [0, -9, 0], [0, 0, -9],
               [-4, 0, 0], [0, -27, 0], [0, 0, -27]]
    target = BatchTargets(AffinityTransform(offsets=offsets, add_mask=True))(lbl.to(DEV))
    assert np.array_equal(target[0].cpu().numpy(), label_ref.affinities(lbl[0, 0].numpy(), offsets, add_mask=True))
    loss_fn = LossWrapper(DiceLoss(), ApplyAndRemoveMask(masking_method="multiply"))

    def run(scale=1.0, xin=x, tgt=target, keep_pred_grad=False):
        model.zero_grad()
        pred = model(xin)
        got = {}
        if keep_pred_grad:
            pred.register_hook(lambda gr: got.setdefault("g", gr.detach().clone()))
        loss = loss_fn(pred, tgt)
        (loss * scale).backward()
        return pred.detach().clone(), float(loss), [p.grad.clone() for p in model.parameters()], got.get("g")

    p1, l1, g1, gp = run(keep_pred_grad=True)
    p2, l2, g2, _ = run()
    assert np.isfinite(l1) and l1 == l2, ("loss not reproducible", l1, l2)
    assert torch.equal(p1, p2), ("prediction not reproducible", float((p1 - p2).abs().max()))
    names = [k for k, _ in model.named_parameters()]
    bad = [(k, float((a - b).abs().max())) for k, a, b in zip(names, g1, g2) if not torch.equal(a, b)]
    assert not bad, ("gradients not reproducible", bad)
    assert float(p1.min()) >= 0.0 and float(p1.max()) <= 1.0                      # Sigmoid epilogue
    assert float((gp * (1.0 - target[:, 12:])).abs().max()) == 0.0                # masked voxels get no gradient
    _, _, g4, _ = run(scale=2.0)
    bad = [(k, float((2.0 * a - b).abs().max())) for k, a, b in zip(names, g1, g4) if not torch.equal(2.0 * a, b)]
    assert not bad, ("backward not linear bit for bit", bad)
    p0, _, _, _ = run(xin=x[:1], tgt=target[:1])
    assert rel_err(p0[0].cpu(), p1[0].cpu()) < 1e-5                                # per-sample statistics only


def test_affinity_config_full_size_sample_matches_oracle():
    """BASELINE cfg 3 against the fp32 CPU oracle at the FULL per-sample size (1x1x64x256x256; the batch of 2 is two independent
    samples under InstanceNorm, which test_affinity_config_full_size_properties checks bit for bit): prediction, masked Dice loss
    and every parameter gradient -- the same robust bounds as the cfg-2 full-size test (both sides are fp32-class
    implementations of an ill-conditioned gradient).  The oracle step is ~11 TFLOP: about a minute on 16 host threads."""
    from oracle import label_ref, loss_ref, unet_ref
    from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper
    from torch_em_amd.model import AnisotropicUNet
    torch.manual_seed(0)
    sf = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    model = AnisotropicUNet(1, 12, scale_factors=sf, initial_features=32, final_activation="Sigmoid").to(DEV)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 1, 64, 256, 256, generator=g)
    lbl = torch.randint(0, 50, (1, 1, 8, 16, 16), generator=g).repeat_interleave(8, 2).repeat_interleave(16, 3) \
        .repeat_interleave(16, 4)
    offsets = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [-3, 0, 0], [0, -9, 0], [0, 0, -9],
               [-4, 0, 0], [0, -27, 0], [0, 0, -27]]
    y = torch.from_numpy(label_ref.affinities(lbl[0, 0].numpy(), offsets, add_mask=True))[None].float()
    pred = model(x.to(DEV))
    loss = LossWrapper(DiceLoss(), ApplyAndRemoveMask(masking_method="multiply"))(pred, y.to(DEV))
    loss.backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    nthreads = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        pr = unet_ref.unet_forward(sd, x, sf, final_activation="Sigmoid")
        lo = loss_ref.masked_dice_loss(pr, y)
        lo.backward()
    finally:
        torch.set_num_threads(nthreads)
    assert rel_err(pred.detach().cpu(), pr.detach()) < TOL
    assert abs(float(loss) - float(lo)) < 1e-4 * max(1.0, abs(float(lo)))
    named = dict(model.named_parameters())
    num = den = 0.0
    gmax_all = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, v in sd.items():
        a = named[k].grad.double().cpu().numpy().ravel()
        r = v.grad.double().cpu().numpy().ravel()
        if float(np.abs(r).max()) < 1e-4 * gmax_all:   # mathematically-zero gradients (a bias in front of an InstanceNorm)
            assert float(np.abs(a).max()) < 1e-3 * gmax_all, k
            continue
        e = float(np.linalg.norm(a - r) / np.linalg.norm(r))
        assert e < 5e-2, (k, e)
        num += float(np.sum((a - r) ** 2))
        den += float(np.sum(r ** 2))
    glob = (num / den) ** 0.5
    print(f"cfg 3 full-size sample: global gradient L2 rel err vs the fp32 CPU oracle {glob:.3e}, loss {float(loss):.7f} vs {float(lo):.7f}")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "fullsize_grad_error_cfg3.json"), "w") as f:
            json.dump({"config": "AnisotropicUNet(1,12,sf=[[1,2,2],[1,2,2],[2,2,2],[2,2,2]],32,Sigmoid) 1x1x64x256x256 masked Dice, default "
                                 "arithmetic", "global_grad_l2_rel_err_vs_fp32_cpu_oracle": glob, "loss_hip": float(loss),
                       "loss_oracle": float(lo)}, f)
    assert glob < 1e-2, glob


def test_side_outputs_golden_and_mfma_size():
    """return_side_outputs=True: the reference's golden case (narrow net), then an MFMA-width net against the float64
    oracle (outputs TOL; gradients: every tensor within 1e-2 and the whole gradient within 2e-3 in relative L2 -- the
    robust criteria of _check_against_fp64, fp32 gradients of ReLU/pool nets being ill-conditioned)."""
    from oracle import loss_ref, unet_ref
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet2d, UNet3d
    g = np.load(os.path.join(GOLDEN, "g4_side_outputs.npz"))
    model = UNet3d(1, 2, depth=2, initial_features=4, return_side_outputs=True, final_activation="Sigmoid")
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    model.to(DEV)
    outs = model(torch.from_numpy(g["x"]).to(DEV))
    assert isinstance(outs, list) and [tuple(o.shape) for o in outs] == [(1, 2, 8, 16, 16), (1, 2, 4, 8, 8)]
    val = sum(DiceLoss()(o, torch.from_numpy(g[f"y{i}"]).to(DEV)) for i, o in enumerate(outs))
    val.backward()
    assert abs(float(val) - float(g["loss"])) < TOL
    for i, o in enumerate(outs):
        assert rel_err(o.detach().cpu(), torch.from_numpy(g[f"out{i}"])) < TOL
    check_grads({k: p.grad.cpu() for k, p in model.named_parameters()},
                {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad.")}, TOL)
    # reference test/model/test_unet.py:32-43 (shapes), at a width that takes the MFMA kernels; one output unused
    torch.manual_seed(1)
    net = UNet2d(1, [3, 2, 1], depth=3, initial_features=32, return_side_outputs=True).to(DEV)
    x = torch.rand(2, 1, 64, 64)
    outs = net(x.to(DEV))
    assert [tuple(o.shape) for o in outs] == [(2, 1, 64, 64), (2, 2, 32, 32), (2, 3, 16, 16)]
    (outs[0].square().mean() + 2.0 * outs[2].mean()).backward()   # outs[1] receives no gradient
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in net.state_dict().items()}
    ro = unet_ref.unet_forward(sd, x.double(), [2, 2, 2])
    (ro[0].square().mean() + 2.0 * ro[2].mean()).backward()
    for a, b in zip(outs, ro):
        assert rel_err(a.detach().cpu(), b.detach().float()) < TOL
    num = den = 0.0
    for k, p in net.named_parameters():
        ref = sd[k].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) < 1e-6, k   # out_conv.1 is unused
            continue
        if float(ref.abs().max()) < 1e-9:
            continue
        e = float((p.grad.cpu().double() - ref).norm() / ref.norm())
        assert e < 1e-2, (k, e)
        num += float((p.grad.cpu().double() - ref).square().sum())
        den += float(ref.square().sum())
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5


@pytest.mark.parametrize("norm", ["BatchNorm", "InstanceNormTrackStats"])
def test_stateful_norms_golden(norm):
    """norm="BatchNorm" / "InstanceNormTrackStats": one training step (prediction, loss, gradients, running statistics
    after the step) and the eval-mode forward, against the reference's golden vectors."""
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    g = np.load(os.path.join(GOLDEN, f"g1c_unet3d_{norm}.npz"))
    model = UNet3d(1, 2, depth=2, initial_features=4, norm=norm)
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    model.to(DEV).train()
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    pred = model(x)
    val = DiceLoss()(pred, y)
    val.backward()
    assert rel_err(pred.detach().cpu(), g["pred"]) < TOL
    assert abs(float(val) - float(g["loss"])) < TOL
    check_grads({k: p.grad.cpu().numpy() for k, p in model.named_parameters()},
                {k[5:]: g[k] for k in g.files if k.startswith("grad.")}, TOL)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("after."):
            if "running" in k:
                assert rel_err(sd[k[6:]].cpu(), g[k]) < 1e-5, k
            else:
                assert int(sd[k[6:]]) == int(g[k]), k
    model.eval()
    with torch.no_grad():
        pe = model(x)
    assert rel_err(pe.cpu(), g["pred_eval"]) < TOL
    with pytest.raises(NotImplementedError):   # frozen statistics have no backward here
        model(x).sum().backward()


def test_postprocessing_accumulate_channels():
    """`postprocessing="affinities_with_foreground_to_boundaries3d"` etc. (reference model/unet.py:15-95)."""
    from torch_em_amd.model import UNet3d
    from torch_em_amd.model.unet import POSTPROCESSING, AccumulateChannels
    torch.manual_seed(0)
    x = torch.randn(2, 5, 4, 6, 8)
    for inv, acc, mode in [((0, 1), (1, 4), "max"), (None, (0, 3), "max"), ((1, 3), (0, 5), "mean"), (None, (2, 4), "min")]:
        want = getattr(torch, mode)(x[:, acc[0]:acc[1]], dim=1, keepdim=True)
        want = want if torch.is_tensor(want) else want.values
        if inv is not None:
            want = torch.cat([x[:, inv[0]:inv[1]], want], dim=1)
        for xin in (x.to(DEV), x.to(DEV).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)):  # NCDHW, channels-last
            got = AccumulateChannels(inv, acc, mode)(xin)
            assert got.shape == want.shape and rel_err(got.cpu(), want) < 1e-6
    assert sorted(POSTPROCESSING) == sorted(["affinities_to_boundaries_anisotropic", "affinities_to_boundaries2d",
                                             "affinities_with_foreground_to_boundaries2d", "affinities_to_boundaries3d",
                                             "affinities_with_foreground_to_boundaries3d"])
    net = UNet3d(1, 4, depth=1, initial_features=4, final_activation="Sigmoid",
                 postprocessing="affinities_with_foreground_to_boundaries3d").to(DEV).eval()
    with torch.no_grad():
        out = net(torch.randn(1, 1, 8, 8, 8, device=DEV))
    assert out.shape == (1, 2, 8, 8, 8)
    with pytest.raises(ValueError):
        UNet3d(1, 4, depth=1, initial_features=4, postprocessing="nope")
