"""GPU: every C-ABI op against the oracle / a plain torch fp32 CPU reference of the same op.
Tolerances are relative fp32 (max|a-b|/max|b|); integer/label ops are bit-exact."""
import itertools
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda"


def to5(x):  # [N,C,D,H,W] cpu -> NDHWC cuda contiguous
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def from5(x5):
    return x5.permute(0, 4, 1, 2, 3).cpu()


def _ops():
    from torch_em_amd import ops
    return ops


CONV_CASES = [
    # N, D, H, W, Cin, Cout, k
    (1, 5, 6, 7, 1, 4, (3, 3, 3)),
    (2, 4, 9, 10, 3, 5, (1, 3, 3)),
    (1, 6, 6, 6, 4, 2, (1, 1, 1)),
    (2, 9, 11, 13, 16, 32, (3, 3, 3)),
    (1, 8, 16, 16, 32, 64, (3, 3, 3)),
    (1, 5, 17, 9, 32, 32, (1, 3, 3)),
    (1, 1, 20, 33, 16, 64, (1, 3, 3)),   # 2-D layout (D == 1)
    (2, 4, 8, 8, 64, 32, (1, 1, 1)),
    (1, 1, 16, 16, 32, 32, (1, 1, 1)),
    (1, 3, 5, 4, 8, 16, (3, 1, 3)),      # generic-only kernel shape
    (2, 8, 8, 8, 128, 128, (3, 3, 3)),   # few workgroups => split-K over input channels
    (1, 4, 8, 8, 256, 64, (1, 1, 1)),    # split-K, 1x1x1
    (2, 9, 17, 10, 1, 32, (3, 3, 3)),    # Cin == 1 first-layer kernels
    (1, 1, 19, 21, 1, 16, (1, 3, 3)),    # Cin == 1, 2-D
    (1, 6, 11, 67, 1, 32, (3, 3, 3)),    # Cin == 1: three 32-wide tiles of the row kernel, the last one with 3 voxels
    (1, 5, 7, 33, 1, 64, (3, 3, 3)),     # Cin == 1, 64 output channels (two row passes), W = 32 + 1
    (1, 3, 9, 40, 1, 8, (1, 3, 3)),      # Cin == 1, 8 output channels (4 pair lanes per row), 2-D kernel on a volume
    (2, 9, 17, 10, 3, 32, (3, 3, 3)),    # Cin == 3 (RGB) through the small-Cin first-layer kernels
    (1, 1, 19, 21, 2, 16, (1, 3, 3)),    # Cin == 2, 2-D
    (1, 5, 9, 9, 4, 32, (3, 3, 3)),      # Cin == 4
    (2, 4, 9, 8, 32, 2, (1, 1, 1)),      # out_conv projection kernels
    (1, 5, 9, 8, 32, 8, (1, 1, 1)),      # projection to 8 embedding channels (SPOCO): proj wgrad NJ=1
    (1, 5, 9, 8, 32, 12, (1, 1, 1)),     # 12 affinity channels from 32 features
    (1, 3, 9, 8, 128, 8, (1, 1, 1)),     # side-output projection from a deep level: proj wgrad NJ=4
    (1, 3, 8, 8, 64, 12, (1, 1, 1)),     # projection to 12 affinity channels
    (2, 9, 17, 10, 32, 1, (3, 3, 3)),    # Cout == 1: dgrad of a first layer (affine first norm)
    (1, 1, 19, 21, 16, 1, (1, 3, 3)),    # Cout == 1, 2-D
    (2, 1, 40, 24, 32, 64, (1, 3, 3)),   # 2-D MFMA weight gradient (1x16x8 patches)
    (1, 20, 24, 17, 32, 32, (3, 3, 3)),  # z-sliding wgrad kernel, one Cout tile (k-halves), ragged H/W
    (1, 16, 16, 24, 64, 64, (3, 3, 3)),  # z-sliding wgrad kernel, two Cout tiles
    (2, 33, 8, 8, 32, 96, (3, 3, 3)),    # z-sliding, odd depth, Cout = 3 tiles (last group half empty)
    (1, 16, 132, 136, 32, 32, (3, 3, 3)),  # z-sliding, 289 columns > 256 CUs: workgroups walk two columns each
    (1, 16, 16, 16, 256, 512, (3, 3, 3)),  # benchmark widths of the deepest levels: split-K forward / dgrad, 16^3 wgrad
    (2, 8, 8, 8, 512, 512, (3, 3, 3)),     # the 8^3 base level: split-K forward, the single-wave-of-workgroups wgrad plan
    (2, 20, 24, 40, 64, 32, (3, 3, 3)),    # patch kernel (150 units < 512), Cin = 64 -> one Cout tile (decoder level 0), ragged
    (2, 12, 24, 24, 128, 64, (3, 3, 3)),   # patch kernel (54 units), Cin = 128 -> two Cout tiles (decoder level 1)
    # (the ping-pong kernel family has its own test below: test_conv_pingpong_family_parity)
    (1, 16, 32, 32, 64, 32, (1, 1, 1)),    # streaming 1x1x1 GEMM (>= 16384 voxels), one Cout tile
    (1, 17, 33, 31, 32, 96, (1, 1, 1)),    # streaming 1x1x1 GEMM, ragged voxel count, three Cout tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("fused", [False, True])
def test_conv_fwd_dgrad_wgrad(case, fused):
    ops = _ops()
    N, D, H, W, Cin, Cout, k = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    scale = (torch.rand(N, Cin, generator=g) + 0.5) if fused else None
    shift = torch.randn(N, Cin, generator=g) if fused else None
    pad = tuple(v // 2 for v in k)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    xh = xr * scale[:, :, None, None, None] + shift[:, :, None, None, None] if fused else xr
    pre = F.conv3d(xh, wr, br, padding=pad)
    yr = F.relu(pre) if fused else pre
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    x5 = to5(x)
    wd = w.to(DEV)
    # mode 4 (fp16x3 with scaled lo planes) is meant for pre-normalised inputs: exercised on the fused cases
    for mfma in (([0, 1, 2, 3, 4] if fused else [0, 1, 2, 3]) if ops.mfma_ok(Cin, Cout, k) else [0]):
        wp = ops.pack_weights(wd, transpose=False, mfma=mfma)
        y5 = ops.new_act(N, D, H, W, Cout, DEV)
        ops.conv_fwd(x5, wp, b.to(DEV), y5, k, Cin, Cout, scale=None if scale is None else scale.to(DEV),
                     shift=None if shift is None else shift.to(DEV), act="relu" if fused else None, mfma=mfma)
        # mode 2 = split-bf16 products (~1e-5 relative each); modes 0/1 are exact fp32 FMA chains
        assert rel_err(from5(y5), yr.detach()) < (1e-4 if mfma == 2 else 2e-5), f"fwd mfma={mfma}"
    # gradient w.r.t. the conv input (of xhat when fused): g' = gy * (y > 0)
    gpre = gy * (yr.detach() > 0) if fused else gy
    g5 = to5(gpre)
    for mfma in ([0, 1, 2] if ops.mfma_ok(Cout, Cin, k) else [0]):
        wpt = ops.pack_weights(wd, transpose=True, mfma=mfma)
        gx5 = ops.new_act(N, D, H, W, Cin, DEV)
        ops.conv_fwd(g5, wpt, None, gx5, k, Cout, Cin, mfma=mfma)
        exp = xr.grad / scale[:, :, None, None, None] if fused else xr.grad
        assert rel_err(from5(gx5), exp) < (1e-4 if mfma == 2 else 2e-5), f"dgrad mfma={mfma}"
    for mfma in ([0, 1, 2] if ops.mfma_ok(Cin, Cout, k, wgrad=True) else [0]):
        dw = torch.empty(w.numel(), device=DEV)
        db = torch.empty(Cout, device=DEV)
        ops.conv_wgrad(x5, g5, k, Cin, Cout, dw, db, scale=None if scale is None else scale.to(DEV),
                       shift=None if shift is None else shift.to(DEV), mfma=mfma)
        assert rel_err(dw.cpu().view(w.shape), wr.grad) < (1e-4 if mfma == 2 else 5e-5), f"wgrad mfma={mfma}"
        assert rel_err(db.cpu(), br.grad) < 5e-5, f"bgrad mfma={mfma}"


MIXED_CASES = [
    (1, 8, 16, 16, 32, 64, (3, 3, 3)),    # shortest z-sliding column (D = 8; below it the patch wgrad kernel stays bf16x3)
    (1, 20, 24, 17, 32, 32, (3, 3, 3)),   # z-sliding wgrad, k-halves
    (1, 16, 16, 24, 64, 64, (3, 3, 3)),   # z-sliding wgrad, two Cout tiles
    (1, 1, 20, 33, 16, 64, (1, 3, 3)),    # 2-D
    (2, 4, 8, 8, 64, 32, (1, 1, 1)),
    (2, 8, 8, 8, 128, 128, (3, 3, 3)),    # split-K
    (1, 16, 32, 32, 64, 32, (1, 1, 1)),   # >= 16384 voxels: the data gradient runs the streaming 1x1x1 GEMM with one fp16 term
    (2, 16, 64, 64, 32, 32, (3, 3, 3)),   # 512 patches: the ping-pong kernel with one fp16 term, one Cout tile
    (2, 16, 64, 64, 32, 64, (3, 3, 3)),   # ... two Cout tiles (forward) / one (data gradient)
    (2, 32, 64, 64, 32, 32, (3, 3, 3)),   # 512 units of the z-reuse kernel: its one-term instantiations
]


@pytest.mark.parametrize("mode", [5, 7])
@pytest.mark.parametrize("case", MIXED_CASES)
def test_conv_mixed_precision_mode(case, mode):
    """use_mfma = 5 / 7 (mixed_precision=True with dtype float16 / bfloat16): operands rounded to fp16 / bf16, one MFMA per
    product, fp32 accumulation -- the arithmetic of torch.autocast(float16 / bfloat16) around nn.Conv3d (reference
    trainer/default_trainer.py:134-142).  Expected values: the fp32 convolution of the ROUNDED operands, so only the
    summation order differs (2e-5)."""
    ops = _ops()
    N, D, H, W, Cin, Cout, k = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    scale = torch.rand(N, Cin, generator=g) + 0.5
    shift = torch.randn(N, Cin, generator=g)
    pad = tuple(v // 2 for v in k)
    r16 = (lambda t: t.half().float()) if mode == 5 else (lambda t: t.bfloat16().float())
    # the kernel applies the pre-norm as ONE fused multiply-add before rounding to 16 bits: float64 reproduces that
    xn = (x.double() * scale[:, :, None, None, None].double() + shift[:, :, None, None, None].double()).float()
    xh = r16(xn)
    exp = F.relu(F.conv3d(xh, r16(w), b, padding=pad))
    x5, wd = to5(x), w.to(DEV)
    y5 = ops.new_act(N, D, H, W, Cout, DEV)
    ops.conv_fwd(x5, ops.pack_weights(wd, transpose=False, mfma=mode), b.to(DEV), y5, k, Cin, Cout, scale=scale.to(DEV),
                 shift=shift.to(DEV), act="relu", mfma=mode)
    assert rel_err(from5(y5), exp) < 2e-5
    # the same numbers are NOT the fp32 result: the mode really rounds (guards against a silent fp32 fallback)
    exact = F.relu(F.conv3d(xn, w, b, padding=pad))
    assert rel_err(from5(y5), exact) > 5e-5
    gy = torch.randn(exp.shape, generator=g)
    if ops.mfma_ok(Cout, Cin, k):
        wr = r16(w).requires_grad_(False)
        gxe = torch.nn.grad.conv3d_input(x.shape, wr, r16(gy), padding=pad)
        gx5 = ops.new_act(N, D, H, W, Cin, DEV)
        ops.conv_fwd(to5(gy), ops.pack_weights(wd, transpose=True, mfma=mode), None, gx5, k, Cout, Cin, mfma=mode)
        assert rel_err(from5(gx5), gxe) < 2e-5
    if ops.mfma_ok(Cin, Cout, k, wgrad=True):
        dw = torch.empty(w.numel(), device=DEV)
        db = torch.empty(Cout, device=DEV)
        ops.conv_wgrad(x5, to5(gy), k, Cin, Cout, dw, db, scale=scale.to(DEV), shift=shift.to(DEV), mfma=mode)
        zs = k == (3, 3, 3) and D >= 8       # the z-sliding kernel has the one-term variants; the patch kernel stays bf16x3
        xe = xh if zs else xn
        dwe = torch.nn.grad.conv3d_weight(xe, w.shape, r16(gy) if zs else gy, padding=pad)
        # (3e-5, not 2e-5: `dwe` is itself an fp32 sum over N*D*H*W terms in ATen's order; against float64 the kernels are
        # checked in test_wgrad_z_sliding_kernels_agree_with_float64 -- 2 x 8^3 x 128 -> 128 measures 2.5e-5 here)
        assert rel_err(dw.cpu().view(w.shape), dwe) < (3e-5 if zs else 1e-4)
        assert rel_err(db.cpu(), gy.sum((0, 2, 3, 4))) < 5e-5   # bias gradient sums the fp32 values


PP_CASES = [
    # N, D, H, W, Cin, Cout, k, (forward CT, dgrad CT) in auto mode: every case has >= 512 units for both directions
    ((2, 16, 64, 64, 32, 32, (3, 3, 3)), (1, 1)),    # level-0 encoder conv2 / decoder conv2 shape class
    ((2, 16, 64, 64, 32, 64, (3, 3, 3)), (2, 1)),    # level-1 encoder conv1: two Cout tiles forward, one tile dgrad
    ((2, 16, 64, 64, 128, 64, (3, 3, 3)), (2, 2)),   # decoder level 1: two tiles both ways, 8 chunks
    ((2, 18, 61, 67, 32, 64, (3, 3, 3)), (2, 1)),    # ragged borders in z, y, x (720 units)
    ((2, 17, 62, 66, 64, 32, (3, 3, 3)), (1, 2)),    # ragged, decoder level-0 conv1 shape class
    ((2, 16, 64, 64, 32, 32, (1, 3, 3)), (1, 1)),    # anisotropic 1x3x3 kernel, one tile
    ((2, 16, 64, 64, 64, 64, (1, 3, 3)), (2, 2)),    # anisotropic, two tiles
    ((2, 32, 64, 64, 32, 32, (3, 3, 3)), (1, 1)),    # 512 z-reuse units both ways: the shape class of the level-0 layers in auto mode
]


@pytest.fixture
def conv_variant_option():
    from torch_em_amd import _lib
    old = _lib.get_option("conv_fwd_variant")
    yield lambda v: _lib.set_option("conv_fwd_variant", v)
    _lib.set_option("conv_fwd_variant", old)


def _expected_family(variant, k, pp_tiles, zr_units):
    """The dispatch rule of tem_conv3d_fwd (csrc/conv_bf16x3.hip): z-reuse for 3x3x3 with >= 512 units (or when forced),
    else ping-pong (every PP_CASES shape has >= 512 ping-pong units), patch kernel only when asked for."""
    if variant == 0:
        return 0
    if variant == 1:
        return pp_tiles
    if k == (3, 3, 3) and (variant == 2 or zr_units >= 512):
        return 3
    return pp_tiles


def _zr_units(N, D, H, W, cout):
    return N * ((D + 3) // 4) * ((H + 15) // 16) * ((W + 7) // 8) * (cout // 32)


@pytest.mark.parametrize("case,cts", PP_CASES)
@pytest.mark.parametrize("variant", [-1, 0, 1, 2])
def test_conv_pingpong_family_parity(case, cts, variant, conv_variant_option):
    """Every instantiation of the two team kernels -- k_conv_zr<NS,F16,MODE> (csrc/conv_zr.hip, 3x3x3, z-reuse) and
    k_conv_pp<KD,3,3,..,CT in {1,2},NS,F16> (csrc/conv_pp.hip) -- that run the forward / data-gradient convolutions of the
    128^3 ... 32^3 levels (reference model/unet.py:417-438), against F.conv3d in fp32, and the SAME shapes on the
    one-patch-per-workgroup kernel (variant 0): bf16x3 1e-4, fp16x3 2e-5, one-term fp16 2e-5 against the convolution of the
    fp16-rounded operands.  tem_conv3d_fwd_kernel() says which family a launch takes (0 patch kernel, 1 / 2 ping-pong with
    that many Cout tiles per team, 3 z-reuse), so a silent fallback cannot pass as coverage."""
    ops = _ops()
    from torch_em_amd import _lib
    lib = _lib.load()
    conv_variant_option(variant)
    N, D, H, W, Cin, Cout, k = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    scale, shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g)
    pad = tuple(v // 2 for v in k)
    xn = (x.double() * scale[:, :, None, None, None].double() + shift[:, :, None, None, None].double()).float()
    r16 = lambda t: t.half().float()
    exp = F.relu(F.conv3d(xn, w, b, padding=pad))
    exp16 = F.relu(F.conv3d(r16(xn), r16(w), b, padding=pad))
    x5, wd = to5(x), w.to(DEV)
    for mode in (2, 4, 5):
        fam = lib.tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, k[0], k[1], k[2], mode)
        assert fam == _expected_family(variant, k, cts[0], _zr_units(N, D, H, W, Cout)), (mode, fam)
        y5 = ops.new_act(N, D, H, W, Cout, DEV)
        ops.conv_fwd(x5, ops.pack_weights(wd, transpose=False, mfma=mode), b.to(DEV), y5, k, Cin, Cout, scale=scale.to(DEV),
                     shift=shift.to(DEV), act="relu", mfma=mode)
        err = rel_err(from5(y5), exp16 if mode == 5 else exp)
        assert err < (1e-4 if mode == 2 else 2e-5), f"fwd mode {mode}: {err}"
    # data gradient = the same kernel on the transposed weights (Cout -> Cin channels), no norm, no activation; the
    # fp16x3 data gradient (mode 4) reads a raw gradient: exercised at two magnitudes (the engine prescales by a power of two)
    gy = torch.randn(exp.shape, generator=g)
    gxe = torch.nn.grad.conv3d_input(x.shape, w, gy, padding=pad)
    gxe16 = torch.nn.grad.conv3d_input(x.shape, r16(w), r16(gy), padding=pad)
    g5 = to5(gy)
    for mode in (2, 4, 5):
        fam = lib.tem_conv3d_fwd_kernel(N, D, H, W, Cout, Cin, k[0], k[1], k[2], mode)
        assert fam == _expected_family(variant, k, cts[1], _zr_units(N, D, H, W, Cin)), (mode, fam)
        gx5 = ops.new_act(N, D, H, W, Cin, DEV)
        ops.conv_fwd(g5, ops.pack_weights(wd, transpose=True, mfma=mode), None, gx5, k, Cout, Cin, mfma=mode)
        err = rel_err(from5(gx5), gxe16 if mode == 5 else gxe)
        assert err < (1e-4 if mode == 2 else 2e-5), f"dgrad mode {mode}: {err}"
        # masked data gradient (the ReLU mask of the producing layer applied in the epilogue: `ref`)
        if mode != 5:
            refm = torch.randn(N, Cin, D, H, W, generator=torch.Generator().manual_seed(3))
            ops.conv_fwd(g5, ops.pack_weights(wd, transpose=True, mfma=mode), None, gx5, k, Cout, Cin, mfma=mode, ref=to5(refm))
            err = rel_err(from5(gx5), gxe * (refm > 0))
            assert err < (1e-4 if mode == 2 else 2e-5), f"masked dgrad mode {mode}: {err}"


@pytest.mark.parametrize("case", [(2, 16, 16, 16, 256, 256), (2, 16, 16, 16, 128, 256), (2, 15, 16, 15, 128, 256),
                                  (1, 32, 32, 32, 128, 64), (2, 8, 8, 8, 512, 512)])
def test_conv_zreuse_split_k(case):
    """Family 4 of tem_conv3d_fwd_kernel: the z-reuse kernel with the input channels split over several units (the 16^3 /
    32^3 levels of the U-Net, reference model/unet.py:417-438: too few 4x16x8 tiles to give every team of every CU one)
    plus the summing epilogue that applies bias / ReLU / the ReLU mask -- forward with the fused pre-norm, data gradient,
    masked data gradient, against F.conv3d; option zr_splitk = 0 sends the same shapes to the split-K patch kernel."""
    ops = _ops()
    from torch_em_amd import _lib
    lib = _lib.load()
    N, D, H, W, Cin, Cout = case
    k, pad = (3, 3, 3), (1, 1, 1)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    scale, shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g)
    xn = (x.double() * scale[:, :, None, None, None].double() + shift[:, :, None, None, None].double()).float()
    r16 = lambda t: t.half().float()  # noqa: E731
    exp = F.relu(F.conv3d(xn, w, b, padding=pad))
    exp16 = F.relu(F.conv3d(r16(xn), r16(w), b, padding=pad))
    gy = torch.randn(exp.shape, generator=g)
    refm = torch.randn(N, Cin, D, H, W, generator=g)
    gxe = torch.nn.grad.conv3d_input(x.shape, w, gy, padding=pad)
    x5, g5, wd = to5(x), to5(gy), w.to(DEV)
    for splitk in (1, 0):
        _lib.set_option("zr_splitk", splitk)
        try:
            for mode in (2, 4, 5):
                fam = lib.tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, 3, 3, 3, mode)
                assert fam == (4 if splitk else 0), (mode, fam)
                # the split-K epilogue writes the statistics partials (round 4): 4 rows of 256 / (Cout / 4) voxels per block;
                # the patch kernel's split-K (option off) cannot
                vb = 4 * (256 // (Cout // 4))
                assert lib.tem_conv3d_fwd_stat_blocks(N, D, H, W, Cin, Cout, 3, 3, 3, mode) == \
                    ((D * H * W + vb - 1) // vb if splitk else 0)
                y5 = torch.full((N, D, H, W, Cout + 4), 3.0, device=DEV)                                # a channel slice of a wider buffer
                ops.conv_fwd(x5, ops.pack_weights(wd, transpose=False, mfma=mode), b.to(DEV), y5[..., :Cout], k, Cin, Cout,
                             scale=scale.to(DEV), shift=shift.to(DEV), act="relu", mfma=mode)
                err = rel_err(from5(y5[..., :Cout]), exp16 if mode == 5 else exp)
                assert err < (1e-4 if mode == 2 else 2e-5), f"fwd mode {mode} splitk {splitk}: {err}"
                assert float(y5[..., Cout:].min()) == 3.0 and float(y5[..., Cout:].max()) == 3.0
            for mode in (2, 5):
                assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cout, Cin, 3, 3, 3, mode) in ((4, 3) if splitk else (0,))
                gx5 = ops.new_act(N, D, H, W, Cin, DEV)
                ops.conv_fwd(g5, ops.pack_weights(wd, transpose=True, mfma=mode), None, gx5, k, Cout, Cin, mfma=mode, ref=to5(refm))
                want = (torch.nn.grad.conv3d_input(x.shape, r16(w), r16(gy), padding=pad) if mode == 5 else gxe) * (refm > 0)
                err = rel_err(from5(gx5), want)
                assert err < (1e-4 if mode == 2 else 2e-5), f"masked dgrad mode {mode} splitk {splitk}: {err}"
        finally:
            _lib.set_option("zr_splitk", 1)


@pytest.mark.parametrize("case", [(2, 32, 64, 64, 32, 32), (2, 37, 61, 70, 64, 32), (2, 16, 64, 64, 96, 64), (2, 32, 64, 64, 48, 32)])
def test_conv_zreuse_wide_chunks_and_tile_order(case):
    """Two dispatch choices of the z-reuse kernel (csrc/conv_zr.hip; forward / data-gradient convolutions of ConvBlock,
    reference model/unet.py:417-438): "zr_wide" (the one-term modes 5 / 7 stage 32 input channels = whole 128-byte lines
    per phase when Cin % 32 == 0; 48 channels fall back to 16 per phase) and "zr_tile_blocks" (tiles walked in 4 x 4 x 4
    blocks, any extent that is not a multiple of 4 / 2 tiles shrinks the block).  The tile order must not change a single
    bit (a unit's arithmetic is the same, the fused statistics are indexed by position); the wide kernel sums the same
    products in another order: fp32-rounding close to the narrow one, and both against F.conv3d of the rounded operands."""
    ops = _ops()
    from torch_em_amd import _lib
    lib = _lib.load()
    N, D, H, W, Cin, Cout = case
    k, pad = (3, 3, 3), (1, 1, 1)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    scale, shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g)
    xn = (x.double() * scale[:, :, None, None, None].double() + shift[:, :, None, None, None].double()).float()
    x5, wd = to5(x), w.to(DEV)
    out = {}
    try:
        for mode in (4, 5, 7):
            rnd = (lambda t: t) if mode == 4 else (lambda t: t.half().float()) if mode == 5 else (lambda t: t.bfloat16().float())
            exp = F.relu(F.conv3d(rnd(xn), rnd(w), b, padding=pad))
            assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, 3, 3, 3, mode) == 3
            for wide in (1, 0):
                for blocks in (1, 0):
                    _lib.set_option("zr_wide", wide)
                    _lib.set_option("zr_tile_blocks", blocks)
                    y5 = ops.new_act(N, D, H, W, Cout, DEV)
                    stat, _nb = ops.conv_fwd(x5, ops.pack_weights(wd, transpose=False, mfma=mode), b.to(DEV), y5, k, Cin, Cout,
                                             scale=scale.to(DEV), shift=shift.to(DEV), act="relu", mfma=mode, want_stats=True)
                    out[(mode, wide, blocks)] = (y5.clone(), stat.clone())
                    err = rel_err(from5(y5), exp)
                    assert err < 2e-5, f"mode {mode} wide {wide} blocks {blocks}: {err}"
                    assert rel_err(stat[..., 0].sum(1).cpu(), exp.sum((2, 3, 4))) < 1e-4
                assert torch.equal(out[(mode, wide, 1)][0], out[(mode, wide, 0)][0]), f"mode {mode}: tile order changed the output"
                assert torch.equal(out[(mode, wide, 1)][1], out[(mode, wide, 0)][1]), f"mode {mode}: tile order changed the statistics"
            d = rel_err(out[(mode, 1, 1)][0].cpu(), out[(mode, 0, 1)][0].cpu())
            assert d < 2e-6, f"mode {mode}: wide vs narrow {d}"
            if mode == 4:   # no wide variant of the two-term kernels
                assert torch.equal(out[(mode, 1, 1)][0], out[(mode, 0, 1)][0])
    finally:
        _lib.set_option("zr_wide", 1)
        _lib.set_option("zr_tile_blocks", 1)


@pytest.mark.parametrize("case", [(2, 32, 64, 64, 32, 32), (2, 18, 61, 67, 64, 32), (2, 16, 64, 64, 64, 64)])
@pytest.mark.parametrize("gscale", [1.0, 3e-7, 5e4])
def test_data_gradient_fp16_two_term_with_device_prescale(case, gscale):
    """tem_conv3d_wgrad_gmax + tem_conv3d_fwd_gscaled: the weight gradient reports max |g| (bit pattern, integer atomicMax),
    the data gradient (dgrad half of convolution_backward, reference model/unet.py:417-438) prescales g by the power of
    two that puts that maximum at 2^14, runs the fp16 two-term layout and unscales: fp32-class (2e-5 against F.conv3d's
    fp32 data gradient, the bf16x3 kernels have 1e-4) for gradients of any magnitude -- 3e-7 sits far below fp16's
    normal range, 5e4 next to its overflow."""
    ops = _ops()
    N, D, H, W, Cin, Cout = case
    k = (3, 3, 3)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.2
    gy = torch.randn(N, Cout, D, H, W, generator=g) * gscale
    gy[0, 0, 0, 0, :4] = 0.0
    refm = torch.randn(N, Cin, D, H, W, generator=g)
    gxe = torch.nn.grad.conv3d_input(x.shape, w, gy, padding=(1, 1, 1))
    dwe = torch.nn.grad.conv3d_weight(x, w.shape, gy, padding=(1, 1, 1))
    x5, g5, wd = to5(x), to5(gy), w.to(DEV)
    assert ops.conv_wgrad_gmax_ok(g5, k, Cin, Cout, 2) and ops.conv_fwd_family(g5, k, Cout, Cin, 4) == 3
    amax = torch.zeros(1, dtype=torch.int32, device=DEV)
    dw = torch.empty(w.numel(), device=DEV)
    db = torch.empty(Cout, device=DEV)
    ops.conv_wgrad_gmax(x5, g5, k, Cin, Cout, dw, db, amax)
    assert rel_err(dw.cpu().view(w.shape), dwe) < 1e-4 and rel_err(db.cpu(), gy.sum((0, 2, 3, 4))) < 5e-5
    assert int(amax.cpu()[0]) == int(gy.abs().max().view(torch.int32))          # exact: the bit pattern of max |g|
    wp = ops.pack_weights(wd, transpose=True, mfma=4)
    gx5 = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd_gscaled(g5, wp, gx5, k, Cout, Cin, amax)
    err = rel_err(from5(gx5), gxe)
    assert err < 2e-5, err
    ops.conv_fwd_gscaled(g5, wp, gx5, k, Cout, Cin, amax, ref=to5(refm))          # ReLU mask of the producing layer
    err = rel_err(from5(gx5), gxe * (refm > 0))
    assert err < 2e-5, err
    # the bf16x3 data gradient of the same operands: the 16-bit class this path replaces (guards against a silent fallback)
    gx2 = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd(g5, ops.pack_weights(wd, transpose=True, mfma=2), None, gx2, k, Cout, Cin, mfma=2)
    assert not torch.equal(gx2, gx5)


@pytest.mark.parametrize("mode", [2, 5])
@pytest.mark.parametrize("case", [(2, 32, 64, 64, 32, 32), (2, 18, 61, 67, 64, 32), (2, 16, 64, 64, 64, 64)])
def test_data_gradient_with_norm_backward_epilogue(case, mode):
    """tem_conv3d_fwd_refnorm: the data gradient of a block's second conv leaves the z-reuse kernel with the backward of the
    norm in front of that conv and the ReLU mask of the first conv's output already applied (reference: autograd's
    native_group_norm_backward + threshold_backward after convolution_backward, model/unet.py:417-438) -- equal to the
    plain data gradient followed by the elementwise pass of tem_norm_bwd_from_sums it replaces."""
    ops = _ops()
    N, D, H, W, Cin, Cout = case              # the conv maps Cin -> Cout; its data gradient Cout -> Cin
    k = (3, 3, 3)
    g = torch.Generator().manual_seed(23)
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
    g5 = to5(torch.randn(N, Cout, D, H, W, generator=g))
    a1 = to5(torch.relu(torch.randn(N, Cin, D, H, W, generator=g) + 0.2))     # a ReLU output: ~45 % zeros
    coef = torch.randn(N, Cin, 4, generator=g).to(DEV)
    assert ops.conv_fwd_family(g5, k, Cout, Cin, mode) == 3
    wp = ops.pack_weights(w, transpose=True, mfma=mode)
    plain = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd(g5, wp, None, plain, k, Cout, Cin, mfma=mode)
    kc = coef.view(N, 1, 1, 1, Cin, 4)
    want = torch.where(a1 > 0, kc[..., 0] * plain - kc[..., 1] - (a1 - kc[..., 3]) * kc[..., 2], torch.zeros_like(plain))
    got = torch.full((N, D, H, W, Cin + 8), 7.0, device=DEV)                  # a channel slice of a wider buffer
    ops.conv_fwd_refnorm(g5, wp, got[..., :Cin], k, Cout, Cin, a1, coef, mode)
    assert float((got[..., :Cin] - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float(got[..., Cin:].min()) == 7.0 and float(got[..., Cin:].max()) == 7.0
    with pytest.raises(ValueError):                                            # a shape the z-reuse kernel does not take
        small = to5(torch.randn(1, Cout, 4, 8, 8, generator=g))
        ops.conv_fwd_refnorm(small, wp, ops.new_act(1, 4, 8, 8, Cin, DEV), k, Cout, Cin,
                             ops.new_act(1, 4, 8, 8, Cin, DEV), coef[:1].contiguous(), mode)


@pytest.mark.parametrize("mode", [2, 5, 7])
def test_norm_backward_epilogue_is_deterministic_over_repeated_launches(mode):
    """Regression test of a store-data hazard (zr_store4, csrc/conv_zr.hip): with an SGPR soffset on the 16-byte buffer
    store the compiler let the next row's v_cndmask overwrite the first data register, and the MODE 3 epilogue (data gradient
    + ReLU mask + norm backward, what autograd does after convolution_backward behind model/unet.py:417-438) stored the x
    component of row m + 1 into row m in a few lanes -- in 1 % to 99 % of the launches depending on register allocation.
    120 launches of the shape that showed it must agree bit for bit, and with the unfused two-pass result."""
    ops = _ops()
    N, D, H, W, Cin, Cout = 2, 18, 61, 67, 64, 32
    k = (3, 3, 3)
    g = torch.Generator().manual_seed(23)
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
    g5 = to5(torch.randn(N, Cout, D, H, W, generator=g))
    a1 = to5(torch.relu(torch.randn(N, Cin, D, H, W, generator=g) + 0.2))
    coef = torch.randn(N, Cin, 4, generator=g).to(DEV)
    wp = ops.pack_weights(w, transpose=True, mfma=mode)
    plain = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd(g5, wp, None, plain, k, Cout, Cin, mfma=mode)
    kc = coef.view(N, 1, 1, 1, Cin, 4)
    want = torch.where(a1 > 0, kc[..., 0] * plain - kc[..., 1] - (a1 - kc[..., 3]) * kc[..., 2], torch.zeros_like(plain))
    first = None
    junk = torch.empty(16 << 20, device=DEV)
    for i in range(120):
        got = torch.full((N, D, H, W, Cin), float("nan"), device=DEV)
        ops.conv_fwd_refnorm(g5, wp, got, k, Cout, Cin, a1, coef, mode)
        if i % 3 == 0:
            junk.normal_()   # other traffic between the launches
        if first is None:
            first = got.clone()
            assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
        else:
            assert torch.equal(got, first), f"launch {i} differs from launch 0 in {int((got != first).sum())} elements"


@pytest.mark.parametrize("case", [
    (2, 18, 61, 67, 32, 64, (3, 3, 3), 32),   # z-reuse kernel (720 units), ragged in z, y and x, two column tiles, GroupNorm(32, 64)
    (2, 32, 64, 64, 32, 32, (3, 3, 3), 32),   # z-reuse kernel, exactly one unit per team
    (2, 17, 50, 66, 32, 32, (3, 3, 3), 32),   # ragged patches, one column tile
    (1, 24, 64, 64, 32, 64, (3, 3, 3), 32),   # 2x2 wave tiling, GroupNorm(32, 64)
    (2, 8, 48, 48, 64, 96, (1, 3, 3), 96),    # three column tiles
    (1, 1, 330, 400, 16, 64, (1, 3, 3), 64),  # 2-D patches
    (2, 16, 64, 64, 64, 32, (1, 1, 1), 1),    # 1x1x1, a single group
    (2, 8, 8, 8, 256, 512, (3, 3, 3), 512),   # z-reuse kernel with split input channels: the split-K epilogue writes them
    (2, 16, 16, 16, 128, 256, (3, 3, 3), 32), # the same at 16^3, GroupNorm(32, 256)
    (4, 6, 12, 12, 128, 128, (3, 3, 3), 128), # ragged split-K tiles (the 6 x 12 x 12 level of cfg 5)
])
@pytest.mark.parametrize("mode", [2, 3, 4])
def test_conv_fused_forward_statistics(case, mode):
    """tem_conv3d_fwd_stats + tem_norm_finalize_partials == tem_norm_stats of the stored output (what the next
    InstanceNorm / GroupNorm / BatchNorm of a ConvBlock needs, reference model/unet.py:429-438), without reading it."""
    ops = _ops()
    N, D, H, W, Cin, Cout, k, groups = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    scale, shift = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV), torch.randn(N, Cin, generator=g).to(DEV)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    x5 = to5(x)
    y5 = ops.new_act(N, D, H, W, Cout, DEV)
    wp = ops.pack_weights(w, transpose=False, mfma=mode)
    got = ops.conv_fwd(x5, wp, b, y5, k, Cin, Cout, scale=scale, shift=shift, act="relu", mfma=mode, want_stats=True)
    if got is None:
        # the only launches that may decline: split-K of the PATCH kernel (family 0: e.g. mode 3 has no z-reuse split-K launch)
        from torch_em_amd import _lib
        assert _lib.load().tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, *k, mode) == 0 and D * H * W <= 16 ** 3
        return
    part, nblk = got
    y_ref = ops.new_act(N, D, H, W, Cout, DEV)
    ops.conv_fwd(x5, wp, b, y_ref, k, Cin, Cout, scale=scale, shift=shift, act="relu", mfma=mode)
    assert torch.equal(y5, y_ref)                                     # the output itself is unchanged
    V = D * H * W
    for rows in (N, 1):                                               # per-sample statistics / BatchNorm over the batch
        if rows == 1:
            yb = y5.reshape(1, N * D, H, W, Cout)
            want = ops.norm_stats(yb, groups, gamma, beta, 1e-5)
        else:
            want = ops.norm_stats(y5, groups, gamma, beta, 1e-5)
        have = ops.norm_stats_from_partials(part, rows, V, Cout, groups, gamma, beta, 1e-5)
        for a, c, name in zip(have, want, ("mean", "rstd", "scale", "shift")):
            assert a.shape == c.shape and rel_err(a.cpu(), c.cpu()) < 2e-6, (name, rows)
    # launches that cannot provide them say so: the VALU kernels here, split-K shapes in the model tests
    assert ops.conv_fwd(x5, ops.pack_weights(w, transpose=False, mfma=0), b, y_ref, k, Cin, Cout, scale=scale,
                        shift=shift, act="relu", mfma=0, want_stats=True) is None


@pytest.mark.parametrize("case", [
    (2, 16, 32, 24, 32, (2, 2, 2), 32),     # InstanceNorm groups
    (1, 8, 20, 12, 64, (1, 2, 2), 8),       # anisotropic factor, GroupNorm(8, 64)
    (3, 6, 6, 12, 20, (3, 3, 3), 20),       # 27-voxel windows (general path), C / 4 = 5 does not divide 256: declines
    (2, 4, 8, 8, 512, (2, 2, 2), 32),       # 128 channel quads
])
def test_maxpool_forward_also_delivers_the_statistics_of_its_output(case):
    """tem_maxpool3d_fwd_stats + tem_norm_finalize_partials == tem_norm_stats of the pooled tensor (the norm in front of the
    next encoder block's first conv, reference model/unet.py:311-321), without the pass over it."""
    ops = _ops()
    N, D, H, W, C, f, groups = case
    gen = torch.Generator().manual_seed(9)
    x5 = to5(torch.randn(N, C, D, H, W, generator=gen) * 2.0 + 0.5)
    Do, Ho, Wo = D // f[0], H // f[1], W // f[2]
    y, y_ref = ops.new_act(N, Do, Ho, Wo, C, DEV), ops.new_act(N, Do, Ho, Wo, C, DEV)
    got = ops.maxpool_fwd(x5, y, f, want_stats=True)
    ops.maxpool_fwd(x5, y_ref, f)
    assert torch.equal(y, y_ref)
    cq = C // 4 if C % 4 == 0 else C
    if 256 % cq:
        assert got is None
        return
    part, nblk = got
    assert nblk == Do * Ho and tuple(part.shape) == (N, nblk, C, 2)
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(DEV), torch.randn(C, generator=gen).to(DEV)
    for rows in (N, 1):
        want = ops.norm_stats(y if rows == N else y.reshape(1, N * Do, Ho, Wo, C), groups, gamma, beta, 1e-5)
        have = ops.norm_stats_from_partials(part, rows, Do * Ho * Wo, C, groups, gamma, beta, 1e-5)
        for a, c, name in zip(have, want, ("mean", "rstd", "scale", "shift")):
            assert a.shape == c.shape and rel_err(a.cpu(), c.cpu()) < 2e-6, (name, rows)


@pytest.mark.parametrize("case", [(2, 9, 17, 10, 1, 32, (3, 3, 3)), (1, 1, 19, 21, 1, 16, (1, 3, 3)),
                                  (2, 5, 9, 12, 3, 32, (3, 3, 3))])
def test_first_layer_fused_forward_statistics(case):
    """the small-Cin first-layer kernel (VALU, HBM-bound) provides the same per-patch partial sums"""
    ops = _ops()
    N, D, H, W, Cin, Cout, k = case
    g = torch.Generator().manual_seed(4)
    x5 = to5(torch.randn(N, Cin, D, H, W, generator=g))
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.3).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    scale, shift = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV), torch.randn(N, Cin, generator=g).to(DEV)
    wp = ops.pack_weights(w, transpose=False, mfma=0)
    y5, y_ref = ops.new_act(N, D, H, W, Cout, DEV), ops.new_act(N, D, H, W, Cout, DEV)
    part, nblk = ops.conv_fwd(x5, wp, b, y5, k, Cin, Cout, scale=scale, shift=shift, act="relu", mfma=0, want_stats=True)
    ops.conv_fwd(x5, wp, b, y_ref, k, Cin, Cout, scale=scale, shift=shift, act="relu", mfma=0)
    tx = 32 if Cin == 1 else 8     # Cin = 1: 4x8x32 tiles of the row kernel (k_conv_fwd_c1rows), else 4x8x8 patches
    assert torch.equal(y5, y_ref) and nblk == ((D + 3) // 4) * ((H + 7) // 8) * ((W + tx - 1) // tx)
    want = ops.norm_stats(y5, Cout, None, None, 1e-5)
    have = ops.norm_stats_from_partials(part, N, D * H * W, Cout, Cout, None, None, 1e-5)
    for a, c in zip(have, want):
        assert rel_err(a.cpu(), c.cpu()) < 2e-6


@pytest.mark.parametrize("case", [
    (2, 16, 16, 24, 32, 32, True),     # k-halves kernel, affine norm (GroupNorm-style gamma / beta)
    (1, 20, 24, 17, 32, 32, False),    # ragged H / W, two z segments, InstanceNorm (no affine)
    (2, 16, 16, 16, 64, 64, True),     # two Cout tiles
    (3, 17, 9, 10, 64, 32, False),     # Cin tiles > 1, odd depth, three samples
])
def test_wgrad_delivers_norm_backward_sums(case):
    """tem_conv3d_wgrad_sums: (sum_v gz, sum_v gz*xn) of the norm in front of a conv from the per-sample weight
    gradient and the boundary shell of g -- against the definition evaluated in float64 (gz = conv data gradient)."""
    ops = _ops()
    N, D, H, W, Cin, Cout, affine = case
    k = (3, 3, 3)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cin, D, H, W, generator=g).double() * 1.5 + 0.3
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.1).double()
    gout = torch.randn(N, Cout, D, H, W, generator=g).double()
    gamma = (torch.rand(Cin, generator=g) + 0.5).double() if affine else torch.ones(Cin, dtype=torch.float64)
    beta = torch.randn(Cin, generator=g).double() if affine else torch.zeros(Cin, dtype=torch.float64)
    mean = x.mean((2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var((2, 3, 4), unbiased=False) + 1e-5)
    xn = (x - mean[:, :, None, None, None]) * rstd[:, :, None, None, None]
    z = xn * gamma[None, :, None, None, None] + beta[None, :, None, None, None]
    gz = torch.nn.grad.conv3d_input(x.shape, w, gout, padding=1)
    A_ref, B_ref = gz.sum((2, 3, 4)), (gz * xn).sum((2, 3, 4))
    dw_ref = torch.nn.grad.conv3d_weight(z, w.shape, gout, padding=1)
    scale = (rstd * gamma[None]).float().to(DEV)
    shift = (beta[None] - mean * rstd * gamma[None]).float().to(DEV)
    x5, g5 = to5(x.float()), to5(gout.float())
    if not ops.conv_wgrad_sums_ok(x5, k, Cin, Cout, 2):
        pytest.skip("the wgrad_sums options exclude this size (tests/conftest.py sets the threshold to 0)")
    dw = torch.empty(w.numel(), device=DEV)
    db = torch.empty(Cout, device=DEV)
    sums = ops.conv_wgrad(x5, g5, k, Cin, Cout, dw, db, scale=scale, shift=shift, mfma=2,
                          sums_from=(w.float().to(DEV), gamma.float().to(DEV) if affine else None,
                                     beta.float().to(DEV) if affine else None))
    assert rel_err(dw.cpu().view(w.shape), dw_ref) < 1e-4 and rel_err(db.cpu(), gout.sum((0, 2, 3, 4))) < 5e-5
    A, B = sums[..., 0].cpu().double(), sums[..., 1].cpu().double()
    # both are sums of ~V signed terms: judge them against the size of the terms, sqrt(V) * rms
    sa = float(gz.pow(2).sum((2, 3, 4)).sqrt().max())
    sb = float((gz * xn).pow(2).sum((2, 3, 4)).sqrt().max())
    assert float((A - A_ref).abs().max()) < 1e-4 * sa, float((A - A_ref).abs().max()) / sa
    assert float((B - B_ref).abs().max()) < 1e-4 * sb, float((B - B_ref).abs().max()) / sb
    # and through the norm backward: same result as the two-pass path
    gz5, xx5 = to5(gz.float()), to5(x.float())
    outs = []
    for sm in (None, sums):
        gx = torch.empty_like(gz5)
        dgam = torch.zeros(Cin, device=DEV) if affine else None
        dbet = torch.zeros(Cin, device=DEV) if affine else None
        ops.norm_bwd(gz5, xx5, Cin, gamma.float().to(DEV) if affine else None, mean.float().to(DEV),
                     rstd.float().to(DEV), True, gx, dgam, dbet, sums=sm)
        outs.append((gx, dgam, dbet))
    assert rel_err(outs[1][0].cpu(), outs[0][0].cpu()) < 2e-5
    if affine:
        assert float((outs[1][1] - outs[0][1]).abs().max()) < 1e-4 * sb and \
            float((outs[1][2] - outs[0][2]).abs().max()) < 1e-4 * sa


@pytest.mark.parametrize("case", [
    (2, 16, 16, 24, 32, 32, 32),    # InstanceNorm: one channel per group
    (2, 16, 16, 16, 64, 64, 16),    # GroupNorm without affine, 4 channels per group, two blocks of 32 channels
    (1, 20, 24, 17, 32, 32, 1),     # one group of 32 channels
    (3, 17, 9, 10, 64, 32, 1),      # one group of 64 channels: wider than a block -> the request stays armed
    (2, 16, 16, 16, 96, 32, 32),    # 3 channels per group: not a power of two -> not delivered
])
def test_wgrad_sums_also_deliver_the_norm_backward_coefficients(case):
    """TEM_BP_NORM_COEF of tem_conv3d_wgrad_ex: the kernel that finishes the norm sums also writes coef [N, C, 4] -- bit for
    bit what tem_norm_bwd_coef(sums=...) derives from them -- and reports it as not delivered when the group layout does not fit."""
    ops = _ops()
    N, D, H, W, Cin, Cout, G = case
    k = (3, 3, 3)
    gen = torch.Generator().manual_seed(11)
    x5 = to5(torch.randn(N, Cin, D, H, W, generator=gen) * 1.5 + 0.3)
    g5 = to5(torch.randn(N, Cout, D, H, W, generator=gen))
    w = (torch.randn(Cout, Cin, *k, generator=gen) * 0.1).to(DEV)
    if not ops.conv_wgrad_sums_ok(x5, k, Cin, Cout, 2):
        pytest.skip("the wgrad_sums options exclude this size (tests/conftest.py sets the threshold to 0)")
    mean, rstd, scale, shift = ops.norm_stats(x5, G, None, None, 1e-5)[:4]
    dw, db = torch.empty(w.numel(), device=DEV), torch.empty(Cout, device=DEV)
    coef = torch.full((N, Cin, 4), float("nan"), device=DEV)
    bp = ops.Byproducts(norm_coef=(G, mean, rstd, coef))
    sums = ops.conv_wgrad(x5, g5, k, Cin, Cout, dw, db, scale=scale, shift=shift, mfma=2, sums_from=(w, None, None), bp=bp)
    left = not bp.coef
    cg = Cin // G
    assert left == (not (cg <= 32 and cg & (cg - 1) == 0))
    assert not (bp.amax or bp.sums)
    want = ops.norm_bwd_coef(x5, x5, G, None, mean, rstd, sums=sums)   # with sums the tensors are not read
    if left:
        assert bool(torch.isnan(coef).all())          # untouched
    else:
        assert torch.equal(coef, want)
    # nothing survives the call: the same launch without the argument leaves coef alone
    coef2 = coef.clone()
    sums2 = ops.conv_wgrad(x5, g5, k, Cin, Cout, dw, db, scale=scale, shift=shift, mfma=2, sums_from=(w, None, None))
    assert torch.equal(sums2, sums)
    assert bool(torch.isnan(coef).all()) if left else torch.equal(coef2, coef)
    # a request that makes no sense for the entry point is an error, not a silent no-op
    with pytest.raises(Exception, match="by-product"):
        ops.conv_wgrad(x5, g5, k, Cin, Cout, dw, db, scale=scale, shift=shift, mfma=2, sums_from=(w, None, None),
                       bp=ops.Byproducts(out_amax=torch.zeros(1, dtype=torch.int32, device=DEV)))


@pytest.mark.parametrize("case", [
    (2, 16, 16, 16, 256, 128, 128, True),   # InstanceNorm-style groups, in-place apply with the ReLU mask
    (2, 8, 8, 8, 512, 256, 32, False),      # GroupNorm(32, 256), coefficients only
    (4, 6, 12, 12, 128, 128, 128, True),    # ragged split-K tiles
])
def test_split_k_data_gradient_delivers_the_norm_backward_rows(case):
    """TEM_BP_NORM_SUMS of tem_conv3d_fwd_ex: the split-K epilogue of a data gradient writes per-block (sum g, sum g * xn) of the norm in
    front of the conv; tem_norm_bwd_from_partials on those rows == tem_norm_bwd on the tensors (its own reduction pass)."""
    ops = _ops()
    N, D, H, W, Cout, Cin, G, apply = case          # the conv is Cin -> Cout; its data gradient Cout -> Cin
    k = (3, 3, 3)
    gen = torch.Generator().manual_seed(5)
    x5 = to5(torch.randn(N, Cin, D, H, W, generator=gen) * 1.3 + 0.2)     # input of the norm in front of the conv
    g5 = to5(torch.randn(N, Cout, D, H, W, generator=gen))
    w = (torch.randn(Cout, Cin, *k, generator=gen) * 0.05).to(DEV)
    mode = 2
    if ops.conv_fwd_family(g5, k, Cout, Cin, mode) != 4:
        pytest.skip("not a split-K launch of the z-reuse kernel on this device")
    wp = ops.pack_weights(w, transpose=True, mfma=mode)
    mean, rstd = ops.norm_stats(x5, G, None, None, 1e-5)[:2]
    nblk = ops.conv_fwd_stat_blocks(g5, k, Cout, Cin, mode)
    assert nblk > 0
    part = torch.full((N, nblk, Cin, 2), float("nan"), device=DEV)
    gx, gx_ref = ops.new_act(N, D, H, W, Cin, DEV), ops.new_act(N, D, H, W, Cin, DEV)
    bp = ops.Byproducts(norm_sums=(x5, G, mean, rstd, part))
    ops.conv_fwd(g5, wp, None, gx, k, Cout, Cin, mfma=mode, bp=bp)
    assert bp.sums and not bp.amax                                # delivered
    ops.conv_fwd(g5, wp, None, gx_ref, k, Cout, Cin, mfma=mode)
    assert torch.equal(gx, gx_ref)                                # the gradient itself is unchanged
    sums = part.sum(1).cpu().double()
    xn = (from5(x5).double() - mean.cpu().double().repeat_interleave(Cin // G, 1)[:, :, None, None, None]) * \
        rstd.cpu().double().repeat_interleave(Cin // G, 1)[:, :, None, None, None]
    gz = from5(gx).double()
    A, B = gz.sum((2, 3, 4)), (gz * xn).sum((2, 3, 4))
    assert float((sums[..., 0] - A).abs().max()) < 2e-5 * float(gz.abs().sum((2, 3, 4)).max())
    assert float((sums[..., 1] - B).abs().max()) < 2e-5 * float((gz * xn).abs().sum((2, 3, 4)).max())
    if apply:
        a, b = gx.clone(), gx.clone()
        ops.norm_bwd(a, x5, G, None, mean, rstd, True, a)
        ops.norm_bwd(b, x5, G, None, mean, rstd, True, b, sums=part)
        assert rel_err(b.cpu(), a.cpu()) < 2e-5
    else:
        assert rel_err(ops.norm_bwd_coef(gx, x5, G, None, mean, rstd, sums=part).cpu(),
                       ops.norm_bwd_coef(gx, x5, G, None, mean, rstd).cpu()) < 2e-5
    # a launch that cannot deliver the rows says so and leaves them alone
    part2 = torch.full_like(part, float("nan"))
    bp = ops.Byproducts(norm_sums=(x5, G, mean, rstd, part2))
    ops.conv_fwd(g5, ops.pack_weights(w, transpose=True, mfma=0), None, gx_ref, k, Cout, Cin, mfma=0, bp=bp)
    assert not bp.sums and bool(torch.isnan(part2).all())


@pytest.fixture
def wgrad_kernel_option():
    """option "wgrad_zs": 1 k_conv_wgrad_zs, 2 k_conv_wgrad_zt, 3 k_conv_wgrad_tr (transposing LDS reads)"""
    from torch_em_amd import _lib
    old = _lib.get_option("wgrad_zs")
    yield lambda v: _lib.set_option("wgrad_zs", v)
    _lib.set_option("wgrad_zs", old)


@pytest.mark.parametrize("kernel", [1, 2, 3])
@pytest.mark.parametrize("mode", [1, 2, 5, 7])
@pytest.mark.parametrize("case", [(2, 16, 16, 24, 32, 32), (1, 20, 24, 17, 32, 32), (2, 16, 16, 16, 64, 64), (2, 33, 8, 8, 32, 96),
                                  (1, 16, 132, 136, 32, 32), (3, 17, 9, 10, 64, 32),
                                  (2, 8, 8, 8, 64, 96), (1, 9, 12, 12, 64, 64), (1, 12, 24, 24, 32, 64)])   # short columns (8 <= D < 16)
def test_wgrad_z_sliding_kernels_agree_with_float64(case, mode, kernel, wgrad_kernel_option):
    """Every z-sliding weight-gradient kernel (round 2: k_conv_wgrad_zs, round 3: _zt with a staging team, round 4: _tr with
    voxel-major LDS records and ds_read_b64_tr_b16 fragment reads) in the exact-fp32 / bf16x3 / one-term fp16 / one-term bf16
    modes against the float64 weight gradient of the operands rounded as the mode rounds them (only the summation order differs)."""
    ops = _ops()
    wgrad_kernel_option(kernel)
    N, D, H, W, Cin, Cout = case
    k = (3, 3, 3)
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(N, Cin, D, H, W, generator=gen)
    gy = torch.randn(N, Cout, D, H, W, generator=gen)
    scale = torch.rand(N, Cin, generator=gen) + 0.5
    shift = torch.randn(N, Cin, generator=gen)
    xn = (x.double() * scale[:, :, None, None, None].double() + shift[:, :, None, None, None].double()).float()
    if mode == 1:   # exact fp32 (kernel 3: k_conv_wgrad_tr<4> on fp32 records; 1 / 2: the patch kernel of conv_mfma.hip)
        exp = torch.nn.grad.conv3d_weight(xn.double(), (Cout, Cin, *k), gy.double(), padding=1)
        tol = 2e-5
    elif mode == 2:
        def r2(t):
            hi = t.bfloat16().float()
            return hi.double() + (t - hi).bfloat16().double()
        xh, gh = r2(xn), r2(gy)
        exp = torch.nn.grad.conv3d_weight(xh, (Cout, Cin, *k), gh, padding=1) - \
            torch.nn.grad.conv3d_weight(xh - xn.bfloat16().double(), (Cout, Cin, *k), gh - gy.bfloat16().double(), padding=1)  # no lo*lo
        tol = 2e-5
    else:
        r16 = (lambda t: t.half().double()) if mode == 5 else (lambda t: t.bfloat16().double())
        exp = torch.nn.grad.conv3d_weight(r16(xn), (Cout, Cin, *k), r16(gy), padding=1)
        tol = 2e-5
    x5, g5 = to5(x), to5(gy)
    dw = torch.empty(exp.numel(), device=DEV)
    db = torch.empty(Cout, device=DEV)
    ops.conv_wgrad(x5, g5, k, Cin, Cout, dw, db, scale=scale.to(DEV), shift=shift.to(DEV), mfma=mode)
    assert rel_err(dw.cpu().view(exp.shape), exp) < tol
    assert rel_err(db.cpu(), gy.sum((0, 2, 3, 4))) < 5e-5
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(x5, g5, k, Cin, Cout, dw2, db, scale=scale.to(DEV), shift=shift.to(DEV), mfma=mode)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("kernel", [1, 3])
@pytest.mark.parametrize("case", [(2, 16, 16, 24, 32, 32), (1, 20, 24, 17, 32, 32), (2, 16, 16, 16, 64, 64), (2, 33, 8, 8, 32, 96),
                                  (1, 16, 132, 136, 32, 32)])
@pytest.mark.parametrize("gscale", [1.0, 3e-7, 5e4])
def test_wgrad_fp16_two_by_one_with_device_prescale(case, gscale, kernel, wgrad_kernel_option):
    """tem_conv3d_wgrad_gscaled (the default weight-gradient arithmetic of the pre-normalised 3x3x3 layers since round 4):
    x^ in two fp16 terms, g in ONE fp16 term after the power-of-two prescale from max |g| (tem_absmax), two MFMAs per
    product.  Expected values: the float64 weight gradient of exactly those rounded operands -- only the fp32 summation
    order differs (2e-5 of max |dw|); against the unrounded float64 gradient the published bound of this arithmetic is
    1e-3 of max |dw| (measured 1e-4 .. 3e-4: an 11-bit g), and it must NOT be the 1e-5 of three products (that would be a
    silent fallback to bf16x3)."""
    ops = _ops()
    wgrad_kernel_option(kernel)
    N, D, H, W, Cin, Cout = case
    k = (3, 3, 3)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, D, H, W, generator=gen)
    gy = torch.randn(N, Cout, D, H, W, generator=gen) * gscale
    scale = torch.rand(N, Cin, generator=gen) + 0.5
    shift = torch.randn(N, Cin, generator=gen)
    xn = (x.double() * scale[:, :, None, None, None].double() + shift[:, :, None, None, None].double()).float()
    hi = xn.half().float()
    lo = (xn - hi).half().float()
    e = int(np.floor(np.log2(float(gy.abs().max()))))
    sc = 2.0 ** (14 - e)
    gr = (gy * sc).half().double() / sc
    exp = torch.nn.grad.conv3d_weight(hi.double() + lo.double(), (Cout, Cin, *k), gr, padding=1)
    exact = torch.nn.grad.conv3d_weight(xn.double(), (Cout, Cin, *k), gy.double(), padding=1)
    x5, g5 = to5(x), to5(gy)
    assert ops.conv_wgrad_gscaled_ok(x5, k, Cin, Cout)
    amax = ops.absmax(g5)
    assert int(amax.item()) == int(gy.abs().max().view(torch.int32).item())   # bit pattern of max |g|, exact
    dw = torch.empty(exp.numel(), device=DEV)
    db = torch.empty(Cout, device=DEV)
    ops.conv_wgrad_gscaled(x5, g5, k, Cin, Cout, dw, db, amax, scale=scale.to(DEV), shift=shift.to(DEV))
    got = dw.cpu().view(exp.shape).double()
    assert rel_err(got, exp) < 2e-5
    assert 3e-5 < rel_err(got, exact) < 1e-3
    assert rel_err(db.cpu(), gy.sum((0, 2, 3, 4))) < 5e-5   # the bias gradient sums the unrounded fp32 values
    # repeated launches are bit-identical (no floating-point atomics)
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad_gscaled(x5, g5, k, Cin, Cout, dw2, db, amax, scale=scale.to(DEV), shift=shift.to(DEV))
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("case", [(2, 5, 9, 8, 32, 2), (1, 3, 8, 8, 64, 2), (1, 4, 9, 7, 32, 4), (1, 1, 17, 16, 32, 1)])
def test_out_conv_backward_in_one_pass(case):
    """tem_conv1x1_out_bwd: weight / bias gradient of the output projection and its ReLU-masked data gradient from ONE pass
    over the projection's input -- bit-identical to the two kernels it replaces (tem_conv3d_wgrad + tem_conv3d_fwd on the
    transposed pack with ref), and equal to torch's convolution_backward / threshold_backward in fp32."""
    ops = _ops()
    N, D, H, W, Cin, Cout = case
    gen = torch.Generator().manual_seed(31)
    x = torch.relu(torch.randn(N, Cin, D, H, W, generator=gen))
    w = torch.randn(Cout, Cin, 1, 1, 1, generator=gen) * 0.3
    gy = torch.randn(N, Cout, D, H, W, generator=gen) * 1e-4
    assert ops.conv1x1_out_bwd_ok(Cin, Cout)
    x5, g5, wd = to5(x), to5(gy), w.to(DEV)
    gx = torch.empty_like(x5)
    dw = torch.empty(w.numel(), device=DEV)
    db = torch.empty(Cout, device=DEV)
    am = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.conv1x1_out_bwd(x5, g5, wd, gx, dw, db, out_amax=am)
    assert int(am.item()) == int(gx.abs().max().view(torch.int32).item())
    gx_plain, dw_plain = torch.empty_like(x5), torch.empty_like(dw)
    ops.conv1x1_out_bwd(x5, g5, wd, gx_plain, dw_plain, torch.empty_like(db))       # without the by-product: same values
    assert torch.equal(gx_plain, gx) and torch.equal(dw_plain, dw)
    # the two separate kernels
    gx2 = torch.empty_like(x5)
    ops.conv_fwd(g5, ops.pack_weights(wd, transpose=True, mfma=0), None, gx2, (1, 1, 1), Cout, Cin, ref=x5, mfma=0)
    dw2 = torch.empty_like(dw)
    db2 = torch.empty_like(db)
    ops.conv_wgrad(x5, g5, (1, 1, 1), Cin, Cout, dw2, db2, mfma=0)
    assert torch.equal(gx, gx2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    # torch
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv3d(xr, wr).backward(gy)
    assert rel_err(from5(gx), xr.grad * (x > 0)) < 2e-5
    assert rel_err(dw.cpu().view(w.shape), wr.grad) < 5e-5 and rel_err(db.cpu(), gy.sum((0, 2, 3, 4))) < 5e-5


def test_output_amax_is_a_by_product_of_the_gradient_producers():
    """TEM_BP_OUT_AMAX (tem_conv3d_fwd_ex) / out_amax (tem_maxpool3d_bwd_st): the kernels that write the data gradients of the big levels (max-pool backward, the 1x1x1
    expanding / streaming data gradients, the z-reuse data gradient with a ReLU mask or a fused norm backward) deliver the
    exact bit pattern of max |output|; a launch that does not support it reports it as not delivered."""
    ops = _ops()
    gen = torch.Generator().manual_seed(21)

    def armed(fn):
        am = torch.zeros(1, dtype=torch.int32, device=DEV)
        bp = ops.Byproducts(out_amax=am)
        out = fn(bp)
        return out, am, not bp.amax

    def bits(t):
        return int(t.abs().max().view(torch.int32).item())

    # max-pool backward merging a skip gradient, ReLU mask
    N, D, H, W, C = 2, 8, 16, 16, 32
    x = to5(torch.randn(N, C, D, H, W, generator=gen))
    gy = to5(torch.randn(N, C, D // 2, H // 2, W // 2, generator=gen) * 3e-6)
    gs = to5(torch.randn(N, C, D, H, W, generator=gen) * 1e-6)
    gx = torch.empty_like(x)
    am = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = ops.maxpool_bwd(gy, x, gx, (2, 2, 2), gskip=gs, relu_mask=True, out_amax=am)
    assert int(am.item()) == bits(out)
    # out_conv data gradient (2 -> 32 channels, masked): expanding kernel
    ref = to5(torch.randn(N, 32, D, H, W, generator=gen))
    w = torch.randn(2, 32, 1, 1, 1, generator=gen).to(DEV)
    g2 = to5(torch.randn(N, 2, D, H, W, generator=gen) * 1e-5)
    y = torch.empty_like(ref)
    out, am, left = armed(lambda bp: ops.conv_fwd(g2, ops.pack_weights(w, transpose=True, mfma=0), None, y, (1, 1, 1), 2, 32, ref=ref, mfma=0, bp=bp))
    assert not left and int(am.item()) == bits(out)
    # sampler data gradient (1x1x1, >= 16384 voxels): streaming kernel
    N, D, H, W = 1, 16, 32, 32
    w = (torch.randn(32, 64, 1, 1, 1, generator=gen) * 0.1).to(DEV)
    g3 = to5(torch.randn(N, 32, D, H, W, generator=gen) * 1e-4)
    ref = to5(torch.randn(N, 64, D, H, W, generator=gen))
    y = torch.empty_like(ref)
    out, am, left = armed(lambda bp: ops.conv_fwd(g3, ops.pack_weights(w, transpose=True, mfma=2), None, y, (1, 1, 1), 32, 64, ref=ref, mfma=2, bp=bp))
    assert not left and int(am.item()) == bits(out)
    # 3x3x3 data gradient on the z-reuse kernel: with a ReLU mask it delivers, without one it does not
    N, D, H, W = 2, 32, 64, 64
    w = (torch.randn(32, 32, 3, 3, 3, generator=gen) * 0.1).to(DEV)
    g4 = to5(torch.randn(N, 32, D, H, W, generator=gen) * 1e-3)
    ref = to5(torch.randn(N, 32, D, H, W, generator=gen))
    y = torch.empty_like(ref)
    wp = ops.pack_weights(w, transpose=True, mfma=2)
    assert ops.conv_fwd_family(g4, (3, 3, 3), 32, 32, 2) == 3
    out, am, left = armed(lambda bp: ops.conv_fwd(g4, wp, None, y, (3, 3, 3), 32, 32, ref=ref, mfma=2, bp=bp))
    assert not left and int(am.item()) == bits(out)
    coef = torch.rand(N, 32, 4, generator=gen).to(DEV)
    out, am, left = armed(lambda bp: ops.conv_fwd_refnorm(g4, wp, y, (3, 3, 3), 32, 32, ref, coef, 2, bp=bp))
    assert not left and int(am.item()) == bits(out)
    out, am, left = armed(lambda bp: ops.conv_fwd(g4, wp, None, y, (3, 3, 3), 32, 32, mfma=2, bp=bp))
    assert left and int(am.item()) == 0


def test_absmax_strided_rows_accumulation_and_non_finite_values():
    """tem_absmax: rows with a leading dimension (a channel slice of a wider buffer), accumulation into a non-zero word
    (integer max), and non-finite values: inf wins (the largest finite-ordered pattern); a NaN is skipped by the float maximum
    -- harmless, because the consumer (the prescaled weight gradient) multiplies the NaN itself through."""
    ops = _ops()
    gen = torch.Generator().manual_seed(41)
    wide = torch.randn(2, 5, 6, 7, 48, generator=gen).to(DEV)
    sl = wide[..., 8:40]                                   # ld = 48, C = 32, 16-byte aligned start
    am = ops.absmax(sl)
    assert int(am.item()) == int(sl.abs().max().view(torch.int32).item())
    assert int(am.item()) <= int(wide.abs().max().view(torch.int32).item())
    big = torch.full((1, 1, 1, 1, 4), 3.0e4, device=DEV)
    am2 = ops.absmax(big, am.clone())
    assert int(am2.item()) == int(torch.tensor(3.0e4).view(torch.int32).item())
    am3 = ops.absmax(sl, am2)                              # smaller values do not lower the word
    assert int(am3.item()) == int(torch.tensor(3.0e4).view(torch.int32).item())
    bad = sl.clone().contiguous()
    bad[1, 2, 3, 4, 5] = float("inf")
    assert int(ops.absmax(bad).item()) == 0x7f800000
    bad2 = sl.clone().contiguous()
    bad2[0, 0, 0, 0, 0] = float("nan")
    assert int(ops.absmax(bad2).item()) == int(sl.abs().max().view(torch.int32).item())
    # ... and the NaN reaches the weight gradient it would be the prescale of
    x5 = torch.randn(1, 16, 8, 8, 32, generator=gen).to(DEV)
    g5 = torch.randn(1, 16, 8, 8, 32, generator=gen).to(DEV)
    g5[0, 3, 3, 3, 7] = float("nan")
    dw = torch.empty(32 * 32 * 27, device=DEV)
    ops.conv_wgrad_gscaled(x5, g5, (3, 3, 3), 32, 32, dw, None, ops.absmax(g5))
    assert bool(torch.isnan(dw.view(32, 32, 27)[7]).any())


def test_wgrad_fp16_two_by_one_zero_gradient_and_norm_sums():
    """all-zero g (max |g| = 0: no prescale) gives zeros; with sums_from the norm-backward sums come out as for bf16x3"""
    ops = _ops()
    N, D, H, W, Cin, Cout = 2, 16, 16, 24, 32, 32
    k = (3, 3, 3)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(N, Cin, D, H, W, generator=gen).double() * 1.5 + 0.3
    w = (torch.randn(Cout, Cin, *k, generator=gen) * 0.1).double()
    gout = torch.randn(N, Cout, D, H, W, generator=gen).double() * 1e-5
    mean = x.mean((2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var((2, 3, 4), unbiased=False) + 1e-5)
    xn = (x - mean[:, :, None, None, None]) * rstd[:, :, None, None, None]
    gz = torch.nn.grad.conv3d_input(x.shape, w, gout, padding=1)
    A_ref, B_ref = gz.sum((2, 3, 4)), (gz * xn).sum((2, 3, 4))
    scale, shift = rstd.float().to(DEV), (-mean * rstd).float().to(DEV)
    x5, g5 = to5(x.float()), to5(gout.float())
    dw = torch.empty(w.numel(), device=DEV)
    db = torch.empty(Cout, device=DEV)
    z5 = torch.zeros_like(g5)
    ops.conv_wgrad_gscaled(x5, z5, k, Cin, Cout, dw, db, ops.absmax(z5), scale=scale, shift=shift)
    assert float(dw.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
    if not ops.conv_wgrad_sums_ok(x5, k, Cin, Cout, 2):
        pytest.skip("the wgrad_sums options exclude this size")
    sums = ops.conv_wgrad_gscaled(x5, g5, k, Cin, Cout, dw, db, ops.absmax(g5), scale=scale, shift=shift,
                                  sums_from=(w.float().to(DEV), None, None))
    A, B = sums[..., 0].cpu().double(), sums[..., 1].cpu().double()
    sa = float(gz.pow(2).sum((2, 3, 4)).sqrt().max())
    sb = float((gz * xn).pow(2).sum((2, 3, 4)).sqrt().max())
    # sum gz comes from the bias gradient (fp32 values); sum gz*xn from dw: the 11-bit g shows here (bound 2e-3 of the term size)
    assert float((A - A_ref).abs().max()) < 1e-4 * sa, float((A - A_ref).abs().max()) / sa
    assert float((B - B_ref).abs().max()) < 2e-3 * sb, float((B - B_ref).abs().max()) / sb


@pytest.mark.parametrize("case", [((2, 2, 2), (2, 6, 5, 7), 32, 32, 32), ((1, 2, 2), (1, 5, 6, 4), 64, 64, 32),
                                  ((1, 2, 2), (2, 1, 9, 8), 32, 32, 1), ((2, 2, 2), (1, 3, 4, 4), 8, 24, 8)])
def test_deferred_concat_norm_backward(case):
    """The backward of the norm in front of a decoder block (input = concat(upsample(u), skip), reference
    Decoder._concat model/unet.py:363-373) without its elementwise pass: tem_norm_bwd_coef + tem_upsample_bwd_norm
    (27-point stencil on the low-resolution u) + tem_maxpool3d_bwd_norm must equal norm_bwd followed by the plain
    upsample / max-pool backward."""
    ops = _ops()
    f, (N, d, h, w), cup, cskip, groups = case
    C = cup + cskip
    D, H, W = d * f[0], h * f[1], w * f[2]
    g = torch.Generator().manual_seed(11)
    u = torch.randn(N, d, h, w, cup, generator=g).to(DEV)
    cat = torch.empty(N, D, H, W, C, device=DEV)
    ops.upsample_fwd(u, cat[..., :cup], f)
    cat[..., cup:] = torch.relu(torch.randn(N, D, H, W, cskip, generator=g)).to(DEV)
    gcat = torch.randn(N, D, H, W, C, generator=g).to(DEV)
    gpool = torch.randn(N, d, h, w, cskip, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = torch.randn(C, generator=g).to(DEV)
    mean, rstd, _, _ = ops.norm_stats(cat, groups, gamma, beta, 1e-5)
    # reference: apply pass, then the plain consumers
    gapp = gcat.clone()
    dg0, db0 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.norm_bwd(gapp, cat, groups, gamma, mean, rstd, False, gapp, dg0, db0)
    gt0 = torch.empty_like(u)
    ops.upsample_bwd(gapp[..., :cup], gt0, f)
    gs0 = torch.empty(N, D, H, W, cskip, device=DEV)
    ops.maxpool_bwd(gpool, cat[..., cup:], gs0, f, gskip=gapp[..., cup:], relu_mask=True)
    # deferred: coefficients only, the consumers apply them
    dg1, db1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    coef = ops.norm_bwd_coef(gcat, cat, groups, gamma, mean, rstd, dg1, db1)
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    gt1 = torch.empty_like(u)
    ops.upsample_bwd(gcat[..., :cup], gt1, f, norm=(u, coef[:, :cup]))
    gs1 = torch.empty(N, D, H, W, cskip, device=DEV)
    ops.maxpool_bwd(gpool, cat[..., cup:], gs1, f, gskip=gcat[..., cup:], relu_mask=True, gskip_coef=coef[:, cup:])
    assert rel_err(gs1.cpu(), gs0.cpu()) < 1e-6
    assert rel_err(gt1.cpu(), gt0.cpu()) < 2e-5
    # the same for the norm whose input is the POOLED tensor (first norm of the next level's block): max-pool backward
    # applies it to the incoming gradient, using the maximum it recomputes
    skipt = cat[..., cup:]
    pooled = torch.empty(N, d, h, w, cskip, device=DEV)
    ops.maxpool_fwd(skipt, pooled, f)
    gp = min(groups, cskip)
    gam2, bet2 = gamma[:cskip].contiguous(), beta[:cskip].contiguous()
    mean2, rstd2, _, _ = ops.norm_stats(pooled, gp, gam2, bet2, 1e-5)
    gapp2 = gpool.clone()
    ops.norm_bwd(gapp2, pooled, gp, gam2, mean2, rstd2, False, gapp2, None, None)
    ref = torch.empty(N, D, H, W, cskip, device=DEV)
    ops.maxpool_bwd(gapp2, skipt, ref, f, gskip=gapp[..., cup:], relu_mask=True)
    coef2 = ops.norm_bwd_coef(gpool, pooled, gp, gam2, mean2, rstd2)
    got = torch.empty(N, D, H, W, cskip, device=DEV)
    ops.maxpool_bwd(gpool, skipt, got, f, gskip=gcat[..., cup:], relu_mask=True, gskip_coef=coef[:, cup:], gy_coef=coef2)
    assert rel_err(got.cpu(), ref.cpu()) < 1e-6
    got2 = torch.empty(N, D, H, W, cskip, device=DEV)
    ops.maxpool_bwd(gpool, skipt, got2, f, gskip=gapp[..., cup:], relu_mask=True, gy_coef=coef2)
    assert rel_err(got2.cpu(), ref.cpu()) < 1e-6


@pytest.mark.parametrize("case", [((2, 2, 2), (2, 6, 5, 7), 32, 32, 64), ((1, 2, 2), (1, 5, 6, 4), 64, 64, 32),
                                  ((1, 2, 2), (2, 1, 9, 8), 32, 32, 2), ((2, 2, 2), (1, 3, 4, 4), 8, 24, 8)])
def test_concat_statistics_without_reading_the_concat(case):
    """Statistics of concat(upsample(u), skip) for the first norm of a decoder block (reference Decoder._concat +
    ConvBlock, model/unet.py:363-373, 429-438): tem_upsample_stats (low-resolution stencil) + per-block partial sums of
    the skip half, merged by tem_norm_finalize_partials2, equal tem_norm_stats of the materialised concat."""
    ops = _ops()
    f, (N, d, h, w), cup, cskip, groups = case
    C = cup + cskip
    D, H, W = d * f[0], h * f[1], w * f[2]
    g = torch.Generator().manual_seed(12)
    u = (torch.randn(N, d, h, w, cup, generator=g) * 2.0 + 0.5).to(DEV)
    cat = torch.empty(N, D, H, W, C, device=DEV)
    ops.upsample_fwd(u, cat[..., :cup], f)
    skip = torch.relu(torch.randn(N, D, H, W, cskip, generator=g) + 0.3).to(DEV)
    cat[..., cup:] = skip
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    # partial sums of the skip half as a producer would emit them: here 3 voxel blocks per sample
    sk = skip.reshape(N, -1, cskip)
    cuts = [0, sk.shape[1] // 3, 2 * sk.shape[1] // 3, sk.shape[1]]
    part_b = torch.stack([torch.stack([sk[:, a:b].sum(1), (sk[:, a:b] ** 2).sum(1)], -1) for a, b in zip(cuts, cuts[1:])], 1)
    assert ops.upsample_stats_ok(u)
    # the upsampled half's partials: from the factor-2 upsampling kernel itself (tem_upsample_fwd_stats, what the engine
    # uses) and from the low-resolution stencil (tem_upsample_stats, any factor)
    cat2 = torch.empty_like(cat)
    part_f = ops.upsample_fwd(u, cat2[..., :cup], f, stats=True)
    assert part_f is not None and torch.equal(cat2[..., :cup], cat[..., :cup])
    for part_a in (part_f, ops.upsample_stats(u, f)):
        assert part_a.shape == (N, d * h, cup, 2)
        for rows in (N, 1):
            want = ops.norm_stats(cat if rows == N else cat.reshape(1, N * D, H, W, C), groups, gamma, beta, 1e-5)
            have = ops.norm_stats_from_partials2(part_a, part_b.contiguous(), rows, D * H * W, groups, gamma, beta, 1e-5)
            for a, c, name in zip(have, want, ("mean", "rstd", "scale", "shift")):
                assert a.shape == c.shape and rel_err(a.cpu(), c.cpu()) < 5e-6, (name, rows, rel_err(a.cpu(), c.cpu()))


def test_conv_relu_mask_ref_and_channel_slices():
    """ref-mask epilogue and leading-dimension (concat-buffer slice) addressing."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N, D, H, W, Cin, Cout, k = 1, 4, 8, 8, 32, 32, (3, 3, 3)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.1
    ref = torch.randn(N, Cout, D, H, W, generator=g)
    exp = F.conv3d(x, w, None, padding=1) * (ref > 0)
    big_in = torch.zeros(N, D, H, W, Cin + 32, device=DEV)
    big_in[..., 32:] = to5(x)
    big_out = torch.full((N, D, H, W, Cout + 16), 7.0, device=DEV)
    big_ref = torch.zeros(N, D, H, W, Cout + 4, device=DEV)
    big_ref[..., 4:] = to5(ref)
    for mfma in (False, True):
        wp = ops.pack_weights(w.to(DEV), False, mfma)
        ops.conv_fwd(big_in[..., 32:], wp, None, big_out[..., :Cout], k, Cin, Cout, ref=big_ref[..., 4:], mfma=mfma)
        assert rel_err(from5(big_out[..., :Cout]), exp) < 2e-5
        assert float(big_out[..., Cout:].min()) == 7.0 and float(big_out[..., Cout:].max()) == 7.0


@pytest.mark.parametrize("C,G,affine,shape", [(1, 1, False, (2, 6, 7, 9)), (4, 4, False, (1, 5, 6, 7)),
                                               (32, 32, False, (2, 8, 8, 8)), (64, 32, True, (2, 4, 8, 8)),
                                               (6, 3, True, (1, 3, 5, 7)), (512, 512, False, (1, 2, 2, 2))])
def test_norm_stats_and_backward(C, G, affine, shape):
    ops = _ops()
    N, D, H, W = shape
    g = torch.Generator().manual_seed(C + G)
    x = torch.randn(N, C, D, H, W, generator=g) * 2 + 0.7
    gamma = (1 + 0.3 * torch.randn(C, generator=g)) if affine else None
    beta = (0.2 * torch.randn(C, generator=g)) if affine else None
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True) if affine else None
    br = beta.clone().requires_grad_(True) if affine else None
    yr = F.group_norm(xr, G, gr, br, eps=1e-5)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    x5 = to5(x)
    mean, rstd, scale, shift = ops.norm_stats(x5, G, None if gamma is None else gamma.to(DEV),
                                              None if beta is None else beta.to(DEV), 1e-5)
    y5 = x5 * scale[:, None, None, None, :] + shift[:, None, None, None, :]
    assert rel_err(from5(y5), yr.detach()) < 2e-5
    for relu_mask in (False, True):
        gx5 = torch.empty_like(x5)
        dgamma = torch.empty(C, device=DEV) if affine else None
        dbeta = torch.empty(C, device=DEV) if affine else None
        ops.norm_bwd(to5(gy), x5, G, None if gamma is None else gamma.to(DEV), mean, rstd, relu_mask, gx5, dgamma,
                     dbeta)
        exp = xr.grad * (x > 0) if relu_mask else xr.grad
        assert rel_err(from5(gx5), exp) < 5e-5
        if affine:
            assert rel_err(dgamma.cpu(), gr.grad) < 5e-5 and rel_err(dbeta.cpu(), br.grad) < 5e-5


@pytest.mark.parametrize("C,f,shape", [(4, (2, 2, 2), (2, 4, 6, 8)), (32, (1, 2, 2), (1, 3, 8, 8)),
                                        (3, (2, 2, 2), (1, 2, 2, 2)), (64, (1, 2, 2), (2, 1, 16, 16))])
def test_maxpool(C, f, shape):
    ops = _ops()
    N, D, H, W = shape
    g = torch.Generator().manual_seed(C)
    x = F.relu(torch.randn(N, C, D, H, W, generator=g))  # many exact ties at 0, as after a ReLU
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, f)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    x5 = to5(x)
    y5 = ops.new_act(N, D // f[0], H // f[1], W // f[2], C, DEV)
    ops.maxpool_fwd(x5, y5, f)
    assert torch.equal(from5(y5), yr.detach())
    gskip = torch.randn(N, C, D, H, W, generator=g)
    gx5 = torch.empty_like(x5)
    ops.maxpool_bwd(to5(gy), x5, gx5, f)
    assert torch.equal(from5(gx5), xr.grad)  # first-max tie rule identical to ATen
    ops.maxpool_bwd(to5(gy), x5, gx5, f, gskip=to5(gskip), relu_mask=True)
    assert rel_err(from5(gx5), (xr.grad + gskip) * (x > 0)) < 1e-6


@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("C,f,shape", [(4, (2, 2, 2), (2, 3, 4, 5)), (32, (1, 2, 2), (1, 3, 4, 4)),
                                        (8, (2, 2, 2), (1, 1, 1, 1)), (64, (1, 2, 2), (1, 1, 8, 8)),
                                        (4, (1, 3, 3), (1, 2, 3, 4)), (32, (2, 2, 2), (2, 5, 7, 9)),
                                        (12, (2, 2, 2), (1, 2, 1, 3)), (32, (1, 2, 2), (2, 3, 9, 6))])
def test_upsample(C, f, shape, generic):
    """F.interpolate(trilinear) forward / adjoint (reference Upsampler3d, model/unet.py:455-458): the factor-2 kernels
    (2x2x2 outputs per thread, separable) and, with option upsample_generic = 1, the any-factor gather kernels."""
    ops = _ops()
    from torch_em_amd import _lib
    _lib.set_option("upsample_generic", generic)
    try:
        _upsample_case(ops, C, f, shape)
    finally:
        _lib.set_option("upsample_generic", 0)


def _upsample_case(ops, C, f, shape):
    N, D, H, W = shape
    g = torch.Generator().manual_seed(C + sum(f))
    x = torch.randn(N, C, D, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=tuple(float(v) for v in f), mode="trilinear", align_corners=False)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    y5 = ops.new_act(N, D * f[0], H * f[1], W * f[2], C, DEV)
    ops.upsample_fwd(to5(x), y5, f)
    assert rel_err(from5(y5), yr.detach()) < 1e-6
    gx5 = ops.new_act(N, D, H, W, C, DEV)
    ops.upsample_bwd(to5(gy), gx5, f)
    assert rel_err(from5(gx5), xr.grad) < 1e-5


def test_layout_roundtrip_and_standardize():
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 37, 3, 5, 71, generator=g)
    x5 = ops.nchw_to_nhwc(x.to(DEV))
    assert torch.equal(from5(x5), x)
    assert torch.equal(ops.nhwc_to_nchw(x5).cpu(), x)
    r = torch.randn(3, 1, 16, 16, 16, generator=g) * 5 + 3
    s = ops.standardize(r.to(DEV), eps=1e-7).cpu()
    exp = (r - r.mean(dim=(1, 2, 3, 4), keepdim=True)) / (r.std(dim=(1, 2, 3, 4), unbiased=False, keepdim=True) + 1e-7)
    assert rel_err(s, exp) < 1e-5


@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
def test_dice_family_against_reference_golden(layout):
    """DiceLossWithLogits, BCEDiceLoss, BCEDiceLossWithLogits (reference loss/dice.py:136-253) against values and input
    gradients the REFERENCE produced (tests/golden/gen_golden_losses.py -> g5b_dice_variants.npz), incl. saturated logits
    (+-40) and the log(0) clamp of binary_cross_entropy."""
    from torch_em_amd.loss import BCEDiceLoss, BCEDiceLossWithLogits, DiceLossWithLogits
    gd = np.load(os.path.join(GOLDEN, "g5b_dice_variants.npz"))
    target = torch.from_numpy(gd["target"]).to(DEV)
    cases = {
        "dwl_default": (DiceLossWithLogits(), "logits"),
        "dwl_mean": (DiceLossWithLogits(reduce_channel="mean"), "logits"),
        "dwl_flat": (DiceLossWithLogits(channelwise=False), "logits"),
        "bce_default": (BCEDiceLoss(), "probs"),
        "bce_weighted": (BCEDiceLoss(alpha=0.7, beta=1.3), "probs"),
        "bce_flat": (BCEDiceLoss(alpha=0.5, beta=2.0, channelwise=False), "probs"),
        "bcel_default": (BCEDiceLossWithLogits(), "logits"),
        "bcel_weighted": (BCEDiceLossWithLogits(alpha=1.5, beta=0.25, channelwise=False), "logits"),
    }
    for name, (loss, key) in cases.items():
        x = torch.from_numpy(gd[key]).to(DEV)
        if layout == "channels_last":
            x = x.contiguous(memory_format=torch.channels_last_3d)
        x.requires_grad_(True)
        val = loss(x, target)
        val.backward()
        ref_loss, ref_grad = float(gd[f"{name}.loss"]), torch.from_numpy(gd[f"{name}.grad"])
        assert abs(float(val) - ref_loss) < 3e-6 * max(1.0, abs(ref_loss)), (name, float(val), ref_loss)
        # the saturated entries of `probs` (p = 0 / 1) have a gradient of +-1e12 / count in the reference: relative check
        assert rel_err(x.grad.cpu(), ref_grad) < 5e-6, (name, rel_err(x.grad.cpu(), ref_grad))
    assert DiceLossWithLogits(reduce_channel="min").init_kwargs == {"channelwise": True, "eps": 1e-7, "reduce_channel": "min"}
    with pytest.raises(ValueError, match="Unsupported channel reduction"):
        DiceLossWithLogits(reduce_channel="median")


def test_dice_against_reference_golden():
    from torch_em_amd.loss import ApplyAndRemoveMask, DiceLoss, LossWrapper, dice_score
    gd = dict(np.load(os.path.join(GOLDEN, "g5_dice.npz")))
    p, t = torch.from_numpy(gd["p"]), torch.from_numpy(gd["t"])
    for layout in ("nchw", "channels_last"):
        for cw in (True, False):
            for red in ("sum", "mean", "max", "min"):
                pp = p.to(DEV)
                if layout == "channels_last":
                    pp = pp.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
                pp.requires_grad_(True)
                val = DiceLoss(channelwise=cw, reduce_channel=red)(pp, t.to(DEV))
                val.backward()
                assert abs(float(val) - float(gd[f"loss_{int(cw)}_{red}"])) < 2e-6, (layout, cw, red)
                assert rel_err(pp.grad.cpu(), gd[f"grad_{int(cw)}_{red}"]) < 2e-5, (layout, cw, red)
    assert rel_err(dice_score(p.to(DEV), t.to(DEV), reduce_channel=None).cpu(), gd["score_none"]) < 1e-6
    pm = torch.from_numpy(gd["pm"]).to(DEV).requires_grad_(True)
    tm = torch.from_numpy(gd["tm"]).to(DEV)
    loss = LossWrapper(DiceLoss(), transform=ApplyAndRemoveMask(masking_method="multiply"))
    val = loss(pm, tm)
    val.backward()
    assert abs(float(val) - float(gd["loss_masked"])) < 2e-6
    assert rel_err(pm.grad.cpu(), gd["grad_masked"]) < 2e-5
    # exactly zero gradient outside the mask (reference test/loss/test_loss_wrapper.py:36-62)
    mask = tm[:, 12:].bool()
    assert float(pm.grad[~mask].abs().max()) == 0.0 and float(pm.grad[mask].abs().sum()) > 0
    # reference known answers (test/loss/test_dice.py:25-38)
    ones, zeros = torch.ones(1, 1, 32, 32, device=DEV), torch.zeros(1, 1, 32, 32, device=DEV)
    assert abs(float(DiceLoss()(ones, ones))) < 1e-7 and abs(float(DiceLoss()(ones, zeros)) - 1.0) < 1e-7
    with pytest.raises(ValueError):
        DiceLoss()(ones, torch.zeros(1, 2, 32, 32, device=DEV))
    with pytest.raises(ValueError, match="_crop only supports a mask with a singleton channel axis"):
        LossWrapper(DiceLoss(), transform=ApplyAndRemoveMask())(pm, tm)


def test_adamw_and_ema_against_oracle():
    ops = _ops()
    from oracle import optim_ref
    rng = np.random.RandomState(0)
    n = 100003
    p, m, v = rng.randn(n).astype("float32"), np.zeros(n, "float32"), np.zeros(n, "float32")
    pd, md, vd = torch.from_numpy(p.copy()).to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        g = rng.randn(n).astype("float32")
        ops.adamw_step(pd, torch.from_numpy(g).to(DEV), md, vd, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
        p, m, v = optim_ref.adamw_step(p, g, m, v, step)
        assert rel_err(pd.cpu().numpy(), p) < 1e-6 and rel_err(md.cpu().numpy(), m) < 1e-6
    k, q = rng.randn(n).astype("float32"), rng.randn(n).astype("float32")
    kd = torch.from_numpy(k.copy()).to(DEV)
    ops.ema_update(kd, torch.from_numpy(q).to(DEV), 0.999)
    assert rel_err(kd.cpu().numpy(), optim_ref.ema(k, q, 0.999)) < 1e-7


def _labels(shape, seed, with_zero=True):
    rng = np.random.RandomState(seed)
    lab = rng.randint(1, 6, size=shape).astype("int64")
    if with_zero:
        lab[rng.rand(*shape) < 0.25] = 0
    return lab


def test_label_targets_bit_exact():
    ops = _ops()
    from oracle import label_ref
    offs2 = [[-1, 0], [0, -1], [-3, 0], [0, -3], [4, 5], [-3, 2]]              # reference test offsets
    offs3 = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [-3, 0, 0], [0, -9, 0],
             [0, 0, -9], [-4, 0, 0], [0, -27, 0], [0, 0, -27]]                  # reference cli.py:86-91
    for shape, offs in (((64, 64), offs2), ((8, 40, 48), offs3), ((1, 1), [[0, 1]]), ((3, 70, 5), offs3)):
        lab = _labels(shape, sum(shape))
        ld = torch.from_numpy(lab).to(DEV)
        for kw in (dict(), dict(ignore_label=0, add_mask=True),
                   dict(ignore_label=0, add_mask=True, include_ignore_transitions=True),
                   dict(ignore_label=0, add_binary_target=True, add_mask=True), dict(add_binary_target=True)):
            exp = label_ref.affinities(lab, offs, **kw)
            got = ops.affinity_target(ld, offs, **kw).cpu().numpy()
            assert got.shape == exp.shape and np.array_equal(got, exp), (shape, kw)
        for add_bin in (False, True):
            exp = label_ref.boundaries(lab, add_bin)
            got = ops.boundary_target(ld, add_bin).cpu().numpy()
            assert np.array_equal(got, exp), (shape, add_bin)
            lab2 = lab.copy()
            lab2[lab2 == lab2.max()] = -1            # an ignore label below the background
            for mode in ("thick", "inner", "outer"):     # BoundaryTransform(mode=...), reference label.py:108,123
                exp = label_ref.boundaries_mode(lab2, mode, add_bin)
                got = ops.boundary_target(torch.from_numpy(lab2).to(DEV), add_bin, mode).cpu().numpy()
                assert np.array_equal(got, exp), (shape, add_bin, mode)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_affinity_targets_match_the_reference_definitions(tag):
    """G8 (tests/golden/gen_golden_trainer.py): what the reference's own tests require AffinityTransform to return --
    outputs of its affs_brute_force / affs_brute_force_with_mask (test/transform/test_label_transforms.py:5-55) -- bit
    for bit from the HIP kernel, through the ops wrapper and through the drop-in AffinityTransform class."""
    from torch_em_amd.transform import AffinityTransform
    ops = _ops()
    g = dict(np.load(os.path.join(GOLDEN, "g8_affinities_bruteforce.npz")))
    seg, offs = g[f"{tag}_seg"].astype("int64"), [list(map(int, o)) for o in g[f"{tag}_offsets"]]
    n = len(offs)
    ld = torch.from_numpy(seg).to(DEV)
    assert np.array_equal(ops.affinity_target(ld, offs).cpu().numpy(), g[f"{tag}_affs"])
    out = ops.affinity_target(ld, offs, ignore_label=0, add_mask=True).cpu().numpy()
    assert np.array_equal(out[:n], g[f"{tag}_affs_ignore0"]) and np.array_equal(out[n:], g[f"{tag}_mask_ignore0"])
    out = AffinityTransform(offs, ignore_label=0, add_mask=True, include_ignore_transitions=True)(ld).cpu().numpy()
    assert np.array_equal(out[:n], g[f"{tag}_affs_trans"]) and np.array_equal(out[n:], g[f"{tag}_mask_trans"])


def test_masked_dice_broadcast_mask_and_ignore_label():
    """The fused DiceLoss-behind-a-multiply-mask path accepts what the reference broadcasts: a mask with a singleton channel
    axis, and MaskIgnoreLabel's boolean mask of a non-contiguous target (test/loss/test_loss_wrapper.py:36-87)."""
    from torch_em_amd.loss import ApplyMask, DiceLoss, LossWrapper, MaskIgnoreLabel
    torch.manual_seed(0)
    p = torch.rand(2, 3, 8, 12, 12, device=DEV, requires_grad=True)
    t = torch.rand(2, 3, 8, 12, 12, device=DEV)
    m1 = (torch.rand(2, 1, 8, 12, 12, device=DEV) > 0.4)
    got = LossWrapper(DiceLoss(), ApplyMask("multiply"))(p, t, mask=m1)
    got.backward()
    pr = p.detach().cpu().double().requires_grad_(True)
    mm = m1.cpu().double()
    num = ((pr * mm) * (t.cpu().double() * mm)).transpose(0, 1).flatten(1).sum(1)
    den = ((pr * mm) ** 2).transpose(0, 1).flatten(1).sum(1) + ((t.cpu().double() * mm) ** 2).transpose(0, 1).flatten(1).sum(1)
    want = (1.0 - 2 * num / den.clamp(min=1e-7)).sum()
    want.backward()
    assert abs(float(got) - float(want)) < 1e-5
    assert rel_err(p.grad.cpu(), pr.grad) < 1e-4 and float(p.grad[(~m1).expand_as(p)].abs().max()) == 0.0
    # MaskIgnoreLabel on a channel-sliced (non-contiguous) target
    big = torch.rand(2, 5, 8, 12, 12, device=DEV)
    tt = big[:, 1:4]
    tt[torch.rand_like(tt) > 0.7] = -1
    p.grad = None
    LossWrapper(DiceLoss(), MaskIgnoreLabel(-1, "multiply"))(p, tt).backward()
    assert float(p.grad[tt == -1].abs().max()) == 0.0 and float(p.grad[tt != -1].abs().max()) > 0.0


def test_batched_weight_repack_is_bit_identical():
    """The one-launch re-pack after an optimizer step (tem_conv_pack_weights_tiles: coalesced tile reads through LDS, and
    tem_conv_pack_weights_batch for the rest) writes exactly what the per-tensor tem_conv_pack_weights writes."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    shapes = [(32, 16, (3, 3, 3)), (64, 32, (1, 3, 3)), (32, 32, (1, 1, 1)), (96, 48, (3, 3, 3)), (256, 128, (3, 3, 3)),
              (128, 256, (3, 3, 3)), (48, 32, (3, 1, 3))]
    jobs, expect = [], []
    for cout, cin, k in shapes:
        w = (torch.randn(cout, cin, *k, generator=g) * 0.1).to(DEV)
        for transpose in (False, True):
            cin_e, cout_e = (cout, cin) if transpose else (cin, cout)
            if cin_e % 16 or cout_e % 32:
                continue
            for mode in (1, 2, 3, 4, 5):   # 1 (round 6): the exact-fp32 layout TEM_WL_MFMA through the tile kernel (code 4)
                if mode == 1 and k[0] * k[1] * k[2] > 27:
                    continue
                ref = ops.pack_weights(w, transpose=transpose, mfma=mode)
                dst = torch.full_like(ref, float("nan"))
                nsplit, fp16 = (3 if mode == 3 else 1 if mode == 5 else 2), {4: 2, 5: 1, 1: 4}.get(mode, 0)
                jobs.append((w, dst, cout, cin, k, int(transpose), nsplit, fp16))
                expect.append((ref, dst, (cout, cin, k, transpose, mode)))
    tab = ops.pack_table(jobs)
    assert tab["n_tiles"] > 0
    ops.pack_weights_batch(tab)
    torch.cuda.synchronize()
    for ref, dst, what in expect:
        n = (ref.view(torch.int32) != dst.view(torch.int32)).sum().item()
        # the packed buffer may be larger than the split layout fills: compare what the reference wrote
        used = ~torch.isnan(dst)
        assert n == int((~used).sum()) or torch.equal(ref.view(torch.int32)[used.view(-1)], dst.view(torch.int32)[used.view(-1)]), what
        assert torch.equal(ref.view(torch.int32)[used.view(-1)], dst.view(torch.int32)[used.view(-1)]), what
        assert int(used.sum()) > 0, what


def _fma32(a, b, c):
    """fmaf(a, b, c) on float32 arrays, bit for bit: the product of two fp32 is exact in float64; TwoSum gives the error of
    the float64 addition, and nudging an inexact sum to its neighbour with an odd last bit (round to odd, 53 >= 24 + 2 bits)
    makes the final rounding to fp32 the correctly rounded one."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c = c.astype(np.float64)
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)
    odd = (s.view(np.int64) & 1) == 1
    nudged = np.nextafter(s, np.where(e > 0, np.inf, -np.inf))
    s = np.where((e != 0) & ~odd, nudged, s)
    return s.astype(np.float32)


ZR32_CASES = [
    (2, 32, 64, 64, 32, 32),    # level-0 shape class, 512 units both ways: one per team
    (2, 36, 61, 67, 64, 32),    # ragged borders in z, y, x; four chunks (648 units; data gradient 1296)
    (2, 32, 64, 64, 64, 64),    # two column tiles both ways
    (2, 24, 70, 72, 96, 32),    # six chunks, ragged (540 units; data gradient three column tiles)
]


@pytest.fixture
def fp32_zr_option():
    from torch_em_amd import _lib
    old = _lib.get_option("fp32_zr")
    yield lambda v: _lib.set_option("fp32_zr", v)
    _lib.set_option("fp32_zr", old)


@pytest.mark.parametrize("case", ZR32_CASES)
@pytest.mark.parametrize("teams", [2, 1])
def test_conv_exact_fp32_on_the_zreuse_kernel(case, teams, fp32_zr_option):
    """(teams = 1: option fp32_zr = 2, the one-team-per-workgroup variant that stages the next chunk from inside its own tap
    loop -- measured 1-4 % slower than the two-team kernel, kept as the documented alternative; same arithmetic, same order.)
    use_mfma 1 (engine precision "fp32": the arithmetic of the reference's CPU path, nn.Conv3d in fp32,
    model/unet.py:417-438) on k_conv_zr<..., X32> (round 6): forward with the fused pre-norm, bias, ReLU and the fused
    statistics; data gradient plain, with the ReLU mask and with the norm-backward epilogue -- against F.conv3d in float64
    (an fp32 FMA chain over 27 x Cin terms: 5e-6), and bit for bit against the patch kernel's launch wherever only the
    epilogue differs.  tem_conv3d_fwd_kernel() must report the z-reuse family; option fp32_zr = 0 restores k_conv_fwd_mfma."""
    ops = _ops()
    from torch_em_amd import _lib
    lib = _lib.load()
    N, D, H, W, Cin, Cout = case
    k, pad = (3, 3, 3), (1, 1, 1)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    scale, shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g)
    xn = torch.addcmul(shift[:, :, None, None, None], x, scale[:, :, None, None, None])      # one fp32 fma per element, as the staging does
    exp = F.relu(F.conv3d(xn.double(), w.double(), b.double(), padding=pad))
    x5, wd = to5(x), w.to(DEV)
    fp32_zr_option(1 if teams == 2 else 2)
    assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, 3, 3, 3, 1) == 3
    nblk = lib.tem_conv3d_fwd_stat_blocks(N, D, H, W, Cin, Cout, 3, 3, 3, 1)
    assert nblk == ((D + 3) // 4) * ((H + 15) // 16) * ((W + 7) // 8) * 4
    wp = ops.pack_weights(wd, transpose=False, mfma=1)
    y5 = torch.full((N, D, H, W, Cout + 4), 3.0, device=DEV)
    stat, nb = ops.conv_fwd(x5, wp, b.to(DEV), y5[..., :Cout], k, Cin, Cout, scale=scale.to(DEV), shift=shift.to(DEV), act="relu",
                            mfma=1, want_stats=True)
    assert nb == nblk
    err = rel_err(from5(y5[..., :Cout]), exp)
    assert err < 5e-6, f"fwd: {err}"
    assert float(y5[..., Cout:].min()) == 3.0 and float(y5[..., Cout:].max()) == 3.0
    got = from5(y5[..., :Cout]).double()
    assert rel_err(stat[..., 0].sum(1).cpu(), got.sum((2, 3, 4))) < 1e-5
    assert rel_err(stat[..., 1].sum(1).cpu(), (got * got).sum((2, 3, 4))) < 1e-5
    y6 = ops.new_act(N, D, H, W, Cout, DEV)
    ops.conv_fwd(x5, wp, b.to(DEV), y6, k, Cin, Cout, scale=scale.to(DEV), shift=shift.to(DEV), act="relu", mfma=1)
    assert torch.equal(y6, y5[..., :Cout])                                      # the statistics epilogue stores the same values
    # the patch kernel (rounds 1-5) on the same operands: fp32 both, another summation order
    _lib.set_option("fp32_zr", 0)
    try:
        assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, 3, 3, 3, 1) == 0
        assert lib.tem_conv3d_fwd_stat_blocks(N, D, H, W, Cin, Cout, 3, 3, 3, 1) == 0
        y7 = ops.new_act(N, D, H, W, Cout, DEV)
        ops.conv_fwd(x5, wp, b.to(DEV), y7, k, Cin, Cout, scale=scale.to(DEV), shift=shift.to(DEV), act="relu", mfma=1)
    finally:
        _lib.set_option("fp32_zr", 1 if teams == 2 else 2)
    assert rel_err(y7.cpu(), y6.cpu()) < 5e-6 and rel_err(from5(y7), exp) < 5e-6
    # data gradient: transposed pack, no norm / bias / activation; masked; with the norm backward of the layer in front
    gy = torch.randn(N, Cout, D, H, W, generator=g)
    gxe = torch.nn.grad.conv3d_input(x.shape, w.double(), gy.double(), padding=pad)
    g5 = to5(gy)
    assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cout, Cin, 3, 3, 3, 1) == 3
    wpt = ops.pack_weights(wd, transpose=True, mfma=1)
    gx5 = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd(g5, wpt, None, gx5, k, Cout, Cin, mfma=1)
    err = rel_err(from5(gx5), gxe)
    assert err < 5e-6, f"dgrad: {err}"
    a1 = torch.relu(torch.randn(N, Cin, D, H, W, generator=g) + 0.2)
    a15 = to5(a1)
    gm5 = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd(g5, wpt, None, gm5, k, Cout, Cin, mfma=1, ref=a15)
    assert torch.equal(gm5, torch.where(a15 > 0, gx5, torch.zeros_like(gx5)))
    coef = torch.randn(N, Cin, 4, generator=g).to(DEV)
    kc = coef.view(N, 1, 1, 1, Cin, 4)
    want = torch.where(a15 > 0, kc[..., 0] * gx5 - kc[..., 1] - (a15 - kc[..., 3]) * kc[..., 2], torch.zeros_like(gx5))
    gn5 = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd_refnorm(g5, wpt, gn5, k, Cout, Cin, a15, coef, 1)
    assert float((gn5 - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("case", [(2, 16, 16, 16, 128, 256), (2, 15, 16, 15, 128, 256), (2, 8, 8, 8, 512, 512)])
@pytest.mark.parametrize("teams", [2, 1])
def test_conv_exact_fp32_zreuse_split_k(case, teams, fp32_zr_option):
    """The 16^3 / 8^3 levels in exact fp32: k_conv_zr<..., KSPLIT, X32> + the summing epilogue (bias, ReLU, ReLU mask, fused
    statistics), against F.conv3d in float64; reference model/unet.py:417-438."""
    ops = _ops()
    from torch_em_amd import _lib
    lib = _lib.load()
    N, D, H, W, Cin, Cout = case
    k, pad = (3, 3, 3), (1, 1, 1)
    g = torch.Generator().manual_seed(32)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    scale, shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g)
    xn = torch.addcmul(shift[:, :, None, None, None], x, scale[:, :, None, None, None])
    exp = F.relu(F.conv3d(xn.double(), w.double(), b.double(), padding=pad))
    x5, wd = to5(x), w.to(DEV)
    fp32_zr_option(1 if teams == 2 else 2)
    assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, 3, 3, 3, 1) == 4
    wp = ops.pack_weights(wd, transpose=False, mfma=1)
    y5 = ops.new_act(N, D, H, W, Cout, DEV)
    stat, nb = ops.conv_fwd(x5, wp, b.to(DEV), y5, k, Cin, Cout, scale=scale.to(DEV), shift=shift.to(DEV), act="relu", mfma=1,
                            want_stats=True)
    err = rel_err(from5(y5), exp)
    assert err < 5e-6, f"fwd: {err}"
    assert rel_err(stat[..., 0].sum(1).cpu(), from5(y5).double().sum((2, 3, 4))) < 1e-5
    gy = torch.randn(N, Cout, D, H, W, generator=g)
    refm = torch.randn(N, Cin, D, H, W, generator=g)
    gxe = torch.nn.grad.conv3d_input(x.shape, w.double(), gy.double(), padding=pad) * (refm > 0)
    assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cout, Cin, 3, 3, 3, 1) in (3, 4)
    gx5 = ops.new_act(N, D, H, W, Cin, DEV)
    ops.conv_fwd(to5(gy), ops.pack_weights(wd, transpose=True, mfma=1), None, gx5, k, Cout, Cin, mfma=1, ref=to5(refm))
    err = rel_err(from5(gx5), gxe)
    assert err < 5e-6, f"masked dgrad: {err}"


@pytest.mark.parametrize("teams", [2, 1])
def test_conv_exact_fp32_zreuse_is_an_fmaf_chain(teams, fp32_zr_option):
    """The exact mode's claim, checked bit for bit: v_mfma_f32_32x32x2_f32 adds its two products to the accumulator as two
    fused multiply-adds in k order, so one output value of k_conv_zr<..., X32> is ONE fp32 fmaf chain in the kernel's
    summation order -- 16-channel chunks in order; inside a chunk the nine (ty, tx) columns; inside a column the channel
    octets p = 0..1 and their four k-steps c = 0..3; per k-step the three z taps, each the channel pair (8 p + c, 8 p + 4 + c)
    -- then the bias, then the ReLU.  The host
    loop below (numpy, _fma32) reproduces a 2 x 8 x 32 x 16 x 32 -> 32 layer with the fused pre-norm exactly."""
    ops = _ops()
    fp32_zr_option(1 if teams == 2 else 2)
    N, D, H, W, Cin, Cout = 2, 32, 64, 64, 32, 32
    k = (3, 3, 3)
    g = torch.Generator().manual_seed(33)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    scale, shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g)
    y5 = ops.new_act(N, D, H, W, Cout, DEV)
    ops.conv_fwd(to5(x), ops.pack_weights(w.to(DEV), transpose=False, mfma=1), b.to(DEV), y5, k, Cin, Cout, scale=scale.to(DEV),
                 shift=shift.to(DEV), act="relu", mfma=1)
    # a sub-block that touches the volume's corner (zero padding) and crosses tile borders in z (4), y (16) and x (8)
    n, Z, Y, X = 1, 8, 20, 12
    xs = x[n, :, :Z + 1, :Y + 1, :X + 1].numpy()
    sc, sf = scale[n].numpy()[:, None, None, None], shift[n].numpy()[:, None, None, None]
    xh = _fma32(xs, np.broadcast_to(sc, xs.shape).copy(), np.broadcast_to(sf, xs.shape).copy())
    xp = np.zeros((Cin, Z + 2, Y + 2, X + 2), np.float32)          # zero padding comes after the norm
    xp[:, 1:, 1:, 1:] = xh
    wn = w.numpy()
    acc = np.zeros((Cout, Z, Y, X), np.float32)
    for z in range(Z):
        a = np.zeros((Cout, Y, X), np.float32)
        for ch in range(Cin // 16):
            for ty in range(3):
                for tx in range(3):
                    for p in range(2):
                        for c in range(4):
                            for tz in range(3):
                                win = xp[:, z + tz, ty:ty + Y, tx:tx + X]
                                for ci in (16 * ch + 8 * p + c, 16 * ch + 8 * p + 4 + c):
                                    a = _fma32(np.broadcast_to(wn[:, ci, tz, ty, tx][:, None, None], a.shape).copy(),
                                               np.broadcast_to(win[ci][None], a.shape).copy(), a)
        acc[:, z] = a
    want = np.maximum(acc + b.numpy()[:, None, None, None], 0.0).astype(np.float32)
    got = y5[n, :Z, :Y, :X, :].permute(3, 0, 1, 2).cpu().numpy()
    nbad = int((got.view(np.int32) != want.view(np.int32)).sum())
    assert nbad == 0, f"{nbad} of {got.size} values differ from the fmaf chain (max |d| {np.abs(got - want).max():.3e})"
