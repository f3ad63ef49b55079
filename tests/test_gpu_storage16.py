"""GPU: 16-bit ACTIVATION STORAGE of the mixed-precision modes (round 5).

The reference trains under torch.autocast(float16 | bfloat16) (trainer/default_trainer.py:134-142, 781-794): every tensor
between two ops is a 16-bit tensor.  Here every kernel of the step takes the element type of its activation tensors
(TEM_ST_F16 / TEM_ST_BF16, include/tem_hip.h); arithmetic stays fp32 and a value is rounded ONCE when it is stored.  That
contract makes the parity statement exact: a kernel that reads 16-bit tensors must return, bit for bit, the ROUNDED result of
the same kernel on fp32 tensors that hold the same (16-bit representable) values -- for the MFMA kernels in the one-term
mode of the same type (the stored values are the operands), for the HBM-bound kernels always.  Statistics by-products
describe the tensor AS STORED and are checked against torch on the stored tensor."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
TYPES = [(torch.float16, 5), (torch.bfloat16, 7)]


def to5(x):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def _rt(t, dt):
    """fp32 tensor holding values representable in dt"""
    return t.to(dt).float()


def _assert_rounded_equal(y16, y32, what="", fused_norm=False):
    """fused_norm: the launch normalised fp16 tensors while staging.  The 16-bit kernels do that with v_fma_mix{lo,hi}_f16 --
    x * scale + shift computed exactly and rounded ONCE to fp16 -- where the fp32-tensor kernels round the fma to fp32 first
    and to fp16 second; the two agree except where the fp32 value sits on an fp16 rounding boundary (~2^-13 of the staged
    operands, each moving an output by 2^-11 of ONE of its ~10^3 terms).  Then: at most 0.5 % of the outputs differ, by one
    unit in the last place."""
    exp = y32.to(y16.dtype)
    same = (y16 == exp) | (y16.isnan() & exp.isnan())
    if fused_norm and y16.dtype == torch.float16 and not bool(same.all()):
        bad = ~same
        assert float(bad.float().mean()) < 5e-3, f"{what}: {int(bad.sum())} of {y16.numel()} outputs differ"
        a, e = y16[bad].float(), exp[bad].float()
        ulp = torch.exp2(torch.floor(torch.log2(torch.maximum(a.abs(), e.abs()).clamp_min(2.0 ** -14))) - 10)
        # (next to the ReLU's zero an output may come out as 0 on one side and as a tiny positive number on the other: the
        # absolute term covers a few perturbed products -- one unit in the last place of a normalised operand near 4 .. 8, 2^-8, times a weight of 0.2 .. 0.8 -- which exceeds the unit in the last place of small outputs)
        assert bool(((a - e).abs() <= ulp * 1.001 + 1e-2).all()), f"{what}: differences beyond one unit in the last place"
        return
    assert bool(same.all()), f"{what}: {int((~same).sum())} of {y16.numel()} elements differ from the rounded fp32 result " \
                             f"(max abs diff {float((y16.float() - exp.float()).abs().max()):.3e})"


FWD_CASES = [
    # N, D, H, W, Cin, Cout, k, family the 16-bit launch takes (None: whatever the dispatch picks)
    (2, 18, 61, 67, 32, 64, (3, 3, 3), 3),     # z-reuse, ragged borders, two column tiles
    (2, 32, 64, 64, 64, 32, (3, 3, 3), 3),     # z-reuse, two chunks of 32 channels
    (2, 16, 16, 16, 128, 256, (3, 3, 3), 4),   # split-K z-reuse (fp32 partial sums, 16-bit output from the epilogue)
    (2, 9, 11, 13, 32, 32, (3, 3, 3), 0),      # patch kernel
    (2, 8, 24, 24, 64, 96, (1, 3, 3), 0),      # in-plane taps: patch kernel under 16-bit storage (no ping-pong instantiation)
    (1, 1, 40, 36, 32, 64, (1, 3, 3), 0),      # 2-D data (D == 1): flat patches
    (1, 32, 32, 32, 64, 32, (1, 1, 1), 0),     # 1x1x1: streaming GEMM
    (1, 8, 8, 8, 64, 32, (1, 1, 1), 0),        # 1x1x1: patch kernel (few voxels)
]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("dt,mode", TYPES)
def test_forward_and_data_gradient_are_the_rounded_fp32_launch(case, dt, mode):
    from torch_em_amd import ops
    N, D, H, W, Cin, Cout, k, fam = case
    g = torch.Generator().manual_seed(5)
    x32 = _rt(to5(torch.randn(N, Cin, D, H, W, generator=g)), dt)
    x16 = x32.to(dt)
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    scale, shift = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV), torch.randn(N, Cin, generator=g).to(DEV)
    ref32 = _rt(to5(torch.randn(N, Cout, D, H, W, generator=g)), dt)
    ref16 = ref32.to(dt)
    wp = ops.pack_weights(w, transpose=False, mfma=mode)
    if fam is not None:
        assert ops.conv_fwd_family(x16, k, Cin, Cout, mode) == fam

    def run(x, ref, ydt, **kw):
        y = torch.full((N, D, H, W, Cout), float("nan"), device=DEV, dtype=ydt)
        out = ops.conv_fwd(x, wp, kw.pop("bias", None), y, k, Cin, Cout, mfma=mode, ref=ref, **kw)
        return y, out

    # forward with norm, bias, ReLU
    if k != (1, 1, 1):   # (the engine never puts a norm in front of a 1x1x1 conv; the streaming kernel declines one)
        y16, _ = run(x16, None, dt, bias=b, scale=scale, shift=shift, act="relu")
        y32, _ = run(x32, None, torch.float32, bias=b, scale=scale, shift=shift, act="relu")
        _assert_rounded_equal(y16, y32, "norm + bias + ReLU", fused_norm=fam in (3, 4))
        # ... with the statistics by-product: they describe the stored tensor
        y16s, got = run(x16, None, dt, bias=b, scale=scale, shift=shift, act="relu", want_stats=True)
        if got is not None:
            assert torch.equal(y16s, y16)
            part = got[0].double().sum(1)   # [N, Cout, 2]
            yf = y16s.double().reshape(N, -1, Cout)
            assert torch.allclose(part[..., 0], yf.sum(1), rtol=1e-4, atol=1e-2)
            assert torch.allclose(part[..., 1], (yf * yf).sum(1), rtol=1e-4, atol=1e-2)
    # plain (a data gradient without a mask) and masked
    y16, _ = run(x16, None, dt, bias=b)
    y32, _ = run(x32, None, torch.float32, bias=b)
    _assert_rounded_equal(y16, y32, "plain")
    y16, _ = run(x16, ref16, dt)
    y32, _ = run(x32, ref32, torch.float32)
    _assert_rounded_equal(y16, y32, "ReLU mask")
    assert bool((y16[ref16 <= 0] == 0).all())


@pytest.mark.parametrize("dt,mode", TYPES)
def test_data_gradient_with_norm_backward_epilogue(dt, mode):
    """tem_conv3d_fwd_refnorm on 16-bit tensors == the rounded fp32 launch"""
    from torch_em_amd import ops
    N, D, H, W, Cin, Cout, k = 2, 32, 64, 64, 32, 32, (3, 3, 3)
    g = torch.Generator().manual_seed(6)
    x32 = _rt(to5(torch.randn(N, Cin, D, H, W, generator=g)), dt)
    ref32 = _rt(to5(torch.randn(N, Cout, D, H, W, generator=g)), dt)
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
    coef = torch.randn(N, Cout, 4, generator=g).to(DEV)
    wp = ops.pack_weights(w, transpose=False, mfma=mode)
    assert ops.conv_fwd_family(x32.to(dt), k, Cin, Cout, mode) == 3
    y16 = torch.full((N, D, H, W, Cout), float("nan"), device=DEV, dtype=dt)
    ops.conv_fwd_refnorm(x32.to(dt), wp, y16, k, Cin, Cout, ref32.to(dt), coef, mode)
    y32 = torch.full((N, D, H, W, Cout), float("nan"), device=DEV)
    ops.conv_fwd_refnorm(x32, wp, y32, k, Cin, Cout, ref32, coef, mode)
    _assert_rounded_equal(y16, y32)


WGRAD_CASES = [
    (2, 16, 24, 40, 32, 64, (3, 3, 3)),    # z-sliding kernel with transposing LDS reads (k_conv_wgrad_tr)
    (1, 9, 19, 21, 64, 32, (3, 3, 3)),     # ... ragged
    (2, 4, 12, 12, 32, 32, (3, 3, 3)),     # patch kernel (D < 8)
    (2, 6, 20, 20, 64, 32, (1, 3, 3)),     # in-plane taps
    (1, 8, 16, 16, 64, 32, (1, 1, 1)),     # 1x1x1
]


@pytest.mark.parametrize("case", WGRAD_CASES)
@pytest.mark.parametrize("dt,mode", TYPES)
def test_weight_gradient_of_16bit_tensors(case, dt, mode):
    """same operands, same kernel structure: bit-identical to the fp32-tensor launch of the same mode"""
    from torch_em_amd import ops
    N, D, H, W, Cin, Cout, k = case
    g = torch.Generator().manual_seed(7)
    x32 = _rt(to5(torch.randn(N, Cin, D, H, W, generator=g)), dt)
    g32 = _rt(to5(torch.randn(N, Cout, D, H, W, generator=g)), dt)
    scale, shift = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV), torch.randn(N, Cin, generator=g).to(DEV)
    n = Cout * Cin * k[0] * k[1] * k[2]
    for sc, sf in ((scale, shift), (None, None)):
        dw16, db16 = torch.zeros(n, device=DEV), torch.zeros(Cout, device=DEV)
        ops.conv_wgrad(x32.to(dt), g32.to(dt), k, Cin, Cout, dw16, db16, scale=sc, shift=sf, mfma=mode)
        dw32, db32 = torch.zeros(n, device=DEV), torch.zeros(Cout, device=DEV)
        ops.conv_wgrad(x32, g32, k, Cin, Cout, dw32, db32, scale=sc, shift=sf, mfma=mode)
        assert torch.equal(dw16, dw32), float((dw16 - dw32).abs().max() / dw32.abs().max())
        assert torch.allclose(db16, db32, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt,mode", TYPES)
def test_weight_gradient_delivers_the_norm_sums_from_16bit_tensors(dt, mode):
    from torch_em_amd import ops
    N, D, H, W, Cin, Cout, k = 2, 16, 32, 32, 32, 32, (3, 3, 3)
    g = torch.Generator().manual_seed(8)
    x32 = _rt(to5(torch.randn(N, Cin, D, H, W, generator=g)), dt)
    g32 = _rt(to5(torch.randn(N, Cout, D, H, W, generator=g)), dt)
    scale, shift = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV), torch.randn(N, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
    if not ops.conv_wgrad_sums_ok(x32.to(dt), k, Cin, Cout, mode):
        pytest.skip("this shape does not deliver the sums")
    outs = []
    for xx, gg in ((x32.to(dt), g32.to(dt)), (x32, g32)):
        dw, db = torch.zeros(Cout * Cin * 27, device=DEV), torch.zeros(Cout, device=DEV)
        sums = ops.conv_wgrad(xx, gg, k, Cin, Cout, dw, db, scale=scale, shift=shift, mfma=mode, sums_from=(w, None, None))
        outs.append((dw, db, sums))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_pool_upsample_norm_kernels_on_16bit_tensors(dt):
    from torch_em_amd import ops
    g = torch.Generator().manual_seed(9)
    N, D, H, W, C = 2, 8, 16, 24, 32
    x32 = _rt(to5(torch.randn(N, C, D, H, W, generator=g)), dt)
    x16 = x32.to(dt)
    f = (2, 2, 2)
    # max-pool forward (+ statistics) and backward (skip gradient, ReLU mask, both norm-backward variants)
    y16 = torch.empty((N, D // 2, H // 2, W // 2, C), device=DEV, dtype=dt)
    got = ops.maxpool_fwd(x16, y16, f, want_stats=True)
    y32 = torch.empty((N, D // 2, H // 2, W // 2, C), device=DEV)
    ops.maxpool_fwd(x32, y32, f)
    _assert_rounded_equal(y16, y32, "maxpool")
    assert got is not None
    part = got[0].double().sum(1)
    yf = y16.double().reshape(N, -1, C)
    assert torch.allclose(part[..., 0], yf.sum(1), rtol=1e-5, atol=1e-3)
    gy32 = _rt(to5(torch.randn(N, C, D // 2, H // 2, W // 2, generator=g)), dt)
    gs32 = _rt(to5(torch.randn(N, C, D, H, W, generator=g)), dt)
    kc = torch.randn(N, C, 4, generator=g).to(DEV)
    for kw in ({}, {"gskip": True, "relu_mask": True}, {"gskip": True, "relu_mask": True, "gskip_coef": kc, "gy_coef": kc}):
        kw16 = dict(kw)
        kw32 = dict(kw)
        if kw.get("gskip"):
            kw16["gskip"], kw32["gskip"] = gs32.to(dt), gs32
        gx16 = torch.empty_like(x16)
        ops.maxpool_bwd(gy32.to(dt), x16, gx16, f, **kw16)
        gx32 = torch.empty_like(x32)
        ops.maxpool_bwd(gy32, x32, gx32, f, **kw32)
        _assert_rounded_equal(gx16, gx32, f"maxpool_bwd {sorted(kw)}")
        # anisotropic window (1, 2, 2): the other instantiation of the packed-word kernel
        gya = _rt(to5(torch.randn(N, C, D, H // 2, W // 2, generator=g)), dt)
        ga16, ga32 = torch.empty_like(x16), torch.empty_like(x32)
        ops.maxpool_bwd(gya.to(dt), x16, ga16, (1, 2, 2), **kw16)
        ops.maxpool_bwd(gya, x32, ga32, (1, 2, 2), **kw32)
        _assert_rounded_equal(ga16, ga32, f"maxpool_bwd (1, 2, 2) {sorted(kw)}")
    # upsampling: factor-2 and generic kernels, forward (+ statistics), backward (+ norm backward)
    for ff in ((2, 2, 2), (1, 2, 2), (1, 3, 3)):
        u16 = torch.empty((N, D * ff[0], H * ff[1], W * ff[2], C), device=DEV, dtype=dt)
        part = ops.upsample_fwd(x16, u16, ff, stats=True)
        u32 = torch.empty((N, D * ff[0], H * ff[1], W * ff[2], C), device=DEV)
        ops.upsample_fwd(x32, u32, ff)
        _assert_rounded_equal(u16, u32, f"upsample {ff}")
        if part is not None:   # the row sums describe the stored tensor
            uf = u16.double().reshape(N, -1, C)
            assert torch.allclose(part.double().sum(1)[..., 0], uf.sum(1), rtol=1e-4, atol=1e-2)
        gu32 = _rt(torch.randn(u32.shape, generator=g).to(DEV), dt)
        for norm in (None, (x16, x32, kc)):
            gx16 = torch.empty_like(x16)
            ops.upsample_bwd(gu32.to(dt), gx16, ff, norm=None if norm is None else (norm[0], norm[2]))
            gx32 = torch.empty_like(x32)
            ops.upsample_bwd(gu32, gx32, ff, norm=None if norm is None else (norm[1], norm[2]))
            _assert_rounded_equal(gx16, gx32, f"upsample_bwd {ff} norm={norm is not None}")
        if ops.upsample_stats_ok(x16):
            assert torch.allclose(ops.upsample_stats(x16, ff), ops.upsample_stats(x32, ff), rtol=1e-6, atol=1e-6)
    # norm statistics and backward
    for groups in (C, 4):
        m16 = ops.norm_stats(x16, groups)
        m32 = ops.norm_stats(x32, groups)
        for a, b in zip(m16, m32):
            assert torch.equal(a, b)
        gamma = torch.randn(C, generator=g).to(DEV) if groups != C else None
        go32 = _rt(to5(torch.randn(N, C, D, H, W, generator=g)), dt)
        gx16 = torch.empty_like(x16)
        ops.norm_bwd(go32.to(dt), x16, groups, gamma, m16[0], m16[1], True, gx16)
        gx32 = torch.empty_like(x32)
        ops.norm_bwd(go32, x32, groups, gamma, m32[0], m32[1], True, gx32)
        _assert_rounded_equal(gx16, gx32, f"norm_bwd groups={groups}")
        assert torch.equal(ops.norm_bwd_coef(go32.to(dt), x16, groups, gamma, m16[0], m16[1]),
                           ops.norm_bwd_coef(go32, x32, groups, gamma, m32[0], m32[1]))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_first_and_last_layer_kernels_convert_between_fp32_and_16bit(dt):
    """first conv: fp32 network input -> 16-bit activations; out_conv: 16-bit activations -> fp32 prediction, and their
    backward kernels (VALU path, use_mfma 0)"""
    from torch_em_amd import ops
    g = torch.Generator().manual_seed(10)
    N, D, H, W, C = 2, 8, 24, 40, 32
    xin = to5(torch.randn(N, 1, D, H, W, generator=g))
    w1 = (torch.randn(C, 1, 3, 3, 3, generator=g) * 0.3).to(DEV)
    b1 = torch.randn(C, generator=g).to(DEV)
    sc, sf = (torch.rand(N, 1, generator=g) + 0.5).to(DEV), torch.randn(N, 1, generator=g).to(DEV)
    wp = ops.pack_weights(w1, transpose=False, mfma=0)
    a16 = torch.empty((N, D, H, W, C), device=DEV, dtype=dt)
    got = ops.conv_fwd(xin, wp, b1, a16, (3, 3, 3), 1, C, scale=sc, shift=sf, act="relu", mfma=0, want_stats=True)
    a32 = torch.empty((N, D, H, W, C), device=DEV)
    ops.conv_fwd(xin, wp, b1, a32, (3, 3, 3), 1, C, scale=sc, shift=sf, act="relu", mfma=0)
    _assert_rounded_equal(a16, a32, "first conv")
    assert got is not None
    af = a16.double().reshape(N, -1, C)
    assert torch.allclose(got[0].double().sum(1)[..., 1], (af * af).sum(1), rtol=1e-4, atol=1e-2)
    # its weight gradient (g 16-bit), with and without the norm backward applied on load
    ga32 = _rt(to5(torch.randn(N, C, D, H, W, generator=g)), dt)
    dw16, db16 = torch.zeros(C * 27, device=DEV), torch.zeros(C, device=DEV)
    ops.conv_wgrad(xin, ga32.to(dt), (3, 3, 3), 1, C, dw16, db16, scale=sc, shift=sf, mfma=0)
    dw32, db32 = torch.zeros(C * 27, device=DEV), torch.zeros(C, device=DEV)
    ops.conv_wgrad(xin, ga32, (3, 3, 3), 1, C, dw32, db32, scale=sc, shift=sf, mfma=0)
    assert torch.equal(dw16, dw32) and torch.equal(db16, db32)
    if ops.conv_wgrad_gnorm_ok((3, 3, 3), 1, C, 0):
        coef = torch.randn(N, C, 4, generator=g).to(DEV)
        a32r = a16.float()
        ops.conv_wgrad_gnorm(xin, ga32.to(dt), a16, coef, (3, 3, 3), 1, C, dw16, db16, scale=sc, shift=sf)
        ops.conv_wgrad_gnorm(xin, ga32, a32r, coef, (3, 3, 3), 1, C, dw32, db32, scale=sc, shift=sf)
        assert torch.equal(dw16, dw32) and torch.equal(db16, db32)
    # out_conv forward, its one-pass backward and the two-kernel fallback
    for cout in (2, 12):
        wo = (torch.randn(cout, C, 1, 1, 1, generator=g) * 0.3).to(DEV)
        bo = torch.randn(cout, generator=g).to(DEV)
        wpo = ops.pack_weights(wo, transpose=False, mfma=0)
        last32 = _rt(torch.relu(to5(torch.randn(N, C, D, H, W, generator=g))), dt)
        p16 = torch.empty((N, D, H, W, cout), device=DEV)
        ops.conv_fwd(last32.to(dt), wpo, bo, p16, (1, 1, 1), C, cout, act="sigmoid", mfma=0)
        p32 = torch.empty((N, D, H, W, cout), device=DEV)
        ops.conv_fwd(last32, wpo, bo, p32, (1, 1, 1), C, cout, act="sigmoid", mfma=0)
        assert torch.equal(p16, p32)
        gp = to5(torch.randn(N, cout, D, H, W, generator=g))
        wpt = ops.pack_weights(wo, transpose=True, mfma=0)
        gx16 = torch.empty((N, D, H, W, C), device=DEV, dtype=dt)
        ops.conv_fwd(gp, wpt, None, gx16, (1, 1, 1), cout, C, mfma=0, ref=last32.to(dt))
        gx32 = torch.empty((N, D, H, W, C), device=DEV)
        ops.conv_fwd(gp, wpt, None, gx32, (1, 1, 1), cout, C, mfma=0, ref=last32)
        _assert_rounded_equal(gx16, gx32, "out_conv data gradient")
        dwo16, dbo16 = torch.zeros(cout * C, device=DEV), torch.zeros(cout, device=DEV)
        ops.conv_wgrad(last32.to(dt), gp, (1, 1, 1), C, cout, dwo16, dbo16, mfma=0)
        dwo32, dbo32 = torch.zeros(cout * C, device=DEV), torch.zeros(cout, device=DEV)
        ops.conv_wgrad(last32, gp, (1, 1, 1), C, cout, dwo32, dbo32, mfma=0)
        assert torch.equal(dwo16, dwo32) and torch.equal(dbo16, dbo32)
        if ops.conv1x1_out_bwd_ok(C, cout):
            gxo = torch.empty((N, D, H, W, C), device=DEV, dtype=dt)
            dwf, dbf = torch.zeros(cout * C, device=DEV), torch.zeros(cout, device=DEV)
            ops.conv1x1_out_bwd(last32.to(dt), gp, wo, gxo, dwf, dbf)
            assert torch.equal(gxo, gx16) and torch.equal(dwf, dwo16) and torch.equal(dbf, dbo16)


def _step(model, x, y, mode):
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import engine
    model.zero_grad(set_to_none=True)
    with engine.precision_scope(mode):
        pred = model(x)
        loss = DiceLoss()(pred, y) * 16384.0
        loss.backward()
    return pred.detach().clone(), float(loss.detach()) / 16384.0, torch.cat([p.grad.flatten() / 16384.0 for p in model.parameters()])


@pytest.mark.parametrize("mode", ["amp", "amp_bf16"])
@pytest.mark.parametrize("norm", ["InstanceNorm", "GroupNorm"])
def test_model_step_with_16bit_storage_against_fp32_storage(mode, norm, monkeypatch):
    """the whole engine: 16-bit tensors between all kernels against the same mode with fp32 tensors (rounds 1-4)"""
    from torch_em_amd.model import UNet3d, engine
    torch.manual_seed(3)
    model = UNet3d(1, 2, depth=3, initial_features=32, norm=norm).to(DEV)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 1, 32, 64, 64, generator=g).to(DEV)
    y = (torch.rand(2, 2, 32, 64, 64, generator=g) > 0.5).float().to(DEV)
    monkeypatch.setattr(engine, "_AMP_STORAGE16", True)
    assert engine.act_dtype() == torch.float32   # the default mode stores fp32
    p16, l16, g16 = _step(model, x, y, mode)
    assert p16.dtype == torch.float32
    monkeypatch.setattr(engine, "_AMP_STORAGE16", False)
    p32, l32, g32 = _step(model, x, y, mode)
    # one more rounding per stored tensor (2^-11 / 2^-8 each).  The gradient of this network amplifies forward perturbations
    # (ReLU masks / pooling arg-maxes of near-ties flip): the mixed modes themselves sit 7e-2 (fp16) from
    # the fp32-class gradient with fp32 tensors; the storage adds the same order.
    ptol, gtol = (1e-2, 0.15) if mode == "amp" else (5e-2, 0.5)
    ep, eg = float((p16 - p32).norm() / p32.norm()), float((g16 - g32).norm() / g32.norm())
    print(f"{mode} {norm}: 16-bit vs fp32 tensors: pred {ep:.2e}, loss {abs(l16 - l32):.2e}, grads {eg:.2e}")
    assert ep < ptol and abs(l16 - l32) < ptol and eg < gtol, (ep, eg)
    assert bool(torch.isfinite(g16).all())


def _family_cases():
    from torch_em_amd.model import AnisotropicUNet, UNet2d, UNet3d
    return {
        "anisotropic_1x3x3": (lambda: AnisotropicUNet(1, 3, [[1, 2, 2], [2, 2, 2]], initial_features=32, final_activation="Sigmoid",
                                                      anisotropic_kernel=True), (2, 1, 8, 32, 32)),
        "unet2d": (lambda: UNet2d(1, 2, depth=3, initial_features=32), (4, 1, 64, 64)),
        "side_outputs": (lambda: UNet3d(1, 2, depth=2, initial_features=32, return_side_outputs=True), (1, 1, 16, 32, 32)),
        "batchnorm": (lambda: UNet3d(1, 2, depth=2, initial_features=32, norm="BatchNorm"), (2, 1, 16, 16, 32)),
        "no_norm_rgb": (lambda: UNet3d(3, 2, depth=2, initial_features=32, norm=None), (1, 3, 16, 32, 32)),
        "narrow_4_features": (lambda: UNet3d(1, 2, depth=2, initial_features=4), (2, 1, 16, 16, 16)),   # no MFMA kernel at all
        # factor 3 on 20 voxels: 6 after the pooling (floor), 18 after the upsampling -> the skip tensor is centre-cropped by 1 per side
        "odd_size_cropped_skip": (lambda: AnisotropicUNet(1, 2, [[3, 3, 3]], initial_features=32), (1, 1, 20, 20, 20)),
    }


@pytest.mark.parametrize("mode", ["amp", "amp_bf16"])
@pytest.mark.parametrize("case", ["anisotropic_1x3x3", "unet2d", "side_outputs", "batchnorm", "no_norm_rgb", "narrow_4_features",
                                  "odd_size_cropped_skip"])
def test_model_families_with_16bit_storage(case, mode, monkeypatch):
    """Every model family / option of the drop-in runs a training step with 16-bit tensors and stays close to the same mode
    with fp32 tensors: the kernels that have no 16-bit instantiation must be reached through a conversion, never crash or
    silently read 16-bit data as fp32 (a wrong element type shows up as garbage far outside these bounds)."""
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import engine
    make, shape = _family_cases()[case]
    torch.manual_seed(5)
    model = make().to(DEV)
    if case == "odd_size_cropped_skip":
        model.check_shape = False
    g = torch.Generator().manual_seed(6)
    x = torch.randn(*shape, generator=g).to(DEV)

    def step():
        model.zero_grad(set_to_none=True)
        with engine.precision_scope(mode):
            pred = model(x)
            preds = pred if isinstance(pred, (list, tuple)) else [pred]
            loss = sum(DiceLoss()(p, (torch.rand(p.shape, generator=torch.Generator().manual_seed(7)) > 0.5).float().to(DEV))
                       for p in preds) * 1024.0
            loss.backward()
        grads = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]) / 1024.0
        return preds[0].detach().clone(), grads
    monkeypatch.setattr(engine, "_AMP_STORAGE16", True)
    p16, g16 = step()
    monkeypatch.setattr(engine, "_AMP_STORAGE16", False)
    p32, g32 = step()
    ep, eg = float((p16 - p32).norm() / p32.norm()), float((g16 - g32).norm() / g32.norm())
    print(f"{case} {mode}: 16-bit vs fp32 tensors: pred {ep:.2e}, grads {eg:.2e}")
    assert bool(torch.isfinite(p16).all()) and bool(torch.isfinite(g16).all())
    ptol, gtol = (2e-2, 0.3) if mode == "amp" else (1e-1, 0.7)
    assert ep < ptol and eg < gtol, (ep, eg)


def _planar_of(t):
    """interleaved [N, D, H, W, 64] 16-bit tensor -> ops.Planar holding the same values"""
    from torch_em_amd import ops
    pl = ops.Planar.empty(*t.shape[:4], t.device, t.dtype)
    pl.halves[0].copy_(t[..., :32])
    pl.halves[1].copy_(t[..., 32:])
    return pl


@pytest.mark.parametrize("dt,mode", TYPES)
def test_planar_concat_halves_through_the_chunk_stride(dt, mode):
    """ops.Planar (x_cs / y_cs of tem_conv3d_fwd_ex / _wgrad_ex): the 2 x 32 channels of a concat buffer as two dense planes.
    Same values at other addresses: forward (+ norm + statistics), data gradient into a planar tensor, weight gradient (+ norm
    sums) must equal the launches on the interleaved tensor BIT FOR BIT; a launch that cannot honour the stride raises."""
    from torch_em_amd import ops
    g = torch.Generator().manual_seed(21)
    N, D, H, W = 2, 32, 64, 64
    k = (3, 3, 3)
    x = to5(torch.randn(N, 64, D, H, W, generator=g)).to(dt)
    xp = _planar_of(x)
    assert xp.cs == N * D * H * W * 32 and tuple(xp.shape) == (N, D, H, W, 64)
    w = (torch.randn(32, 64, *k, generator=g) * 0.05).to(DEV)
    bias = torch.randn(32, generator=g).to(DEV)
    mean, rstd, scale, shift = ops.norm_stats(x, 64)
    # forward, norm fused, statistics rows
    assert ops.conv_fwd_family(xp, k, 64, 32, mode) == 3
    wp = ops.pack_weights(w, transpose=False, mfma=mode)
    y_a, y_b = (torch.empty((N, D, H, W, 32), device=DEV, dtype=dt) for _ in range(2))
    pa = ops.conv_fwd(x, wp, bias, y_a, k, 64, 32, scale=scale, shift=shift, act="relu", mfma=mode, want_stats=True)
    pb = ops.conv_fwd(xp, wp, bias, y_b, k, 64, 32, scale=scale, shift=shift, act="relu", mfma=mode, want_stats=True)
    assert torch.equal(y_a, y_b) and pa is not None and pb is not None and torch.equal(pa[0], pb[0])
    # data gradient: 32 -> 64 channels written into a planar tensor
    gy = to5(torch.randn(N, 32, D, H, W, generator=g) * 1e-2).to(dt)
    wt = ops.pack_weights(w, transpose=True, mfma=mode)
    gx_a = torch.empty_like(x)
    gx_p = xp.empty_like()
    ops.conv_fwd(gy, wt, None, gx_a, k, 32, 64, mfma=mode)
    ops.conv_fwd(gy, wt, None, gx_p, k, 32, 64, mfma=mode)
    assert torch.equal(gx_a[..., :32], gx_p.halves[0]) and torch.equal(gx_a[..., 32:], gx_p.halves[1])
    # weight gradient with the norm sums, and the plain one
    assert ops.conv_wgrad_gscaled_ok(xp, k, 64, 32)
    for sums_from in ((w, None, None), None):
        if sums_from is not None and not ops.conv_wgrad_sums_ok(x, k, 64, 32, mode):
            continue
        dwa, dba = torch.zeros(w.numel(), device=DEV), torch.zeros(32, device=DEV)
        dwb, dbb = torch.zeros(w.numel(), device=DEV), torch.zeros(32, device=DEV)
        sa = ops.conv_wgrad(x, gy, k, 64, 32, dwa, dba, scale=scale, shift=shift, mfma=mode, sums_from=sums_from)
        sb = ops.conv_wgrad(xp, gy, k, 64, 32, dwb, dbb, scale=scale, shift=shift, mfma=mode, sums_from=sums_from)
        assert torch.equal(dwa, dwb) and torch.equal(dba, dbb)
        if sums_from is not None:
            assert torch.equal(sa, sb)
            assert torch.equal(ops.norm_bwd_coef(gx_a, x, 64, None, mean, rstd, sums=sa), ops.norm_bwd_coef(gx_p, xp, 64, None, mean, rstd, sums=sb))
    # without sums the reduction runs per half
    ca, cb = ops.norm_bwd_coef(gx_a, x, 32, None, *ops.norm_stats(x, 32)[:2]), ops.norm_bwd_coef(gx_p, xp, 32, None, *ops.norm_stats(x, 32)[:2])
    assert torch.allclose(ca, cb, rtol=1e-4, atol=1e-6)
    # a launch that cannot take the stride refuses it (too few units for the z-reuse kernel here), nothing is written
    small = ops.Planar.empty(1, 8, 8, 8, DEV, dt)
    with pytest.raises(Exception):
        ops.conv_fwd(small, wp, bias, torch.empty((1, 8, 8, 8, 32), device=DEV, dtype=dt), k, 64, 32, mfma=mode)
    with pytest.raises(Exception):   # and fp32-class modes never see one
        ops.conv_fwd(xp, ops.pack_weights(w, transpose=False, mfma=2), bias, y_b, k, 64, 32, mfma=2)


@pytest.mark.parametrize("mode", ["amp", "amp_bf16"])
@pytest.mark.parametrize("norm", ["InstanceNorm", "GroupNorm"])
def test_model_step_is_bit_identical_with_planar_concat(mode, norm, monkeypatch):
    """engine: the 2 x 32-channel concat of the top level as two dense planes (16-bit storage) changes addresses only"""
    from torch_em_amd import ops
    from torch_em_amd.model import UNet3d, engine
    torch.manual_seed(3)
    model = UNet3d(1, 2, depth=2, initial_features=32, norm=norm).to(DEV)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 1, 32, 64, 64, generator=g).to(DEV)
    y = (torch.rand(2, 2, 32, 64, 64, generator=g) > 0.5).float().to(DEV)
    monkeypatch.setattr(engine, "_PLANAR_CONCAT", True)
    with engine.precision_scope(mode):
        _, st = engine._forward_impl(model, x, keep=True)
    assert isinstance(st["levels"][0]["cat"], ops.Planar) and not isinstance(st["levels"][1]["cat"], ops.Planar)
    del st
    pa, la, ga = _step(model, x, y, mode)
    monkeypatch.setattr(engine, "_PLANAR_CONCAT", False)
    pb, lb, gb = _step(model, x, y, mode)
    assert torch.equal(pa, pb) and la == lb and torch.equal(ga, gb)


def test_graphed_step_with_16bit_storage_and_planar_concat_equals_eager():
    """the fp16-storage step (planar concat halves at the top level) replayed as ONE HIP graph: bit-identical to eager"""
    from torch_em_amd import ops
    from torch_em_amd.graph import GraphedTrainStep
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d, engine
    from torch_em_amd.optim import FusedAdamW, GradScaler
    torch.manual_seed(2)
    sd0 = {k: v.detach().clone() for k, v in UNet3d(1, 2, depth=2, initial_features=32).to(DEV).state_dict().items()}
    g = torch.Generator().manual_seed(8)
    xs = [torch.randn(2, 1, 32, 64, 64, generator=g).to(DEV) for _ in range(4)]
    ys = [(torch.rand(2, 2, 32, 64, 64, generator=g) > 0.5).float().to(DEV) for _ in range(4)]
    loss_fn = DiceLoss()
    m0 = UNet3d(1, 2, depth=2, initial_features=32).to(DEV)
    m0.load_state_dict(sd0)
    with engine.precision_scope("amp"):
        _, st = engine._forward_impl(m0, xs[0], keep=True)
    assert isinstance(st["levels"][0]["cat"], ops.Planar)
    del st
    opt0, sc0, l0 = FusedAdamW(m0.parameters(), lr=1e-3), GradScaler(init_scale=2.0 ** 10), []
    for x, y in zip(xs, ys):
        opt0.zero_grad()
        with engine.precision_scope("amp"):
            loss = loss_fn(m0(x), y)
            sc0.scale(loss).backward()
            sc0.step(opt0)
            sc0.update()
        l0.append(loss.detach().clone())
    m1 = UNet3d(1, 2, depth=2, initial_features=32).to(DEV)
    m1.load_state_dict(sd0)
    opt1, sc1 = FusedAdamW(m1.parameters(), lr=1e-3), GradScaler(init_scale=2.0 ** 10)
    step = GraphedTrainStep(m1, loss_fn, opt1, xs[0], ys[0], scaler=sc1, precision="amp")
    l1 = [step(x, y)[1].detach().clone() for x, y in zip(xs, ys)]
    for a, b in zip(l0, l1):
        assert torch.equal(a, b), (float(a), float(b))
    for (k, a), b in zip(m0.state_dict().items(), m1.state_dict().values()):
        assert torch.equal(a, b), k


def test_default_mode_never_allocates_16bit_tensors():
    from torch_em_amd.model import engine
    for mode in ("fp32", "mixed", "split", "split16", "bf16x3"):
        with engine.precision_scope(mode):
            assert engine.act_dtype() == torch.float32
    with engine.precision_scope("amp"):
        assert engine.act_dtype() == (torch.float16 if engine._AMP_STORAGE16 else torch.float32)
