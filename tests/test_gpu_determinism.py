"""GPU: repeated launches of the convolution kernels must agree bit for bit.

Every kernel of the library is deterministic by construction (no floating-point atomics, fixed reduction orders), so a
launch that differs from the first one is a race or an instruction hazard, not noise.  One such hazard lived in the
epilogue stores of the team kernels for two rounds and only showed in a fraction of the
launches -- single-launch parity tests pass over it.  The shapes pick every forward / data-gradient family
(tem_conv3d_fwd_kernel: patch, ping-pong, z-reuse, split-K z-reuse), all epilogue modes (plain, fused statistics, ReLU mask,
norm backward) and the weight-gradient kernels of the reference's ConvBlock (model/unet.py:417-438)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
REPEATS = 40


def to5(x):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def _same_every_time(launch, repeats=REPEATS):
    junk = torch.empty(16 << 20, device=DEV)
    first = None
    for i in range(repeats):
        out = launch()
        out = [t.clone() for t in (out if isinstance(out, (list, tuple)) else [out])]
        if i % 3 == 0:
            junk.normal_()   # other traffic between the launches
        if first is None:
            first = out
            continue
        for a, b in zip(out, first):
            assert torch.equal(a, b), f"launch {i} differs from launch 0 in {int((a != b).sum())} of {a.numel()} elements"


FWD_CASES = [
    # N, D, H, W, Cin, Cout, k, expected family (None: whatever the dispatch picks)
    (2, 18, 61, 67, 32, 64, (3, 3, 3), 3),     # z-reuse, ragged, two column tiles
    (2, 32, 64, 64, 64, 32, (3, 3, 3), 3),     # z-reuse, exactly one unit per team
    (2, 16, 16, 16, 128, 256, (3, 3, 3), 4),   # split-K z-reuse
    (2, 8, 48, 48, 64, 96, (1, 3, 3), None),   # 2-D taps: ping-pong kernel, three column tiles
    (1, 24, 64, 64, 32, 64, (3, 3, 3), None),
    (2, 9, 11, 13, 16, 32, (3, 3, 3), None),   # patch kernel
]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("mode", [2, 4, 5, 7])
def test_forward_and_data_gradient_launches_repeat_bit_for_bit(case, mode):
    from torch_em_amd import _lib, ops
    lib = _lib.load()
    N, D, H, W, Cin, Cout, k, fam = case
    if fam is not None:
        assert lib.tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, k[0], k[1], k[2], mode) == fam
    g = torch.Generator().manual_seed(31)
    x5 = to5(torch.randn(N, Cin, D, H, W, generator=g))
    w = (torch.randn(Cout, Cin, *k, generator=g) * 0.2).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    scale, shift = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV), torch.randn(N, Cin, generator=g).to(DEV)
    ref = to5(torch.randn(N, Cout, D, H, W, generator=g))
    wp = ops.pack_weights(w, transpose=False, mfma=mode)

    def plain():
        y = torch.full((N, D, H, W, Cout), float("nan"), device=DEV)
        ops.conv_fwd(x5, wp, b, y, k, Cin, Cout, scale=scale, shift=shift, act="relu", mfma=mode)
        return y

    def with_stats():
        y = torch.full((N, D, H, W, Cout), float("nan"), device=DEV)
        got = ops.conv_fwd(x5, wp, b, y, k, Cin, Cout, scale=scale, shift=shift, act="relu", mfma=mode, want_stats=True)
        return [y] if got is None else [y, got[0]]

    def masked():
        y = torch.full((N, D, H, W, Cout), float("nan"), device=DEV)
        ops.conv_fwd(x5, wp, None, y, k, Cin, Cout, mfma=mode, ref=ref)
        return y

    for launch in (plain, with_stats, masked):
        _same_every_time(launch)


@pytest.mark.parametrize("case", [(2, 32, 64, 64, 32, 32, (3, 3, 3)), (2, 16, 32, 32, 64, 128, (3, 3, 3)),
                                  (2, 17, 30, 34, 64, 32, (3, 3, 3)), (2, 1, 40, 24, 32, 64, (1, 3, 3)),
                                  (2, 9, 17, 10, 1, 32, (3, 3, 3))])
@pytest.mark.parametrize("mode", [2, 5, 7])
def test_weight_gradient_launches_repeat_bit_for_bit(case, mode):
    from torch_em_amd import ops
    N, D, H, W, Cin, Cout, k = case
    if Cin == 1 and mode != 2:
        pytest.skip("the first-layer weight gradient has one (fp32) arithmetic")
    g = torch.Generator().manual_seed(32)
    x5 = to5(torch.randn(N, Cin, D, H, W, generator=g))
    g5 = to5(torch.randn(N, Cout, D, H, W, generator=g))
    scale, shift = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV), torch.randn(N, Cin, generator=g).to(DEV)

    def launch():
        dw = torch.full((Cout, Cin, *k), float("nan"), device=DEV)
        db = torch.full((Cout,), float("nan"), device=DEV)
        ops.conv_wgrad(x5, g5, k, Cin, Cout, dw, db, scale=scale, shift=shift, mfma=(mode if Cin > 1 else 0))
        return [dw, db]

    _same_every_time(launch)
