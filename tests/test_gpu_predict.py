"""GPU parity of tiled inference (util/prediction.py + csrc/predict.hip) against oracle/predict_ref.py driving the CPU
oracle U-Net.  Block gather: bit-exact.  Predictions: max-norm 1e-4 (fp32 model forward, tests/test_gpu_unet.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(dim3=True, cin=1, cout=2):
    from oracle import unet_ref
    from torch_em_amd.model import UNet2d, UNet3d
    torch.manual_seed(0)
    model = (UNet3d if dim3 else UNet2d)(cin, cout, depth=2, initial_features=4).to(DEV).eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sf = [2, 2]
    return model, (lambda x: unet_ref.unet_forward(sd, x, sf))


def test_block_gather_matches_numpy_reflect():
    from oracle import predict_ref
    from torch_em_amd.util.prediction import _load_block_device
    rng = np.random.default_rng(0)
    for shape, bs, halo in [((7, 9, 11), (4, 4, 4), (2, 3, 1)), ((3, 20, 5), (2, 8, 4), (3, 4, 6)), ((12, 13), (5, 6), (4, 8))]:
        nd = len(shape)
        for with_c in (False, True):
            x = rng.standard_normal(((2,) if with_c else ()) + shape).astype("float32")
            vol = torch.from_numpy(x if with_c else x[None]).to(DEV)
            while vol.dim() < 4:
                vol = vol[:, None]
            vol = vol.contiguous()
            for _ in range(6):
                off = [int(rng.integers(0, s)) for s in shape]
                got = _load_block_device(vol, off, list(bs), list(halo), nd).cpu().numpy()
                want = predict_ref.load_block(x, off, bs, halo, with_channels=with_c)
                assert np.array_equal(got if with_c else got[0], want), (shape, off)


@pytest.mark.parametrize("case", ["plain", "mask", "roi", "channels", "list_output", "grid_shift"])
def test_predict_with_halo_3d(case):
    from oracle import predict_ref
    from torch_em_amd.util import predict_with_halo
    cin = 2 if case == "channels" else 1
    model, ref_model = _model(True, cin, 2)
    rng = np.random.default_rng(1)
    shape = (20, 36, 28)
    x = rng.standard_normal(((cin,) if cin > 1 else ()) + shape).astype("float32")
    bs, halo = (8, 16, 16), (4, 8, 8)
    kw, rkw = {}, {}
    if case == "mask":
        m = np.zeros(shape, dtype="uint8")
        m[2:14, 5:30, :20] = 1
        kw["mask"], rkw["mask"] = m, m
    if case == "roi":
        kw["roi"] = rkw["roi"] = (slice(4, 20), slice(None, 32), slice(4, None))
    if case == "channels":
        kw["with_channels"] = rkw["with_channels"] = True
    if case == "grid_shift":
        kw["grid_shift"] = (0, 0.25, 0.5)
        pad = [int(np.rint(g * b)) for g, b in zip(kw["grid_shift"], bs)]
        xp = np.pad(x, tuple((p, 0) for p in pad))
        want = predict_ref.predict_with_halo(xp, ref_model, bs, halo, 2)[(slice(None),) + tuple(slice(p, None) for p in pad)]
    else:
        want = predict_ref.predict_with_halo(x, ref_model, bs, halo, 2, **rkw)
    if case == "list_output":
        o0, o1 = np.zeros(shape, "float32"), np.zeros((1,) + shape, "float32")
        predict_with_halo(x, model, [DEV], bs, halo, output=[(o0, 0), (o1, slice(1, 2))], disable_tqdm=True)
        got = np.concatenate([o0[None], o1])
    else:
        got = predict_with_halo(x, model, [DEV], bs, halo, disable_tqdm=True, **kw)
    assert got.shape == want.shape and got.dtype == np.float32
    assert float(np.abs(got - want).max()) < 1e-4
    if case == "mask":
        assert float(np.abs(got[:, 15:]).max()) == 0.0


@pytest.mark.parametrize("case", ["plain", "mask"])
def test_predict_with_halo_over_several_gpu_ids(case):
    """`gpu_ids` with more than one entry: one worker thread per entry, blocks dealt round-robin, per-device output volumes
    merged box by box (reference util/prediction.py:188-193, 313).  On a one-GPU box the entries name the same device
    twice -- the threads, the block split and the merge are what is under test; the result must be bit-identical to the
    single-worker run (and to more workers than blocks)."""
    from torch_em_amd.util import predict_with_halo
    model, _ = _model(True, 1, 2)
    rng = np.random.default_rng(5)
    shape = (20, 36, 28)
    x = rng.standard_normal(shape).astype("float32")
    bs, halo = (8, 16, 16), (4, 8, 8)
    kw = {}
    if case == "mask":
        m = np.zeros(shape, dtype="uint8")
        m[2:14, 5:30, :20] = 1
        kw["mask"] = m
    one = predict_with_halo(x, model, [DEV], bs, halo, disable_tqdm=True, **kw)
    ids = [DEV, 0] if torch.cuda.device_count() < 2 else [0, 1]
    two = predict_with_halo(x, model, ids, bs, halo, disable_tqdm=True, **kw)
    many = predict_with_halo(x, model, [DEV] * 16, bs, halo, disable_tqdm=True, **kw)
    assert np.array_equal(one, two) and np.array_equal(one, many)
    out = np.full((2,) + shape, 7.0, dtype="float32")
    predict_with_halo(x, model, ids, bs, halo, output=out, disable_tqdm=True, iter_list=[0, 1, 2, 3], **kw)
    ref = np.full((2,) + shape, 7.0, dtype="float32")
    predict_with_halo(x, model, [DEV], bs, halo, output=ref, disable_tqdm=True, iter_list=[0, 1, 2, 3], **kw)
    assert np.array_equal(out, ref)


def test_predict_with_halo_2d_and_padding():
    from oracle import predict_ref
    from torch_em_amd.util import predict_with_halo, predict_with_padding
    model, ref_model = _model(False, 1, 3)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((70, 52)).astype("float32")
    want = predict_ref.predict_with_halo(x, ref_model, (32, 32), (8, 8), 3)
    got = predict_with_halo(x, model, [0], (32, 32), (8, 8), disable_tqdm=True)
    assert float(np.abs(got - want).max()) < 1e-4
    xs = rng.standard_normal((37, 50)).astype("float32")
    wantp = predict_ref.predict_with_padding(ref_model, xs, (4, 4))
    gotp = predict_with_padding(model, xs, (4, 4))
    assert gotp.shape == wantp.shape == (1, 3, 37, 50)
    assert float(np.abs(gotp - wantp).max()) < 1e-4
