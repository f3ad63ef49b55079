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
    """`gpu_ids` with more than one entry: one worker thread per DEVICE (each with its own model copy, made before the
    workers start, and its own HIP stream), blocks dealt round-robin, per-device output volumes merged box by box (reference
    util/prediction.py:188-193, 313).  Repeated entries of one device share its worker (round 5, advisor: two workers on
    one device shared the model and the library's per-stream scratch buffers), so on a one-GPU box this checks the argument
    handling and that the result is bit-identical to the single-entry run; the threads and the merge need two devices."""
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


def test_block_kernels_against_the_reference_helpers():
    """G9 (tests/golden/gen_golden_predict.py): what the reference's own `_load_block` and `_write_prediction` return
    (util/prediction.py:98-142, 420-447) for halos crossing no / one / both borders, 2-D, channel axes, mask blocks -- the
    gather kernel (tem_block_load_reflect) and the masked inner-box scatter (tem_block_store_inner) reproduce them bit for bit."""
    import os
    from conftest import GOLDEN
    from torch_em_amd import _lib
    from torch_em_amd.util.prediction import _load_block_device, _store_inner
    g = dict(np.load(os.path.join(GOLDEN, "g9_predict_helpers.npz")))
    for case in g["load_block_cases"]:
        name, key, wc = str(case).split("|")
        wc = bool(int(wc))
        off, bs, ha = (list(map(int, r)) for r in g[name + ".args"])
        nd = 2 if key == "vol2" else 3
        off, bs, ha = off[:nd], bs[:nd], ha[:nd]
        x = g[key]
        vol = torch.from_numpy(x if wc else x[None]).to(DEV)
        while vol.dim() < 4:
            vol = vol[:, None]
        got = _load_block_device(vol.contiguous(), off, bs, ha, nd).cpu().numpy()
        assert np.array_equal(got if wc else got[0], g[name + ".data"]), name
    begin, end, halo = (list(map(int, r)) for r in g["wp.args"])
    size = [e - b for b, e in zip(begin, end)]
    pred = torch.from_numpy(g["wp.pred"]).to(DEV)
    lib = _lib.load()
    out = torch.full((3, 7, 9, 11), -7.0, device=DEV)
    _store_inner(lib, pred, out, None, halo, begin, size)
    assert np.array_equal(out.cpu().numpy(), g["wp.out_channels"])
    mask = torch.zeros((7, 9, 11), dtype=torch.uint8, device=DEV)
    mask[tuple(slice(b, e) for b, e in zip(begin, end))] = torch.from_numpy(g["wp.mask_block"].astype("uint8")).to(DEV)
    out = torch.full((3, 7, 9, 11), -7.0, device=DEV)
    _store_inner(lib, pred, out, mask, halo, begin, size)
    assert np.array_equal(out.cpu().numpy(), g["wp.out_masked"])


@pytest.mark.parametrize("case", ["plain", "mask", "roi", "channels", "list_output"])
@pytest.mark.parametrize("batch_size", [1, 3])
def test_predict_with_halo_pipelined(case, batch_size):
    """reference util/prediction.py:487-759: same arguments, same result as predict_with_halo -- here bit for bit (the same
    kernels, staged over a prefetch / compute / write-back stream; `batch_size` blocks per forward pass)."""
    from torch_em_amd.util import predict_with_halo, predict_with_halo_pipelined
    cin = 2 if case == "channels" else 1
    model, _ = _model(True, cin, 3)
    rng = np.random.default_rng(7)
    shape = (20, 36, 28)
    x = rng.standard_normal(((cin,) if case == "channels" else ()) + shape).astype("float32")
    bs, halo = (8, 16, 16), (4, 8, 8)
    kw = dict(with_channels=case == "channels", disable_tqdm=True)
    if case == "mask":
        m = np.zeros(shape, dtype="uint8")
        m[2:14, 5:30, :20] = 1
        kw["mask"] = m
    if case == "roi":
        kw["roi"] = (slice(4, 20), slice(None, 30), slice(6, None))
    if case == "list_output":
        oa, ob = np.zeros(shape, "float32"), np.zeros((2,) + shape, "float32")
        pa, pb = np.zeros(shape, "float32"), np.zeros((2,) + shape, "float32")
        predict_with_halo(x, model, [DEV], bs, halo, output=[(oa, 0), (ob, slice(1, 3))], **kw)
        predict_with_halo_pipelined(x, model, [DEV], bs, halo, output=[(pa, 0), (pb, slice(1, 3))], batch_size=batch_size, **kw)
        assert np.array_equal(oa, pa) and np.array_equal(ob, pb) and float(np.abs(pb).max()) > 0
        return
    want = predict_with_halo(x, model, [DEV], bs, halo, **kw)
    got = predict_with_halo_pipelined(x, model, [DEV], bs, halo, batch_size=batch_size, num_prefetch_workers=2,
                                      num_write_workers=2, **kw)
    assert got.shape == want.shape and np.array_equal(got, want) and float(np.abs(got).max()) > 0
    with pytest.raises(NotImplementedError):
        predict_with_halo_pipelined(x, model, [DEV], bs, halo, grid_shift=(0.5, 0.5, 0.5), **kw)


@pytest.mark.parametrize("pipelined", [False, True])
def test_prediction_is_ordered_behind_the_callers_stream(pipelined):
    """The workers run on streams of their own; a weight update the caller has ENQUEUED but not finished (an optimizer step,
    `load_state_dict`) must still be seen by the first forward pass.  The caller's stream is kept busy for tens of
    milliseconds in front of the update, so an unordered worker stream would read the old weights."""
    from torch_em_amd.util import predict_with_halo, predict_with_halo_pipelined
    fn = predict_with_halo_pipelined if pipelined else predict_with_halo
    model, _ = _model(True, 1, 2)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((16, 32, 32)).astype("float32")
    bs, halo = (8, 16, 16), (4, 8, 8)
    new = {k: torch.randn_like(v) * 0.3 for k, v in model.state_dict().items()}
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        busy = torch.randn(4096, 4096, device=DEV)
        for _ in range(60):                      # ~tens of ms of queued work in front of the weight update
            busy = busy @ busy * 1e-2
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(new[k])
        got = fn(x, model, [DEV], bs, halo, disable_tqdm=True)
    torch.cuda.synchronize()
    want = fn(x, model, [DEV], bs, halo, disable_tqdm=True)
    assert np.array_equal(got, want)
    assert float(np.abs(want).max()) > 0
