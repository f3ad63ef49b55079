"""Tiled inference for the MI355X path.

Mirrors `torch_em.util.prediction.predict_with_padding` (reference util/prediction.py:21-76) and `predict_with_halo`
(:145-330): same arguments, block order (row-major blocks of `block_shape` over the volume / roi, the last block of
an axis clipped), reflect padding of border blocks (`_load_block`, :98-142), per-block `standardize` preprocessing,
halo cropping, mask zeroing, single or channel-split outputs, `grid_shift`.

What is different: the input volume is uploaded to HBM ONCE (288 GB per GPU hold any volume torch-em is used on), a
block is gathered (with the reflection) by one HIP kernel, normalised by the standardize kernel, run through the
U-Net forward kernels, and scattered into a device output volume by one HIP kernel (csrc/predict.hip); the result
comes back over PCIe once.  The reference pads, normalises and crops every block with numpy on the host and copies it
H2D and D2H.  Callables: `preprocess` may be this package's `standardize` (runs on device), None, or a function of
a CUDA tensor; `postprocess`, `skip_block` and `prediction_function` receive CUDA tensors.  Several `gpu_ids`: one worker
thread per entry (as the reference, :188-193, 313), each with its own copy of the model and of the input volume on its
device and the blocks dealt round-robin; the per-device output volumes are merged box by box on the host.
"""
import ctypes
from typing import Any, Callable, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib, ops
from ..transform.raw import standardize


def _i3(v):
    return (ctypes.c_int * 3)(*[int(a) for a in v])


def _pad3(v, fill):
    v = [int(a) for a in v]
    return [fill] * (3 - len(v)) + v


class _Blocking:
    """Row-major grid of `block_shape` blocks over [start, stop) (bioimage_cpp.utils.Blocking as the reference uses it)."""

    def __init__(self, start, stop, block_shape):
        self.start, self.stop, self.block_shape = list(start), list(stop), list(block_shape)
        self.grid = [max((sp - st + bs - 1) // bs, 0) for st, sp, bs in zip(start, stop, block_shape)]
        self.number_of_blocks = int(np.prod(self.grid)) if self.grid else 0

    def get_block(self, block_id):
        idx = np.unravel_index(int(block_id), self.grid)
        begin = [st + i * bs for st, i, bs in zip(self.start, idx, self.block_shape)]
        end = [min(b + bs, sp) for b, bs, sp in zip(begin, self.block_shape, self.stop)]
        return begin, end


def _load_block_device(vol, offset, block_shape, halo, ndim):
    """vol: CUDA [C, D, H, W] (D == 1 for 2-D).  Returns [C, *padded block] like reference `_load_block`."""
    shape = list(vol.shape[1:])[3 - ndim:]
    starts = [off - ha for off, ha in zip(offset, halo)]
    stops = [off + bs + ha for off, bs, ha in zip(offset, block_shape, halo)]
    pad_left = [max(0, -s) for s in starts]
    seg_start = [max(0, s) for s in starts]
    seg_stop = [min(sh, s) for sh, s in zip(shape, stops)]
    seg_len = [b - a for a, b in zip(seg_start, seg_stop)]
    out_shape = [bs + 2 * ha for bs, ha in zip(block_shape, halo)]
    C = vol.shape[0]
    dst = torch.empty([C] + _pad3(out_shape, 1), dtype=torch.float32, device=vol.device)
    lib = _lib.load()
    _lib.check(lib.tem_block_load_reflect(ops._p(vol), ops._p(dst), C, vol.shape[1], vol.shape[2], vol.shape[3],
                                          _i3(_pad3(seg_start, 0)), _i3(_pad3(seg_len, 1)), _i3(_pad3(pad_left, 0)),
                                          _i3(_pad3(out_shape, 1)), ops._stream(vol)), "tem_block_load_reflect")
    return dst.reshape([C] + out_shape)


def _to_volume(arr, with_channels, ndim, device):
    t = torch.as_tensor(np.asarray(arr)).to(device=device, dtype=torch.float32)
    if not with_channels:
        t = t[None]
    while t.dim() < 4:
        t = t[:, None]
    return t.contiguous()


def predict_with_padding(model: torch.nn.Module, input_: np.ndarray, min_divisible: Tuple[int, ...],
                         device: Optional[Union[torch.device, str]] = None, with_channels: bool = False,
                         prediction_function: Callable[[Any], Any] = None) -> np.ndarray:
    """Prediction for inputs whose shape is not divisible by the model's factors: reflect-pad on the right, predict,
    crop (reference :21-76; the padding is the same gather kernel as the halo blocks)."""
    input_ = np.asarray(input_)
    if with_channels:
        assert len(min_divisible) + 1 == input_.ndim, f"{min_divisible}, {input_.ndim}"
    else:
        assert len(min_divisible) == input_.ndim
    ndim = len(min_divisible)
    if device is None:
        device = next(model.parameters()).device
    spatial = list(input_.shape[1:] if with_channels else input_.shape)
    padded = [sh if sh % md == 0 else sh + md - sh % md for sh, md in zip(spatial, min_divisible)]
    vol = _to_volume(input_, with_channels, ndim, device)
    with torch.no_grad():
        block = _load_block_device(vol, [0] * ndim, padded, [0] * ndim, ndim)
        model_input = block[None]
        output = model(model_input) if prediction_function is None else prediction_function(model, model_input)
        output = output.cpu().numpy()
    crop = (slice(None),) * (output.ndim - ndim) + tuple(slice(0, sh) for sh in spatial)
    return output[crop]


def predict_with_halo(input_, model: torch.nn.Module, gpu_ids: List[Union[str, int]], block_shape: Tuple[int, ...],
                      halo: Tuple[int, ...], output=None, preprocess: Optional[Callable] = standardize,
                      postprocess: Optional[Callable] = None, with_channels: bool = False,
                      skip_block: Optional[Callable] = None, mask=None, disable_tqdm: bool = False,
                      tqdm_desc: str = "predict with halo", prediction_function: Optional[Callable] = None,
                      roi: Optional[Tuple[slice]] = None, iter_list: Optional[List[int]] = None,
                      grid_shift: Optional[Tuple[float, ...]] = None):
    """Block-wise network prediction with a halo; see the module docstring (reference :145-330)."""
    if len(gpu_ids) < 1:
        raise ValueError("predict_with_halo: gpu_ids is empty")
    devices = [torch.device(g if not isinstance(g, int) else f"cuda:{g}") for g in gpu_ids]
    if any(d.type != "cuda" for d in devices):
        raise RuntimeError("torch_em_amd.predict_with_halo runs on MI355X only; there is no CPU fallback")
    shape_spatial0 = tuple(input_.shape[1:] if with_channels else input_.shape)
    ndim = len(shape_spatial0)
    assert len(block_shape) == len(halo) == ndim

    input_eff, mask_eff = input_, mask
    pad_left = (0,) * ndim
    if grid_shift is not None:
        assert len(grid_shift) == ndim, "grid_shift must match number of spatial dims"
        pad_left = tuple(int(np.rint(abs(gs) * bs)) for gs, bs in zip(grid_shift, block_shape))
        if not isinstance(input_eff, np.ndarray):
            raise TypeError("grid_shift padding currently requires input_ to be a numpy array")
        pw = tuple((pl, 0) for pl in pad_left)
        input_eff = np.pad(input_eff, (((0, 0),) + pw) if with_channels else pw, mode="constant", constant_values=0)
        if mask_eff is not None:
            if not isinstance(mask_eff, np.ndarray):
                raise TypeError("grid_shift padding currently requires mask to be a numpy array")
            mask_eff = np.pad(mask_eff, pw, mode="constant", constant_values=0)
    shape_spatial = tuple(input_eff.shape[1:] if with_channels else input_eff.shape)

    if roi is None:
        blocking = _Blocking([0] * ndim, list(shape_spatial), block_shape)
    else:
        assert len(roi) == ndim
        blocking = _Blocking([0 if ro.start is None else ro.start for ro in roi],
                             [sh if ro.stop is None else ro.stop for ro, sh in zip(roi, shape_spatial)], block_shape)

    user_output = output
    if output is not None and grid_shift:
        raise ValueError(
            "grid_shift is not supported together with a user-provided `output`, because "
            "grid_shift requires internal zero-padding and a final cropping step. "
            "Pass `output=None` (let this function allocate the output) or disable `grid_shift`. "
            "Or pad the input manually beforehand."
        )

    block_ids = list(range(blocking.number_of_blocks)) if iter_list is None else [int(b) for b in iter_list]
    try:
        from tqdm import tqdm
    except ImportError:  # pragma: no cover
        def tqdm(it, **kw):
            return it
    progress = tqdm(total=len(block_ids), disable=disable_tqdm, desc=tqdm_desc)
    lib = _lib.load()

    def run_on(device, my_blocks, my_model):
        """The blocks `my_blocks` on ONE device: the whole input is uploaded once, every block is a gather kernel, the
        forward pass and a masked scatter into the device's output volume -> (output volume or None, boxes written)."""
        if next(my_model.parameters()).device != device:
            from copy import deepcopy
            my_model = deepcopy(my_model).to(device)
        vol = _to_volume(input_eff, with_channels, ndim, device)
        mask_dev = None
        if mask_eff is not None:
            mask_dev = torch.as_tensor(np.asarray(mask_eff) != 0).to(device=device, dtype=torch.uint8).contiguous()
        out_dev, written = None, []
        with torch.no_grad(), torch.cuda.device(device):
            for block_id in my_blocks:
                progress.update(1)
                begin, end = blocking.get_block(block_id)
                size = [e - b for b, e in zip(begin, end)]
                if mask_dev is not None:
                    sl = tuple(slice(b, e) for b, e in zip(begin, end))
                    if not bool(mask_dev[sl].any()):
                        continue
                inp = _load_block_device(vol, begin, block_shape, halo, ndim)
                if not with_channels:
                    inp = inp[0]
                if skip_block is not None and skip_block(inp):
                    continue
                if preprocess is standardize:
                    inp = ops.standardize(inp.reshape(1, -1), 1e-7).reshape(inp.shape)  # whole-block statistics
                elif preprocess is not None:
                    inp = preprocess(inp)
                model_in = inp[None] if with_channels else inp[None, None]
                pred = my_model(model_in) if prediction_function is None else prediction_function(my_model, model_in)
                if not torch.is_tensor(pred):
                    pred = pred[0]
                pred = pred.squeeze(0)
                if postprocess is not None:
                    pred = postprocess(pred)
                if pred.dim() == ndim:
                    pred = pred[None]
                pred = pred.float().contiguous()
                n_out = pred.shape[0]
                if out_dev is None:
                    out_dev = torch.zeros([n_out] + _pad3(shape_spatial, 1), dtype=torch.float32, device=device)
                _lib.check(lib.tem_block_store_inner(
                    ops._p(pred), _i3(_pad3(pred.shape[1:], 1)), ops._p(out_dev), n_out, out_dev.shape[1], out_dev.shape[2],
                    out_dev.shape[3], ops._p(mask_dev), _i3(_pad3(halo, 0)), _i3(_pad3(begin, 0)), _i3(_pad3(size, 1)),
                    ops._stream(pred)), "tem_block_store_inner")
                written.append(tuple(slice(b, e) for b, e in zip(begin, end)))
        return out_dev, written

    if len(devices) == 1:
        parts = [run_on(devices[0], block_ids, model)]
    else:
        # one worker thread per entry of gpu_ids, block list dealt round-robin (the reference: a thread pool with one
        # worker per device pulling blocks, util/prediction.py:188-193, 313); the inner boxes of the blocks are disjoint,
        # so the per-device output volumes merge by copying each device's boxes
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(devices)) as pool:
            futs = [pool.submit(run_on, d, block_ids[i::len(devices)], model) for i, d in enumerate(devices)]
            parts = [f.result() for f in futs]
    progress.close()
    written = [bb for _, w in parts for bb in w]
    out_dev = None
    result = None
    for od, w in parts:
        if od is None:
            continue
        host = od.reshape([od.shape[0]] + list(shape_spatial)).cpu().numpy()
        if result is None:
            result, out_dev = host, od
        else:
            for bb in w:
                result[(slice(None),) + bb] = host[(slice(None),) + bb]

    if out_dev is None:  # nothing was predicted (everything masked / skipped)
        n_out = getattr(model, "out_channels", 1)
        result = np.zeros((n_out,) + shape_spatial, dtype="float32")
    if user_output is None:
        output = result
    else:  # copy only the boxes that were predicted into the caller's array(s), like the reference's in-place writes
        output = user_output
        for bb in written:
            if isinstance(output, list):
                for out, channel_slice in output:
                    this_bb = bb if out.ndim == ndim else (slice(None),) + bb
                    out[this_bb] = result[(channel_slice,) + bb]
            elif output.ndim == ndim + 1:
                output[(slice(None),) + bb] = result[(slice(None),) + bb]
            else:
                output[bb] = result[(0,) + bb]
    if grid_shift is not None:
        crop = tuple(slice(pl, pl + sh) for pl, sh in zip(pad_left, shape_spatial0))
        output = output[(slice(None),) + crop] if output.ndim == ndim + 1 else output[crop]
    return output
