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
thread per DEVICE (as the reference, :188-193, 313; repeated entries of one device share its worker), each with its own copy
of the model -- made on the calling thread before the workers start -- and of the input volume, its own HIP stream (the
library's workspaces are per stream) and the blocks dealt round-robin; the per-device output volumes are merged box by box
on the host.

`predict_with_halo_pipelined` (reference :487-759: producer threads -> GPU consumers -> writer threads over queues) is the
same three stages as HIP streams of one worker per device: the gather + preprocessing of the NEXT batch of blocks runs on a
prefetch stream, the forward pass on the worker's compute stream, the masked scatter of the PREVIOUS batch on a write-back
stream, joined by events -- the role of the reference's queues, without the PCIe copies between the stages; `batch_size`
blocks are stacked into one forward pass.
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


class _Setup:
    """argument handling shared by predict_with_halo and predict_with_halo_pipelined (reference :145-260 / :487-560)"""

    def __init__(self, input_, model, gpu_ids, block_shape, halo, output, with_channels, mask, roi, iter_list, grid_shift,
                 disable_tqdm, tqdm_desc):
        if len(gpu_ids) < 1:
            raise ValueError("predict_with_halo: gpu_ids is empty")
        devices = [torch.device(g if not isinstance(g, int) else f"cuda:{g}") for g in gpu_ids]
        if any(d.type != "cuda" for d in devices):
            raise RuntimeError("torch_em_amd.predict_with_halo runs on MI355X only; there is no CPU fallback")
        # one worker per DEVICE: two workers on one device would share the model object (whose packed weights are built on
        # first use) and gain nothing -- a device runs one block at a time
        self.devices = []
        for d in devices:
            d = torch.device("cuda", torch.cuda.current_device()) if d.index is None else d
            if d not in self.devices:
                self.devices.append(d)
        self.shape_spatial0 = tuple(input_.shape[1:] if with_channels else input_.shape)
        self.ndim = ndim = len(self.shape_spatial0)
        assert len(block_shape) == len(halo) == ndim
        self.input_eff, self.mask_eff, self.pad_left = input_, mask, (0,) * ndim
        if grid_shift is not None:
            assert len(grid_shift) == ndim, "grid_shift must match number of spatial dims"
            self.pad_left = tuple(int(np.rint(abs(gs) * bs)) for gs, bs in zip(grid_shift, block_shape))
            if not isinstance(input_, np.ndarray):
                raise TypeError("grid_shift padding currently requires input_ to be a numpy array")
            pw = tuple((pl, 0) for pl in self.pad_left)
            self.input_eff = np.pad(input_, (((0, 0),) + pw) if with_channels else pw, mode="constant", constant_values=0)
            if mask is not None:
                if not isinstance(mask, np.ndarray):
                    raise TypeError("grid_shift padding currently requires mask to be a numpy array")
                self.mask_eff = np.pad(mask, pw, mode="constant", constant_values=0)
        self.shape_spatial = tuple(self.input_eff.shape[1:] if with_channels else self.input_eff.shape)
        if roi is None:
            self.blocking = _Blocking([0] * ndim, list(self.shape_spatial), block_shape)
        else:
            assert len(roi) == ndim
            self.blocking = _Blocking([0 if ro.start is None else ro.start for ro in roi],
                                      [sh if ro.stop is None else ro.stop for ro, sh in zip(roi, self.shape_spatial)],
                                      block_shape)
        if output is not None and grid_shift:
            raise ValueError(
                "grid_shift is not supported together with a user-provided `output`, because "
                "grid_shift requires internal zero-padding and a final cropping step. "
                "Pass `output=None` (let this function allocate the output) or disable `grid_shift`. "
                "Or pad the input manually beforehand."
            )
        self.block_ids = list(range(self.blocking.number_of_blocks)) if iter_list is None else [int(b) for b in iter_list]
        try:
            from tqdm import tqdm
        except ImportError:  # pragma: no cover
            tqdm = _NoProgress
        self.progress = tqdm(total=len(self.block_ids), disable=disable_tqdm, desc=tqdm_desc)
        # the per-device model copies are made HERE, on the calling thread, before any worker runs a forward pass (which
        # attaches packed weights to the conv modules of ITS model: a deepcopy racing with that sees half-built entries);
        # the reference builds its (model, device) pairs up front as well (:188-192)
        self.models = []
        for d in self.devices:
            if next(model.parameters()).device == d:
                self.models.append(model)
            else:
                from copy import deepcopy
                copy = deepcopy(model)
                for mod in copy.modules():   # the packed weights are device buffers of the source model: the copy builds its own
                    mod.__dict__.pop("_tem_pack", None)
                self.models.append(copy.to(d))
        # Every worker runs on a stream of its own, and a fresh torch.cuda.Stream does not wait for anything: whatever the
        # CALLER still has in flight on its current stream of a device (the peer copies just above, an optimizer step or a
        # load_state_dict before a train-then-predict call) must be ordered in front of the workers' first upload / forward.
        self.ready = {}
        for d in self.devices:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(d))
            self.ready[d] = ev
        self.model, self.with_channels, self.user_output, self.grid_shift = model, with_channels, output, grid_shift

    def worker_stream(self, device):
        """A new stream of `device` that starts behind everything the calling thread had enqueued when the setup ended."""
        s = torch.cuda.Stream(device)
        s.wait_event(self.ready[device])
        return s

    def upload(self, device):
        vol = _to_volume(self.input_eff, self.with_channels, self.ndim, device)
        mask_dev = None
        if self.mask_eff is not None:
            mask_dev = torch.as_tensor(np.asarray(self.mask_eff) != 0).to(device=device, dtype=torch.uint8).contiguous()
        return vol, mask_dev

    def run(self, run_on):
        """run_on(device, block ids, model) -> (device output volume or None, boxes written), one worker per device"""
        devs = self.devices
        if len(devs) == 1:
            parts = [run_on(devs[0], self.block_ids, self.models[0])]
        else:
            # block list dealt round-robin (the reference: one consumer per device pulling blocks, util/prediction.py:188-193,
            # 313); the inner boxes of the blocks are disjoint, so the per-device output volumes merge by copying boxes
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(len(devs)) as pool:
                futs = [pool.submit(run_on, d, self.block_ids[i::len(devs)], self.models[i]) for i, d in enumerate(devs)]
                parts = [f.result() for f in futs]
        self.progress.close()
        return self.finish(parts)

    def finish(self, parts):
        ndim, shape_spatial = self.ndim, self.shape_spatial
        written = [bb for _, w in parts for bb in w]
        out_dev, result = None, None
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
            result = np.zeros((getattr(self.model, "out_channels", 1),) + shape_spatial, dtype="float32")
        if self.user_output is None:
            output = result
        else:  # copy only the boxes that were predicted into the caller's array(s), like the reference's in-place writes
            output = self.user_output
            for bb in written:
                if isinstance(output, list):
                    for out, channel_slice in output:
                        this_bb = bb if out.ndim == ndim else (slice(None),) + bb
                        out[this_bb] = result[(channel_slice,) + bb]
                elif output.ndim == ndim + 1:
                    output[(slice(None),) + bb] = result[(slice(None),) + bb]
                else:
                    output[bb] = result[(0,) + bb]
        if self.grid_shift is not None:
            crop = tuple(slice(pl, pl + sh) for pl, sh in zip(self.pad_left, self.shape_spatial0))
            output = output[(slice(None),) + crop] if output.ndim == ndim + 1 else output[crop]
        return output


class _NoProgress:
    """Stand-in for tqdm(total=...) when tqdm is not installed: the two methods the workers call."""

    def __init__(self, *args, **kwargs):
        pass

    def update(self, n=1):
        pass

    def close(self):
        pass


def _store_inner(lib, pred, out_dev, mask_dev, halo, begin, size):
    _lib.check(lib.tem_block_store_inner(
        ops._p(pred), _i3(_pad3(pred.shape[1:], 1)), ops._p(out_dev), pred.shape[0], out_dev.shape[1], out_dev.shape[2],
        out_dev.shape[3], ops._p(mask_dev), _i3(_pad3(halo, 0)), _i3(_pad3(begin, 0)), _i3(_pad3(size, 1)),
        ops._stream(pred)), "tem_block_store_inner")


def _gather_block(su, vol, mask_dev, block_id, block_shape, halo, preprocess, skip_block):
    """One block up to the network input: None (masked out / skipped) or (input [C, *block + 2 halo] or [*...], begin, size)."""
    begin, end = su.blocking.get_block(block_id)
    size = [e - b for b, e in zip(begin, end)]
    if mask_dev is not None:
        sl = tuple(slice(b, e) for b, e in zip(begin, end))
        if not bool(mask_dev[sl].any()):
            return None
    inp = _load_block_device(vol, begin, block_shape, halo, su.ndim)
    if not su.with_channels:
        inp = inp[0]
    if skip_block is not None and skip_block(inp):
        return None
    if preprocess is standardize:
        inp = ops.standardize(inp.reshape(1, -1), 1e-7).reshape(inp.shape)  # whole-block statistics
    elif preprocess is not None:
        inp = preprocess(inp)
    return inp, begin, size


def _postprocessed(pred, postprocess, ndim):
    if postprocess is not None:
        pred = postprocess(pred)
    if pred.dim() == ndim:
        pred = pred[None]
    return pred.float().contiguous()


def predict_with_halo(input_, model: torch.nn.Module, gpu_ids: List[Union[str, int]], block_shape: Tuple[int, ...],
                      halo: Tuple[int, ...], output=None, preprocess: Optional[Callable] = standardize,
                      postprocess: Optional[Callable] = None, with_channels: bool = False,
                      skip_block: Optional[Callable] = None, mask=None, disable_tqdm: bool = False,
                      tqdm_desc: str = "predict with halo", prediction_function: Optional[Callable] = None,
                      roi: Optional[Tuple[slice]] = None, iter_list: Optional[List[int]] = None,
                      grid_shift: Optional[Tuple[float, ...]] = None):
    """Block-wise network prediction with a halo; see the module docstring (reference :145-330)."""
    su = _Setup(input_, model, gpu_ids, block_shape, halo, output, with_channels, mask, roi, iter_list, grid_shift,
                disable_tqdm, tqdm_desc)
    lib = _lib.load()
    ndim = su.ndim

    def run_on(device, my_blocks, my_model):
        """The blocks `my_blocks` on ONE device: the whole input is uploaded once, every block is a gather kernel, the
        forward pass and a masked scatter into the device's output volume -> (output volume or None, boxes written)."""
        out_dev, written = None, []
        # the worker's own stream: the library's scratch buffers are keyed by (device, stream), so two workers never share one
        with torch.no_grad(), torch.cuda.device(device), torch.cuda.stream(su.worker_stream(device)):
            vol, mask_dev = su.upload(device)
            for block_id in my_blocks:
                su.progress.update(1)
                got = _gather_block(su, vol, mask_dev, block_id, block_shape, halo, preprocess, skip_block)
                if got is None:
                    continue
                inp, begin, size = got
                model_in = inp[None] if with_channels else inp[None, None]
                pred = my_model(model_in) if prediction_function is None else prediction_function(my_model, model_in)
                if not torch.is_tensor(pred):
                    pred = pred[0]
                pred = _postprocessed(pred.squeeze(0), postprocess, ndim)
                if out_dev is None:
                    out_dev = torch.zeros([pred.shape[0]] + _pad3(su.shape_spatial, 1), dtype=torch.float32, device=device)
                _store_inner(lib, pred, out_dev, mask_dev, halo, begin, size)
                written.append(tuple(slice(b, b + s) for b, s in zip(begin, size)))
            torch.cuda.current_stream(device).synchronize()
        return out_dev, written

    return su.run(run_on)


def predict_with_halo_pipelined(input_, model: torch.nn.Module, gpu_ids: List[Union[str, int]], block_shape: Tuple[int, ...],
                                halo: Tuple[int, ...], output=None, preprocess: Optional[Callable] = standardize,
                                postprocess: Optional[Callable] = None, with_channels: bool = False,
                                skip_block: Optional[Callable] = None, mask=None, disable_tqdm: bool = False,
                                tqdm_desc: str = "predict with halo (pipelined)",
                                prediction_function: Optional[Callable] = None, roi: Optional[Tuple[slice]] = None,
                                iter_list: Optional[List[int]] = None, batch_size: int = 1, num_prefetch_workers: int = 4,
                                queue_size: Optional[int] = None, num_write_workers: int = 1,
                                write_queue_size: Optional[int] = None, grid_shift: Optional[Tuple[float, ...]] = None):
    """`predict_with_halo` as a three-stage pipeline (reference util/prediction.py:487-759), same arguments and results.

    The reference decouples loading + preprocessing (producer threads), prediction (one consumer per GPU, `batch_size`
    blocks per forward pass) and postprocessing + writing (writer threads) with queues, because each stage works on host
    arrays and crosses PCIe.  Here the volume, every block and the output live in HBM, so the stages are HIP STREAMS of the
    device's worker thread: the gather / reflect padding / `standardize` of batch i + 1 run on a prefetch stream, the forward
    pass of batch i on the compute stream, the masked scatter of batch i - 1 on a write-back stream; events order them.
    `num_prefetch_workers`, `queue_size`, `num_write_workers`, `write_queue_size` are accepted for signature compatibility:
    they size host threads and queues this design does not have (the prefetch depth is one batch).  Blocks of a batch are
    stacked along the batch axis, so `prediction_function` and `postprocess` see what the reference's see; a batch whose
    blocks cannot be stacked is never formed (every block has the shape block_shape + 2 halo)."""
    if grid_shift is not None:
        raise NotImplementedError(
            "grid_shift is not supported by predict_with_halo_pipelined. "
            "Use predict_with_halo for grid_shift, or pre-pad the input and use roi."
        )
    batch_size = max(1, int(batch_size))
    su = _Setup(input_, model, gpu_ids, block_shape, halo, output, with_channels, mask, roi, iter_list, None, disable_tqdm,
                tqdm_desc)
    lib = _lib.load()
    ndim = su.ndim

    def run_on(device, my_blocks, my_model):
        out_dev, written = None, []
        with torch.no_grad(), torch.cuda.device(device):
            s_main, s_pre, s_post = (su.worker_stream(device) for _ in range(3))
            with torch.cuda.stream(s_main):
                vol, mask_dev = su.upload(device)
            s_pre.wait_stream(s_main)
            pending = list(my_blocks)

            def prefetch():
                """the next batch, gathered and preprocessed on the prefetch stream -> (stacked input, [(begin, size)], event)"""
                jobs = []
                with torch.cuda.stream(s_pre):
                    while pending and len(jobs) < batch_size:
                        su.progress.update(1)
                        got = _gather_block(su, vol, mask_dev, pending.pop(0), block_shape, halo, preprocess, skip_block)
                        if got is not None:
                            jobs.append(got)
                    if not jobs:
                        return None
                    batch = torch.stack([j[0] for j in jobs]) if with_channels else torch.stack([j[0] for j in jobs])[:, None]
                    ev = torch.cuda.Event()
                    ev.record(s_pre)
                return batch, [(j[1], j[2]) for j in jobs], ev

            nxt = prefetch()
            while nxt is None and pending:
                nxt = prefetch()
            while nxt is not None:
                batch, boxes, ev = nxt
                with torch.cuda.stream(s_main):
                    s_main.wait_event(ev)
                    pred = my_model(batch) if prediction_function is None else prediction_function(my_model, batch)
                    if not torch.is_tensor(pred):
                        pred = pred[0]
                    done = torch.cuda.Event()
                    done.record(s_main)
                    batch.record_stream(s_main)
                nxt = None
                while nxt is None and pending:   # the gather of the next batch is enqueued while this forward pass runs
                    nxt = prefetch()
                with torch.cuda.stream(s_post):
                    s_post.wait_event(done)
                    for k, (begin, size) in enumerate(boxes):
                        pk = _postprocessed(pred[k], postprocess, ndim)
                        if out_dev is None:
                            out_dev = torch.zeros([pk.shape[0]] + _pad3(su.shape_spatial, 1), dtype=torch.float32, device=device)
                        _store_inner(lib, pk, out_dev, mask_dev, halo, begin, size)
                        written.append(tuple(slice(b, b + s) for b, s in zip(begin, size)))
                    pred.record_stream(s_post)
            for st in (s_pre, s_main, s_post):
                st.synchronize()
        return out_dev, written

    return su.run(run_on)
