"""CPU restatement of the reference's tiled prediction (TEST INFRASTRUCTURE, see oracle/__init__.py).

  _load_block        /root/reference/torch_em/util/prediction.py:98-142
  _prepare_block_input / _write_prediction (the producer / writer halves of predict_with_halo_pipelined) :388-447
  _pad_for_shift_left / _crop_after_shift_left                                                           :79-95
Pinned by tests/golden/g9_predict_helpers.npz = outputs of the reference's own functions (tests/test_oracle_golden.py);
  predict_with_halo  /root/reference/torch_em/util/prediction.py:145-330  (single worker, numpy in / numpy out)
  predict_with_padding                                                :21-76
bioimage_cpp's Blocking (not installed here) is restated as a row-major grid whose last block per axis is clipped;
`model` is any callable on a float32 CPU tensor [1, C, *S] (the tests pass the oracle U-Net).
"""
import numpy as np
import torch


def load_block(input_, offset, block_shape, halo, with_channels=False):
    shape = input_.shape[1:] if with_channels else input_.shape
    starts = [off - ha for off, ha in zip(offset, halo)]
    stops = [off + bs + ha for off, bs, ha in zip(offset, block_shape, halo)]
    pad_left = [max(0, -s) for s in starts]
    pad_right = [max(0, s - sh) for s, sh in zip(stops, shape)]
    bb = tuple(slice(max(0, a), min(sh, b)) for a, b, sh in zip(starts, stops, shape))
    data = input_[(slice(None),) + bb] if with_channels else input_[bb]
    if any(pad_left) or any(pad_right):
        pw = tuple(zip(pad_left, pad_right))
        data = np.pad(data, (((0, 0),) + pw) if with_channels else pw, mode="reflect")
    return data


def blocks(start, stop, block_shape):
    grid = [(sp - st + bs - 1) // bs for st, sp, bs in zip(start, stop, block_shape)]
    for idx in np.ndindex(*grid):
        begin = [st + i * bs for st, i, bs in zip(start, idx, block_shape)]
        yield begin, [min(b + bs, sp) for b, bs, sp in zip(begin, block_shape, stop)]


def standardize(x, eps=1e-7):
    x = x.astype("float32")
    return (x - x.mean()) / (x.std() + eps)


def predict_with_halo(input_, model, block_shape, halo, n_out, with_channels=False, mask=None, roi=None,
                      preprocess=standardize):
    shape = input_.shape[1:] if with_channels else input_.shape
    ndim = len(shape)
    start = [0] * ndim if roi is None else [0 if r.start is None else r.start for r in roi]
    stop = list(shape) if roi is None else [sh if r.stop is None else r.stop for r, sh in zip(roi, shape)]
    out = np.zeros((n_out,) + tuple(shape), dtype="float32")
    for begin, end in blocks(start, stop, block_shape):
        size = [e - b for b, e in zip(begin, end)]
        inner = tuple(slice(ha, ha + s) for ha, s in zip(halo, size))
        if mask is not None:
            mb = load_block(mask, begin, block_shape, halo)[inner].astype(bool)
            if mb.sum() == 0:
                continue
        inp = load_block(input_, begin, block_shape, halo, with_channels)
        if preprocess is not None:
            inp = preprocess(inp)
        x = torch.from_numpy(np.ascontiguousarray(inp[None] if with_channels else inp[None, None]))
        with torch.no_grad():
            pred = model(x).numpy()[0]
        pred = pred[(slice(None),) + inner]
        if mask is not None:
            pred[~np.broadcast_to(mb[None], pred.shape)] = 0
        out[(slice(None),) + tuple(slice(b, e) for b, e in zip(begin, end))] = pred
    return out


def predict_with_padding(model, input_, min_divisible, with_channels=False):
    md = ((1,) + tuple(min_divisible)) if with_channels else tuple(min_divisible)
    pad = tuple((0, 0 if sh % m == 0 else m - sh % m) for sh, m in zip(input_.shape, md))
    x = np.pad(input_, pad, mode="reflect")
    x = torch.from_numpy(x[None] if with_channels else x[None, None])
    with torch.no_grad():
        out = model(x).numpy()
    crop = (slice(None),) * (out.ndim - len(input_.shape) + (1 if with_channels else 0)) + \
        tuple(slice(0, sh) for sh in (input_.shape[1:] if with_channels else input_.shape))
    return out[crop]


def load_block_bb(input_, offset, block_shape, halo, with_channels=False):
    """(data, bounding box incl. the padding) as the reference's _load_block returns them"""
    shape = input_.shape[1:] if with_channels else input_.shape
    starts = [off - ha for off, ha in zip(offset, halo)]
    stops = [off + bs + ha for off, bs, ha in zip(offset, block_shape, halo)]
    clipped = any(s < 0 for s in starts) or any(s > sh for s, sh in zip(stops, shape))
    if clipped:   # the box is extended by the padding on both sides
        bb = [(max(0, a) - max(0, -a), min(sh, b) + max(0, b - sh)) for a, b, sh in zip(starts, stops, shape)]
    else:
        bb = list(zip(starts, stops))
    return load_block(input_, offset, block_shape, halo, with_channels), bb


def pad_for_shift_left(arr, pad_vox, with_channels):
    pw = tuple((int(p), 0) for p in pad_vox)
    return np.pad(arr, (((0, 0),) + pw) if with_channels else pw, mode="constant", constant_values=0.0), tuple(pad_vox)


def crop_after_shift_left(arr, pad_left, with_channels, original_shape_spatial):
    sl = tuple(slice(p, p + sh) for p, sh in zip(pad_left, original_shape_spatial))
    return arr[(slice(None),) + sl] if with_channels else arr[sl]


def prepare_block_input(input_, mask, begin, end, block_shape, halo, with_channels, skip_block, preprocess):
    """-> None (block skipped) or (array [1, (C,) *spatial], mask_block or None, inner box as [(start, stop)])"""
    size = [e - b for b, e in zip(begin, end)]
    inner = tuple(slice(ha, ha + s) for ha, s in zip(halo, size))
    mask_block = None
    if mask is not None:
        mask_block = load_block(mask, begin, block_shape, halo)[inner].astype(bool)
        if mask_block.sum() == 0:
            return None
    inp = load_block(input_, begin, block_shape, halo, with_channels)
    if skip_block is not None and skip_block(inp):
        return None
    if preprocess is not None:
        inp = preprocess(inp)
    return (inp[None] if with_channels else inp[None, None]), mask_block, [(s.start, s.stop) for s in inner]


def write_prediction(prediction, begin, end, output, ndim, mask_block, inner, postprocess):
    if postprocess is not None:
        prediction = postprocess(prediction)
    prediction = prediction[((slice(None),) + inner) if prediction.ndim == ndim + 1 else inner]
    if mask_block is not None:
        prediction[~(np.broadcast_to(mask_block[None], prediction.shape) if prediction.ndim == ndim + 1 else mask_block)] = 0
    bb = tuple(slice(b, e) for b, e in zip(begin, end))
    if isinstance(output, list):
        for out, channel_slice in output:
            out[bb if out.ndim == ndim else (slice(None),) + bb] = prediction[channel_slice]
    else:
        output[((slice(None),) + bb) if output.ndim == ndim + 1 else bb] = prediction
