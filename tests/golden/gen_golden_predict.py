"""Golden vectors of the reference's OWN tiled-prediction helpers (build container only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_predict.py

G9  `torch_em.util.prediction` -- `_load_block` (util/prediction.py:98-142: halo crossing no / the left / the right / both
    borders, 2-D and 3-D, with and without a channel axis), `_pad_for_shift_left` / `_crop_after_shift_left` (:79-95),
    `_prepare_block_input` (:388-417: mask check, skip, the default `standardize` preprocessing) and `_write_prediction`
    (:420-447: inner crop, mask zeroing, single / channel-axis / channel-split outputs, postprocess) -- called on seeded
    arrays; inputs, arguments and everything they return are stored in g9_predict_helpers.npz.

None of these functions touches `bioimage_cpp` (absent here: it only provides the block grid `Blocking`, which therefore
stays the one unpinned piece of the tiled prediction); the module imports behind the stub finder of gen_golden_trainer.py
(SURVEY.md 8c route B).  A block is handed over as a plain object with `begin` / `end` / `shape` attributes -- what the
functions read from a `Blocking` block.  The fixture is data; no reference source is copied.
"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_trainer import OUT, import_reference  # noqa: E402


def block(begin, end):
    return types.SimpleNamespace(begin=list(begin), end=list(end), shape=[e - b for b, e in zip(begin, end)])


def main():
    import_reference()
    from torch_em.util import prediction as P
    rng = np.random.default_rng(90)
    out = {}
    # ---- _load_block -------------------------------------------------------------------------------------------------
    vol3 = rng.standard_normal((7, 9, 11)).astype("float32")
    vol3c = rng.standard_normal((2, 7, 9, 11)).astype("float32")
    vol2 = rng.standard_normal((13, 10)).astype("float32")
    out.update(vol3=vol3, vol3c=vol3c, vol2=vol2)
    cases = [
        # name, volume key, with_channels, offset, block_shape, halo
        ("lb_inner", "vol3", False, (2, 3, 4), (2, 3, 3), (1, 2, 2)),
        ("lb_left", "vol3", False, (0, 0, 0), (3, 4, 5), (2, 1, 3)),
        ("lb_right", "vol3", False, (4, 6, 8), (3, 3, 3), (1, 2, 2)),
        ("lb_both", "vol3", False, (0, 0, 0), (7, 9, 11), (2, 3, 4)),
        ("lb_clipped_last", "vol3", False, (6, 8, 10), (2, 2, 2), (1, 1, 1)),   # the last block of an axis: block_shape runs past the volume
        ("lb_nohalo", "vol3", False, (3, 3, 3), (4, 6, 8), (0, 0, 0)),
        ("lb_ch_left", "vol3c", True, (0, 2, 0), (3, 3, 6), (1, 1, 2)),
        ("lb_ch_both", "vol3c", True, (0, 0, 0), (7, 9, 11), (1, 2, 3)),
        ("lb_2d_left", "vol2", False, (0, 0), (5, 4), (3, 2)),
        ("lb_2d_right", "vol2", False, (10, 8), (5, 4), (2, 1)),
    ]
    names = []
    for name, key, wc, off, bs, ha in cases:
        data, bb = P._load_block(out[key], list(off), bs, ha, with_channels=wc)
        out[name + ".args"] = np.array([list(off), list(bs), list(ha)], dtype=np.int64)
        out[name + ".data"] = np.ascontiguousarray(data)
        out[name + ".bb"] = np.array([[s.start, s.stop] for s in bb], dtype=np.int64)
        names.append(f"{name}|{key}|{int(wc)}")
    out["load_block_cases"] = np.array(names)
    # ---- grid-shift padding -------------------------------------------------------------------------------------------
    padded, pad_left = P._pad_for_shift_left(vol3, (1, 0, 3), False)
    out["shift.pad"], out["shift.pad_left"] = padded, np.array(pad_left)
    out["shift.crop"] = P._crop_after_shift_left(padded, pad_left, False, vol3.shape)
    padded_c, pad_left_c = P._pad_for_shift_left(vol3c, (2, 1, 0), True)
    out["shift.pad_c"] = padded_c
    out["shift.crop_c"] = P._crop_after_shift_left(padded_c, pad_left_c, True, vol3c.shape[1:])
    # ---- _prepare_block_input ---------------------------------------------------------------------------------------------
    mask = np.zeros(vol3.shape, dtype="uint8")
    mask[:4, 2:8, :5] = 1
    out["mask3"] = mask
    prep = []
    for name, blk, bs, ha, use_mask in (("pb_masked", block((0, 0, 0), (3, 4, 5)), (3, 4, 5), (1, 1, 2), True),
                                        ("pb_masked_out", block((4, 4, 5), (7, 8, 10)), (3, 4, 5), (1, 1, 2), True),
                                        ("pb_plain_last", block((6, 8, 10), (7, 9, 11)), (3, 4, 5), (2, 2, 2), False)):
        res = P._prepare_block_input(vol3, mask if use_mask else None, blk, bs, ha, False, None, P.standardize)
        skipped = res is P._SKIP
        out[name + ".args"] = np.array([blk.begin, blk.end, list(bs), list(ha)], dtype=np.int64)
        out[name + ".skipped"] = np.array(skipped)
        if not skipped:
            tensor, mask_block, inner_bb = res
            out[name + ".tensor"] = tensor.numpy()
            out[name + ".inner_bb"] = np.array([[s.start, s.stop] for s in inner_bb], dtype=np.int64)
            if mask_block is not None:
                out[name + ".mask_block"] = mask_block
        prep.append(f"{name}|{int(use_mask)}")
    out["prepare_cases"] = np.array(prep)
    # with a channel axis and a skip_block callable
    res = P._prepare_block_input(vol3c, None, block((0, 3, 0), (3, 6, 6)), (3, 3, 6), (1, 1, 2), True, lambda a: a.mean() > 1e9, None)
    out["pb_channels.tensor"] = res[0].numpy()
    res = P._prepare_block_input(vol3c, None, block((0, 3, 0), (3, 6, 6)), (3, 3, 6), (1, 1, 2), True, lambda a: True, None)
    out["pb_channels.skip_all"] = np.array(res is P._SKIP)
    # ---- _write_prediction ------------------------------------------------------------------------------------------------
    halo = (1, 2, 2)
    blk = block((2, 3, 4), (4, 6, 7))
    inner_bb = tuple(slice(h, h + s) for h, s in zip(halo, blk.shape))
    pred = rng.standard_normal((3, 4, 7, 7)).astype("float32")       # [C, block + 2 halo]
    mb = rng.random((2, 3, 3)) > 0.4
    out["wp.pred"], out["wp.mask_block"] = pred, mb
    out["wp.args"] = np.array([blk.begin, blk.end, list(halo)], dtype=np.int64)
    o1 = np.full((3, 7, 9, 11), -7.0, dtype="float32")
    P._write_prediction(pred.copy(), blk, o1, 3, None, inner_bb, None)
    out["wp.out_channels"] = o1
    o2 = np.full((3, 7, 9, 11), -7.0, dtype="float32")
    P._write_prediction(pred.copy(), blk, o2, 3, mb, inner_bb, None)
    out["wp.out_masked"] = o2
    oa, ob = np.full((7, 9, 11), -7.0, dtype="float32"), np.full((2, 7, 9, 11), -7.0, dtype="float32")
    P._write_prediction(pred.copy(), blk, [(oa, 0), (ob, slice(1, 3))], 3, mb, inner_bb, None)
    out["wp.out_list_a"], out["wp.out_list_b"] = oa, ob
    o3 = np.full((7, 9, 11), -7.0, dtype="float32")
    P._write_prediction(pred.copy(), blk, o3, 3, mb, inner_bb, lambda p: p[1] * 2.0)   # postprocess drops the channel axis
    out["wp.out_post"] = o3
    np.savez_compressed(os.path.join(OUT, "g9_predict_helpers.npz"), **out)
    print("G9:", len(out), "arrays;", ", ".join(names))


if __name__ == "__main__":
    main()
