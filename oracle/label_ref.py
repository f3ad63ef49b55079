"""CPU restatement of the label targets (TEST INFRASTRUCTURE, see oracle/__init__.py).

affinities(): AffinityTransform.__call__  /root/reference/torch_em/transform/label.py:290-327.
  The arithmetic is in bioimage-cpp (un-vendored C++, unpinned in setup.py:7-23 /
  environment.yaml:5-18), so the restatement follows the brute-force DEFINITIONS that the
  reference's own test holds: test/transform/test_label_transforms.py:5-20 (no ignore label) and
  :23-55 (ignore label 0; mask_bg_transition <-> include_ignore_transitions=False); 3-D is the
  same rule with one more axis.  Channel order / dtype / "1 - affs" follow label.py:299-325.
boundaries(): BoundaryTransform.__call__  /root/reference/torch_em/transform/label.py:113-129 =
  skimage.segmentation.find_boundaries(mode="thick") = grey-dilation != grey-erosion with the
  connectivity-1 footprint (scikit-image, un-vendored).  PARITY UNPINNED: no reference test
  exercises it; checked here against scipy.ndimage's dilation/erosion and hand-made KATs.
"""
import numpy as np


def affinities(labels, offsets, ignore_label=None, add_binary_target=False, add_mask=False,
               include_ignore_transitions=False):
    labels = np.asarray(labels)
    nd = labels.ndim
    shape = labels.shape
    n_off = len(offsets)
    affs = np.ones((n_off,) + shape, dtype="float32")
    mask = np.zeros((n_off,) + shape, dtype="float32")
    for c, off in enumerate(offsets):
        assert len(off) == nd
        src, dst = [], []
        empty = False
        for d in range(nd):
            o = int(off[d])
            lo, hi = max(0, -o), min(shape[d], shape[d] - o)
            if hi <= lo:
                empty = True
                break
            src.append(slice(lo, hi))
            dst.append(slice(lo + o, hi + o))
        if empty:
            continue
        src, dst = tuple(src), tuple(dst)
        val, oval = labels[src], labels[dst]
        a = (val != oval).astype("float32")
        m = np.ones_like(a)
        if ignore_label is not None:
            n_ign = (val == ignore_label).astype(int) + (oval == ignore_label).astype(int)
            bad = n_ign == 2
            if not include_ignore_transitions:
                bad |= n_ign == 1
            a[bad] = 1.0
            m[bad] = 0.0
        affs[(c,) + src] = a
        mask[(c,) + src] = m
    out = affs
    if add_binary_target:
        out = np.concatenate([(labels != 0)[None].astype("float32"), out], axis=0)
    if add_mask:
        if add_binary_target:
            mb = np.ones((1,) + shape, "float32") if ignore_label is None else \
                (labels != ignore_label)[None].astype("float32")
            mask = np.concatenate([mb, mask], axis=0)
        out = np.concatenate([out, mask], axis=0)
    return out


def affinities_brute_force(labels, offsets, ignore_label=None, include_ignore_transitions=False):
    """Pure-Python loops in the style of the reference's test definitions (small inputs only)."""
    labels = np.asarray(labels)
    shape = labels.shape
    affs = np.zeros((len(offsets),) + shape, "float32")
    mask = np.zeros((len(offsets),) + shape, "float32")
    for idx in np.ndindex(*shape):
        val = labels[idx]
        for c, off in enumerate(offsets):
            o = tuple(i + int(d) for i, d in zip(idx, off))
            if any(v < 0 or v >= s for v, s in zip(o, shape)):
                affs[(c,) + idx], mask[(c,) + idx] = 1.0, 0.0
                continue
            oval = labels[o]
            if ignore_label is not None:
                n_ign = int(val == ignore_label) + int(oval == ignore_label)
                if n_ign == 2 or (n_ign == 1 and not include_ignore_transitions):
                    affs[(c,) + idx], mask[(c,) + idx] = 1.0, 0.0
                    continue
            affs[(c,) + idx] = 0.0 if val == oval else 1.0
            mask[(c,) + idx] = 1.0
    return affs, mask


def boundaries(labels, add_binary_target=False):
    labels = np.asarray(labels)
    b = np.zeros(labels.shape, dtype=bool)
    for d in range(labels.ndim):
        a = [slice(None)] * labels.ndim
        c = [slice(None)] * labels.ndim
        a[d], c[d] = slice(0, -1), slice(1, None)
        diff = labels[tuple(a)] != labels[tuple(c)]
        b[tuple(a)] |= diff
        b[tuple(c)] |= diff
    out = b[None].astype("float32")
    if add_binary_target:
        out = np.concatenate([(labels != 0)[None].astype("float32"), out], axis=0)
    return out


def boundaries_mode(labels, mode="thick", add_binary_target=False):
    """find_boundaries(labels, mode) for the same-shape modes, from the definitions (neighbour loops, no morphology):
    thick = some face neighbour differs; inner = thick & foreground; outer = thick & (background | the full 3^ndim
    window of a foreground voxel holds max(label) != min(label, background -> dtype max)).  scikit-image is absent from
    this image (reference setup.py:17, unpinned): pinned by the three known-answer arrays of its find_boundaries
    docstring (tests/test_oracle_golden.py) and by `boundaries_morphology`, the published algorithm on scipy.ndimage."""
    labels = np.asarray(labels)
    thick = boundaries(labels)[0].astype(bool)
    if mode == "thick":
        b = thick
    elif mode == "inner":
        b = thick & (labels != 0)
    elif mode == "outer":
        big = np.iinfo(labels.dtype).max
        inv = np.where(labels == 0, big, labels)
        pad_l = np.pad(labels, 1, mode="edge")     # edge padding never changes a window's max / min
        pad_i = np.pad(inv, 1, mode="edge")
        mx, mn = labels.copy(), inv.copy()
        import itertools
        for off in itertools.product((0, 1, 2), repeat=labels.ndim):
            sl = tuple(slice(o, o + n) for o, n in zip(off, labels.shape))
            mx = np.maximum(mx, pad_l[sl])
            mn = np.minimum(mn, pad_i[sl])
        b = thick & ((labels == 0) | ((mx != mn) & (labels != 0)))
    else:
        raise ValueError(mode)
    out = b[None].astype("float32")
    if add_binary_target:
        out = np.concatenate([(labels != 0)[None].astype("float32"), out], axis=0)
    return out


def boundaries_morphology(labels, mode="thick"):
    """The documented find_boundaries(mode='thick') algorithm via scipy.ndimage (cross-check)."""
    from scipy import ndimage as ndi
    # scipy's separable min / max filters keep their intermediate results in the input dtype after computing in double:
    # the "outer" stand-in value iinfo(int64).max is not a double and comes back as INT64_MIN (so find_boundaries itself
    # is off for int64 / uint64 label images there).  The cross-check therefore runs on int32, where the published
    # algorithm is exact; `boundaries_mode` and the HIP kernel evaluate the definition in integers for every dtype.
    assert np.abs(labels).max() < 2 ** 31 - 1
    labels = labels.astype("int32")
    fp = ndi.generate_binary_structure(labels.ndim, 1)
    b = ndi.grey_dilation(labels, footprint=fp) != ndi.grey_erosion(labels, footprint=fp)
    if mode == "inner":
        b &= labels != 0
    elif mode == "outer":
        bg = labels == 0
        fp = ndi.generate_binary_structure(labels.ndim, labels.ndim)
        inv = np.array(labels, copy=True)
        inv[bg] = np.iinfo(labels.dtype).max
        b &= bg | ((ndi.grey_dilation(labels, footprint=fp) != ndi.grey_erosion(inv, footprint=fp)) & ~bg)
    return b[None].astype("float32")
