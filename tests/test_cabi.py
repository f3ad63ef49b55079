"""CPU: the C-ABI shared library loads and exports every symbol include/tem_hip.h declares
(no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tem_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tem_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = _declared_symbols()
    assert "tem_conv3d_fwd" in syms and "tem_dice_sums" in syms and len(syms) >= 25


def test_library_exports_every_declared_symbol():
    from torch_em_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"not exported: {missing}"


def test_ctypes_signatures_cover_the_header():
    from torch_em_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_version_and_error_string():
    from torch_em_amd import _lib
    lib = _lib.load()
    assert lib.tem_version() >= 100
    assert isinstance(lib.tem_last_error(), bytes)
    # argument validation happens before any HIP call: exercise the error path on CPU
    rc = lib.tem_conv3d_fwd(None, 0, None, None, None, None, None, 0, None, 0, None, 0, 1, 1, 1, 1, 1, 1, 3, 3, 3, 0, 0,
                            None)
    assert rc == -1 and b"null pointer" in lib.tem_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "tem_conv3d_fwd")


def test_cfg2_layers_select_the_team_kernels():
    """Dispatch guard (no GPU needed: tem_conv3d_fwd_kernel is host logic, 256 CUs assumed without a device): every 3x3x3
    forward / data-gradient convolution of the benchmark network (UNet3d(1->2, 32 features, depth 4) on 2x1x128^3,
    BASELINE.json cfg 2) runs on the z-reuse team kernel -- family 3 with fused statistics at the 128^3 ... 32^3 levels,
    family 4 (split input channels, statistics from the split-K epilogue) where there are too few tiles: a change that silently sends one of
    them back to the one-patch-per-workgroup kernel costs 0.1-1 ms per step."""
    from torch_em_amd import _lib
    lib = _lib.load()
    # (size, features) per level; each block has cin->f and f->f convs, the decoder's first conv reads 2f channels
    levels = [(128, 32), (64, 64), (32, 128), (16, 256)]
    layers = []
    for i, (s, f) in enumerate(levels):
        layers += [(s, f, f), (s, 2 * f, f)]                       # conv2 of both blocks / decoder conv1
        if i:
            layers.append((s, f // 2, f))                          # encoder conv1 (level 0 has Cin = 1: the row kernel)
    layers += [(8, 256, 512), (8, 512, 512)]                       # base block
    for s, cin, cout in layers:
        for a, b in ((cin, cout), (cout, cin)):                    # forward and its data gradient (transposed channels)
            for mode in (2, 4, 5, 7):
                fam = lib.tem_conv3d_fwd_kernel(2, s, s, s, a, b, 3, 3, 3, mode)
                units = 2 * (s // 4) * max(s // 16, 1) * (s // 8) * (b // 32)
                assert fam == (3 if units >= 512 else 4), (s, a, b, mode, fam)
                blocks = lib.tem_conv3d_fwd_stat_blocks(2, s, s, s, a, b, 3, 3, 3, mode)
                # statistics partials come from the team kernel's epilogue or, round 4, from the split-K epilogue (blocks of
                # 4 rows of 256 / (Cout / 4) voxels)
                vb = 4 * (256 // (b // 4))
                assert blocks > 0 and (fam == 3 or blocks == (s ** 3 + vb - 1) // vb), (s, a, b, mode, blocks)
