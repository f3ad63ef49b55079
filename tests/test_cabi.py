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
