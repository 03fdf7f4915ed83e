"""C-ABI checks that need no GPU: the built library loads and exports every symbol include/uegan_hip.h declares,
and the ctypes table in uegan_amd/_lib.py covers exactly that set."""
import ctypes
import os
import re

import pytest

from helpers import ROOT
from uegan_amd import _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "uegan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uegan_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_ctypes_table():
    assert header_symbols() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.uegan_version.restype = ctypes.c_int
    assert lib.uegan_version() == 105          # host-only call, no GPU needed
    # the fp16-storage build of the same sources (uegan_amd.set_compute_dtype(torch.float16)) has the same ABI
    lib16 = ctypes.CDLL(_lib.LIB_PATH_F16)
    missing = [s for s in header_symbols() if not hasattr(lib16, s)]
    assert not missing, missing


def test_missing_library_fails_loudly(tmp_path):
    _lib._reset_for_tests()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "nope.so"))
    _lib._reset_for_tests()


def test_ops_refuse_cpu_tensors_without_emulator():
    import torch
    from uegan_amd import ops
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    _lib._reset_for_tests()
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.mul(torch.zeros(4), torch.zeros(4))
    _lib._reset_for_tests()
