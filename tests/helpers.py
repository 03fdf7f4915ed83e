"""Shared test plumbing: backend selection (real GPU library vs CPU fiber emulator), tolerances, fixtures."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libuegan_emu.so")
EMU_LIB_F16 = os.path.join(ROOT, "tests", "emu", "_build", "libuegan_emu_f16.so")      # the same sources with fp16 as the 16-bit storage format

from uegan_amd import _lib  # noqa: E402

_emu_built = [False]


def build_emu():
    if not _emu_built[0]:
        r = subprocess.run(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.fail("emulator build failed:\n" + r.stdout + r.stderr)
        _emu_built[0] = True


# Launch-variant thresholds (include/uegan_hip.h: uegan_set_tuning).  The library never reads the environment; a test asks for a value
# with set_tuning(), use_backend() applies the wishes to whichever library it loads, conftest.py resets everything after each test.
TUNING = {"SMALL_GRID": (0, 256), "FOLD_MAX": (1, -1), "HEADS_NO_CG": (2, 0), "WIDE_MIN_GRID": (3, 192), "TALL_MIN_GRID": (4, 192), "TALL_RPW": (5, 0), "TALL_REFLECT": (6, 1), "FLAT_S2": (7, 1), "TOEP_HEADS": (8, 1), "HEADS_MFMA": (9, 1), "FWD_STATS": (10, 1), "WGRAD_XCD": (11, 1)}
_want_tuning = {}


def _apply_tuning():
    if _lib._lib is None:
        return
    for lib in (_lib._lib, _lib._lib_f16, getattr(_lib, "_emu_f16", None)):      # every loaded build: bf16- and fp16-format libraries
        if lib is None:
            continue
        for name, (knob, default) in TUNING.items():
            _lib.check(lib.uegan_set_tuning(knob, _want_tuning.get(name, default), None))


def set_tuning(name, value):
    assert name in TUNING, name
    _want_tuning[name] = int(value)
    _apply_tuning()


def reset_tuning():
    _want_tuning.clear()
    _apply_tuning()


def _install_dtype_hook():
    """Every ops.set_compute_dtype() in a test re-applies the requested tuning to the build that dtype selects.  The fp16-format library is
    loaded lazily (by the first op after set_compute_dtype(float16)), so a set_tuning() made BEFORE the switch would otherwise never reach it
    and a test run alone with -k would pass through a different kernel than the one it names."""
    import uegan_amd
    from uegan_amd import ops
    if getattr(ops.set_compute_dtype, "_tuned", False):
        return
    orig = ops.set_compute_dtype

    def set_compute_dtype(dt):
        orig(dt)
        if _lib._lib is not None and dt != torch.float32 and (_lib.is_emulated() or torch.cuda.is_available()):
            _lib.load()               # (binds the fp16-format build now instead of at the first op)
        _apply_tuning()

    set_compute_dtype._tuned = True
    set_compute_dtype.__doc__ = orig.__doc__
    ops.set_compute_dtype = set_compute_dtype
    uegan_amd.set_compute_dtype = set_compute_dtype


_install_dtype_hook()


def use_backend(kind):
    """kind 'gpu': the real libuegan_hip.so on cuda:0; kind 'emu': the same kernel sources on the CPU emulator."""
    if kind == "gpu":
        if _lib.is_emulated():
            _lib._reset_for_tests()
        _lib.load()
        _apply_tuning()
        return torch.device("cuda:0")
    build_emu()
    if not _lib.is_emulated():
        _lib._inject_for_tests(EMU_LIB, EMU_LIB_F16)
    _apply_tuning()
    return torch.device("cpu")


# every kernel-level test runs twice: on the emulator (CPU CI, -m "not gpu") and on the MI355X (-m gpu)
BACKENDS = [pytest.param("emu", id="emu"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def tens(z, key, dev=None):
    t = torch.from_numpy(np.asarray(z[key]))
    return t.to(dev) if dev is not None else t


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def half_round(t, dtype):
    """round to the 16-bit storage dtype (bf16 or fp16) and back"""
    return t.to(dtype).to(torch.float32)
