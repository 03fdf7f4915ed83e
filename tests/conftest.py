import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libuegan_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the kernel sources on the CPU fiber emulator (tests/emu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    has_gpu = _gpu_available()
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))


@pytest.fixture(scope="session")
def emu_lib():
    """Build (if needed) and inject the CPU-emulated kernel library. CPU-only sessions."""
    if _gpu_available():
        pytest.skip("emulator tests run only where no GPU is visible")
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("emulator build failed:\n" + r.stdout + r.stderr)
    from uegan_amd import _lib
    _lib._inject_for_tests(EMU_LIB, os.path.join(os.path.dirname(EMU_LIB), "libuegan_emu_f16.so"))
    return _lib.load()


@pytest.fixture(autouse=True)
def _reset_library_tuning():
    """launch-variant thresholds a test set through helpers.set_tuning() do not leak into the next test"""
    yield
    import helpers
    helpers.reset_tuning()
    # ... nor does the 16-bit storage format (uegan_amd.set_compute_dtype(torch.float16) routes to the fp16-format build)
    import torch
    from uegan_amd import _lib, ops
    if ops.get_compute_dtype() == torch.float16:
        ops.set_compute_dtype(torch.float32)
    _lib.use_half_format("bf16")
