"""uegan_amd: MI355X-native (gfx950 HIP) implementation of the UEGAN training / inference hot path.

Drop-in mirror of the reference's `models.py` / `losses.py` class API (`uegan_amd.models`, `uegan_amd.losses`) plus
the step driver (`uegan_amd.trainer`) and inference helper (`uegan_amd.tester`).  All device arithmetic lives in
libuegan_hip.so (uegan_amd/csrc, C ABI in include/uegan_hip.h); there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .ops import set_compute_dtype, get_compute_dtype, invalidate_weight_caches, set_precise, precise  # noqa: F401
from . import models, losses, trainer, tester  # noqa: F401
from .models import Generator, Discriminator  # noqa: F401
from .losses import PerceptualLoss, GANLoss, MultiscaleRecLoss, TVLoss  # noqa: F401

__all__ = ["Generator", "Discriminator", "PerceptualLoss", "GANLoss", "MultiscaleRecLoss", "TVLoss", "set_compute_dtype",
           "get_compute_dtype", "invalidate_weight_caches", "set_precise", "precise", "models", "losses", "trainer", "tester"]
