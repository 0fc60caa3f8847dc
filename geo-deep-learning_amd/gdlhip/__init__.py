"""gdlhip -- Python host side of the MI355X-native segmentation hot path.

``gdlhip._lib``   ctypes binding of libgdlhip.so (C-ABI in include/gdlhip.h)
``gdlhip.ops``    raw tensor-level wrappers (no autograd)
``gdlhip.nn``     precision policy, packed-weight cache and the autograd-aware fused ops the
                  reference-shaped modules under ``geo_deep_learning/`` are built from
"""

from . import _lib, ops  # noqa: F401
from ._lib import LIB_PATH, GdlHipError  # noqa: F401


def is_built() -> bool:
    return LIB_PATH.exists()
