"""roctx ranges around the kernel groups of a step ("roctx ranges per kernel group"), so that a
`rocprofv3 --marker-trace --kernel-trace` timeline reads as encoder / neck / decoder / heads / loss / backward / optimizer instead
of 500 kernel names.  Off unless GDL_ROCTX=1 (the ranges are host calls into libroctx64.so: nothing is launched, nothing enters a
hipGraph); with it off `rng` costs one attribute test.

    GDL_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats -d out -- python bench.py --steps 5 --no-extras
"""

from __future__ import annotations

import ctypes
import os
from contextlib import contextmanager

_lib = None
ENABLED = os.environ.get("GDL_ROCTX") == "1"


def _load():
    global _lib, ENABLED
    if _lib is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                _lib = ctypes.CDLL(name)
                _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                _lib.roctxRangePushA.restype = ctypes.c_int
                _lib.roctxRangePop.restype = ctypes.c_int
                break
            except (OSError, AttributeError):
                _lib = None
        if _lib is None:
            ENABLED = False          # asked for, not available: say so once and carry on unmarked
            import logging
            logging.getLogger(__name__).warning("GDL_ROCTX=1 but libroctx64.so could not be loaded: no ranges")
    return _lib


@contextmanager
def rng(name: str):
    """`with rng("neck"):` -- a roctx range named gdl/<name> around the host code that launches the group's kernels."""
    if not ENABLED:
        yield
        return
    lib = _load()
    if lib is None:
        yield
        return
    lib.roctxRangePushA(f"gdl/{name}".encode())
    try:
        yield
    finally:
        lib.roctxRangePop()
