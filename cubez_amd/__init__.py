"""cubez_amd -- MI355X-native (gfx950) implementation of the CubeZ iterative-solver hot path.

The product is the C-ABI library ``libczhip_{f32,f64}.so`` (include/cz_hip.h): hand-written HIP
kernels behind the reference's own ``cz_Ffunc.h`` operator boundary plus the restated solver
loops.  This package is only the thin ctypes binding used by the tests, ``bench.py`` and the
Python launcher; there is no CPU fallback -- loading fails loudly when the library is missing.
"""
from .driver import CZ  # noqa: F401
from .lib import CzHip, DeviceArray, GUIDE, lib_path, load  # noqa: F401

__all__ = ["CZ", "CzHip", "DeviceArray", "GUIDE", "lib_path", "load"]
