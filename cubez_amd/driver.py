"""ctypes binding of parts 4-5 of include/cz_hip.h: the restated CubeZ driver (class CZ) behind a handle.

Mirrors the reference CLI (src/main.cpp:15-60): ``CZ(prec).evaluate(["64","64","64","jacobi","4000","0.8"])``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .lib import GUIDE, load


class CZ:
    def __init__(self, prec: str = "f32", quiet: bool = True, device: int = -1):
        self.prec = prec
        self.real = np.float32 if prec == "f32" else np.float64
        self.lib = lib = load(prec)
        lib.cz_create.restype = C.c_void_p
        for name in ("cz_destroy", "cz_solve", "cz_result_iter"):
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.cz_evaluate.argtypes = lib.cz_setup.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]
        lib.cz_sweeps.argtypes = [C.c_void_p, C.c_int]
        lib.cz_result_res.argtypes = [C.c_void_p]
        lib.cz_result_res.restype = C.c_double
        lib.cz_history.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        lib.cz_field.argtypes = [C.c_void_p, C.c_void_p]
        lib.cz_local_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        lib.cz_error_max.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.cz_error_max.restype = C.c_double
        lib.cz_set_quiet.argtypes = lib.cz_set_debug.argtypes = [C.c_void_p, C.c_int]
        lib.cz_last_solve_seconds.argtypes = [C.c_void_p]
        lib.cz_last_solve_seconds.restype = C.c_double
        lib.cz_kernel_ms.argtypes = [C.c_void_p, C.c_char_p]
        lib.cz_kernel_ms.restype = C.c_double
        lib.czhip_timing_read.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
        lib.cz_info.argtypes = [C.c_void_p, C.c_int]
        if lib.czhip_init(int(device)) != 0:
            raise RuntimeError("czhip_init failed")
        self.h = lib.cz_create()
        lib.cz_set_quiet(self.h, 1 if quiet else 0)

    @staticmethod
    def _argv(args):
        argv = [b"cz"] + [str(a).encode() for a in args]
        arr = (C.c_char_p * len(argv))(*argv)
        return len(argv), arr

    def evaluate(self, args) -> int:
        n, arr = self._argv(args)
        return self.lib.cz_evaluate(self.h, n, arr)

    def setup(self, args) -> int:
        n, arr = self._argv(args)
        return self.lib.cz_setup(self.h, n, arr)

    def solve(self) -> int:
        return self.lib.cz_solve(self.h)

    def sweeps(self, n: int) -> int:
        return self.lib.cz_sweeps(self.h, int(n))

    @property
    def iter(self) -> int:
        return self.lib.cz_result_iter(self.h)

    @property
    def res(self) -> float:
        return self.lib.cz_result_res(self.h)

    @property
    def solve_seconds(self) -> float:
        return self.lib.cz_last_solve_seconds(self.h)

    def history(self):
        n = self.lib.cz_history(self.h, None, 0)
        out = (C.c_double * max(n, 1))()
        self.lib.cz_history(self.h, out, n)
        return [out[i] for i in range(n)]

    def history_text(self) -> str:
        return "Itration      Residual\n" + "".join("%6d, %13.6e\n" % (i + 1, r) for i, r in enumerate(self.history()))

    def local(self):
        a = [(C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 6)(), (C.c_int * 6)()]
        self.lib.cz_local_size(self.h, *a)
        return dict(size=list(a[0]), head=list(a[1]), nID=list(a[2]), inner=list(a[3]))

    def field(self) -> np.ndarray:
        sz = self.local()["size"]
        out = np.empty((sz[1] + 2 * GUIDE, sz[0] + 2 * GUIDE, sz[2] + 2 * GUIDE), dtype=self.real)
        self.lib.cz_field(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def error_max(self):
        loc = (C.c_int * 3)()
        d = self.lib.cz_error_max(self.h, loc)
        return d, tuple(loc)

    def info(self) -> dict:
        """what a (multi-GPU) run decided (cz_info of include/cz_hip.h)"""
        keys = ("ranks", "fused_pass", "shell_slabs", "overlap", "lagged_reduce", "rccl_ranks", "comm_cus", "pass_kind", "exchange_depth", "buffers", "bicg_fused", "rb4_passes")
        return {k: self.lib.cz_info(self.h, i) for i, k in enumerate(keys)}

    def timing(self, enable: bool):
        self.lib.czhip_timing(1 if enable else 0)

    def timing_read(self, label: str):
        tot = C.c_double(0.0)
        n = self.lib.czhip_timing_read(label.encode(), C.byref(tot))
        return n, tot.value

    def close(self):
        if self.h:
            self.lib.cz_destroy(self.h)
            self.h = None
