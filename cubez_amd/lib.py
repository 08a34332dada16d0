"""ctypes binding of include/cz_hip.h (parts 1-3: drop-in kernels, runtime, async operations)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

GUIDE = 2  # /root/reference/src/cz_cpp/cz_Define.h:40
_HERE = os.path.dirname(os.path.abspath(__file__))
_LOADED: dict = {}

# every symbol include/cz_hip.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "bc_k_", "jacobi_", "psor2sma_core_", "blas_clear_", "blas_copy_", "blas_triad_", "blas_dot1_", "blas_dot2_",
    "blas_bicg_1_", "blas_bicg_2_", "blas_calc_ax_", "blas_calc_rk_",
    "jacobi_maf_", "psor2sma_core_maf_", "calc_rk_maf_", "calc_ax_maf_", "search_pivot_", "pcr_rb_", "imask_k_",
    "czhip_real_bytes", "czhip_arch", "czhip_init", "czhip_finalize", "czhip_alloc_s3d", "czhip_free", "czhip_h2d",
    "czhip_d2h", "czhip_sync", "czhip_stream", "czhip_set_tuning", "czhip_get_tuning",
    "czhip_jacobi_async", "czhip_rbsor_async", "czhip_check_async", "czhip_jacobi_checked_async",
    "czhip_rbsor_checked_async", "czhip_jacobi2_async", "czhip_set_tuning2", "czhip_set_pcr_mode", "czhip_set_pcr_lex", "czhip_set_pcr_lex_timeout", "czhip_set_pcr_lex_limits", "czhip_set_psor", "czhip_set_psor_ahead", "czhip_use_t2", "czhip_set_pair_window", "czhip_set_pair_preload", "czhip_set_unit_coef", "czhip_config_describe", "czhip_set_comm_cus", "czhip_selftest_fastdiv", "czhip_pair_maf_async", "czhip_rbsor2_async", "czhip_rbsor4_async", "czhip_set_rb4", "czhip_jacobi2_from_zero_async", "czhip_jacobi2_from_zero_made_async", "czhip_check2_async", "czhip_pair_split_async", "psor_", "psor_maf_", "pcr_", "pcr_eda_", "pcr_esa_", "pcr_rb_esa_", "pcr_j_esa_", "pcr_rb_maf_", "pcr_rb_esa_maf_", "pcr_maf_", "pcr_eda_maf_", "pcr_esa_maf_",
    "cz_create", "cz_destroy", "cz_evaluate", "cz_setup", "cz_solve", "cz_sweeps", "cz_result_iter", "cz_result_res",
    "cz_history", "cz_field", "cz_local_size", "cz_error_max", "cz_set_quiet", "cz_last_solve_seconds", "cz_kernel_ms",
    "cz_set_debug", "cz_set_profile", "cz_info", "czhip_timing", "czhip_timing_read",
    "cz_comm_unique_id_bytes", "cz_comm_get_unique_id", "cz_comm_bootstrap", "cz_comm_shutdown", "cz_comm_selftest", "cz_comm_auto_division",
    "cz_comm_decompose", "cz_comm_local_world", "cz_comm_local_world_free", "cz_comm_bootstrap_local",
]


def lib_path(prec: str) -> str:
    return os.path.join(_HERE, f"libczhip_{prec}.so")


def load(prec: str = "f32") -> C.CDLL:
    """dlopen the HIP library of one precision.  Raises if it has not been built: there is no fallback."""
    if prec not in _LOADED:
        path = lib_path(prec)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing -- build it with `make -C cubez_amd/csrc` "
                               "(or __graft_entry__.build()); cubez_amd has no CPU fallback")
        _LOADED[prec] = C.CDLL(path, mode=C.RTLD_LOCAL)
    return _LOADED[prec]


class DeviceArray:
    """A device-resident S3D array, shape (NJ+4, NI+4, NK+4) K-fastest (czAllocR_S3D, cz.h:209-232)."""

    def __init__(self, hip: "CzHip", sz):
        self.hip, self.sz = hip, tuple(int(v) for v in sz)
        self.shape = (self.sz[1] + 2 * GUIDE, self.sz[0] + 2 * GUIDE, self.sz[2] + 2 * GUIDE)
        self.nbytes = int(np.prod(self.shape)) * hip.real().itemsize
        arr = (C.c_int * 3)(*self.sz)
        self.ptr = hip.lib.czhip_alloc_s3d(arr)

    def put(self, host: np.ndarray):
        assert host.shape == self.shape and host.dtype == self.hip.real
        host = np.ascontiguousarray(host)
        self.hip.lib.czhip_h2d(self.ptr, host.ctypes.data_as(C.c_void_p), self.nbytes)
        return self

    def get(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.hip.real)
        self.hip.lib.czhip_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes)
        return out

    def free(self):
        if self.ptr:
            self.hip.lib.czhip_free(self.ptr)
            self.ptr = None


class CzHip:
    """The drop-in kernels with the reference's argument conventions (everything by pointer)."""

    def __init__(self, prec: str = "f32", device: int = -1):
        self.prec = prec
        self.real = np.float32 if prec == "f32" else np.float64
        self.creal = C.c_float if prec == "f32" else C.c_double
        self.lib = lib = load(prec)
        lib.czhip_alloc_s3d.restype = C.c_void_p
        lib.czhip_alloc_s3d.argtypes = [C.POINTER(C.c_int)]
        lib.czhip_free.argtypes = [C.c_void_p]
        lib.czhip_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.czhip_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.czhip_stream.restype = C.c_void_p
        lib.czhip_arch.restype = C.c_char_p
        assert lib.czhip_real_bytes() == np.dtype(self.real).itemsize
        if lib.czhip_init(int(device)) != 0:
            raise RuntimeError("czhip_init failed")

    # -- helpers
    def alloc(self, sz, host=None) -> DeviceArray:
        a = DeviceArray(self, sz)
        if host is not None:
            a.put(host)
        return a

    @staticmethod
    def _i(v):
        v = np.ascontiguousarray(v, dtype=np.int32)
        return v, v.ctypes.data_as(C.POINTER(C.c_int))

    def _r(self, v):
        v = np.ascontiguousarray(v, dtype=self.real)
        return v, v.ctypes.data_as(C.c_void_p)

    def _s(self, v):
        return C.byref(self.creal(float(v)))

    def set_tuning(self, threads=0, m=0, tj=-1, pf=-1) -> bool:
        return self.lib.czhip_set_tuning(int(threads), int(m), int(tj), int(pf)) == 0

    def sync(self):
        self.lib.czhip_sync()

    # -- part 3 (two fused sweeps)
    def jacobi2(self, u, w, b, sz, idx, cf, omg, idx1=None, read=True):
        """u -> w = two Jacobi sweeps in one launch; returns (launched, res_first, res_second)."""
        (_, szp), (_, idxp), (_, cfp) = self._i(sz), self._i(idx), self._r(cf)
        idx1p = self._i(idx1)[1] if idx1 is not None else None
        if not hasattr(self, "_dres"):
            self._dres = self.lib.czhip_alloc_s3d((C.c_int * 3)(4, 4, 4))
        self.lib.czhip_jacobi2_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, self.creal, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        ok = self.lib.czhip_jacobi2_async(u.ptr, w.ptr, b.ptr, szp, idxp, idx1p, GUIDE, cfp, float(omg), self._dres, 0.0, 0.0, 0,
                                          None, None, None, None)
        if not read:
            return bool(ok), None, None
        out = (C.c_double * 2)()
        self.lib.czhip_d2h(out, self._dres, 16)
        return bool(ok), out[0], out[1]

    def pair_maf(self, u, w, b, sz, idx, x, y, z, omg, rb_ofst=-1):
        """MAF flavour of jacobi2 / rbsor2: u -> w; returns (launched, res_first, res_second) (rb_ofst >= 0: res_first is the iteration's sum)"""
        (_, szp), (_, idxp) = self._i(sz), self._i(idx)
        if not hasattr(self, "_dres"):
            self._dres = self.lib.czhip_alloc_s3d((C.c_int * 3)(4, 4, 4))
        xs = [np.ascontiguousarray(a, dtype=self.real) for a in (x, y, z)]
        self.lib.czhip_pair_maf_async.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3 + [self.creal, C.c_int, C.c_void_p]
        ok = self.lib.czhip_pair_maf_async(u.ptr, w.ptr, b.ptr, szp, idxp, GUIDE, *[a.ctypes.data_as(C.c_void_p) for a in xs], float(omg), int(rb_ofst),
                                           self._dres)
        out = (C.c_double * 2)()
        self.lib.czhip_d2h(out, self._dres, 16)
        return bool(ok), out[0], out[1]

    def rbsor2(self, u, w, b, sz, idx, cf, ofst, omg, idx1=None):
        """u -> w = one red-black iteration (colour 0 then 1) in one launch; returns (launched, res)."""
        (_, szp), (_, idxp), (_, cfp) = self._i(sz), self._i(idx), self._r(cf)
        idx1p = self._i(idx1)[1] if idx1 is not None else None
        if not hasattr(self, "_dres"):
            self._dres = self.lib.czhip_alloc_s3d((C.c_int * 3)(4, 4, 4))
        self.lib.czhip_rbsor2_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_int, self.creal, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        ok = self.lib.czhip_rbsor2_async(u.ptr, w.ptr, b.ptr, szp, idxp, idx1p, GUIDE, cfp, int(ofst), float(omg), self._dres, 0.0,
                                         0.0, 0, None, None, None, None)
        out = (C.c_double * 2)()
        self.lib.czhip_d2h(out, self._dres, 16)
        return bool(ok), out[0]

    def rbsor4(self, u, w, b, sz, idx, cf, ofst, omg, probe=0):
        """u -> w = TWO red-black iterations (colour 0, 1, 0, 1) in one launch (rb4_k); returns (launched, res of iteration 1, of iteration 2)."""
        (_, szp), (_, idxp), (_, cfp) = self._i(sz), self._i(idx), self._r(cf)
        if not hasattr(self, "_dres"):
            self._dres = self.lib.czhip_alloc_s3d((C.c_int * 3)(4, 4, 4))
        self.lib.czhip_rbsor4_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, self.creal,
                                                C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        ok = self.lib.czhip_rbsor4_async(u.ptr, w.ptr, b.ptr, szp, idxp, GUIDE, cfp, int(ofst), float(omg), self._dres, 0.0, 0.0, 0, None, None, None,
                                         None, int(probe))
        out = (C.c_double * 2)()
        self.lib.czhip_d2h(out, self._dres, 16)
        return bool(ok), out[0], out[1]

    def pass_from_zero(self, w, b_out, sz, idx, cf, omg, op=0, x=None, y=None, z=None, a=0.0, bb=0.0, rb_ofst=-1):
        """first pass of a preconditioner solve: start vector a literal zero, right-hand side read (op 0) or made from x, y, z (op 1: a*x + y,
        op 2: x + a*(z - bb*y)) and stored to b_out; two Jacobi sweeps (rb_ofst < 0) or one red-black iteration; returns launched"""
        (_, szp), (_, idxp), (_, cfp) = self._i(sz), self._i(idx), self._r(cf)
        if not hasattr(self, "_dres"):
            self._dres = self.lib.czhip_alloc_s3d((C.c_int * 3)(4, 4, 4))
        f = self.lib.czhip_jacobi2_from_zero_made_async
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, self.creal, self.creal, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_int, C.c_void_p, self.creal, C.c_int, C.c_void_p, C.c_int]
        ptr = lambda d: d.ptr if d is not None else None  # noqa: E731
        return bool(f(w.ptr, w.ptr, b_out.ptr, int(op), ptr(x), ptr(y), ptr(z), float(a), float(bb), szp, idxp, None, GUIDE, cfp, float(omg), int(rb_ofst),
                      self._dres, 0))

    def pair_split(self, u, w, b, sz, idx, idx1, nID, cf, omg, rb_ofst=-1, read=True):
        """the fused pass as shell slabs + interior (what a decomposed brick runs); returns (launched, res0, res1)."""
        (_, szp), (_, idxp), (_, idx1p), (_, nidp), (_, cfp) = self._i(sz), self._i(idx), self._i(idx1), self._i(nID), self._r(cf)
        if not hasattr(self, "_dres"):
            self._dres = self.lib.czhip_alloc_s3d((C.c_int * 3)(4, 4, 4))
        self.lib.czhip_pair_split_async.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_void_p, self.creal, C.c_int, C.c_void_p]
        ok = self.lib.czhip_pair_split_async(u.ptr, w.ptr, b.ptr, szp, idxp, idx1p, nidp, GUIDE, cfp, float(omg), int(rb_ofst), self._dres)
        if not read:  # leave the queue running (timing loops)
            return bool(ok), None, None
        out = (C.c_double * 2)()
        self.lib.czhip_d2h(out, self._dres, 16)
        return bool(ok), out[0], out[1]

    def timing(self, enable: bool):
        """HIP-event timing of the library's launches, per label (czhip_timing)."""
        self.lib.czhip_timing(1 if enable else 0)

    def timing_read(self, label: str):
        """(number of launches, total ms) recorded under `label` since timing was enabled."""
        tot = C.c_double(0.0)
        self.lib.czhip_timing_read.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
        n = self.lib.czhip_timing_read(label.encode(), C.byref(tot))
        return n, tot.value

    def set_tuning2(self, threads=0, mv=0, tj=-1, enable=-1) -> bool:
        return self.lib.czhip_set_tuning2(int(threads), int(mv), int(tj), int(enable)) == 0

    # -- part 1: same call shapes as oracle.cz_oracle.Kernels
    def bc_k(self, sz, p: DeviceArray, dh, org, nID):
        (_, szp), (_, nidp), (_, orgp), g = self._i(sz), self._i(nID), self._r(org), C.c_int(GUIDE)
        self.lib.bc_k_(szp, C.byref(g), C.c_void_p(p.ptr), self._s(dh), orgp, nidp)

    def jacobi(self, p, sz, idx, cf, omg, b, wk2, res=0.0):
        (_, szp), (_, idxp), (_, cfp), g = self._i(sz), self._i(idx), self._r(cf), C.c_int(GUIDE)
        r, fl = C.c_double(res), C.c_double(0.0)
        self.lib.jacobi_(C.c_void_p(p.ptr), szp, idxp, C.byref(g), cfp, self._s(omg), C.c_void_p(b.ptr), C.byref(r),
                         C.c_void_p(wk2.ptr), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def psor2sma_core(self, p, sz, idx, cf, ofst, color, omg, b, res=0.0):
        (_, szp), (_, idxp), (_, cfp), g = self._i(sz), self._i(idx), self._r(cf), C.c_int(GUIDE)
        r, fl, o, c = C.c_double(res), C.c_double(0.0), C.c_int(ofst), C.c_int(color)
        self.lib.psor2sma_core_(C.c_void_p(p.ptr), szp, idxp, C.byref(g), cfp, C.byref(o), C.byref(c), self._s(omg),
                                C.c_void_p(b.ptr), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    # MAF flavour: x, y, z are host numpy arrays (length N+4), like the reference's xc, yc, zc
    def jacobi_maf(self, p, sz, idx, x, y, z, omg, b, wk2, res=0.0):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        (xk, xp), (yk, yp), (zk, zp) = self._r(x), self._r(y), self._r(z)
        tmp = np.zeros(sz[2] + 4, dtype=self.real)
        r, fl = C.c_double(res), C.c_double(0.0)
        self.lib.jacobi_maf_(C.c_void_p(p.ptr), szp, idxp, C.byref(g), xp, yp, zp, self._s(omg), C.c_void_p(b.ptr), C.byref(r),
                             C.c_void_p(wk2.ptr), tmp.ctypes.data_as(C.c_void_p), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def psor(self, p, sz, idx, cf, omg, b, res=0.0):
        (_, szp), (_, idxp), (_, cfp), g = self._i(sz), self._i(idx), self._r(cf), C.c_int(GUIDE)
        r, fl = C.c_double(res), C.c_double(0.0)
        self.lib.psor_(C.c_void_p(p.ptr), szp, idxp, C.byref(g), cfp, self._s(omg), C.c_void_p(b.ptr), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def psor_maf(self, p, sz, idx, x, y, z, omg, b, res=0.0):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        (xk, xp), (yk, yp), (zk, zp) = self._r(x), self._r(y), self._r(z)
        r, fl = C.c_double(res), C.c_double(0.0)
        self.lib.psor_maf_(C.c_void_p(p.ptr), szp, idxp, C.byref(g), xp, yp, zp, self._s(omg), C.c_void_p(b.ptr), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def psor2sma_core_maf(self, p, sz, idx, x, y, z, ofst, color, omg, b, res=0.0):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        (xk, xp), (yk, yp), (zk, zp) = self._r(x), self._r(y), self._r(z)
        tmp = np.zeros(sz[2] + 4, dtype=self.real)
        r, fl, o, c = C.c_double(res), C.c_double(0.0), C.c_int(ofst), C.c_int(color)
        self.lib.psor2sma_core_maf_(C.c_void_p(p.ptr), szp, idxp, C.byref(g), xp, yp, zp, C.byref(o), C.byref(c), self._s(omg),
                                    C.c_void_p(b.ptr), C.byref(r), tmp.ctypes.data_as(C.c_void_p), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def calc_rk_maf(self, r, p, b, sz, idx, x, y, z, pvt):
        (_, szp), (_, idxp), g, fl = self._i(sz), self._i(idx), C.c_int(GUIDE), C.c_double(0.0)
        (xk, xp), (yk, yp), (zk, zp) = self._r(x), self._r(y), self._r(z)
        self.lib.calc_rk_maf_(C.c_void_p(r.ptr), C.c_void_p(p.ptr), C.c_void_p(b.ptr), szp, idxp, C.byref(g), xp, yp, zp,
                              C.c_void_p(pvt.ptr), C.byref(fl))

    def calc_ax_maf(self, ap, p, sz, idx, x, y, z, pvt):
        (_, szp), (_, idxp), g, fl = self._i(sz), self._i(idx), C.c_int(GUIDE), C.c_double(0.0)
        (xk, xp), (yk, yp), (zk, zp) = self._r(x), self._r(y), self._r(z)
        self.lib.calc_ax_maf_(C.c_void_p(ap.ptr), C.c_void_p(p.ptr), szp, idxp, C.byref(g), xp, yp, zp, C.c_void_p(pvt.ptr),
                              C.byref(fl))

    def search_pivot(self, pvt, sz, idx, x, y, z):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        (xk, xp), (yk, yp), (zk, zp) = self._r(x), self._r(y), self._r(z)
        self.lib.search_pivot_(C.c_void_p(pvt.ptr), szp, idxp, C.byref(g), xp, yp, zp)

    def imask_k(self, x, sz, idx):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        self.lib.imask_k_(C.c_void_p(x.ptr), szp, idxp, C.byref(g))

    def pcr_rb(self, sz, idx, pn, ofst, color, x, msk, rhs, omg, res=0.0):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        w = [np.zeros(sz[2] + 4, dtype=self.real) for _ in range(6)]
        r, fl, pnc, o, c = C.c_double(res), C.c_double(0.0), C.c_int(pn), C.c_int(ofst), C.c_int(color)
        self.lib.pcr_rb_(szp, idxp, C.byref(g), C.byref(pnc), C.byref(o), C.byref(c), C.c_void_p(x.ptr), C.c_void_p(msk.ptr),
                         C.c_void_p(rhs.ptr), *[v.ctypes.data_as(C.c_void_p) for v in w], self._s(omg), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def _pcr_call(self, fn, sz, idx, pn, pre, x, msk, rhs, extra, omg, res):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        w = [np.zeros(sz[2] + 4, dtype=self.real) for _ in range(6)]  # the reference's host scratch: ignored by the library
        r, fl, pnc = C.c_double(res), C.c_double(0.0), C.c_int(pn)
        ints = [C.c_int(v) for v in pre]
        fn(szp, idxp, C.byref(g), C.byref(pnc), *[C.byref(v) for v in ints], C.c_void_p(x.ptr), C.c_void_p(msk.ptr), C.c_void_p(rhs.ptr),
           *[v.ctypes.data_as(C.c_void_p) for v in w], *[C.c_void_p(e.ptr) for e in extra], self._s(omg), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def pcr(self, sz, idx, pn, x, msk, rhs, omg, res=0.0):
        return self._pcr_call(self.lib.pcr_, sz, idx, pn, [], x, msk, rhs, [], omg, res)

    def pcr_eda(self, sz, idx, pn, x, msk, rhs, omg, res=0.0):
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        w = [np.zeros(sz[2] + 4, dtype=self.real) for _ in range(3)]
        r, fl, pnc = C.c_double(res), C.c_double(0.0), C.c_int(pn)
        self.lib.pcr_eda_(szp, idxp, C.byref(g), C.byref(pnc), C.c_void_p(x.ptr), C.c_void_p(msk.ptr), C.c_void_p(rhs.ptr),
                          *[v.ctypes.data_as(C.c_void_p) for v in w], self._s(omg), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def pcr_esa(self, sz, idx, pn, x, msk, rhs, omg, res=0.0):
        return self._pcr_call(self.lib.pcr_esa_, sz, idx, pn, [(1 << pn) >> 2], x, msk, rhs, [], omg, res)

    def pcr_rb_esa(self, sz, idx, pn, ofst, color, x, msk, rhs, omg, res=0.0):
        return self._pcr_call(self.lib.pcr_rb_esa_, sz, idx, pn, [ofst, color, (1 << pn) >> 2], x, msk, rhs, [], omg, res)

    def pcr_j_esa(self, sz, idx, pn, x, msk, rhs, src, wrk, omg, res=0.0):
        return self._pcr_call(self.lib.pcr_j_esa_, sz, idx, pn, [(1 << pn) >> 2], x, msk, rhs, [src, wrk], omg, res)

    def pcr_maf(self, name, sz, idx, pn, color, x, msk, rhs, xc, yc, zc, omg, res=0.0):
        """the MAF line solvers: name in pcr_rb_maf, pcr_rb_esa_maf, pcr_maf, pcr_eda_maf, pcr_esa_maf (same call shape as the oracle)"""
        (_, szp), (_, idxp), g = self._i(sz), self._i(idx), C.c_int(GUIDE)
        (xk, xp), (yk, yp), (zk, zp) = self._r(xc), self._r(yc), self._r(zc)
        nw = 3 if name == "pcr_eda_maf" else 6
        w = [np.zeros(sz[2] + 4, dtype=self.real) for _ in range(nw)]
        tmp = np.zeros(sz[2] + 4, dtype=self.real)
        r, fl, pnc = C.c_double(res), C.c_double(0.0), C.c_int(pn)
        o, c, ss = C.c_int(0), C.c_int(color), C.c_int((1 << pn) >> 2)
        pre = {"pcr_rb_maf": [o, c], "pcr_rb_esa_maf": [o, c, ss], "pcr_maf": [], "pcr_eda_maf": [], "pcr_esa_maf": [ss]}[name]
        getattr(self.lib, name + "_")(szp, idxp, C.byref(g), C.byref(pnc), *[C.byref(v) for v in pre], C.c_void_p(x.ptr), C.c_void_p(msk.ptr),
                                      C.c_void_p(rhs.ptr), xp, yp, zp, *[v.ctypes.data_as(C.c_void_p) for v in w], self._s(omg), C.byref(r),
                                      tmp.ctypes.data_as(C.c_void_p), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def blas_clear(self, x, sz):
        (_, szp), g = self._i(sz), C.c_int(GUIDE)
        self.lib.blas_clear_(C.c_void_p(x.ptr), szp, C.byref(g))

    def blas_copy(self, y, x, sz):
        (_, szp), g = self._i(sz), C.c_int(GUIDE)
        self.lib.blas_copy_(C.c_void_p(y.ptr), C.c_void_p(x.ptr), szp, C.byref(g))

    def blas_triad(self, z, x, y, a, sz, idx):
        (_, szp), (_, idxp), g, fl = self._i(sz), self._i(idx), C.c_int(GUIDE), C.c_double(0.0)
        self.lib.blas_triad_(C.c_void_p(z.ptr), C.c_void_p(x.ptr), C.c_void_p(y.ptr), self._s(a), szp, idxp, C.byref(g),
                             C.byref(fl))

    def blas_dot1(self, p, sz, idx):
        (_, szp), (_, idxp), g, fl = self._i(sz), self._i(idx), C.c_int(GUIDE), C.c_double(0.0)
        r = self.creal(0.0)
        self.lib.blas_dot1_(C.byref(r), C.c_void_p(p.ptr), szp, idxp, C.byref(g), C.byref(fl))
        return self.real(r.value)

    def blas_dot2(self, p, q, sz, idx):
        (_, szp), (_, idxp), g, fl = self._i(sz), self._i(idx), C.c_int(GUIDE), C.c_double(0.0)
        r = self.creal(0.0)
        self.lib.blas_dot2_(C.byref(r), C.c_void_p(p.ptr), C.c_void_p(q.ptr), szp, idxp, C.byref(g), C.byref(fl))
        return self.real(r.value)

    def blas_bicg_1(self, p, r, q, beta, omg, sz, idx):
        (_, szp), (_, idxp), g, fl = self._i(sz), self._i(idx), C.c_int(GUIDE), C.c_double(0.0)
        self.lib.blas_bicg_1_(C.c_void_p(p.ptr), C.c_void_p(r.ptr), C.c_void_p(q.ptr), self._s(beta), self._s(omg), szp,
                              idxp, C.byref(g), C.byref(fl))

    def blas_bicg_2(self, z, x, y, a, b, sz, idx):
        (_, szp), (_, idxp), g, fl = self._i(sz), self._i(idx), C.c_int(GUIDE), C.c_double(0.0)
        self.lib.blas_bicg_2_(C.c_void_p(z.ptr), C.c_void_p(x.ptr), C.c_void_p(y.ptr), self._s(a), self._s(b), szp, idxp,
                              C.byref(g), C.byref(fl))

    def blas_calc_ax(self, ap, p, sz, idx, cf):
        (_, szp), (_, idxp), (_, cfp), g, fl = self._i(sz), self._i(idx), self._r(cf), C.c_int(GUIDE), C.c_double(0.0)
        self.lib.blas_calc_ax_(C.c_void_p(ap.ptr), C.c_void_p(p.ptr), szp, idxp, C.byref(g), cfp, C.byref(fl))

    def blas_calc_rk(self, r, p, b, sz, idx, cf):
        (_, szp), (_, idxp), (_, cfp), g, fl = self._i(sz), self._i(idx), self._r(cf), C.c_int(GUIDE), C.c_double(0.0)
        self.lib.blas_calc_rk_(C.c_void_p(r.ptr), C.c_void_p(p.ptr), C.c_void_p(b.ptr), szp, idxp, C.byref(g), cfp,
                               C.byref(fl))
