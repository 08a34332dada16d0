"""ctypes front-end of the parity oracle + a restatement of the reference's host loops.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from cubez_amd/ (the product).

Two interchangeable kernel back-ends, same call signatures:

* ``kind="oracle"``  oracle/liboracle_{f32,f64}.so -- the C restatement (cz_oracle.c)
* ``kind="ref"``     oracle/_ref/libczref_{f32,f64}.so -- the reference's own Fortran
                      (cz_solver.f90 / cz_blas.f90 / cz_utility.f90) compiled by oracle/Makefile

The host loops below restate /root/reference/src/cz_cpp/cz_Poisson.cpp
(JACOBI :30-82, RBSOR :159-235, Fdot1/2 :239-270, Preconditioner :273-322,
PBiCGSTAB :332-504) and the set-up part of cz_Evaluate.cpp (:56-99, :160-177,
:222-224, :375-391) for the single-domain case, with REAL_TYPE scalar arithmetic
emulated by numpy scalars of the build precision.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GUIDE = 2  # cz_Define.h:40
FLT_MIN = 1.17549435e-38
EPS = 1.0e-5  # cz.h:162
LC_MAX = 8  # cz_Poisson.cpp:280

_c_int_p = C.POINTER(C.c_int)
_c_dbl_p = C.POINTER(C.c_double)


def lib_path(kind: str, prec: str) -> str:
    if kind == "oracle":
        return os.path.join(_HERE, f"liboracle_{prec}.so")
    if kind == "ref":
        return os.path.join(_HERE, "_ref", f"libczref_{prec}.so")
    if kind == "ref_serial":  # the reference built without OpenMP (pins the PCR line solvers, see oracle/Makefile)
        return os.path.join(_HERE, "_ref", f"libczref_serial_{prec}.so")
    if kind == "ref_sph":  # cz_utility.f90 with -D_aurora_=1: the .sph writer (fileout_t)
        return os.path.join(_HERE, "_ref", f"libczref_sph_{prec}.so")
    raise ValueError(kind)


def have(kind: str, prec: str = "f32") -> bool:
    return os.path.exists(lib_path(kind, prec))


def _ia(v):
    return np.ascontiguousarray(v, dtype=np.int32)


class Kernels:
    """Grid kernels of one back-end/precision.  Arrays are numpy, shape (NJ+4, NI+4, NK+4), C order
    (= the reference's K-fastest Fortran layout)."""

    def __init__(self, kind: str = "oracle", prec: str = "f32"):
        self.kind, self.prec = kind, prec
        self.real = np.float32 if prec == "f32" else np.float64
        self.creal = C.c_float if prec == "f32" else C.c_double
        self.lib = C.CDLL(lib_path(kind, prec))
        self._pre, self._suf = ("oracle_", "") if kind == "oracle" else ("", "_")

    # -- helpers -----------------------------------------------------------------
    def _f(self, name):
        return getattr(self.lib, f"{self._pre}{name}{self._suf}")

    def _rp(self, a):
        assert a.dtype == self.real and a.flags["C_CONTIGUOUS"], (a.dtype, self.real)
        return a.ctypes.data_as(C.c_void_p)

    def _rs(self, v):
        return C.byref(self.creal(float(v)))

    def _ip(self, a):
        return a.ctypes.data_as(_c_int_p)

    def alloc(self, sz):
        """czAllocR_S3D (cz.h:209-232): zero-filled (NJ+4, NI+4, NK+4)."""
        return np.zeros((sz[1] + 2 * GUIDE, sz[0] + 2 * GUIDE, sz[2] + 2 * GUIDE), dtype=self.real)

    # -- kernels -----------------------------------------------------------------
    def bc_k(self, sz, p, dh, org, nID):
        sz, nID, g = _ia(sz), _ia(nID), C.c_int(GUIDE)
        org = np.ascontiguousarray(org, dtype=self.real)
        self._f("bc_k")(self._ip(sz), C.byref(g), self._rp(p), self._rs(dh), self._rp(org), self._ip(nID))

    def jacobi(self, p, sz, idx, cf, omg, b, wk2, res=0.0, wide=None):
        """returns res (+= dble(res1)); wide: optional 1-element float64 array accumulating in double
        (oracle back-end only)."""
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        cf = np.ascontiguousarray(cf, dtype=self.real)
        r, fl = C.c_double(res), C.c_double(0.0)
        args = [self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(cf), self._rs(omg), self._rp(b),
                C.byref(r), self._rp(wk2), C.byref(fl)]
        if wide is not None:
            assert self.kind == "oracle"
            self.lib.oracle_jacobi_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("jacobi")(*args)
        self.last_flop = fl.value
        return r.value

    def psor2sma_core(self, p, sz, idx, cf, ofst, color, omg, b, res=0.0, wide=None):
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        cf = np.ascontiguousarray(cf, dtype=self.real)
        r, fl = C.c_double(res), C.c_double(0.0)
        o, c = C.c_int(ofst), C.c_int(color)
        args = [self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(cf), C.byref(o), C.byref(c),
                self._rs(omg), self._rp(b), C.byref(r), C.byref(fl)]
        if wide is not None:
            assert self.kind == "oracle"
            self.lib.oracle_psor2sma_core_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("psor2sma_core")(*args)
        self.last_flop = fl.value
        return r.value

    def psor(self, p, sz, idx, cf, omg, b, res=0.0, wide=None):
        """lexicographic in-place SOR, one thread (cz_solver.f90:207-269)"""
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        cf = np.ascontiguousarray(cf, dtype=self.real)
        r, fl = C.c_double(res), C.c_double(0.0)
        args = [self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(cf), self._rs(omg), self._rp(b), C.byref(r), C.byref(fl)]
        if wide is not None:
            assert self.kind == "oracle"
            self.lib.oracle_psor_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("psor")(*args)
        self.last_flop = fl.value
        return r.value

    def psor_maf(self, p, sz, idx, x, y, z, omg, b, res=0.0, wide=None):
        """cz_maf.f90:23-112"""
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        r, fl = C.c_double(res), C.c_double(0.0)
        args = [self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(x), self._rp(y), self._rp(z), self._rs(omg), self._rp(b),
                C.byref(r), C.byref(fl)]
        if wide is not None:
            assert self.kind == "oracle"
            self.lib.oracle_psor_maf_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("psor_maf")(*args)
        self.last_flop = fl.value
        return r.value

    def blas_clear(self, x, sz):
        sz, g = _ia(sz), C.c_int(GUIDE)
        self._f("blas_clear")(self._rp(x), self._ip(sz), C.byref(g))

    def blas_copy(self, y, x, sz):
        sz, g = _ia(sz), C.c_int(GUIDE)
        self._f("blas_copy")(self._rp(y), self._rp(x), self._ip(sz), C.byref(g))

    def blas_triad(self, z, x, y, a, sz, idx):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        self._f("blas_triad")(self._rp(z), self._rp(x), self._rp(y), self._rs(a), self._ip(sz), self._ip(idx),
                              C.byref(g), C.byref(fl))

    def blas_dot1(self, p, sz, idx, wide=None):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        r = self.creal(0.0)
        args = [C.byref(r), self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), C.byref(fl)]
        if wide is not None:
            assert self.kind == "oracle"
            self.lib.oracle_blas_dot1_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("blas_dot1")(*args)
        return self.real(r.value)

    def blas_dot2(self, p, q, sz, idx, wide=None):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        r = self.creal(0.0)
        args = [C.byref(r), self._rp(p), self._rp(q), self._ip(sz), self._ip(idx), C.byref(g), C.byref(fl)]
        if wide is not None:
            assert self.kind == "oracle"
            self.lib.oracle_blas_dot2_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("blas_dot2")(*args)
        return self.real(r.value)

    def blas_bicg_1(self, p, r, q, beta, omg, sz, idx):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        self._f("blas_bicg_1")(self._rp(p), self._rp(r), self._rp(q), self._rs(beta), self._rs(omg), self._ip(sz),
                               self._ip(idx), C.byref(g), C.byref(fl))

    def blas_bicg_2(self, z, x, y, a, b, sz, idx):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        self._f("blas_bicg_2")(self._rp(z), self._rp(x), self._rp(y), self._rs(a), self._rs(b), self._ip(sz),
                               self._ip(idx), C.byref(g), C.byref(fl))

    def blas_calc_ax(self, ap, p, sz, idx, cf):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        cf = np.ascontiguousarray(cf, dtype=self.real)
        self._f("blas_calc_ax")(self._rp(ap), self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(cf),
                                C.byref(fl))

    def blas_calc_rk(self, r, p, b, sz, idx, cf):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        cf = np.ascontiguousarray(cf, dtype=self.real)
        self._f("blas_calc_rk")(self._rp(r), self._rp(p), self._rp(b), self._ip(sz), self._ip(idx), C.byref(g),
                                self._rp(cf), C.byref(fl))

    # -- MAF flavour (cz_maf.f90, cz_blas.f90:738-1039); x, y, z: 1-D coordinate arrays of length N+4 (X(-1:N+2))
    def jacobi_maf(self, p, sz, idx, x, y, z, omg, b, wk2, res=0.0, wide=None):
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        r, fl = C.c_double(res), C.c_double(0.0)
        tmp = np.zeros(sz[2] + 4, dtype=self.real)
        args = [self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(x), self._rp(y), self._rp(z), self._rs(omg),
                self._rp(b), C.byref(r), self._rp(wk2), self._rp(tmp), C.byref(fl)]
        if wide is not None:
            self.lib.oracle_jacobi_maf_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("jacobi_maf")(*args)
        self.last_flop = fl.value
        return r.value

    def psor2sma_core_maf(self, p, sz, idx, x, y, z, ofst, color, omg, b, res=0.0, wide=None):
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        r, fl, o, c = C.c_double(res), C.c_double(0.0), C.c_int(ofst), C.c_int(color)
        tmp = np.zeros(sz[2] + 4, dtype=self.real)
        args = [self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(x), self._rp(y), self._rp(z), C.byref(o),
                C.byref(c), self._rs(omg), self._rp(b), C.byref(r), self._rp(tmp), C.byref(fl)]
        if wide is not None:
            self.lib.oracle_psor2sma_core_maf_w(*args, wide.ctypes.data_as(_c_dbl_p))
        else:
            self._f("psor2sma_core_maf")(*args)
        self.last_flop = fl.value
        return r.value

    def calc_rk_maf(self, r, p, b, sz, idx, x, y, z, pvt):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        self._f("calc_rk_maf")(self._rp(r), self._rp(p), self._rp(b), self._ip(sz), self._ip(idx), C.byref(g), self._rp(x),
                               self._rp(y), self._rp(z), self._rp(pvt), C.byref(fl))
        self.last_flop = fl.value

    def calc_ax_maf(self, ap, p, sz, idx, x, y, z, pvt):
        sz, idx, g, fl = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_double(0.0)
        self._f("calc_ax_maf")(self._rp(ap), self._rp(p), self._ip(sz), self._ip(idx), C.byref(g), self._rp(x), self._rp(y),
                               self._rp(z), self._rp(pvt), C.byref(fl))
        self.last_flop = fl.value

    def search_pivot(self, pvt, sz, idx, x, y, z):
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        self._f("search_pivot")(self._rp(pvt), self._ip(sz), self._ip(idx), C.byref(g), self._rp(x), self._rp(y), self._rp(z))

    # -- line SOR by parallel cyclic reduction (cz_solver.f90:497-662), pinned against the SERIAL reference build
    def imask_k(self, x, sz, idx):
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        self._f("imask_k")(self._rp(x), self._ip(sz), self._ip(idx), C.byref(g))

    def pcr_rb(self, sz, idx, pn, ofst, color, x, msk, rhs, omg, res=0.0):
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        w = [np.zeros(sz[2] + 4, dtype=self.real) for _ in range(6)]  # WA, WC, WD, WAA, WCC, WDD (cz_Evaluate.cpp:257-262)
        r, fl = C.c_double(res), C.c_double(0.0)
        pnc, o, c = C.c_int(pn), C.c_int(ofst), C.c_int(color)
        self._f("pcr_rb")(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), C.byref(o), C.byref(c), self._rp(x), self._rp(msk),
                          self._rp(rhs), *[self._rp(v) for v in w], self._rs(omg), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def _pcr_work(self, sz, pn, nwide=3):
        """work arrays of the line-SOR kernels: three of NK+4 entries (cz_Evaluate.cpp:257-262) and three ESA arrays, which
        the reference declares with n + 2*2^(pn-2) entries (cz_Evaluate.cpp:304-311) but indexes further when n < 3/4 * 2^pn:
        they are allocated with room (zeros) for those accesses."""
        nk = int(sz[2]) + 4
        small = [np.zeros(nk + (1 << pn) + 8, dtype=self.real) for _ in range(3)]
        wide = [np.zeros(nk + 4 * (1 << pn) + 8, dtype=self.real) for _ in range(nwide)]
        return small, wide

    def pcr(self, sz, idx, pn, x, msk, rhs, omg, res=0.0):
        """cz_solver.f90:666-878: lexicographic line SOR, pn-2 PCR stages + 4x4 systems"""
        sz, idx, g, pnc = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_int(pn)
        w1, w = self._pcr_work(sz, pn)
        r, fl = C.c_double(res), C.c_double(0.0)
        self._f("pcr")(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), self._rp(x), self._rp(msk), self._rp(rhs),
                       *[self._rp(v) for v in w], *[self._rp(v) for v in w1], self._rs(omg), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def pcr_eda(self, sz, idx, pn, x, msk, rhs, omg, res=0.0):
        """cz_solver.f90:883-1045 (allocates its own extended arrays; defined for n >= 3/4 * 2^pn)"""
        sz, idx, g, pnc = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_int(pn)
        w1, _ = self._pcr_work(sz, pn)
        r, fl = C.c_double(res), C.c_double(0.0)
        self._f("pcr_eda")(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), self._rp(x), self._rp(msk), self._rp(rhs),
                           *[self._rp(v) for v in w1], self._rs(omg), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def pcr_esa(self, sz, idx, pn, x, msk, rhs, omg, res=0.0):
        """cz_solver.f90:1050-1257"""
        sz, idx, g, pnc, ss = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_int(pn), C.c_int((1 << pn) >> 2)
        w1, w = self._pcr_work(sz, pn)
        r, fl = C.c_double(res), C.c_double(0.0)
        self._f("pcr_esa")(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), C.byref(ss), self._rp(x), self._rp(msk), self._rp(rhs),
                           *[self._rp(v) for v in w], *[self._rp(v) for v in w1], self._rs(omg), C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def pcr_rb_esa(self, sz, idx, pn, ofst, color, x, msk, rhs, omg, res=0.0):
        """cz_solver.f90:1261-1469"""
        sz, idx, g, pnc, ss = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_int(pn), C.c_int((1 << pn) >> 2)
        o, c = C.c_int(ofst), C.c_int(color)
        w1, w = self._pcr_work(sz, pn)
        r, fl = C.c_double(res), C.c_double(0.0)
        self._f("pcr_rb_esa")(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), C.byref(o), C.byref(c), C.byref(ss), self._rp(x),
                              self._rp(msk), self._rp(rhs), *[self._rp(v) for v in w], *[self._rp(v) for v in w1], self._rs(omg),
                              C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def pcr_j_esa(self, sz, idx, pn, x, msk, rhs, src, wrk, omg, res=0.0):
        """cz_solver.f90:1473-1676"""
        sz, idx, g, pnc, ss = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_int(pn), C.c_int((1 << pn) >> 2)
        w1, w = self._pcr_work(sz, pn)
        r, fl = C.c_double(res), C.c_double(0.0)
        self._f("pcr_j_esa")(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), C.byref(ss), self._rp(x), self._rp(msk), self._rp(rhs),
                             *[self._rp(v) for v in w], *[self._rp(v) for v in w1], self._rp(src), self._rp(wrk), self._rs(omg),
                             C.byref(r), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def pcr_sweep_wide(self, name, sz, idx, pn, color, x, msk, rhs, wrk, omg, wide, res=0.0):
        """one sweep of a line-SOR variant with sum dp^2 also accumulated in double (oracle only)"""
        assert self.kind == "oracle"
        order, final4 = {"pcr": (0, 1), "pcr_esa": (0, 1), "pcr_eda": (0, 0), "pcr_rb_esa": (1, 1), "pcr_j_esa": (2, 0), "pcr_rb_2x2": (1, 0)}[name]
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        ints = [C.c_int(v) for v in (pn, order, color, final4)]
        r = C.c_double(res)
        self.lib.oracle_pcr_sweep_w(self._ip(sz), self._ip(idx), C.byref(g), *[C.byref(v) for v in ints], self._rp(x), self._rp(msk),
                                    self._rp(rhs), self._rp(wrk) if wrk is not None else None, self._rs(omg), C.byref(r),
                                    wide.ctypes.data_as(_c_dbl_p))
        return r.value

    def pcr_maf(self, name, sz, idx, pn, color, x, msk, rhs, xc, yc, zc, omg, res=0.0, wide=None):
        """the MAF line-SOR kernels, cz_maf.f90:442-1560; name in pcr_rb_maf, pcr_rb_esa_maf, pcr_maf, pcr_eda_maf, pcr_esa_maf"""
        sz, idx, g, pnc = _ia(sz), _ia(idx), C.c_int(GUIDE), C.c_int(pn)
        if wide is not None:
            assert self.kind == "oracle"
            order = C.c_int(1 if "_rb" in name else 0)
            col, r = C.c_int(color), C.c_double(res)
            self.lib.oracle_pcr_maf_sweep_w(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), C.byref(order), C.byref(col), self._rp(x),
                                            self._rp(msk), self._rp(rhs), self._rp(xc), self._rp(yc), self._rp(zc), self._rs(omg), C.byref(r),
                                            wide.ctypes.data_as(_c_dbl_p))
            return r.value
        w1, w = self._pcr_work(sz, pn)
        tmp = np.zeros(int(sz[2]) + 4, dtype=self.real)
        r, fl = C.c_double(res), C.c_double(0.0)
        o, c, ss = C.c_int(0), C.c_int(color), C.c_int((1 << pn) >> 2)
        pre = {"pcr_rb_maf": [o, c], "pcr_rb_esa_maf": [o, c, ss], "pcr_maf": [], "pcr_eda_maf": [], "pcr_esa_maf": [ss]}[name]
        work = [self._rp(v) for v in w1] if name == "pcr_eda_maf" else [self._rp(v) for v in w] + [self._rp(v) for v in w1]
        self._f(name)(self._ip(sz), self._ip(idx), C.byref(g), C.byref(pnc), *[C.byref(v) for v in pre], self._rp(x), self._rp(msk),
                      self._rp(rhs), self._rp(xc), self._rp(yc), self._rp(zc), *work, self._rs(omg), C.byref(r), self._rp(tmp), C.byref(fl))
        self.last_flop = fl.value
        return r.value

    def fileout_t(self, sz, s, dh, org, fname):
        """cz_utility.f90:17-47 (reference build with -D_aurora_=1 only): writes the .sph file `fname` (<= 20 characters)."""
        assert self.kind == "ref_sph" and len(fname) <= 20
        sz, g = _ia(sz), C.c_int(GUIDE)
        org = np.ascontiguousarray(org, dtype=self.real)
        name = fname.encode().ljust(20)
        f = self._f("fileout_t")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]  # hidden length last
        f(self._ip(sz), C.byref(g), self._rp(s), self._rs(dh), self._rp(org), name, 20)

    def exact_t(self, sz, e, dh, org):
        sz, g = _ia(sz), C.c_int(GUIDE)
        org = np.ascontiguousarray(org, dtype=self.real)
        self._f("exact_t")(self._ip(sz), C.byref(g), self._rp(e), self._rs(dh), self._rp(org))

    def err_t(self, sz, idx, p, e):
        sz, idx, g = _ia(sz), _ia(idx), C.c_int(GUIDE)
        d = C.c_double(0.0)
        loc = np.zeros(3, dtype=np.int32)
        self._f("err_t")(self._ip(sz), self._ip(idx), C.byref(g), C.byref(d), self._rp(p), self._rp(e), self._ip(loc))
        return d.value, tuple(int(v) for v in loc)


def sph_bytes(sz, s, dh, org):
    """Restatement of fileout_t (cz_utility.f90:33-44): the bytes of the Fortran sequential unformatted file.
    Records (each framed by its byte length as a 4-byte integer before and after): (1, 1) | (ix, jx, kx) | org(1:3) |
    (dh, dh, dh) | (nn = 0, rtime = 0.0) | s(1:kx, 1:ix, 1:jx) written with i fastest, then j, then k.  Reals have the
    width of the build (4 or 8 bytes), integers are 4 bytes."""
    ni, nj, nk = (int(v) for v in sz)
    R = s.dtype.type
    g = GUIDE

    def rec(*parts):
        body = b"".join(np.ascontiguousarray(p).tobytes() for p in parts)
        n = np.int32(len(body)).tobytes()
        return n + body + n

    core = s[g:g + nj, g:g + ni, g:g + nk]          # [j, i, k]
    data = np.ascontiguousarray(core.transpose(2, 0, 1))  # [k, j, i]: i fastest
    i4 = np.int32
    return (rec(np.array([1, 1], dtype=i4)) + rec(np.array([ni, nj, nk], dtype=i4)) + rec(np.asarray(org, dtype=R)) +
            rec(np.array([dh, dh, dh], dtype=R)) + rec(np.array([0], dtype=i4), np.array([0.0], dtype=R)) + rec(data))


# ----------------------------------------------------------------------------------
# host loops (single domain)
# ----------------------------------------------------------------------------------
def get_num_stage(n):
    """cz.h:293-300: smallest i with n < 2**i."""
    b = 1
    for i in range(1, 20):
        b *= 2
        if n < b:
            return i
    return -1


def range_inner_index(size, nID):
    """cz_miscel.cpp:20-52 -> (innerFidx[6], number of inner points)."""
    ist = jst = kst = 2
    ied, jed, ked = size
    if nID[1] < 0:
        ied = size[0] - 1
    if nID[3] < 0:
        jed = size[1] - 1
    if nID[5] < 0:
        ked = size[2] - 1
    idx = [ist, ied, jst, jed, kst, ked]
    return idx, float(ied - ist + 1) * float(jed - jst + 1) * float(ked - kst + 1)


@dataclass
class Result:
    itr: int
    res: float
    history: list = field(default_factory=list)  # [(itr, res)]
    P: np.ndarray | None = None
    errmax: float | None = None
    errloc: tuple | None = None

    def history_text(self) -> str:
        """the reference's history file (cz_Evaluate.cpp:218, cz_Poisson.cpp:71)."""
        return "Itration      Residual\n" + "".join("%6d, %13.6e\n" % (i, r) for i, r in self.history)


class CZ:
    """Single-domain restatement of class CZ's solve path (cz.h:84-181)."""

    def __init__(self, kernels: Kernels, wide: bool = False):
        """wide=True: residuals and dot products come from the double accumulation of the same REAL-rounded
        terms (oracle back-end only) -- the mode the GPU path is compared with (SURVEY.md 8c tolerance chain)."""
        self.k = kernels
        self.wide = wide
        R = kernels.real
        self.R = R
        self.cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=R)  # cz.h:169-172
        self.eps = EPS
        self.nID = [-1] * 6  # DomainInfo.h:61
        self.history = []

    # cz_Evaluate.cpp:56-99,160-177,222-224,239-288,375-391
    def setup(self, gsz, coef):
        R = self.R
        self.size = [int(v) for v in gsz]
        self.pitch = R(1.0 / float(R(self.size[2] - 1)))  # :88
        self.origin = np.zeros(3, dtype=R)
        self.ac1 = R(float(coef))  # :99
        self.idx, npts = range_inner_index(self.size, self.nID)
        self.res_normal = 1.0 / npts
        k = self.k
        self.P, self.RHS, self.WRK = k.alloc(self.size), k.alloc(self.size), k.alloc(self.size)
        # 1-D grid (cz_Evaluate.cpp:342-363) and pivot array (:369) for the MAF flavours
        self.xc, self.yc, self.zc = (np.array([R(i - 1) * self.pitch for i in range(n + 2 * GUIDE)], dtype=R) for n in self.size)
        self.pvt = k.alloc(self.size)
        k.search_pivot(self.pvt, self.size, self.idx, self.xc, self.yc, self.zc)
        k.bc_k(self.size, self.P, self.pitch, self.origin, self.nID)
        k.bc_k(self.size, self.RHS, self.pitch, self.origin, self.nID)

    # cz_Poisson.cpp:30-82
    def JACOBI(self, X, B, itr_max, converge_check=True, maf=False):
        k, res, itr = self.k, 0.0, 1
        while itr <= itr_max:
            w = np.zeros(1) if self.wide else None
            if maf:  # cz_Poisson.cpp:45-53
                res = k.jacobi_maf(X, self.size, self.idx, self.xc, self.yc, self.zc, self.ac1, B, self.WRK, res=0.0, wide=w)
            else:
                res = k.jacobi(X, self.size, self.idx, self.cf, self.ac1, B, self.WRK, res=0.0, wide=w)
            if self.wide:
                res = float(w[0])
            if converge_check:
                res = math.sqrt(res * self.res_normal)
                self.history.append((itr, res))
                k.bc_k(self.size, X, self.pitch, self.origin, self.nID)
                if res < self.eps:
                    break
            itr += 1
        return itr, res

    # cz_Poisson.cpp:159-235
    def RBSOR(self, X, B, itr_max, converge_check=True, maf=False):
        k, res, itr = self.k, 0.0, 1
        ip = 0  # numProc == 1 (:183-186)
        while itr <= itr_max:
            res = 0.0
            w = np.zeros(1) if self.wide else None
            for color in (0, 1):
                if maf:  # cz_Poisson.cpp:190-200
                    res = k.psor2sma_core_maf(X, self.size, self.idx, self.xc, self.yc, self.zc, ip, color, self.ac1, B, res=res,
                                              wide=w)
                else:
                    res = k.psor2sma_core(X, self.size, self.idx, self.cf, ip, color, self.ac1, B, res=res, wide=w)
            if self.wide:
                res = float(w[0])
            if converge_check:
                res = math.sqrt(res * self.res_normal)
                self.history.append((itr, res))
                k.bc_k(self.size, X, self.pitch, self.origin, self.nID)
                if res < self.eps:
                    break
            itr += 1
        return itr, res

    # cz_Poisson.cpp:95-146
    def PSOR(self, X, B, itr_max, converge_check=True, maf=False):
        k, res, itr = self.k, 0.0, 1
        while itr <= itr_max:
            w = np.zeros(1) if self.wide else None
            if maf:
                res = k.psor_maf(X, self.size, self.idx, self.xc, self.yc, self.zc, self.ac1, B, res=0.0, wide=w)
            else:
                res = k.psor(X, self.size, self.idx, self.cf, self.ac1, B, res=0.0, wide=w)
            if self.wide:
                res = float(w[0])
            if converge_check:
                res = math.sqrt(res * self.res_normal)
                self.history.append((itr, res))
                k.bc_k(self.size, X, self.pitch, self.origin, self.nID)
                if res < self.eps:
                    break
            itr += 1
        return itr, res

    # cz_Poisson.cpp:518-611
    def LSOR_PCR_RB(self, X, B, itr_max, converge_check=True):
        k, res, itr = self.k, 0.0, 1
        n = self.idx[5] - self.idx[4] + 1
        pn = get_num_stage(n)
        if not hasattr(self, "MSK"):
            self.MSK = k.alloc(self.size)
            k.imask_k(self.MSK, self.size, self.idx)  # cz_Evaluate.cpp:389
        while itr <= itr_max:
            res = 0.0
            for color in (0, 1):
                res = k.pcr_rb(self.size, self.idx, pn, 0, color, X, self.MSK, B, self.ac1, res=res)
            if converge_check:
                res = math.sqrt(res * self.res_normal)
                self.history.append((itr, res))
                k.bc_k(self.size, X, self.pitch, self.origin, self.nID)
                if res < self.eps:
                    break
            itr += 1
        return itr, res

    # cz_Poisson.cpp:273-322
    # cz_Poisson.cpp:621-742 (pcr_rb_esa), :745-826 (pcr), :910-1005 (pcr_esa), :1008-1095 (pcr_j_esa)
    def LSOR_PCR_VARIANT(self, X, B, itr_max, name, converge_check=True):
        k, res, itr = self.k, 0.0, 1
        pn = get_num_stage(self.idx[5] - self.idx[4] + 1)
        if not hasattr(self, "MSK"):
            self.MSK = k.alloc(self.size)
            k.imask_k(self.MSK, self.size, self.idx)  # cz_Evaluate.cpp:389
        while itr <= itr_max:
            res = 0.0
            if self.wide:
                w = np.zeros(1)
                for color in ((0, 1) if name == "pcr_rb_esa" else (0,)):
                    k.pcr_sweep_wide(name, self.size, self.idx, pn, color, X, self.MSK, B, self.WRK, self.ac1, w)
                res = float(w[0])
            elif name == "pcr":
                res = k.pcr(self.size, self.idx, pn, X, self.MSK, B, self.ac1)
            elif name == "pcr_esa":
                res = k.pcr_esa(self.size, self.idx, pn, X, self.MSK, B, self.ac1)
            elif name == "pcr_eda":
                res = k.pcr_eda(self.size, self.idx, pn, X, self.MSK, B, self.ac1)
            elif name == "pcr_rb_esa":
                for color in (0, 1):
                    res = k.pcr_rb_esa(self.size, self.idx, pn, 0, color, X, self.MSK, B, self.ac1, res=res)
            elif name == "pcr_j_esa":
                if not hasattr(self, "SRC"):
                    self.SRC = k.alloc(self.size)
                res = k.pcr_j_esa(self.size, self.idx, pn, X, self.MSK, B, self.SRC, self.WRK, self.ac1)
            else:
                raise ValueError(name)
            if converge_check:
                res = math.sqrt(res * self.res_normal)
                self.history.append((itr, res))
                k.bc_k(self.size, X, self.pitch, self.origin, self.nID)
                if res < self.eps:
                    break
            itr += 1
        return itr, res

    # cz_Poisson.cpp: the MAF branches of LSOR_PCR_RB (:549-557), LSOR_PCR_RB_ESA (:665-683), LSOR_PCR (:770-776), LSOR_PCR_EDA, LSOR_PCR_ESA
    def LSOR_PCR_MAF(self, X, B, itr_max, name, converge_check=True):
        k, res, itr = self.k, 0.0, 1
        pn = get_num_stage(self.idx[5] - self.idx[4] + 1)
        if not hasattr(self, "MSK"):
            self.MSK = k.alloc(self.size)
            k.imask_k(self.MSK, self.size, self.idx)
        while itr <= itr_max:
            res = 0.0
            w = np.zeros(1) if self.wide else None
            for color in ((0, 1) if "_rb" in name else (0,)):
                res = k.pcr_maf(name, self.size, self.idx, pn, color, X, self.MSK, B, self.xc, self.yc, self.zc, self.ac1, res=res, wide=w)
            if self.wide:
                res = float(w[0])
            if converge_check:
                res = math.sqrt(res * self.res_normal)
                self.history.append((itr, res))
                k.bc_k(self.size, X, self.pitch, self.origin, self.nID)
                if res < self.eps:
                    break
            itr += 1
        return itr, res

    def Preconditioner(self, xx, bb, pc):
        if pc in ("jacobi", "jacobi_maf"):
            self.JACOBI(xx, bb, LC_MAX, converge_check=False, maf=pc.endswith("_maf"))
        elif pc in ("sor2sma", "sor2sma_maf"):
            self.RBSOR(xx, bb, LC_MAX, converge_check=False, maf=pc.endswith("_maf"))
        elif pc in ("psor", "psor_maf"):
            self.PSOR(xx, bb, LC_MAX, converge_check=False, maf=pc.endswith("_maf"))
        elif pc == "pcr_rb":
            self.LSOR_PCR_RB(xx, bb, LC_MAX, converge_check=False)
        elif pc in ("pcr_rb_maf", "pcr_rb_esa_maf", "pcr_maf", "pcr_eda_maf"):
            self.LSOR_PCR_MAF(xx, bb, LC_MAX, pc, converge_check=False)
        elif pc in ("pcr", "pcr_rb_esa", "pcr_eda"):  # cz_Poisson.cpp:300-316, cz_Evaluate.cpp:585-592
            self.LSOR_PCR_VARIANT(xx, bb, LC_MAX, pc, converge_check=False)
        # "pcr_j_esa" is accepted by cz_Evaluate.cpp:588-590 but CZ::Preconditioner has no case for it: it copies
        else:
            self.k.blas_copy(xx, bb, self.size)

    # cz_Poisson.cpp:332-504
    def PBiCGSTAB(self, X, B, ItrMax, pc, maf=False):
        k, R, sz, idx = self.k, self.R, self.size, self.idx

        def calc_ax(ap, p):  # cz_Poisson.cpp:415-422
            if maf:
                k.calc_ax_maf(ap, p, sz, idx, self.xc, self.yc, self.zc, self.pvt)
            else:
                k.blas_calc_ax(ap, p, sz, idx, self.cf)

        if self.wide:
            def dot1(x):
                w = np.zeros(1)
                k.blas_dot1(x, sz, idx, wide=w)
                return R(w[0])

            def dot2(x, y):
                w = np.zeros(1)
                k.blas_dot2(x, y, sz, idx, wide=w)
                return R(w[0])
        else:
            def dot1(x):
                return k.blas_dot1(x, sz, idx)

            def dot2(x, y):
                return k.blas_dot2(x, y, sz, idx)
        a = {n: k.alloc(sz) for n in ("p", "p_", "r", "r0", "q", "s", "s_", "t_")}
        res = 0.0
        k.blas_clear(a["q"], sz)
        if maf:  # cz_Poisson.cpp:350-353
            k.calc_rk_maf(a["r"], X, B, sz, idx, self.xc, self.yc, self.zc, self.pvt)
        else:
            k.blas_calc_rk(a["r"], X, B, sz, idx, self.cf)
        k.blas_copy(a["r0"], a["r"], sz)
        rho_old, alpha, omega = R(1.0), R(0.0), R(1.0)
        itr = 1
        while itr < ItrMax:  # strict '<' (:373)
            rho = dot2(a["r"], a["r0"])
            if abs(float(rho)) < FLT_MIN:
                itr = 0
                break
            if itr == 1:
                k.blas_copy(a["p"], a["r"], sz)
            else:
                beta = R(R(R(rho / rho_old) * alpha) / omega)  # :394
                k.blas_bicg_1(a["p"], a["r"], a["q"], beta, omega, sz, idx)
            k.blas_clear(a["p_"], sz)
            self.Preconditioner(a["p_"], a["p"], pc)
            calc_ax(a["q"], a["p_"])
            alpha = R(rho / dot2(a["q"], a["r0"]))  # :427
            k.blas_triad(a["s"], a["q"], a["r"], R(-alpha), sz, idx)
            k.blas_clear(a["s_"], sz)
            self.Preconditioner(a["s_"], a["s"], pc)
            calc_ax(a["t_"], a["s_"])
            omega = R(dot2(a["t_"], a["s"]) / dot1(a["t_"]))  # :464
            k.blas_bicg_2(X, a["p_"], a["s_"], alpha, omega, sz, idx)
            k.blas_triad(a["r"], a["t_"], a["s"], R(-omega), sz, idx)
            res = float(dot1(a["r"]))
            res = math.sqrt(res * self.res_normal)
            self.history.append((itr, res))
            k.bc_k(sz, X, self.pitch, self.origin, self.nID)
            if res < self.eps:
                break
            rho_old = rho
            itr += 1
        return itr, res

    def error_max(self):
        """debug epilogue, cz_Evaluate.cpp:550-563."""
        # the serial reference build rounds exact_t's sinh/sin one ulp differently from the OpenMP build (other vector
        # math path); the OpenMP build is the reference binary, so its exact_t is the one fixtures hold
        k = Kernels("ref", self.k.prec) if self.k.kind == "ref_serial" else self.k
        e = k.alloc(self.size)
        k.exact_t(self.size, e, self.pitch, self.origin)
        return k.err_t(self.size, self.idx, self.P, e)


def run(gsz, solver, itr_max, coef, precond=None, kind="oracle", prec="f32", with_error=False, wide=False) -> Result:
    """``cz gsz_x gsz_y gsz_z solver ItrMax coef [precond]`` on the chosen back-end, one thread semantics."""
    cz = CZ(Kernels(kind, prec), wide=wide)
    cz.setup(gsz, coef)
    if solver in ("jacobi", "jacobi_maf"):
        itr, res = cz.JACOBI(cz.P, cz.RHS, itr_max, maf=solver.endswith("_maf"))
    elif solver in ("sor2sma", "sor2sma_maf"):
        itr, res = cz.RBSOR(cz.P, cz.RHS, itr_max, maf=solver.endswith("_maf"))
    elif solver in ("psor", "psor_maf"):
        itr, res = cz.PSOR(cz.P, cz.RHS, itr_max, maf=solver.endswith("_maf"))
    elif solver == "pcr_rb":
        itr, res = cz.LSOR_PCR_RB(cz.P, cz.RHS, itr_max)
    elif solver in ("pcr_rb_maf", "pcr_rb_esa_maf", "pcr_maf", "pcr_eda_maf", "pcr_esa_maf"):
        itr, res = cz.LSOR_PCR_MAF(cz.P, cz.RHS, itr_max, solver)
    elif solver in ("pcr", "pcr_esa", "pcr_eda", "pcr_rb_esa", "pcr_j_esa"):
        itr, res = cz.LSOR_PCR_VARIANT(cz.P, cz.RHS, itr_max, solver)
    elif solver in ("pbicgstab", "pbicgstab_maf"):
        itr, res = cz.PBiCGSTAB(cz.P, cz.RHS, itr_max, precond or "none", maf=solver.endswith("_maf"))
    else:
        raise ValueError(solver)
    out = Result(itr=itr, res=res, history=cz.history, P=cz.P)
    if with_error:
        out.errmax, out.errloc = cz.error_max()
    return out
