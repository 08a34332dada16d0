#!/usr/bin/env python3
"""Experiment: temporal blocking of the Jacobi sweep through the 256 MiB Infinity Cache.

Two sweeps are applied slab by slab (W planes of j at a time): sweep n -> a small scratch ring that stays cache
resident, sweep n+1 from the scratch to the output array, so that per PAIR of sweeps HBM sees one read of p, one of b
and one write.  Uses only the existing single-sweep kernel (index ranges + offset pointers).  Timing only (the
residual of the redundant planes is double counted here); prints ms per sweep for each slab width."""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np  # noqa: E402

from cubez_amd import CzHip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--pairs", type=int, default=10)
ap.add_argument("--widths", default="16,32,48,64")
ap.add_argument("--tunings", default="512,2,16,0;512,2,8,0;512,2,4,0;256,2,8,0;256,2,4,0;512,1,8,0;512,1,4,0")
args = ap.parse_args()

h = CzHip("f32")
lib = h.lib
N = args.n
sz = [N, N, N]
csz = (C.c_int * 3)(*sz)
PS = (N + 4) * (N + 4)  # elements per plane
host = np.random.default_rng(0).uniform(-1, 1, (N + 4, N + 4, N + 4)).astype(np.float32)
U, Wout, RHS = h.alloc(sz, host), h.alloc(sz, host), h.alloc(sz)
del host
cf = (C.c_float * 7)(1, 1, 1, 1, 1, 1, 6)
lib.czhip_alloc_s3d.restype = C.c_void_p
res = C.c_void_p(lib.czhip_alloc_s3d((C.c_int * 3)(1, 1, 1)))
lib.czhip_jacobi_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                   C.c_void_p, C.c_int, C.c_void_p]


def idx(j0, j1):
    return (C.c_int * 6)(2, N - 1, max(j0, 2), min(j1, N - 1), 2, N - 1)


def plain(npairs):
    bufs = [U.ptr, Wout.ptr]
    for s in range(2 * npairs):
        lib.czhip_jacobi_async(bufs[s & 1], bufs[(s + 1) & 1], RHS.ptr, csz, idx(2, N - 1), 2, cf, 0.8, res, 0, None)
    lib.czhip_sync()


def slabbed(npairs, W, scratch):
    bufs = [U.ptr, Wout.ptr]
    for p in range(npairs):
        src, dst = bufs[p & 1], bufs[(p + 1) & 1]
        a = 2
        while a <= N - 1:
            b = min(a + W - 1, N - 1)
            # scratch slot 0 <-> padded plane index of 1-based plane (a-2): offset the base pointer accordingly
            off = (a - 2 + 1) * PS * 4  # 1-based plane j sits at padded index j+1
            seff = scratch - off
            lib.czhip_jacobi_async(src, seff, RHS.ptr, csz, idx(a - 1, b + 1), 2, cf, 0.8, res, 0, None)
            lib.czhip_jacobi_async(seff, dst, RHS.ptr, csz, idx(a, b), 2, cf, 0.8, res, 1, None)
            a = b + 1
    lib.czhip_sync()


lups = float(N - 2) ** 3
for tu in args.tunings.split(";"):
    tb, m, tj, pf = (int(v) for v in tu.split(","))
    assert h.set_tuning(tb, m, tj, pf)
    plain(2)
    t0 = time.perf_counter()
    plain(args.pairs)
    t_plain = (time.perf_counter() - t0) / (2 * args.pairs)
    line = f"tuning {tu:14s} plain {t_plain*1e3:7.4f} ms/sweep {lups/t_plain/1e6:9.0f} MLUPS |"
    for W in (int(w) for w in args.widths.split(",")):
        nbytes = (W + 6) * PS * 4
        lib.czhip_alloc_s3d.restype = C.c_void_p
        import ctypes
        # raw scratch allocation through the S3D allocator: enough planes
        planes = W + 6
        scr = h.alloc([N, planes - 4, N])
        slabbed(1, W, scr.ptr)
        t0 = time.perf_counter()
        slabbed(args.pairs, W, scr.ptr)
        t = (time.perf_counter() - t0) / (2 * args.pairs)
        line += f" W={W}: {t*1e3:7.4f} ms {lups/t/1e6:8.0f} |"
        scr.free()
    print(line, flush=True)
