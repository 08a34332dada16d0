#!/usr/bin/env python3
"""Sweep the compiled (threads, vectors/thread, planes/chunk, prefetch) variants of the Jacobi / RB-SOR sweep on
one GPU and print ms/sweep, MLUPS and algorithmic GB/s (12 B/LUP FP32 Jacobi).  Interleaved rounds in
one process (cdna_hip_programming.md rule 24)."""
import argparse
import ctypes as C
import json
import sys
import time
import os

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np  # noqa: E402

from cubez_amd import CzHip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--prec", default="f32")
ap.add_argument("--sweeps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--mode", default="jacobi", choices=["jacobi", "rbsor"])
ap.add_argument("--tunings", default="")
args = ap.parse_args()

h = CzHip(args.prec)
lib = h.lib
N = args.n
sz = [N, N, N]
idx = (C.c_int * 6)(2, N - 1, 2, N - 1, 2, N - 1)
csz = (C.c_int * 3)(*sz)
R = h.real
rng = np.random.default_rng(0)
shape = (N + 4, N + 4, N + 4)
host = rng.uniform(-1, 1, shape).astype(R)
A, B2, RHS = h.alloc(sz, host), h.alloc(sz, host), h.alloc(sz)
del host
cf = (h.creal * 7)(1, 1, 1, 1, 1, 1, 6)
lib.czhip_alloc_s3d.restype = C.c_void_p
res = C.c_void_p(lib.czhip_alloc_s3d((C.c_int * 3)(1, 1, 1)))
lib.czhip_jacobi_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, h.creal,
                                   C.c_void_p, C.c_int, C.c_void_p]
lib.czhip_rbsor_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                  h.creal, C.c_void_p, C.c_int, C.c_void_p]


def run(nsweep):
    bufs = [A.ptr, B2.ptr]
    for s in range(nsweep):
        if args.mode == "jacobi":
            lib.czhip_jacobi_async(bufs[s & 1], bufs[(s + 1) & 1], RHS.ptr, csz, idx, 2, cf, 0.8, res, 0, None)
        else:
            for color in (0, 1):
                lib.czhip_rbsor_async(A.ptr, RHS.ptr, csz, idx, 2, cf, 0, color, 1.5, res, color, None)
    lib.czhip_sync()


if args.tunings:
    tunings = [tuple(int(v) for v in t.split(",")) for t in args.tunings.split(";")]
else:
    tunings = [(tb, m, tj, pf) for tb in (256, 512, 1024) for m in (1, 2, 4) if not (tb == 1024 and m == 4)
               for tj in (0, 16, 64) for pf in (0, 1)]
lups = float(N - 2) ** 3
bytes_per_lup = (12 if args.mode == "jacobi" else 16) * (1 if args.prec == "f32" else 2)
best = {}
for rnd in range(args.rounds):
    for tu in tunings:
        if not h.set_tuning(*tu):
            continue
        run(2)
        t0 = time.perf_counter()
        run(args.sweeps)
        dt = (time.perf_counter() - t0) / args.sweeps
        best.setdefault(tu, []).append(dt)
rows = []
for tu, v in best.items():
    med = sorted(v)[len(v) // 2]
    rows.append((med, tu, min(v)))
rows.sort()
print(f"# {args.mode} {N}^3 {args.prec}: ms/sweep(median)  ms(min)  MLUPS  alg-GB/s  tuning(threads,m,tj,pf)")
for med, tu, mn in rows:
    print("%8.4f %8.4f %10.0f %8.0f  %s" % (med * 1e3, mn * 1e3, lups / med / 1e6, lups * bytes_per_lup / med / 1e9, tu))
print(json.dumps({"best": rows[0][1], "ms": rows[0][0] * 1e3}))
