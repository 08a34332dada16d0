#!/usr/bin/env python3
"""Sweep the (threads, vectors/thread, planes/chunk) variants of the two-sweep kernel through the driver's bench leg."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from cubez_amd import CZ
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
solver = sys.argv[3] if len(sys.argv) > 3 else "jacobi"
cz = CZ(prec, quiet=True)
assert cz.setup([N, N, N, solver, 1000, 0.8 if solver == "jacobi" else 1.5]) == 1
lib = cz.lib
lups = float(N - 2) ** 3
rows = []
cfgs = [(0, 0, 0, 0)] + [(tb, mv, tj, 1) for (tb, mv) in ((512, 2), (1024, 2), (512, 3), (256, 4)) for tj in (8, 16, 32, 64)]
for rnd in range(2):
    for (tb, mv, tj, en) in cfgs:
        if lib.czhip_set_tuning2(tb, mv, tj, en) != 0:
            continue
        cz.sweeps(4)
        lib.czhip_sync()
        t0 = time.perf_counter()
        cz.sweeps(40)
        lib.czhip_sync()
        dt = (time.perf_counter() - t0) / 40
        rows.append((dt, (tb, mv, tj, en)))
best = {}
for dt, k in rows:
    best[k] = min(best.get(k, 1e9), dt)
for k, dt in sorted(best.items(), key=lambda kv: kv[1]):
    print("%8.4f ms/sweep %9.0f MLUPS  alg %6.0f GB/s  %s" % (dt * 1e3, lups / dt / 1e6, lups * ((12 if solver == "jacobi" else 16) * (1 if prec == "f32" else 2)) / dt / 1e9, k))
