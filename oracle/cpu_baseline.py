#!/usr/bin/env python3
"""Time the CPU path beside the GPU numbers (bench.py's cpu_baseline leg; TEST/BENCH INFRASTRUCTURE).

kind "reference": oracle/_ref/libczref_f32.so -- the reference's own jacobi_/psor2sma_core_ (Fortran + OpenMP,
                  cz_solver.f90:284-493) compiled by oracle/Makefile -- run multi-threaded (its Jacobi then has the
                  NOWAIT race of cz_solver.f90:355, irrelevant for timing: same work per sweep).
kind "port":      oracle/liboracle_f32.so (the C restatement) when the reference library is not present.

Run as a subprocess so that OMP_NUM_THREADS / OMP_PROC_BIND take effect before libomp starts.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--solver", default="jacobi")
ap.add_argument("--prec", default="f32")
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("--threads", type=int, default=0)
args = ap.parse_args()

# default: the CPU share of a one-GPU box (16) or what the affinity mask allows, whichever is smaller
cores = args.threads or min(16, len(os.sched_getaffinity(0)))
os.environ["OMP_NUM_THREADS"] = str(cores)
os.environ.setdefault("OMP_PROC_BIND", "close")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import cz_oracle as O  # noqa: E402

kind = "ref" if O.have("ref", args.prec) else "oracle"
cz = O.CZ(O.Kernels(kind, args.prec))
N = args.n
cz.setup((N, N, N), 0.8 if args.solver == "jacobi" else 1.5)
k = cz.k


def one():
    if args.solver == "jacobi":
        k.jacobi(cz.P, cz.size, cz.idx, cz.cf, cz.ac1, cz.RHS, cz.WRK)
    else:
        for color in (0, 1):
            k.psor2sma_core(cz.P, cz.size, cz.idx, cz.cf, 0, color, cz.ac1, cz.RHS)


one()  # warm-up (first touch)
t0 = time.perf_counter()
n = 0
while True:
    one()
    n += 1
    dt = time.perf_counter() - t0
    if dt >= args.seconds or n >= 1000:
        break
lups = float(N - 2) ** 3 * n
cpu_model = "unknown CPU"
try:
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu_model = line.split(":", 1)[1].strip()
            break
except OSError:
    pass
print(json.dumps({"value": lups / dt / 1e6, "unit": "MLUPS", "cores": cores, "kind": "reference" if kind == "ref" else "port",
                  "sample": f"{n} {args.solver} sweeps of the {N}^3 {args.prec} grid in {dt:.1f} s, OMP_NUM_THREADS={cores}, {cpu_model}"}))
