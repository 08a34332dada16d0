#!/usr/bin/env python3
"""Level-1 integration cost: the synchronous drop-in jacobi_ / psor2sma_core_ symbols (reference semantics:
result in p AND wk2, host residual) against the device-resident loops of the driver."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from cubez_amd import CzHip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = CzHip("f32")
sz = [N, N, N]; idx = [2, N - 1, 2, N - 1, 2, N - 1]
host = np.random.default_rng(0).uniform(-1, 1, (N + 4, N + 4, N + 4)).astype(np.float32)
p, wk, b = h.alloc(sz, host), h.alloc(sz, host), h.alloc(sz)
cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=np.float32)
lups = float(N - 2) ** 3
for _ in range(3): h.jacobi(p, sz, idx, cf, 0.8, b, wk)
t0 = time.perf_counter()
for _ in range(20): h.jacobi(p, sz, idx, cf, 0.8, b, wk)
dt = (time.perf_counter() - t0) / 20
print("drop-in jacobi_        : %.4f ms/call  %.0f MLUPS (sweep + copy-back + host residual, 20 B/LUP)" % (dt * 1e3, lups / dt / 1e6))
t0 = time.perf_counter()
for _ in range(20):
    r = 0.0
    for c in (0, 1): r = h.psor2sma_core(p, sz, idx, cf, 0, c, 1.5, b, res=r)
dt = (time.perf_counter() - t0) / 20
print("drop-in psor2sma_core_ : %.4f ms/iteration (2 colour calls)  %.0f MLUPS" % (dt * 1e3, lups / dt / 1e6))
