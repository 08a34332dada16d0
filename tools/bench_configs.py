#!/usr/bin/env python3
"""Time the other BASELINE.json configs on one GPU (they are parity cases, not bench lines): RB-SOR 512^3 FP32 and
BiCGSTAB+Jacobi 512^3 FP64, plus FP64 Jacobi.  Prints MLUPS / seconds per iteration and algorithmic GB/s."""
import os, sys, time, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from cubez_amd import CZ
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lups = float(N - 2) ** 3
out = {}
for solver, prec, coef, bpl in (("jacobi", "f32", 0.8, 12), ("sor2sma", "f32", 1.5, 16), ("jacobi", "f64", 0.8, 24), ("sor2sma", "f64", 1.5, 32),
                                ("jacobi_maf", "f32", 0.8, 12), ("sor2sma_maf", "f32", 1.5, 16), ("jacobi_maf", "f64", 0.8, 24)):
    cz = CZ(prec, quiet=True)
    assert cz.setup([N, N, N, solver, 1000, coef]) == 1
    cz.sweeps(10); cz.lib.czhip_sync()
    t0 = time.perf_counter(); cz.sweeps(100); cz.lib.czhip_sync(); dt = (time.perf_counter() - t0) / 100
    out[f"{solver}_{prec}"] = dict(ms_per_iter=dt * 1e3, mlups=lups / dt / 1e6, alg_GBs=lups * bpl / dt / 1e9)
    print(solver, prec, "%.4f ms/iter %.0f MLUPS alg %.0f GB/s" % (dt * 1e3, lups / dt / 1e6, lups * bpl / dt / 1e9), flush=True)
    cz.close()
for pc, coef in (("jacobi", 0.8), ("sor2sma", 1.5)):
    cz = CZ("f64", quiet=True)
    assert cz.setup([N, N, N, "pbicgstab", 12, coef, pc]) == 1   # 11 iterations
    cz.lib.czhip_sync(); t0 = time.perf_counter(); it = cz.solve(); dt = time.perf_counter() - t0
    nit = len(cz.history())
    # algorithmic bytes per iteration: 76 words per point with the Jacobi preconditioner
    words = 76 if pc == "jacobi" else 16 * 4 + 28
    out[f"pbicgstab_{pc}_f64"] = dict(s_per_iter=dt / nit, iters=nit, alg_GBs=lups * words * 8 / (dt / nit) / 1e9)
    print("pbicgstab", pc, "f64: %d its, %.4f s/iter, alg %.0f GB/s, last res %.3e" % (nit, dt / nit, lups * words * 8 / (dt / nit) / 1e9, cz.res), flush=True)
    cz.close()
print(json.dumps(out))
