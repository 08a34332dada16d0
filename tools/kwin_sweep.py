"""k windows of the two-stage pass (Geom2, cz_k_pair.h): MLUPS of the fused Jacobi pass / red-black iteration by window length, per box.
    python3 tools/kwin_sweep.py [quick]
kwin: 0 = whole rows wherever they fit (rounds 1-3), -1 = the launcher's rule, n = windows of n vectors."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubez_amd import CZ  # noqa: E402

quick = "quick" in sys.argv
KW = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or [0, -1, 32, 43, 64, 96, 128]
cases = [("f32", (512, 512, 512)), ("f32", (256, 256, 256)), ("f32", (384, 384, 384)), ("f32", (512, 512, 1020)), ("f32", (384, 384, 2100)),
         ("f32", (1024, 1024, 1024)), ("f64", (512, 512, 512)), ("f64", (384, 384, 1100)), ("f64", (1024, 1024, 1024))]
if quick:
    cases = [c for c in cases if max(c[1]) <= 1100 or c[0] == "f32" and c[1][2] == 2100]
for prec, gsz in cases:
    for solver, coef in (("jacobi", 0.8), ("sor2sma", 1.5)):
        if solver == "sor2sma" and (prec == "f64" or gsz[0] >= 1024):
            continue
        row = []
        for kwin in KW:
            cz = CZ(prec, quiet=True)
            cz.lib.czhip_set_pair_window(kwin)
            assert cz.setup(list(gsz) + [solver, 100000, coef]) == 1
            cz.sweeps(20)
            cz.lib.czhip_sync()
            cz.timing(True)
            best = 1e9
            nst = 60 if gsz[0] * gsz[1] * gsz[2] <= 600 ** 3 else 20
            for rep in range(3):
                t0 = time.perf_counter()
                cz.sweeps(nst)
                cz.lib.czhip_sync()
                best = min(best, (time.perf_counter() - t0) / nst)
            nk, ms = cz.timing_read("jacobi2" if solver == "jacobi" else "rbsor2")
            n1, ms1 = cz.timing_read("jacobi" if solver == "jacobi" else "rbsor")
            cz.timing(False)
            info = cz.info()
            cz.lib.czhip_set_pair_window(-1)
            cz.close()
            pts = (gsz[0] - 2) * (gsz[1] - 2) * (gsz[2] - 2)
            row.append(f"kwin {kwin:4d}: {pts / best / 1e6:8.0f} MLUPS ({'pass %.3f ms' % (ms / nk) if nk else 'single sweeps'})")
        print(f"{prec} {'x'.join(map(str, gsz)):>15s} {solver:8s} | " + " | ".join(row), flush=True)
