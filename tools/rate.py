"""Rates of arbitrary runs of the restated driver, one line each:
    python3 tools/rate.py prec:nx,ny,nz:solver[:precond][@ENV=VALUE,...] ...
stationary solvers: MLUPS over unchecked sweeps (cz_sweeps) + mean kernel time by label; pbicgstab: ms per iteration of a 10-iteration solve."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubez_amd import CZ  # noqa: E402

COEF = {"jacobi": 0.8, "jacobi_maf": 0.8, "sor2sma": 1.5, "sor2sma_maf": 1.5, "pcr_j_esa": 0.9}
for spec in sys.argv[1:]:
    env = {}
    if "@" in spec:
        spec, e = spec.split("@", 1)
        env = dict(kv.split("=", 1) for kv in e.split(";"))  # several: @A=1;B=2
    to_conv = spec.endswith("!solve")  # a whole solve to convergence instead of unchecked sweeps
    if to_conv:
        spec = spec[:-6]
    parts = spec.split(":")
    prec, gsz, solver = parts[0], [int(v) for v in parts[1].split(",")], parts[2]
    pc = parts[3] if len(parts) > 3 else None
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cz = CZ(prec, quiet=True)
        # (the kernel context reads the environment once per thread: the kernel switches go through the API)
        if env.get("CZHIP_PCR"):
            f, v = (env["CZHIP_PCR"].split(",") + ["0"])[:2]
            cz.lib.czhip_set_pcr_mode(int(f), int(v))
        if env.get("CZHIP_T2"):
            a = [int(v) for v in env["CZHIP_T2"].split(",")] + [0, 0, 0]
            assert cz.lib.czhip_set_tuning2(a[1] if a[1] else -2, a[2], a[3], a[0]) == 0
        if env.get("CZHIP_PSOR"):
            a = [int(v) for v in env["CZHIP_PSOR"].split(",")] + [0, 0]  # one_launch, workgroups per CU, ask-ahead steps
            cz.lib.czhip_set_psor(a[0], a[1])
            cz.lib.czhip_set_psor_ahead(a[2])
        if "CZHIP_T2_PRE" in env:
            cz.lib.czhip_set_pair_preload(int(env["CZHIP_T2_PRE"]))
        if "CZHIP_T2_KWIN" in env:
            cz.lib.czhip_set_pair_window(int(env["CZHIP_T2_KWIN"]))
        if "CZHIP_RB4" in env:
            a = [int(v) for v in env["CZHIP_RB4"].split(",")] + [0, 0]
            cz.lib.czhip_set_rb4(a[0], a[1], a[2])
        if "CZHIP_UNIT_COEF" in env:
            cz.lib.czhip_set_unit_coef(int(env["CZHIP_UNIT_COEF"]))
        pts = (gsz[0] - 2) * (gsz[1] - 2) * (gsz[2] - 2)
        if solver.startswith("pbicgstab"):
            assert cz.setup(gsz + [solver, 3, COEF.get(pc, 0.8), pc or "jacobi"]) == 1
            cz.solve()
            cz.close()
            cz = CZ(prec, quiet=True)
            assert cz.setup(gsz + [solver, 11, COEF.get(pc, 0.8), pc or "jacobi"]) == 1
            cz.lib.czhip_sync()
            t0 = time.perf_counter()
            cz.solve()
            dt = time.perf_counter() - t0
            n = len(cz.history())
            print(f"{spec:48s} {env} {dt / n * 1e3:9.3f} ms per iteration ({n} iterations)", flush=True)
        elif to_conv:
            best = 1e9
            for rep in range(3):
                assert cz.setup(gsz + [solver, 100000, COEF.get(solver, 1.2)]) == 1
                cz.lib.czhip_sync()
                t0 = time.perf_counter()
                itr = cz.solve()
                best = min(best, time.perf_counter() - t0)
                if rep < 2:
                    cz.close()
                    cz = CZ(prec, quiet=True)
            print(f"{spec:48s} {env} to convergence: {itr} iterations in {best * 1e3:8.2f} ms ({best / itr * 1e6:.2f} us per iteration)", flush=True)
        else:
            assert cz.setup(gsz + [solver, 100000, COEF.get(solver, 1.2)]) == 1
            nst = max(4, min(100, int(4e9 / pts)))
            cz.sweeps(max(2, nst // 5))
            cz.lib.czhip_sync()
            cz.timing("notime" not in env)  # (HIP events around every launch cost microseconds: @notime=1 for small grids)
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                cz.sweeps(nst)
                cz.lib.czhip_sync()
                best = min(best, (time.perf_counter() - t0) / nst)
            lab = []
            for lb in ("jacobi", "rbsor", "jacobi2", "rbsor2", "pcr_rb", "psor"):
                nk, ms = cz.timing_read(lb)
                if nk:
                    lab.append(f"{lb} {ms / nk:.4f} ms x {nk / (3 * nst):.2f}/sweep")
            cz.timing(False)
            print(f"{spec:48s} {env} {pts / best / 1e6:9.0f} MLUPS  {best * 1e3:8.4f} ms per sweep  [{'; '.join(lab)}]", flush=True)
        if env.get("CZHIP_PCR"):
            cz.lib.czhip_set_pcr_mode(2, 0)
        if env.get("CZHIP_T2"):
            cz.lib.czhip_set_tuning2(-2, 2, 0, 1)
        cz.lib.czhip_set_pair_preload(1)
        cz.lib.czhip_set_pair_window(-1)
        cz.lib.czhip_set_unit_coef(1)
        cz.lib.czhip_set_rb4(1, 0, 0)
        cz.lib.czhip_set_psor(1, 0)
        cz.lib.czhip_set_psor_ahead(0)
        cz.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
