"""What does the CU reservation of decomposed runs (CZ_COMM_CUS = k CUs per XCD left to the exchange stream) cost the sweeps?
One rank, 512^3: ms per launch of the fused Jacobi pass and of the fused red-black iteration with k = 0 .. 6, reserved through the launch
geometry (pair_tj_model counts fewer slots per XCD; the default) and through a CU mask on the compute stream.  python3 tools/cu_reserve_cost.py [n]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubez_amd import CZ  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
out = {}
MASK = "--mask" in sys.argv  # also measure the CU-mask form (queues with a CU mask hung twice in round 3: not by default)
cases = [("f32", "jacobi", 0.8, 0), ("f64", "jacobi", 0.8, 0), ("f32", "sor2sma", 1.5, 0)]
if MASK:
    cases += [("f32", "jacobi", 0.8, 1), ("f64", "jacobi", 0.8, 1)]
for prec, solver, coef, hard in cases:
    for k in (0, 1, 2, 3, 4, 6):
        if k == 0 and hard:
            continue
        cz = CZ(prec, quiet=True)
        assert cz.setup([n, n, n, solver, 1000, coef]) == 1
        assert cz.lib.czhip_set_comm_cus(k, hard) == k
        cz.sweeps(40)
        cz.lib.czhip_sync()
        cz.timing(True)
        best = 1e9
        for rep in range(2):
            t0 = time.perf_counter()
            cz.sweeps(100)
            cz.lib.czhip_sync()
            best = min(best, (time.perf_counter() - t0) / 100)
        nk, ms = cz.timing_read("jacobi2" if solver == "jacobi" else "rbsor2")
        cz.timing(False)
        cz.lib.czhip_set_comm_cus(0, 0)
        cz.close()
        out[f"{solver}_{prec}_{'mask' if hard else 'geometry'}_k{k}"] = dict(ms_per_sweep=best * 1e3, kernel_ms=ms / max(nk, 1), mlups=(n - 2) ** 3 / best / 1e6)
        print(f"{solver:8s} {prec} {'CU mask ' if hard else 'geometry'} k = {k}: {best * 1e3:.4f} ms per sweep, kernel {ms / max(nk, 1):.4f} ms per launch, {(n - 2) ** 3 / best / 1e6:9.0f} MLUPS", flush=True)
print(json.dumps(out))
