"""What does the CU reservation of decomposed runs (CZ_COMM_CUS = k CUs per XCD left to the exchange stream) cost the sweeps?
One rank, 512^3: ms per launch of the fused Jacobi pass and of the fused red-black iteration with k = 0 .. 6, reserved through the launch
geometry (pair_tj_model counts fewer slots per XCD).  python3 tools/cu_reserve_cost.py [n]
(The CU-mask form this script also measured in round 3 was removed from the library in round 4; profiles/r03/cu_reserve_cost.txt has its numbers.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubez_amd import CZ  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
out = {}
cases = [("f32", "jacobi", 0.8), ("f64", "jacobi", 0.8), ("f32", "sor2sma", 1.5)]
for prec, solver, coef in cases:
    for k in (0, 1, 2, 3, 4, 6):
        cz = CZ(prec, quiet=True)
        assert cz.setup([n, n, n, solver, 1000, coef]) == 1
        assert cz.lib.czhip_set_comm_cus(k) == k
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
        cz.lib.czhip_set_comm_cus(0)
        cz.close()
        out[f"{solver}_{prec}_geometry_k{k}"] = dict(ms_per_sweep=best * 1e3, kernel_ms=ms / max(nk, 1), mlups=(n - 2) ** 3 / best / 1e6)
        print(f"{solver:8s} {prec} geometry k = {k}: {best * 1e3:.4f} ms per sweep, kernel {ms / max(nk, 1):.4f} ms per launch, {(n - 2) ** 3 / best / 1e6:9.0f} MLUPS", flush=True)
print(json.dumps(out))
