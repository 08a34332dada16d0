"""Cost of running the fused pass as shell slabs + interior (the overlapped form of a decomposed brick) against the unsplit
launch, on one GPU: python tools/split_cost.py [--n 512] [--prec f32].  Prints kernel times from the library's own HIP-event timing."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubez_amd import CzHip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--prec", default="f32")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
h = CzHip(a.prec)
n = a.n
sz = [n, n, n]
R = np.float32 if a.prec == "f32" else np.float64
rng = np.random.default_rng(1)
u0 = rng.uniform(-1, 1, (n + 4, n + 4, n + 4)).astype(R)
du, db, dw = h.alloc(sz, u0), h.alloc(sz, u0), h.alloc(sz, u0)
cf = [1, 1, 1, 1, 1, 1, 6]
out = {}
for name, pat in (("warm-up", [1] * 6), ("corner brick of 2x2x2 (I+,J+,K+)", [0, 1, 0, 1, 0, 1]), ("inner brick (all six faces)", [1] * 6),
                  ("slab 1x8x1 (J-,J+)", [0, 0, 1, 1, 0, 0]), ("I+ only", [0, 1, 0, 0, 0, 0]), ("J+ only", [0, 0, 0, 1, 0, 0]),
                  ("K+ only", [0, 0, 0, 0, 0, 1]), ("K- only", [0, 0, 0, 0, 1, 0])):
    nID = [(3 if v else -1) for v in pat]
    idx = [(1 if v else 2) if f % 2 == 0 else (n if v else n - 1) for f, v in enumerate(pat)]
    idx1 = [idx[f] + ((1 if f % 2 else -1) if v else 0) for f, v in enumerate(pat)]
    for _ in range(3):
        h.jacobi2(du, dw, db, sz, idx, cf, 0.8, idx1=idx1, read=False)
        h.pair_split(du, dw, db, sz, idx, idx1, nID, cf, 0.8, read=False)
    h.timing(True)
    for _ in range(a.reps):
        h.jacobi2(du, dw, db, sz, idx, cf, 0.8, idx1=idx1, read=False)
    h.lib.czhip_sync()
    n_full, ms_full = h.timing_read("jacobi2")
    h.timing(False)
    h.timing(True)
    for _ in range(a.reps):
        h.pair_split(du, dw, db, sz, idx, idx1, nID, cf, 0.8, read=False)
    h.lib.czhip_sync()
    n_in, ms_in = h.timing_read("jacobi2")
    n_sh, ms_sh = h.timing_read("pair_shell")
    h.timing(False)
    out[name] = {"unsplit_ms": ms_full / n_full, "interior_ms": ms_in / n_in, "shell_ms": ms_sh / n_sh}
print(json.dumps(out, indent=1))
