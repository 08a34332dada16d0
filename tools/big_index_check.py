"""Arrays of more than 2^31 elements (1300^3 cells, 8.9 GB per FP32 array): the fused-pair path and the single-sweep path must agree
bit for bit, and a k-face value must come out where the 64-bit index says.  python tools/big_index_check.py [N]"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubez_amd import CZ  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
assert (N + 4) ** 3 > 2 ** 31
out = []
for t2 in (1, 0):
    cz = CZ("f32", quiet=True)
    cz.lib.czhip_set_tuning2(0, 0, -1, t2)
    t0 = time.time()
    assert cz.setup([N, N, N, "jacobi", 4, 0.8]) == 1
    itr = cz.solve()
    P = cz.field()
    out.append((itr, cz.history(), hashlib.sha256(P.tobytes()).hexdigest(), float(P[N // 2 + 2, N // 2 + 2, 2]), float(P[N + 1, N + 1, N + 1])))
    print("t2 =", t2, "iter", itr, "hist", cz.history(), "seconds", round(time.time() - t0, 1), flush=True)
    cz.lib.czhip_set_tuning2(0, 0, -1, 1)
    cz.close()
    del P
assert out[0][0] == out[1][0] == 5
assert out[0][2] == out[1][2], "fused-pair and single-sweep paths differ"
assert np.allclose(out[0][1], out[1][1], rtol=1e-12, atol=0)
x = (N // 2) / (N - 1)
assert abs(out[0][3] - np.sin(np.pi * x) ** 2) < 1e-5, out[0][3]   # Dirichlet value on the k = 1 face, centre of the plane
print("OK: %d^3 cells, %.2e elements per array" % (N, (N + 4) ** 3))
