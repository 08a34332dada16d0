"""rb4_k (two red-black iterations per pass) against two fused iterations (jacobi2p_k<RB = 1>): ms per iteration, MLUPS, by window length and chunk.
    python3 tools/rb4_rate.py prec n [kwin,tj ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cubez_amd import CzHip

prec, n = sys.argv[1], int(sys.argv[2])
forms = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or [(0, 0)]
h = CzHip(prec)
R = h.real
sz, idx = [n, n, n], [2, n - 1, 2, n - 1, 2, n - 1]
rng = np.random.default_rng(1)
shape = (n + 4, n + 4, n + 4)
cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=R)
p = rng.uniform(-1, 1, shape).astype(R)
du, db, dw = h.alloc(sz, p), h.alloc(sz, p * 0), h.alloc(sz, p)
(_, szp), (_, idxp), (_, cfp) = h._i(sz), h._i(idx), h._r(cf)
dres = h.lib.czhip_alloc_s3d((C.c_int * 3)(4, 4, 4))
f4, f2 = h.lib.czhip_rbsor4_async, h.lib.czhip_rbsor2_async
f4.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, h.creal, C.c_void_p, C.c_double, C.c_double, C.c_int,
               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
f2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, h.creal, C.c_void_p, C.c_double, C.c_double,
               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
pts = (n - 2) ** 3


def run(fn, reps):
    a, b = du, dw
    for _ in range(5):
        fn(a, b)
        a, b = b, a
    h.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(a, b)
            a, b = b, a
        h.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


t2 = run(lambda a, b: f2(a.ptr, b.ptr, db.ptr, szp, idxp, None, 2, cfp, 0, 1.5, dres, 0.0, 0.0, 0, None, None, None, None), 60)
print(f"{prec} {n}^3  one iteration per pass (jacobi2p_k<RB>): {t2 * 1e3:.4f} ms per iteration  {pts / t2 / 1e6:9.0f} MLUPS", flush=True)
for kw, tj in forms:
    h.lib.czhip_set_rb4(2, kw, tj)
    ok = f4(du.ptr, dw.ptr, db.ptr, szp, idxp, 2, cfp, 0, 1.5, dres, 0.0, 0.0, 0, None, None, None, None, 1)
    if not ok:
        print(f"   rb4 window {kw} chunk {tj}: refused")
        continue
    t4 = run(lambda a, b: f4(a.ptr, b.ptr, db.ptr, szp, idxp, 2, cfp, 0, 1.5, dres, 0.0, 0.0, 0, None, None, None, None, 0), 40)
    print(f"   rb4 window {kw:3d} chunk {tj:3d}: {t4 * 1e3:.4f} ms per pass = {t4 * 5e2:.4f} ms per iteration  {2 * pts / t4 / 1e6:9.0f} MLUPS  ({t2 / (t4 / 2):.2f} x)", flush=True)
h.lib.czhip_set_rb4(1, 0, 0)
