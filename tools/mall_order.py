"""Does the order in which a pass walks the j planes matter to the 256 MB memory-side cache?  The fused Jacobi pass (u -> w, ping-pong) as P
sub-launches over j ranges, in the same order every pass or in alternating order (the part written last is read first by the next pass):
    python3 tools/mall_order.py [n] [prec]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cubez_amd import CzHip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
h = CzHip(prec)
R = h.real
sz = [n, n, n]
rng = np.random.default_rng(1)
p = rng.uniform(-1, 1, (n + 4, n + 4, n + 4)).astype(R)
du, dw, db = h.alloc(sz, p), h.alloc(sz, p), h.alloc(sz, p * 0)
cf = [1, 1, 1, 1, 1, 1, 6]
full = [2, n - 1, 2, n - 1, 2, n - 1]


def parts(P):
    lo, hi = 2, n - 1
    cuts = [lo + (hi - lo + 1) * q // P for q in range(P)] + [hi + 1]
    out = []
    for q in range(P):
        a, b = cuts[q], cuts[q + 1] - 1
        idx = [2, n - 1, a, b, 2, n - 1]
        idx1 = [2, n - 1, max(lo, a - 1), min(hi, b + 1), 2, n - 1]
        out.append((idx, idx1))
    return out


def run(P, alternate, passes=40):
    pp = parts(P)
    a, b = du, dw
    def one(k):
        order = pp if (not alternate or k % 2 == 0) else pp[::-1]
        for idx, idx1 in order:
            ok, _, _ = h.jacobi2(a, b, db, sz, idx, cf, 0.8, idx1=idx1 if P > 1 else None, read=False)
            assert ok
    for k in range(6):
        one(k)
        a, b = b, a
    h.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(passes):
            one(k)
            a, b = b, a
        h.sync()
        best = min(best, (time.perf_counter() - t0) / passes)
    return best * 1e3


print(f"{n}^3 {prec}: ms per pass (two sweeps)")
print(f"  whole pass, one launch                : {run(1, False):.4f}")
for P in (2, 3, 4, 6):
    print(f"  {P} parts, same order every pass        : {run(P, False):.4f}      alternating order: {run(P, True):.4f}")
