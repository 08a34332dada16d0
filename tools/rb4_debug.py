"""where does rb4_k differ from four colour sweeps of the oracle?  python3 tools/rb4_debug.py prec ni nj nk kwin tj"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cubez_amd import CzHip
from oracle import cz_oracle as O
prec, ni, nj, nk, kw, tj = sys.argv[1], *[int(v) for v in sys.argv[2:7]]
h, ko = CzHip(prec), O.Kernels("oracle", prec)
R = ko.real
sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
rng = np.random.default_rng(1)
shape = (nj + 4, ni + 4, nk + 4)
cf = rng.uniform(0.5, 1.5, 7).astype(R); cf[6] = 6.2
p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
a1 = p.copy()
for it in range(2):
    for color in (0, 1):
        ko.psor2sma_core(a1, sz, idx, cf, 0, color, 1.3, b)
du, db, dw = h.alloc(sz, p), h.alloc(sz, b), h.alloc(sz, p)
h.lib.czhip_set_rb4(2, kw, tj)
ok, r1, r2 = h.rbsor4(du, dw, db, sz, idx, cf, 0, 1.3)
w = dw.get()
bad = np.argwhere(w != a1)
print("launched", ok, "differing", len(bad), "of", w.size)
if len(bad):
    js, is_, ks = bad[:, 0], bad[:, 1], bad[:, 2]
    print("j range", js.min(), js.max(), "i range", is_.min(), is_.max(), "k values", sorted(set(ks.tolist()))[:40])
    print("first", bad[:10].tolist())
    # is the wrong value the input (not updated) or something else?
    same_as_in = int((w[tuple(bad.T)] == p[tuple(bad.T)]).sum())
    print("equal to the input at", same_as_in, "of them")
