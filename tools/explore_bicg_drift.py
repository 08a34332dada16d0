"""How far the GPU BiCGSTAB histories are from the reference, next to how far the reference itself moves when only the summation order of
its dot products changes (tests/golden/perm_cases.json, large_cases.json).  Evidence for the bounds in tests/test_gpu_solvers.py."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cubez_amd import CZ
G = os.path.join(ROOT, "tests", "golden")
perm = json.load(open(f"{G}/perm_cases.json"))
cases = {c["tag"]: c for c in json.load(open(f"{G}/solver_cases.json"))}
for tag, p in perm.items():
    c = cases[tag]
    cz = CZ(c["prec"], quiet=True)
    a = list(c["gsz"]) + [c["solver"], c["itr_max"], c["coef"]] + ([c["precond"]] if c["precond"] else [])
    assert cz.setup(a) == 1
    itr = cz.solve()
    h = np.array(cz.history()); r = np.array(p["hist_reference_full_precision"])
    ph = np.array([float(l.split(",")[1]) for l in open(f"{G}/{p['hist']}").read().splitlines()[1:]])
    m = min(len(h), len(r)); mp = min(len(ph), len(r))
    print(f"{tag:42s} ref {c['iter']:3d} perm {p['iter']:3d} gpu {itr:3d} | dev gpu {np.max(np.abs(h[:m]-r[:m])/r[:m]):.2e} perm {np.max(np.abs(ph[:mp]-r[:mp])/r[:mp]):.2e} | final res rel gpu {abs(cz.res-c['res'])/c['res']:.2e} perm {abs(p['res']-c['res'])/c['res']:.2e}")
    cz.close()
L = json.load(open(f"{G}/large_cases.json"))
for tag, c in L.items():
    cz = CZ("f64", quiet=True)
    assert cz.setup(list(c["gsz"]) + ["pbicgstab", c["itr_max"], 0.8, "jacobi"]) == 1
    itr = cz.solve()
    h = np.array(cz.history())
    r = np.array([float(l.split(",")[1]) for l in open(f"{G}/{c['hist']}").read().splitlines()[1:]])
    m = min(len(h), len(r))
    dev = np.abs(h[:m]-r[:m])/r[:m]
    print(tag, "ref iter", c["iter"], "gpu", itr, "max dev %.2e" % dev.max(), "dev at 4,10,20,40,60,last:", [float("%.1e" % dev[min(i, m-1)]) for i in (3, 9, 19, 39, 59, m-1)])
    cz.close()
