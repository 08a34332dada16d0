#!/usr/bin/env python3
"""Print how far the GPU BiCGSTAB residual histories drift from the reference-generated golden histories."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from cubez_amd import CZ
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
for c in json.load(open(os.path.join(G, "solver_cases.json"))):
    if c["solver"] != "pbicgstab":
        continue
    cz = CZ(c["prec"], quiet=True)
    a = list(c["gsz"]) + [c["solver"], c["itr_max"], c["coef"], c["precond"]]
    cz.setup(a); itr = cz.solve(); h = np.array(cz.history())
    ref = np.array([float(l.split(",")[1]) for l in open(os.path.join(G, f"hist_{c['tag']}.txt")).read().splitlines()[1:]])
    n = min(len(h), len(ref))
    rel = np.abs(h[:n] - ref[:n]) / ref[:n]
    print(c["tag"], "itr", itr, "ref", c["iter"], "final res %.9e ref %.9e rel %.2e" % (cz.res, c["res"], abs(cz.res - c["res"]) / c["res"]),
          "max hist rel %.2e at %d" % (rel.max(), rel.argmax() + 1), "t=%.3fs" % cz.solve_seconds)
    cz.close()
