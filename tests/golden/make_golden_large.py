#!/usr/bin/env python3
"""Golden fixtures for BASELINE.json configs[3] at (near) full size, and the measured bound of the BiCGSTAB tolerance.

Runs only in the build container (needs oracle/_ref/libczref_f64.so = the reference's own Fortran kernels, one thread).  Writes DATA:

  large_cases.json                      index: per case Iter, Res, sha256(X), history file
  hist_pbicgstab_jacobi_256x256x256_f64.txt      `cz 256 256 256 pbicgstab 1000 0.8 jacobi` to convergence (reference kernels driven by the
                                                 restated host loop of oracle/cz_oracle.py, which the 14 CLI pins of make_golden.py validate)
  hist_pbicgstab_jacobi_512x512x512_f64_4it.txt  the first 4 iterations of `cz 512 512 512 pbicgstab ... 0.8 jacobi` (ItrMax = 5: the loop
                                                 is `itr < ItrMax`, cz_Poisson.cpp:373) + sha256 of X after them
  perm_cases.json + hist_*_permuted_dots.txt     every BiCGSTAB case of solver_cases.json run a second time with the C restatement
                                                 (== reference, bit for bit: tests/test_oracle.py) whose two dot products sum the same
                                                 terms with j descending (oracle_set_dot_order(1)).  |permuted - reference| is how far the
                                                 reference moves by its own summation-order rounding; the GPU tests bound |GPU - reference|
                                                 by a small multiple of it instead of by a number chosen by hand.

usage: make_golden_large.py [perm] [256] [512] [perm256]     (default: the first three)
"""
import hashlib
import json
import os
import sys
import time

os.environ["OMP_NUM_THREADS"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import numpy as np  # noqa: E402

from oracle import cz_oracle as O  # noqa: E402


def _load(name):
    try:
        return json.load(open(os.path.join(HERE, name)))
    except (OSError, ValueError):
        return {}


def large(n, itr_max, suffix):
    t0 = time.time()
    r = O.run((n, n, n), "pbicgstab", itr_max, 0.8, "jacobi", kind="ref", prec="f64")
    tag = f"pbicgstab_jacobi_{n}x{n}x{n}_f64{suffix}"
    with open(os.path.join(HERE, f"hist_{tag}.txt"), "w") as f:
        f.write("Itration      Residual\n" + "".join("%6d, %23.16e\n" % (i, v) for i, v in r.history))
    rec = _load("large_cases.json")
    rec[tag] = dict(gsz=[n, n, n], solver="pbicgstab", precond="jacobi", coef=0.8, itr_max=itr_max, prec="f64", iter=r.itr, res=r.res,
                    n_history=len(r.history), sha256_X=hashlib.sha256(r.P.tobytes()).hexdigest(), hist=f"hist_{tag}.txt",
                    backend="oracle/_ref/libczref_f64.so (reference Fortran, OMP_NUM_THREADS=1)", seconds=round(time.time() - t0, 1))
    json.dump(rec, open(os.path.join(HERE, "large_cases.json"), "w"), indent=1)
    print(tag, r.itr, "%e" % r.res, "%.0f s" % (time.time() - t0), flush=True)


def permuted():
    cases = [c for c in json.load(open(os.path.join(HERE, "solver_cases.json"))) if c["solver"] in ("pbicgstab", "pbicgstab_maf")]
    out = _load("perm_cases.json")
    for c in cases:
        k = O.Kernels("oracle", c["prec"])
        res = {}
        for order in (0, 1):
            k.lib.oracle_set_dot_order(order)
            try:
                r = O.run(c["gsz"], c["solver"], c["itr_max"], c["coef"], c["precond"], kind="oracle", prec=c["prec"])
            finally:
                k.lib.oracle_set_dot_order(0)
            res[order] = r
        # order 0 must be the fixture the reference library produced
        assert res[0].itr == c["iter"] and res[0].res == c["res"], (c["tag"], res[0].itr, res[0].res, c["iter"], c["res"])
        h0 = np.array([v for _, v in res[0].history])
        h1 = np.array([v for _, v in res[1].history])
        m = min(len(h0), len(h1))
        with open(os.path.join(HERE, f"hist_{c['tag']}_permuted_dots.txt"), "w") as f:
            f.write("Itration      Residual\n" + "".join("%6d, %23.16e\n" % (i, v) for i, v in res[1].history))
        out[c["tag"]] = dict(iter=res[1].itr, res=res[1].res, iter_reference=res[0].itr, res_reference=res[0].res,
                             max_rel_dev_history=float(np.max(np.abs(h1[:m] - h0[:m]) / h0[:m])) if m else 0.0,
                             max_abs_dev_X=float(np.abs(res[1].P.astype(np.float64) - res[0].P.astype(np.float64)).max()),
                             hist=f"hist_{c['tag']}_permuted_dots.txt", hist_reference_full_precision=[float(v) for v in h0])
        print(c["tag"], "reference", res[0].itr, "%e" % res[0].res, "| j-descending dots", res[1].itr, "%e" % res[1].res,
              "| max rel dev of the history %.2e" % out[c["tag"]]["max_rel_dev_history"], flush=True)
    json.dump(out, open(os.path.join(HERE, "perm_cases.json"), "w"), indent=1)


def permuted_large(n):
    """the 256^3 solve again with j-descending dots (C restatement): the measured bound for the full-size history"""
    t0 = time.time()
    k = O.Kernels("oracle", "f64")
    k.lib.oracle_set_dot_order(1)
    try:
        r = O.run((n, n, n), "pbicgstab", 1000, 0.8, "jacobi", kind="oracle", prec="f64")
    finally:
        k.lib.oracle_set_dot_order(0)
    tag = f"pbicgstab_jacobi_{n}x{n}x{n}_f64"
    with open(os.path.join(HERE, f"hist_{tag}_permuted_dots.txt"), "w") as f:
        f.write("Itration      Residual\n" + "".join("%6d, %23.16e\n" % (i, v) for i, v in r.history))
    rec = _load("large_cases.json")
    rec[tag]["permuted_dots"] = dict(iter=r.itr, res=r.res, hist=f"hist_{tag}_permuted_dots.txt", seconds=round(time.time() - t0, 1))
    json.dump(rec, open(os.path.join(HERE, "large_cases.json"), "w"), indent=1)
    print(tag, "j-descending dots", r.itr, "%e" % r.res, "%.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["perm", "256", "512"]
    if "perm" in what:
        permuted()
    assert O.have("ref", "f64"), "build oracle/_ref first: make -C oracle ref"
    if "256" in what:
        large(256, 1000, "")
    if "512" in what:
        large(512, 5, "_4it")
    if "perm256" in what:
        permuted_large(256)
