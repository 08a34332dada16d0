"""CPU tests: the oracle (C restatement) against the golden fixtures generated from the reference's
own Fortran (tests/golden/make_golden.py), and -- when oracle/_ref is present -- against the
reference library directly on fresh random inputs.  All bit-exact (single thread, no contraction)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import cz_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = json.load(open(os.path.join(GOLDEN, "solver_cases.json")))


def _beq(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_kernels_match_golden(prec):
    g = np.load(os.path.join(GOLDEN, f"kernels_{prec}.npz"))
    k = O.Kernels("oracle", prec)
    sz, idx, cf, omg = list(g["sz"]), list(g["idx"]), g["cf"], float(g["omg"])
    p, b, q, x, y, z = (g[n] for n in ("in_p", "in_b", "in_q", "in_x", "in_y", "in_z"))

    pj, wk = p.copy(), np.zeros_like(p)
    res = k.jacobi(pj, sz, idx, cf, omg, b, wk, res=0.25)
    assert _beq(pj, g["jacobi_p"]) and _beq(wk, g["jacobi_wk2"])
    assert res == float(g["jacobi_res"]) and k.last_flop == float(g["jacobi_flop"])

    for ofst in (0, 1):
        ps, r = p.copy(), 0.0
        for color in (0, 1):
            r = k.psor2sma_core(ps, sz, idx, cf, ofst, color, omg, b, res=r)
            assert _beq(ps, g[f"rb{ofst}_p_c{color}"])
            assert r == float(g[f"rb{ofst}_res_c{color}"])

    ap = np.zeros_like(p)
    k.blas_calc_ax(ap, p, sz, idx, cf)
    assert _beq(ap, g["calc_ax"])
    rk = np.zeros_like(p)
    k.blas_calc_rk(rk, p, b, sz, idx, cf)
    assert _beq(rk, g["calc_rk"])
    assert k.blas_dot1(p, sz, idx) == g["dot1"]
    assert k.blas_dot2(p, q, sz, idx) == g["dot2"]
    zt = z.copy()
    k.blas_triad(zt, x, y, -0.37, sz, idx)
    assert _beq(zt, g["triad"])
    pb = p.copy()
    k.blas_bicg_1(pb, x, q, 0.61, -1.3, sz, idx)
    assert _beq(pb, g["bicg_1"])
    zb = z.copy()
    k.blas_bicg_2(zb, x, y, 0.45, -0.77, sz, idx)
    assert _beq(zb, g["bicg_2"])
    c = p.copy()
    k.blas_clear(c, sz)
    assert _beq(c, g["clear"])
    d = np.zeros_like(p)
    k.blas_copy(d, p, sz)
    assert _beq(d, g["copy"])
    for tag, nid, org in (("all", [-1] * 6, [0.0, 0.0, 0.0]), ("mix", [3, -1, -1, 5, -1, 2], [0.25, 0.5, 0.0])):
        pc = p.copy()
        k.bc_k(sz, pc, 1.0 / (sz[2] - 1), org, nid)
        assert _beq(pc, g[f"bc_{tag}"])
    e = np.zeros_like(p)
    k.exact_t(sz, e, 1.0 / (sz[2] - 1), [0.0, 0.0, 0.0])
    assert _beq(e, g["exact"])

    # MAF flavour on the stretched grid
    xc, yc, zc = g["maf_x"], g["maf_y"], g["maf_z"]
    pv = g["maf_pvt_in"].copy()
    k.search_pivot(pv, sz, idx, xc, yc, zc)
    assert _beq(pv, g["maf_pvt"])
    pm, wm = p.copy(), np.zeros_like(p)
    res = k.jacobi_maf(pm, sz, idx, xc, yc, zc, omg, b, wm, res=0.5)
    assert _beq(pm, g["maf_jacobi_p"]) and _beq(wm, g["maf_jacobi_wk2"])
    assert res == float(g["maf_jacobi_res"]) and k.last_flop == float(g["maf_jacobi_flop"])
    for ofst in (0, 1):
        ps, r = p.copy(), 0.0
        for color in (0, 1):
            r = k.psor2sma_core_maf(ps, sz, idx, xc, yc, zc, ofst, color, omg, b, res=r)
            assert _beq(ps, g[f"maf_rb{ofst}_p_c{color}"]) and r == float(g[f"maf_rb{ofst}_res_c{color}"])
    a = g["maf_ax_in"].copy()
    k.calc_ax_maf(a, p, sz, idx, xc, yc, zc, pv)
    assert _beq(a, g["maf_ax"])
    a = g["maf_rk_in"].copy()
    k.calc_rk_maf(a, p, b, sz, idx, xc, yc, zc, pv)
    assert _beq(a, g["maf_rk"])

    # lexicographic point SOR, one thread
    pp = g["in_p"].copy()
    assert k.psor(pp, sz, idx, cf, omg, b, res=0.125) == float(g["psor_res"]) and _beq(pp, g["psor_p"])
    assert k.last_flop == float(g["psor_flop"])
    pp = g["in_p"].copy()
    assert k.psor_maf(pp, sz, idx, xc, yc, zc, omg, b, res=0.125) == float(g["maf_psor_res"]) and _beq(pp, g["maf_psor_p"])

    # line SOR by PCR (fixtures from the reference's serial build)
    for (ni, nj, nk) in ((9, 8, 13), (12, 10, 37), (6, 7, 64)):
        szp, idp, tag = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1], f"pcr_{ni}x{nj}x{nk}"
        xx, mk = g[tag + "_x_in"].copy(), g[tag + "_x_in"].copy()
        k.imask_k(mk, szp, idp)
        assert _beq(mk, g[tag + "_msk"])
        r = 0.0
        for color in (0, 1):
            r = k.pcr_rb(szp, idp, O.get_num_stage(idp[5] - idp[4] + 1), 0, color, xx, mk, g[tag + "_rhs"], 1.1, res=r)
            assert _beq(xx, g[tag + f"_x_c{color}"]) and r == float(g[tag + f"_res_c{color}"])
        assert k.last_flop == float(g[tag + "_flop"])


def test_wide_accumulators_consistent():
    """the *_w entry points return the same REAL result plus a double accumulation of the same terms."""
    g = np.load(os.path.join(GOLDEN, "kernels_f32.npz"))
    k = O.Kernels("oracle", "f32")
    sz, idx = list(g["sz"]), list(g["idx"])
    p, b = g["in_p"], g["in_b"]
    w = np.zeros(1)
    pj, wk = p.copy(), np.zeros_like(p)
    res = k.jacobi(pj, sz, idx, g["cf"], float(g["omg"]), b, wk, res=0.25, wide=w)
    assert res == float(g["jacobi_res"])
    dp = (pj - p).astype(np.float32)[2:-2, 2:-2, 2:-2]
    # dp*dp rounded to float then summed in double == the wide accumulator up to order of summation
    assert abs(w[0] - float(np.sum((dp * dp).astype(np.float64)))) < 5e-5 * w[0]
    w[:] = 0
    assert k.blas_dot1(p, sz, idx, wide=w) == g["dot1"]
    inner = p[3:-3, 3:-3, 3:-3]
    assert abs(w[0] - float(np.sum((inner * inner).astype(np.float64)))) < 1e-12 * w[0]


# 128^3 BiCGSTAB cases cost ~10 s each on one core: keep two, skip nothing silently
SMALL = [c for c in CASES if max(c["gsz"]) <= 64 or c["tag"] in ("jacobi_128x128x128_f32", "pbicgstab_sor2sma_128x128x128_f64")]


@pytest.mark.parametrize("case", SMALL, ids=[c["tag"] for c in SMALL])
def test_solver_histories_match_golden(case):
    r = O.run(case["gsz"], case["solver"], case["itr_max"], case["coef"], case["precond"], kind="oracle",
              prec=case["prec"], with_error=True)
    assert r.itr == case["iter"]
    assert r.res == case["res"]
    if case["cli"]:  # what the reference CLI printed
        assert r.itr == case["cli"][0] and "%e" % r.res == case["cli"][1]
    assert r.history_text() == open(os.path.join(GOLDEN, f"hist_{case['tag']}.txt")).read()
    assert hashlib.sha256(r.P.tobytes()).hexdigest() == case["sha256_P"]
    if "field" in case:
        assert _beq(r.P, np.load(os.path.join(GOLDEN, case["field"])))
    assert r.errmax == case["errmax"] and list(r.errloc) == case["errloc"]


@pytest.mark.skipif(not O.have("ref"), reason="oracle/_ref not built (reference sources absent)")
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_oracle_vs_reference_random_boxes(prec):
    """fresh random inputs, anisotropic boxes, both colours/offsets: restatement == reference, bit for bit."""
    ko, kr = O.Kernels("oracle", prec), O.Kernels("ref", prec)
    rng = np.random.default_rng(7)
    for (ni, nj, nk) in ((5, 4, 6), (9, 12, 7), (16, 8, 33)):
        sz = [ni, nj, nk]
        shape = (nj + 4, ni + 4, nk + 4)
        idx = [2, ni - 1, 2, nj - 1, 2, nk - 1] if ni > 5 else [1, ni, 1, nj, 1, nk]  # also the interior-rank range
        cf = rng.uniform(0.5, 1.5, 7).astype(ko.real)
        cf[6] = 6.2
        p = rng.uniform(-1, 1, shape).astype(ko.real)
        b = rng.uniform(-1, 1, shape).astype(ko.real)
        q = rng.uniform(-1, 1, shape).astype(ko.real)
        a1, w1, a2, w2 = p.copy(), np.zeros_like(p), p.copy(), np.zeros_like(p)
        assert ko.jacobi(a1, sz, idx, cf, 0.9, b, w1, res=1.5) == kr.jacobi(a2, sz, idx, cf, 0.9, b, w2, res=1.5)
        assert _beq(a1, a2) and _beq(w1, w2)
        for ofst in (0, 1):
            a1, a2, r1, r2 = p.copy(), p.copy(), 0.0, 0.0
            for color in (0, 1):
                r1 = ko.psor2sma_core(a1, sz, idx, cf, ofst, color, 1.3, b, res=r1)
                r2 = kr.psor2sma_core(a2, sz, idx, cf, ofst, color, 1.3, b, res=r2)
                assert r1 == r2 and _beq(a1, a2)
        a1, a2 = np.zeros_like(p), np.zeros_like(p)
        ko.blas_calc_ax(a1, p, sz, idx, cf), kr.blas_calc_ax(a2, p, sz, idx, cf)
        assert _beq(a1, a2)
        ko.blas_calc_rk(a1, p, b, sz, idx, cf), kr.blas_calc_rk(a2, p, b, sz, idx, cf)
        assert _beq(a1, a2)
        assert ko.blas_dot1(p, sz, idx) == kr.blas_dot1(p, sz, idx)
        assert ko.blas_dot2(p, q, sz, idx) == kr.blas_dot2(p, q, sz, idx)
        a1, a2 = q.copy(), q.copy()
        ko.blas_bicg_1(a1, p, b, 0.3, 0.7, sz, idx), kr.blas_bicg_1(a2, p, b, 0.3, 0.7, sz, idx)
        assert _beq(a1, a2)
        ko.blas_bicg_2(a1, p, b, 0.3, 0.7, sz, idx), kr.blas_bicg_2(a2, p, b, 0.3, 0.7, sz, idx)
        assert _beq(a1, a2)
        ko.blas_triad(a1, p, b, -0.3, sz, idx), kr.blas_triad(a2, p, b, -0.3, sz, idx)
        assert _beq(a1, a2)
        a1, a2 = p.copy(), p.copy()
        for nid in ([-1] * 6, [0, 1, -1, -1, 2, -1]):
            ko.bc_k(sz, a1, 0.125, [0.1, 0.2, 0.3], nid), kr.bc_k(sz, a2, 0.125, [0.1, 0.2, 0.3], nid)
            assert _beq(a1, a2)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_sph_writer_restatement_matches_the_reference_file(prec):
    """.sph field file (cz_utility.f90:17-47, written by the reference built with -D_aurora_=1): the restatement used to check
    the GPU driver's p_00000.sph / e_00000.sph produces the reference's bytes."""
    g = np.load(os.path.join(GOLDEN, f"kernels_{prec}.npz"))
    R = np.float32 if prec == "f32" else np.float64
    want = open(os.path.join(GOLDEN, f"sph_small_{prec}.sph"), "rb").read()
    got = O.sph_bytes([5, 4, 6], g["sph_in"], R(0.25), np.array([0.5, 0.25, 0.125], dtype=R))
    assert got == want
