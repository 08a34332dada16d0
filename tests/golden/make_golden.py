#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container: it needs oracle/_ref/libczref_{f32,f64}.so, i.e. the
reference's own Fortran kernels (cz_solver.f90, cz_blas.f90, cz_utility.f90, cz_maf.f90) compiled in
place by oracle/Makefile with amdflang, called with OMP_NUM_THREADS=1 (the deterministic
mode, SURVEY.md finding 2).  Fixtures are DATA (inputs + expected outputs), committed so the
tests can run where /root/reference does not exist.

    kernels_{f32,f64}.npz    seeded random inputs on an 11x7x13 box, non-unit coefficients,
                             and the reference output of every hot-path kernel
    hist_*.txt               residual histories in the reference's file format
                             (cz_Evaluate.cpp:218 header, cz_Poisson.cpp:71 lines), produced by
                             the host loops of oracle/cz_oracle.py driving the reference kernels;
                             the final Iter/Res of each agrees with the reference CLI runs
                             recorded in BASELINE.md section 2b (checked in tests/test_oracle.py)
    field_*.npy              final P of the <=32^3 cases
    solver_cases.json        index of the solver-level cases + expected Iter/Res/sha256(P)
"""
import hashlib
import json
import os
import sys

os.environ["OMP_NUM_THREADS"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import numpy as np  # noqa: E402

from oracle import cz_oracle as O  # noqa: E402

BOX = (11, 7, 13)  # NI, NJ, NK (SURVEY.md 8c probe)
CF = (1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3)
OMG = 0.8


def kernel_vectors(prec):
    k = O.Kernels("ref", prec)
    R = k.real
    rng = np.random.default_rng(12345)
    shape = (BOX[1] + 4, BOX[0] + 4, BOX[2] + 4)

    def rnd():
        return rng.uniform(-1.0, 1.0, size=shape).astype(R)

    sz = list(BOX)
    idx = [2, BOX[0] - 1, 2, BOX[1] - 1, 2, BOX[2] - 1]
    cf = np.array(CF, dtype=R)
    out = dict(sz=np.array(sz), idx=np.array(idx), cf=cf, omg=np.array(OMG, dtype=R))
    p, b, q, x, y = rnd(), rnd(), rnd(), rnd(), rnd()
    out.update(in_p=p, in_b=b, in_q=q, in_x=x, in_y=y)

    # jacobi
    pj, wk = p.copy(), np.zeros_like(p)
    out["jacobi_res"] = np.array(k.jacobi(pj, sz, idx, cf, OMG, b, wk, res=0.25))  # in/out accumulator
    out["jacobi_p"], out["jacobi_wk2"], out["jacobi_flop"] = pj, wk, np.array(k.last_flop)

    # psor2sma_core colour 0 then 1, ofst 0 and 1
    for ofst in (0, 1):
        ps = p.copy()
        r = 0.0
        for color in (0, 1):
            r = k.psor2sma_core(ps, sz, idx, cf, ofst, color, OMG, b, res=r)
            out[f"rb{ofst}_p_c{color}"] = ps.copy()
            out[f"rb{ofst}_res_c{color}"] = np.array(r)

    # blas
    ap = np.zeros_like(p)
    k.blas_calc_ax(ap, p, sz, idx, cf)
    out["calc_ax"] = ap
    rk = np.zeros_like(p)
    k.blas_calc_rk(rk, p, b, sz, idx, cf)
    out["calc_rk"] = rk
    out["dot1"] = np.array(k.blas_dot1(p, sz, idx))
    out["dot2"] = np.array(k.blas_dot2(p, q, sz, idx))
    z = rnd()
    out["in_z"] = z.copy()
    zt = z.copy()
    k.blas_triad(zt, x, y, -0.37, sz, idx)
    out["triad"] = zt
    pb = p.copy()
    k.blas_bicg_1(pb, x, q, 0.61, -1.3, sz, idx)
    out["bicg_1"] = pb
    zb = z.copy()
    k.blas_bicg_2(zb, x, y, 0.45, -0.77, sz, idx)
    out["bicg_2"] = zb
    c = rnd()
    k.blas_clear(c, sz)
    out["clear"] = c
    d = np.zeros_like(p)
    k.blas_copy(d, p, sz)
    out["copy"] = d

    # bc_k: all faces physical, then a mixed neighbour table, non-zero origin
    for tag, nid, org in (("all", [-1] * 6, [0.0, 0.0, 0.0]), ("mix", [3, -1, -1, 5, -1, 2], [0.25, 0.5, 0.0])):
        pc = p.copy()
        k.bc_k(sz, pc, 1.0 / (BOX[2] - 1), org, nid)
        out[f"bc_{tag}"] = pc
    # analytic solution used by the reference's debug epilogue
    e = np.zeros_like(p)
    k.exact_t(sz, e, 1.0 / (BOX[2] - 1), [0.0, 0.0, 0.0])
    out["exact"] = e
    # MAF flavour (cz_maf.f90, cz_blas.f90:738-1039) on a stretched grid: monotone random coordinates so that the
    # second-difference terms (XGG, YEE, ZTT) are exercised, which the uniform benchmark grid leaves at rounding level
    def coords(n):
        return np.cumsum(rng.uniform(0.5, 1.5, n + 4)).astype(R) * R(0.05)

    xc, yc, zc = coords(BOX[0]), coords(BOX[1]), coords(BOX[2])
    out.update(maf_x=xc, maf_y=yc, maf_z=zc)
    pv = rnd()
    out["maf_pvt_in"] = pv.copy()
    k.search_pivot(pv, sz, idx, xc, yc, zc)
    out["maf_pvt"] = pv
    pm_, wm = p.copy(), np.zeros_like(p)
    out["maf_jacobi_res"] = np.array(k.jacobi_maf(pm_, sz, idx, xc, yc, zc, OMG, b, wm, res=0.5))
    out["maf_jacobi_p"], out["maf_jacobi_wk2"], out["maf_jacobi_flop"] = pm_, wm, np.array(k.last_flop)
    for ofst in (0, 1):
        ps, r = p.copy(), 0.0
        for color in (0, 1):
            r = k.psor2sma_core_maf(ps, sz, idx, xc, yc, zc, ofst, color, OMG, b, res=r)
            out[f"maf_rb{ofst}_p_c{color}"] = ps.copy()
            out[f"maf_rb{ofst}_res_c{color}"] = np.array(r)
    apm = rnd()
    # psor / psor_maf: lexicographic in-place SOR, ONE thread
    pp_ = p.copy()
    out["psor_res"] = np.array(k.psor(pp_, sz, idx, cf, OMG, b, res=0.125))
    out["psor_p"], out["psor_flop"] = pp_, np.array(k.last_flop)
    pp_ = p.copy()
    out["maf_psor_res"] = np.array(k.psor_maf(pp_, sz, idx, xc, yc, zc, OMG, b, res=0.125))
    out["maf_psor_p"], out["maf_psor_flop"] = pp_, np.array(k.last_flop)
    out["maf_ax_in"] = apm.copy()
    k.calc_ax_maf(apm, p, sz, idx, xc, yc, zc, pv)
    out["maf_ax"] = apm
    rkm = rnd()
    out["maf_rk_in"] = rkm.copy()
    k.calc_rk_maf(rkm, p, b, sz, idx, xc, yc, zc, pv)
    out["maf_rk"] = rkm
    # line SOR by PCR (cz_solver.f90:497-662), from the SERIAL reference build (see oracle/Makefile: the OpenMP form of
    # these routines reads uninitialised private work arrays)
    ks = O.Kernels("ref_serial", prec)
    for (ni, nj, nk) in ((9, 8, 13), (12, 10, 37), (6, 7, 64)):
        szp = [ni, nj, nk]
        idp = [2, ni - 1, 2, nj - 1, 2, nk - 1]
        shp = (nj + 4, ni + 4, nk + 4)
        xx = rng.uniform(-1.0, 1.0, size=shp).astype(R)
        rh = rng.uniform(-1.0, 1.0, size=shp).astype(R)
        mk = rng.uniform(-1.0, 1.0, size=shp).astype(R)
        ks.imask_k(mk, szp, idp)
        tag = f"pcr_{ni}x{nj}x{nk}"
        out[tag + "_x_in"], out[tag + "_rhs"], out[tag + "_msk"] = xx.copy(), rh, mk
        pn = O.get_num_stage(idp[5] - idp[4] + 1)
        r = 0.0
        for color in (0, 1):
            r = ks.pcr_rb(szp, idp, pn, 0, color, xx, mk, rh, 1.1, res=r)
            out[tag + f"_x_c{color}"] = xx.copy()
            out[tag + f"_res_c{color}"] = np.array(r)
        out[tag + "_flop"] = np.array(ks.last_flop)
    # .sph field file: the reference's writer, cz_utility.f90 built with -D_aurora_=1
    if O.have("ref_sph", prec):
        kw = O.Kernels("ref_sph", prec)
        szs = [5, 4, 6]
        fld = rng.uniform(-1, 1, (szs[1] + 4, szs[0] + 4, szs[2] + 4)).astype(R)
        cwd = os.getcwd()
        os.chdir(HERE)
        try:
            kw.fileout_t(szs, fld, 0.25, [0.5, 0.25, 0.125], f"sph_small_{prec}.sph")
        finally:
            os.chdir(cwd)
        out["sph_in"] = fld
    np.savez_compressed(os.path.join(HERE, f"kernels_{prec}.npz"), **out)


SOLVER_CASES = [
    # gsz, solver, ItrMax, coef, precond, prec, (Iter, Res) printed by the reference CLI
    ((32, 32, 32), "jacobi", 50, 0.8, None, "f32", (51, "1.179653e-03")),
    ((32, 32, 32), "jacobi", 50, 0.8, None, "f64", (51, "1.179668e-03")),
    ((32, 32, 32), "sor2sma", 50, 1.5, None, "f32", (51, "1.069268e-03")),
    ((32, 32, 32), "sor2sma", 50, 1.5, None, "f64", (51, "1.069267e-03")),
    ((48, 40, 36), "jacobi", 30, 0.9, None, "f32", (31, "1.506343e-03")),
    ((32, 32, 32), "pbicgstab", 200, 0.8, "jacobi", "f32", (9, "8.793215e-06")),
    ((32, 32, 32), "pbicgstab", 200, 0.8, "jacobi", "f64", (9, "8.800833e-06")),
    ((64, 64, 64), "pbicgstab", 500, 0.8, "jacobi", "f64", (18, "1.492440e-07")),
    ((64, 64, 64), "sor2sma", 100000, 1.5, None, "f64", (635, "9.937159e-06")),
    ((64, 64, 64), "sor2sma", 100000, 1.5, None, "f32", (635, "9.937164e-06")),
    ((128, 128, 128), "jacobi", 100, 0.8, None, "f32", (101, "3.765854e-04")),
    ((128, 128, 128), "sor2sma", 100, 1.5, None, "f32", (101, "5.656343e-04")),
    ((128, 128, 128), "pbicgstab", 1000, 0.8, "jacobi", "f64", (33, "6.982146e-06")),
    ((128, 128, 128), "pbicgstab", 1000, 1.5, "sor2sma", "f64", (13, "4.176314e-08")),
    # not in BASELINE.md: extra shapes (no CLI pin; reference kernels + restated loops only)
    ((20, 24, 28), "sor2sma", 40, 1.2, None, "f32", None),
    ((24, 20, 36), "pbicgstab", 60, 0.9, "sor2sma", "f64", None),
    ((24, 20, 36), "pbicgstab", 60, 0.9, "none", "f64", None),
    # MAF flavour (no CLI pin)
    ((32, 32, 32), "jacobi_maf", 40, 0.8, None, "f32", None),
    ((24, 20, 28), "jacobi_maf", 30, 0.8, None, "f64", None),
    ((32, 32, 32), "sor2sma_maf", 40, 1.5, None, "f32", None),
    ((24, 20, 28), "sor2sma_maf", 30, 1.5, None, "f64", None),
    ((32, 32, 32), "pbicgstab_maf", 100, 0.8, "jacobi_maf", "f64", None),
    ((32, 32, 32), "pbicgstab_maf", 100, 1.5, "sor2sma_maf", "f64", None),
    ((32, 32, 32), "pbicgstab_maf", 100, 0.8, "jacobi", "f32", None),
    # lexicographic point SOR, one thread
    ((32, 32, 32), "psor", 40, 1.1, None, "f32", None),
    ((24, 20, 36), "psor", 30, 1.3, None, "f64", None),
    ((64, 64, 64), "psor", 100000, 1.5, None, "f64", None),
    ((32, 32, 32), "psor_maf", 30, 1.1, None, "f32", None),
    ((32, 32, 32), "pbicgstab", 100, 1.2, "psor", "f64", None),
    # line SOR by PCR, serial reference build
    ((32, 32, 32), "pcr_rb", 40, 1.2, None, "f32", None),
    ((24, 20, 36), "pcr_rb", 30, 1.1, None, "f64", None),
    ((64, 64, 64), "pcr_rb", 100000, 1.5, None, "f64", None),
    ((32, 32, 32), "pbicgstab", 100, 1.2, "pcr_rb", "f64", None),
    ((32, 32, 32), "pcr", 30, 1.2, None, "f32", None),
    ((32, 32, 32), "pcr_esa", 30, 1.2, None, "f64", None),
    ((32, 32, 32), "pcr_eda", 30, 1.2, None, "f32", None),
    ((32, 32, 32), "pcr_rb_esa", 30, 1.2, None, "f32", None),
    ((64, 64, 64), "pcr_rb_esa", 100000, 1.5, None, "f64", None),
    ((32, 32, 32), "pcr_j_esa", 30, 0.9, None, "f32", None),
    ((32, 32, 32), "pcr_j_esa", 30, 0.9, None, "f64", None),
    ((32, 32, 32), "pbicgstab", 100, 1.2, "pcr_rb_esa", "f64", None),
    ((32, 32, 32), "pbicgstab", 100, 1.2, "pcr", "f64", None),
    # MAF line solvers (cz_maf.f90:442-1560), serial reference build
    ((32, 32, 32), "pcr_rb_maf", 30, 1.2, None, "f32", None),
    ((32, 32, 32), "pcr_rb_esa_maf", 20, 1.2, None, "f64", None),
    ((32, 32, 32), "pcr_maf", 20, 1.2, None, "f32", None),
    ((32, 32, 32), "pcr_eda_maf", 20, 1.2, None, "f64", None),
    ((32, 32, 32), "pcr_esa_maf", 20, 1.2, None, "f32", None),
    ((32, 32, 32), "pbicgstab_maf", 100, 1.2, "pcr_rb_maf", "f64", None),
]


def solver_cases():
    index = []
    for gsz, solver, itmax, coef, pc, prec, cli in SOLVER_CASES:
        kind = "ref_serial" if "pcr" in solver or (pc and "pcr" in pc) else "ref"
        r = O.run(gsz, solver, itmax, coef, pc, kind=kind, prec=prec, with_error=True)
        tag = f"{solver}{'_' + pc if pc else ''}_{gsz[0]}x{gsz[1]}x{gsz[2]}_{prec}"
        with open(os.path.join(HERE, f"hist_{tag}.txt"), "w") as f:
            f.write(r.history_text())
        entry = dict(tag=tag, gsz=list(gsz), solver=solver, itr_max=itmax, coef=coef, precond=pc, prec=prec,
                     iter=r.itr, res=r.res, res_str="%e" % r.res, sha256_P=hashlib.sha256(r.P.tobytes()).hexdigest(),
                     errmax=r.errmax, errloc=list(r.errloc), cli=list(cli) if cli else None)
        if cli:
            assert r.itr == cli[0] and ("%e" % r.res) == cli[1], (tag, r.itr, "%e" % r.res, cli)
        # the whole field of a few cases (every case carries sha256(P), which pins all of them bit for bit)
        if max(gsz) <= 28 or tag in ("jacobi_32x32x32_f32", "sor2sma_32x32x32_f64", "pbicgstab_jacobi_32x32x32_f64", "pcr_rb_32x32x32_f32",
                                     "psor_32x32x32_f32", "jacobi_maf_32x32x32_f32"):
            np.save(os.path.join(HERE, f"field_{tag}.npy"), r.P)
            entry["field"] = f"field_{tag}.npy"
        index.append(entry)
        print(tag, r.itr, "%e" % r.res, "errmax %e" % r.errmax, r.errloc)
    with open(os.path.join(HERE, "solver_cases.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    assert O.have("ref", "f32") and O.have("ref", "f64"), "build oracle/_ref first: make -C oracle ref"
    kernel_vectors("f32")
    kernel_vectors("f64")
    solver_cases()
