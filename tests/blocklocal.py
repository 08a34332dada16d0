"""The decomposed loops that are block-local by nature, restated with the oracle's kernels: the lexicographic solvers (psor, pcr, pcr_eda,
pcr_esa: every brick sweeps its own cells in lexicographic order with the ghost values of the last exchange, one one-layer exchange per
iteration) and the colour / Jacobi line solvers under a cut along k (pcr_rb, pcr_rb_esa: global (i+j) colouring, exchange after each
colour; pcr_j_esa: one exchange per iteration; every brick solves its piece of a k-line) -- the reference's MPI semantics
(cz_Poisson.cpp:95-146 PSOR, :518-611 LSOR_PCR_RB, :745-826 LSOR_PCR: local sweep, Comm_S, Comm_SUM_1), on this build's cell-ownership
bricks (cubez_amd/decomp.py).  The ranks are emulated one after the other in this process: within an
iteration no brick reads what another brick writes, so the order does not matter.  TEST INFRASTRUCTURE."""
import numpy as np

from cubez_amd import decomp
from oracle import cz_oracle as O


def run(gsz, div, solver, nit, coef, prec):
    k = O.Kernels("oracle", prec)
    R = k.real
    world = div[0] * div[1] * div[2]
    bricks = [decomp.decompose(gsz, div, world, r) for r in range(world)]
    pitch = R(1.0 / float(R(gsz[2] - 1)))
    cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=R)
    Pg = k.alloc(gsz)
    k.bc_k(gsz, Pg, pitch, np.zeros(3, dtype=R), [-1] * 6)  # Dirichlet data from GLOBAL indices, as the GPU driver does
    st = []
    for r, b in enumerate(bricks):
        size, head, nID = b["size"], b["head"], b["nID"]
        (ni, nj, nk), (hi, hj, hk) = size, head
        P, RHS = k.alloc(size), k.alloc(size)
        own = (slice(2, 2 + nj), slice(2, 2 + ni), slice(2, 2 + nk))
        glob = (slice(hj + 1, hj + 1 + nj), slice(hi + 1, hi + 1 + ni), slice(hk + 1, hk + 1 + nk))
        P[own] = Pg[glob]
        RHS[own] = Pg[glob]
        idx = decomp.inner_range(size, nID)
        d = dict(size=size, head=head, nID=nID, idx=idx, P=P, RHS=RHS, msgs=decomp.exchange_boxes(size, div, r, depth=1, edges=False))
        if solver.startswith("pcr"):
            d["MSK"] = k.alloc(size)
            k.imask_k(d["MSK"], size, idx)
            d["pn"] = O.get_num_stage(idx[5] - idx[4] + 1)
            d["WRK"] = k.alloc(size)
        st.append(d)

    def halo(name):
        sent = [{tuple(m["dir"]): np.ascontiguousarray(d[name][m["send"]]) for m in d["msgs"]} for d in st]
        for d in st:
            for m in d["msgs"]:
                d[name][m["recv"]] = sent[m["peer"]][tuple(-v for v in m["dir"])]

    halo("P")
    halo("RHS")
    npts = sum(float(d["idx"][1] - d["idx"][0] + 1) * (d["idx"][3] - d["idx"][2] + 1) * (d["idx"][5] - d["idx"][4] + 1) for d in st)
    hist = []
    for _ in range(nit):
        tot = 0.0
        for d in st:
            w = np.zeros(1)
            if solver == "psor":
                k.psor(d["P"], d["size"], d["idx"], cf, coef, d["RHS"], wide=w)
            elif solver in ("pcr_rb", "pcr_rb_esa"):
                continue  # colour by colour below
            else:
                k.pcr_sweep_wide(solver, d["size"], d["idx"], d["pn"], 0, d["P"], d["MSK"], d["RHS"], d["WRK"], coef, w)
                if solver == "pcr_j_esa":  # all columns from the old field into WRK, then back (cz_Poisson.cpp:1061-1068)
                    i = d["idx"]
                    inner = (slice(i[2] + 1, i[3] + 2), slice(i[0] + 1, i[1] + 2), slice(i[4] + 1, i[5] + 2))
                    d["P"][inner] = d["WRK"][inner]
            tot += float(w[0])
        if solver in ("pcr_rb", "pcr_rb_esa"):
            name = "pcr_rb_2x2" if solver == "pcr_rb" else "pcr_rb_esa"
            for color in (0, 1):
                for d in st:
                    w = np.zeros(1)
                    k.pcr_sweep_wide(name, d["size"], d["idx"], d["pn"], (color + d["head"][0] + d["head"][1]) & 1, d["P"], d["MSK"], d["RHS"], None,
                                     coef, w)
                    tot += float(w[0])
                halo("P")
        else:
            halo("P")
        hist.append(float(np.sqrt(tot / npts)))
    g = 2
    G = np.zeros((gsz[1] + 4, gsz[0] + 4, gsz[2] + 4), dtype=R)
    for d in st:
        (ni, nj, nk), (hi, hj, hk) = d["size"], d["head"]
        G[g + hj - 1:g + hj - 1 + nj, g + hi - 1:g + hi - 1 + ni, g + hk - 1:g + hk - 1 + nk] = d["P"][g:g + nj, g:g + ni, g:g + nk]
    return hist, G
