"""GPU test (-m gpu): every solver name on small, ragged grids (odd extents, extents below the tile / wave sizes of the kernels, k-extents
that are not a multiple of the vector width) against the oracle driver, bit for bit.  Catches geometry assumptions, not performance."""
import numpy as np
import pytest

from oracle import cz_oracle as O

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8, 8), (9, 7, 12), (16, 5, 33), (6, 6, 6), (33, 18, 21), (12, 40, 10)]
STATIONARY = ["jacobi", "psor", "sor2sma", "pcr_rb", "pcr_rb_esa", "pcr_j_esa", "pcr", "pcr_esa", "pcr_eda", "jacobi_maf", "psor_maf", "sor2sma_maf",
              "pcr_rb_maf", "pcr_rb_esa_maf", "pcr_maf", "pcr_eda_maf", "pcr_esa_maf"]


def _gpu(prec, gsz, solver, nit, coef, pc=None):
    from cubez_amd import CZ
    cz = CZ(prec, quiet=True)
    try:
        assert cz.setup(list(gsz) + [solver, nit, coef] + ([pc] if pc else [])) == 1
        itr = cz.solve()
        return itr, cz.history(), cz.field()
    finally:
        cz.close()


@pytest.mark.parametrize("gsz", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("solver", STATIONARY)
def test_stationary_solvers_on_ragged_grids(solver, gsz):
    prec = "f32" if (sum(gsz) + len(solver)) % 2 else "f64"
    coef = 0.8 if solver.startswith("jacobi") else 0.9 if solver == "pcr_j_esa" else 1.2
    # (where n < 3/4 * 2^pn the reference's _esa / _eda forms index past their arrays; oracle and GPU read zeros there)
    itr, hist, P = _gpu(prec, gsz, solver, 5, coef)
    o = O.run(gsz, solver, 5, coef, kind="oracle", prec=prec, wide=True)
    assert itr == o.itr
    assert P.tobytes() == o.P.tobytes(), (solver, gsz, prec)
    assert np.allclose(hist, [r for _, r in o.history], rtol=1e-10, atol=0)


@pytest.mark.parametrize("gsz", [(9, 7, 12), (16, 16, 16)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("pc", ["none", "jacobi", "psor", "sor2sma", "pcr_rb", "pcr", "pcr_rb_esa", "pcr_eda"])
def test_bicgstab_preconditioners_on_small_grids(pc, gsz):
    coef = 0.8 if pc == "jacobi" else 1.2
    itr, hist, P = _gpu("f64", gsz, "pbicgstab", 6, coef, pc)
    o = O.run(gsz, "pbicgstab", 6, coef, pc, kind="oracle", prec="f64", wide=True)
    assert abs(itr - o.itr) <= (1 if pc == "none" else 0)
    m = min(len(hist), len(o.history))
    assert np.allclose(hist[:m - 1], [r for _, r in o.history][:m - 1], rtol=1e-6, atol=0)
