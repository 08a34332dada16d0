"""GPU parity tests (run with -m gpu on an MI355X): every drop-in kernel of libczhip, called through
the C-ABI with the reference's argument conventions, against the CPU oracle on the same inputs.

Bar (SURVEY.md 8c tolerance chain): fields BIT-EXACT (kernels are built with -ffp-contract=off and
IEEE division); residuals / dot products are accumulated in double on the GPU in a fixed tree order,
so they are compared with the oracle's double accumulation of the same REAL-rounded terms to
1e-12 relative (pure summation-order noise in double), and with the reference's REAL-accumulated
value to the order-of-summation tolerance of the precision (1e-3 FP32 / 1e-10 FP64)."""
import os

import numpy as np
import pytest

from oracle import cz_oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RTOL_WIDE = 1e-12


def _beq(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def _hip(prec):
    from cubez_amd import CzHip
    return CzHip(prec)


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def _real_tol(prec):
    return 1e-3 if prec == "f32" else 1e-10


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_golden_vectors(prec):
    """the reference-generated vectors of tests/golden/kernels_*.npz through the HIP library."""
    g = np.load(os.path.join(GOLDEN, f"kernels_{prec}.npz"))
    h = _hip(prec)
    sz, idx, cf, omg = list(g["sz"]), list(g["idx"]), g["cf"], float(g["omg"])
    p, b, q, x, y, z = (g[n] for n in ("in_p", "in_b", "in_q", "in_x", "in_y", "in_z"))
    dp, db, dq, dx, dy = (h.alloc(sz, a) for a in (p, b, q, x, y))

    pj, wk = h.alloc(sz, p), h.alloc(sz, np.zeros_like(p))
    res = h.jacobi(pj, sz, idx, cf, omg, db, wk, res=0.25)
    assert _beq(pj.get(), g["jacobi_p"]) and _beq(wk.get(), g["jacobi_wk2"])
    assert h.last_flop == float(g["jacobi_flop"])
    assert _rel(res, float(g["jacobi_res"])) < _real_tol(prec)

    for ofst in (0, 1):
        ps, r = h.alloc(sz, p), 0.0
        for color in (0, 1):
            r = h.psor2sma_core(ps, sz, idx, cf, ofst, color, omg, db, res=r)
            assert _beq(ps.get(), g[f"rb{ofst}_p_c{color}"])
            assert _rel(r, float(g[f"rb{ofst}_res_c{color}"])) < _real_tol(prec)

    out = h.alloc(sz, np.zeros_like(p))
    h.blas_calc_ax(out, dp, sz, idx, cf)
    assert _beq(out.get(), g["calc_ax"])
    out.put(np.zeros_like(p))
    h.blas_calc_rk(out, dp, db, sz, idx, cf)
    assert _beq(out.get(), g["calc_rk"])
    assert _rel(float(h.blas_dot1(dp, sz, idx)), float(g["dot1"])) < _real_tol(prec)
    assert _rel(float(h.blas_dot2(dp, dq, sz, idx)), float(g["dot2"])) < _real_tol(prec)
    zt = h.alloc(sz, z)
    h.blas_triad(zt, dx, dy, -0.37, sz, idx)
    assert _beq(zt.get(), g["triad"])
    pb = h.alloc(sz, p)
    h.blas_bicg_1(pb, dx, dq, 0.61, -1.3, sz, idx)
    assert _beq(pb.get(), g["bicg_1"])
    zb = h.alloc(sz, z)
    h.blas_bicg_2(zb, dx, dy, 0.45, -0.77, sz, idx)
    assert _beq(zb.get(), g["bicg_2"])
    c = h.alloc(sz, p)
    h.blas_clear(c, sz)
    assert _beq(c.get(), g["clear"])
    d = h.alloc(sz, np.zeros_like(p))
    h.blas_copy(d, dp, sz)
    assert _beq(d.get(), g["copy"])
    for tag, nid, org in (("all", [-1] * 6, [0.0, 0.0, 0.0]), ("mix", [3, -1, -1, 5, -1, 2], [0.25, 0.5, 0.0])):
        pc = h.alloc(sz, p)
        h.bc_k(sz, pc, 1.0 / (sz[2] - 1), org, nid)
        assert _beq(pc.get(), g[f"bc_{tag}"])


# (NI, NJ, NK), idx or None for the single-domain inner box
BOXES = [
    ((5, 4, 6), None),               # tiny, V=1 path (NK+4 = 10)
    ((9, 12, 7), None),              # odd sizes, V=1 (f32) path
    ((16, 8, 32), None),             # NK+4 = 36: vector path, single segment
    ((40, 36, 60), None),            # several segments and chunks
    ((33, 70, 124), None),           # NK+4 = 128
    ((24, 20, 28), (1, 24, 1, 20, 1, 28)),   # interior rank of a decomposition: box = all owned cells
    ((24, 20, 28), (1, 23, 2, 20, 1, 27)),   # mixed physical / interior faces
    ((12, 10, 20), (3, 2, 2, 9, 2, 19)),     # empty range (ied < ist): nothing may change
    ((130, 20, 252), None),          # long rows: R = 64
]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", BOXES, ids=[f"{b[0][0]}x{b[0][1]}x{b[0][2]}{'' if b[1] is None else '_idx'+str(i)}" for i, b in enumerate(BOXES)])
def test_random_boxes_vs_oracle(prec, box):
    (ni, nj, nk), idx = box
    sz = [ni, nj, nk]
    if idx is None:
        idx = [2, ni - 1, 2, nj - 1, 2, nk - 1]
    idx = list(idx)
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni * 10007 + nj * 101 + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    cf = rng.uniform(0.5, 1.5, 7).astype(R)
    cf[6] = 6.2
    p, b, q = (rng.uniform(-1, 1, shape).astype(R) for _ in range(3))
    dp, db, dq = h.alloc(sz, p), h.alloc(sz, b), h.alloc(sz, q)
    sentinel = rng.uniform(-1, 1, shape).astype(R)  # outputs must keep their non-inner elements

    # jacobi (drop-in semantics: result in p and wk2, res accumulated)
    a1, w1, wide = p.copy(), sentinel.copy(), np.zeros(1)
    r1 = ko.jacobi(a1, sz, idx, cf, 0.9, b, w1, res=1.5, wide=wide)
    a2, w2 = h.alloc(sz, p), h.alloc(sz, sentinel)
    r2 = h.jacobi(a2, sz, idx, cf, 0.9, db, w2, res=1.5)
    assert _beq(a2.get(), a1) and _beq(w2.get(), w1)
    assert h.last_flop == ko.last_flop
    if wide[0] > 0:
        assert _rel(r2 - 1.5, wide[0]) < RTOL_WIDE * 10
        assert _rel(r2, r1) < _real_tol(prec)
    else:
        assert r2 == 1.5

    # red-black, both offsets, both colours
    for ofst in (0, 1):
        a1, a2, r1, r2, wide = p.copy(), h.alloc(sz, p), 0.0, 0.0, np.zeros(1)
        for color in (0, 1):
            r1 = ko.psor2sma_core(a1, sz, idx, cf, ofst, color, 1.3, b, res=r1, wide=wide)
            r2 = h.psor2sma_core(a2, sz, idx, cf, ofst, color, 1.3, db, res=r2)
            assert _beq(a2.get(), a1), (ofst, color)
            if wide[0] > 0:
                assert _rel(r2, wide[0]) < RTOL_WIDE * 10

    # SpMV / residual
    o1, o2 = sentinel.copy(), h.alloc(sz, sentinel)
    ko.blas_calc_ax(o1, p, sz, idx, cf), h.blas_calc_ax(o2, dp, sz, idx, cf)
    assert _beq(o2.get(), o1)
    o1, o2 = sentinel.copy(), h.alloc(sz, sentinel)
    ko.blas_calc_rk(o1, p, b, sz, idx, cf), h.blas_calc_rk(o2, dp, db, sz, idx, cf)
    assert _beq(o2.get(), o1)

    # dots: GPU accumulates in double, rounds once to REAL
    wide = np.zeros(1)
    d1 = ko.blas_dot1(p, sz, idx, wide=wide)
    d2 = h.blas_dot1(dp, sz, idx)
    assert d2 == R(wide[0]) or _rel(float(d2), wide[0]) < (1.3e-7 if prec == "f32" else RTOL_WIDE * 10)
    assert _rel(float(d2), float(d1)) < _real_tol(prec) or float(d1) == 0.0
    wide = np.zeros(1)
    d1 = ko.blas_dot2(p, q, sz, idx, wide=wide)
    d2 = h.blas_dot2(dp, dq, sz, idx)
    assert d2 == R(wide[0]) or abs(float(d2) - wide[0]) <= (1e-6 if prec == "f32" else 1e-11) * max(1.0, abs(wide[0]))

    # axpy family
    o1, o2 = sentinel.copy(), h.alloc(sz, sentinel)
    ko.blas_triad(o1, p, b, -0.3, sz, idx), h.blas_triad(o2, dp, db, -0.3, sz, idx)
    assert _beq(o2.get(), o1)
    o1, o2 = q.copy(), h.alloc(sz, q)
    ko.blas_bicg_1(o1, p, b, 0.3, 0.7, sz, idx), h.blas_bicg_1(o2, dp, db, 0.3, 0.7, sz, idx)
    assert _beq(o2.get(), o1)
    ko.blas_bicg_2(o1, p, b, 0.3, 0.7, sz, idx), h.blas_bicg_2(o2, dp, db, 0.3, 0.7, sz, idx)
    assert _beq(o2.get(), o1)

    # boundary condition with a mixed neighbour table and non-zero origin
    for nid in ([-1] * 6, [0, 1, -1, -1, 2, -1], [-1, 4, 2, -1, -1, 7]):
        o1, o2 = p.copy(), h.alloc(sz, p)
        ko.bc_k(sz, o1, 0.125, [0.1, 0.2, 0.3], nid), h.bc_k(sz, o2, 0.125, [0.1, 0.2, 0.3], nid)
        assert _beq(o2.get(), o1)
    for a in (dp, db, dq):
        a.free()


TUNINGS = [(256, 1, 0, 0), (256, 1, 3, 1), (256, 2, 0, 0), (256, 2, 5, 1), (256, 4, 0, 0), (256, 4, 2, 1),
           (512, 1, 0, 1), (512, 2, 7, 0), (512, 4, 0, 1), (1024, 1, 0, 0), (1024, 2, 4, 1)]


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_all_tunings_bit_identical(prec):
    """every compiled (threads, vectors/thread, chunk, prefetch) variant of the sweep gives the same bits."""
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    sz = [70, 45, 124]
    idx = [2, 69, 2, 44, 2, 123]
    rng = np.random.default_rng(99)
    shape = (sz[1] + 4, sz[0] + 4, sz[2] + 4)
    cf = np.array([1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3], dtype=R)
    p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    db = h.alloc(sz, b)
    a1, w1, wide = p.copy(), np.zeros_like(p), np.zeros(1)
    ko.jacobi(a1, sz, idx, cf, 0.8, b, w1, wide=wide)
    s1 = p.copy()
    for color in (0, 1):
        ko.psor2sma_core(s1, sz, idx, cf, 0, color, 1.4, b)
    try:
        for (tb, m, tj, pf) in TUNINGS:
            assert h.set_tuning(tb, m, tj, pf), (tb, m, tj, pf)
            a2, w2 = h.alloc(sz, p), h.alloc(sz, np.zeros_like(p))
            r2 = h.jacobi(a2, sz, idx, cf, 0.8, db, w2)
            assert _beq(a2.get(), a1), (tb, m, tj, pf)
            assert _rel(r2, wide[0]) < RTOL_WIDE * 10
            s2 = h.alloc(sz, p)
            for color in (0, 1):
                h.psor2sma_core(s2, sz, idx, cf, 0, color, 1.4, db)
            assert _beq(s2.get(), s1), (tb, m, tj, pf)
            for a in (a2, w2, s2):
                a.free()
    finally:
        h.set_tuning(256, 2, 0, 1)


def test_residual_is_run_to_run_deterministic():
    h = _hip("f32")
    sz = [64, 64, 60]
    idx = [2, 63, 2, 63, 2, 59]
    rng = np.random.default_rng(5)
    shape = (sz[1] + 4, sz[0] + 4, sz[2] + 4)
    p, b = (rng.uniform(-1, 1, shape).astype(np.float32) for _ in range(2))
    cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=np.float32)
    db = h.alloc(sz, b)
    vals = set()
    for _ in range(5):
        a, w = h.alloc(sz, p), h.alloc(sz, np.zeros_like(p))
        vals.add(h.jacobi(a, sz, idx, cf, 0.8, db, w))
        a.free(), w.free()
    assert len(vals) == 1


T2_BOXES = [((40, 36, 60), None), ((33, 70, 124), None), ((130, 20, 252), None), ((70, 45, 124), None),
            ((24, 20, 28), (1, 24, 1, 20, 1, 28)), ((64, 9, 60), None), ((96, 40, 508), None),
            # row lengths that are no multiple of the vector width (round 3: the pass takes them; nk + 4 = 65, 127, 130, 257, 63, 511)
            ((40, 36, 61), None), ((33, 50, 123), None), ((70, 20, 126), None), ((29, 31, 253), None),
            ((24, 20, 59), (1, 24, 1, 20, 1, 59)), ((31, 23, 507), None),
            # rows beyond what a segment of whole rows holds (round 4: the pass cuts k into windows; VERDICT r3 missing 2): 1 104 and 2 104 elements
            ((9, 7, 1100), None), ((7, 6, 2100), None)]
T2_TUNINGS = [(512, 2, 32), (512, 2, 5), (512, 2, 7), (512, 2, 16), (-2, 2, 0), (1024, 2, 16), (1024, 2, 11), (1024, 2, 4)]  # (-2, 2, 0): shape and chunk chosen by the library


@pytest.fixture(params=[-1, 6, 17], ids=["rule", "win6", "win17"])
def kwin(request):
    """k windows of the two-stage pass (Geom2): the launcher's rule (whole rows where a segment of them is a decent share of a workgroup, else
    windows of 64 vectors), and windows of 6 / 17 vectors forced on every box -- 3 to 90 windows per row, partial last windows, windows that
    start at no multiple of anything.  Results must not depend on it."""
    hs = [_hip(p) for p in ("f32", "f64")]
    for h in hs:
        h.lib.czhip_set_pair_window(request.param)
    yield request.param
    for h in hs:
        h.lib.czhip_set_pair_window(-1)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", T2_BOXES, ids=[f"{b[0][0]}x{b[0][1]}x{b[0][2]}{'' if b[1] is None else '_idx'}" for b in T2_BOXES])
def test_two_fused_sweeps_equal_two_oracle_sweeps(prec, box, kwin):
    """czhip_jacobi2_async (temporal blocking) == two applications of the oracle's jacobi, bit for bit; both residuals."""
    (ni, nj, nk), idx = box
    sz = [ni, nj, nk]
    idx = list(idx) if idx else [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 7 * nj + 13 * nk)
    shape = (nj + 4, ni + 4, nk + 4)
    cf = rng.uniform(0.5, 1.5, 7).astype(R)
    cf[6] = 6.2
    p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    a1, w1, r = p.copy(), np.zeros_like(p), []
    for _ in range(2):
        wide = np.zeros(1)
        ko.jacobi(a1, sz, idx, cf, 0.9, b, w1, wide=wide)
        r.append(wide[0])
    du, db = h.alloc(sz, p), h.alloc(sz, b)
    launched = 0
    try:
        for (tb, mv, tj) in T2_TUNINGS:
            assert h.set_tuning2(tb, mv, tj, 1)
            dw = h.alloc(sz, p)  # ping-pong partner: same non-inner elements as the input
            ok, r1, r2 = h.jacobi2(du, dw, db, sz, idx, cf, 0.9)
            if ok:
                launched += 1
                assert _beq(dw.get(), a1), (tb, mv, tj)
                assert _beq(du.get(), p)  # the input is never modified
                assert _rel(r1, r[0]) < RTOL_WIDE * 10 and _rel(r2, r[1]) < RTOL_WIDE * 10, (tb, mv, tj)
            dw.free()
    finally:
        h.set_tuning2(-2, 2, 0, 1)
    if nk + 4 >= 64:
        assert launched > 0


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", T2_BOXES, ids=[f"{b[0][0]}x{b[0][1]}x{b[0][2]}{'' if b[1] is None else '_idx'}" for b in T2_BOXES])
def test_fused_red_black_iteration_equals_two_colour_calls(prec, box, kwin):
    """czhip_rbsor2_async (both colours in one pass, out of place) == psor2sma_core colour 0 + colour 1 of the oracle."""
    (ni, nj, nk), idx = box
    sz = [ni, nj, nk]
    idx = list(idx) if idx else [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(3 * ni + 5 * nj + 11 * nk)
    shape = (nj + 4, ni + 4, nk + 4)
    cf = rng.uniform(0.5, 1.5, 7).astype(R)
    cf[6] = 6.2
    p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    du, db = h.alloc(sz, p), h.alloc(sz, b)
    launched = 0
    try:
        for ofst in (0, 1):
            a1, wide = p.copy(), np.zeros(1)
            for color in (0, 1):
                ko.psor2sma_core(a1, sz, idx, cf, ofst, color, 1.3, b, wide=wide)
            for (tb, mv, tj) in T2_TUNINGS[::2] + T2_TUNINGS[-1:]:
                assert h.set_tuning2(tb, mv, tj, 1)
                dw = h.alloc(sz, p)
                ok, r = h.rbsor2(du, dw, db, sz, idx, cf, ofst, 1.3)
                if ok:
                    launched += 1
                    assert _beq(dw.get(), a1), (ofst, tb, mv, tj)
                    assert _rel(r, wide[0]) < RTOL_WIDE * 10
                dw.free()
    finally:
        h.set_tuning2(-2, 2, 0, 1)
    if nk + 4 >= 64:
        assert launched > 0


RB4_FORMS = [(0, 0), (5, 0), (9, 3), (32, 2), (0, 7)]  # (vectors per k window, planes per chunk); 0 = the launcher's rule


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", T2_BOXES, ids=[f"{b[0][0]}x{b[0][1]}x{b[0][2]}{'' if b[1] is None else '_idx'}" for b in T2_BOXES])
def test_two_red_black_iterations_per_pass_equal_four_colour_calls(prec, box):
    """czhip_rbsor4_async (rb4_k, round 4: colour 0, 1, 0, 1 in ONE pass over memory, four stages deep, k cut into windows) == four psor2sma_core
    calls of the oracle, bit for bit, and the residuals of both iterations; both colour offsets, windows of 5 / 9 / 32 vectors and the
    launcher's own, chunks of 2, 3, 7 planes; rows that are no multiple of the vector width and rows of 1 104 / 2 104 elements included."""
    (ni, nj, nk), idx = box
    sz = [ni, nj, nk]
    idx = list(idx) if idx else [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(7 * ni + 3 * nj + 5 * nk)
    shape = (nj + 4, ni + 4, nk + 4)
    cf = rng.uniform(0.5, 1.5, 7).astype(R)
    cf[6] = 6.2
    p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    du, db = h.alloc(sz, p), h.alloc(sz, b)
    launched = 0
    try:
        for ofst in (0, 1):
            a1, r = p.copy(), []
            for it in range(2):
                wide = np.zeros(1)
                for color in (0, 1):
                    ko.psor2sma_core(a1, sz, idx, cf, ofst, color, 1.3, b, wide=wide)
                r.append(wide[0])
            for (kw, tj) in RB4_FORMS:
                assert h.lib.czhip_set_rb4(2, kw, tj) == 0  # (2: also where the launcher would leave a small box to the preloaded one-iteration pass)
                dw = h.alloc(sz, p)
                ok, r1, r2 = h.rbsor4(du, dw, db, sz, idx, cf, ofst, 1.3)
                if ok:
                    launched += 1
                    assert _beq(dw.get(), a1), (ofst, kw, tj)
                    assert _beq(du.get(), p)  # the input is never modified
                    assert _rel(r1, r[0]) < RTOL_WIDE * 10 and _rel(r2, r[1]) < RTOL_WIDE * 10, (ofst, kw, tj)
                dw.free()
    finally:
        h.lib.czhip_set_rb4(1, 0, 0)
    if idx[0] >= 2 and idx[2] >= 2:
        assert launched > 0


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", T2_BOXES, ids=[f"{b[0][0]}x{b[0][1]}x{b[0][2]}{'' if b[1] is None else '_idx'}" for b in T2_BOXES])
def test_unit_coefficient_forms_equal_the_oracle_and_the_general_forms(prec, box):
    """The reference's coefficients are c1 .. c6 = 1, dd = 6 (cz.h:169-172).  Where the six are exactly 1 the Jacobi pair and the two-iteration
    red-black pass leave the six multiplications out (offdiag_sum<UNIT>): == the oracle with those coefficients and == the general form
    (czhip_set_unit_coef(0)), bit for bit, residuals to the last bit between the two forms.  A coefficient one ULP from 1 takes the general form
    and differs from the unit result somewhere (the switch looks at the values, not at a promise)."""
    (ni, nj, nk), idx = box
    sz = [ni, nj, nk]
    idx = list(idx) if idx else [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(11 * ni + 3 * nj + 7 * nk)
    shape = (nj + 4, ni + 4, nk + 4)
    cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=R)
    p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    aj, wj = p.copy(), np.zeros_like(p)
    for _ in range(2):
        ko.jacobi(aj, sz, idx, cf, 0.9, b, wj, wide=np.zeros(1))
    ar = p.copy()
    for it in range(2):
        for color in (0, 1):
            ko.psor2sma_core(ar, sz, idx, cf, 0, color, 1.3, b, wide=np.zeros(1))
    du, db = h.alloc(sz, p), h.alloc(sz, b)
    launched = 0
    try:
        assert h.lib.czhip_set_rb4(2, 0, 0) == 0
        res = {}
        for unit in (1, 0):
            h.lib.czhip_set_unit_coef(unit)
            dw = h.alloc(sz, p)
            ok, r1, r2 = h.jacobi2(du, dw, db, sz, idx, cf, 0.9)
            if ok:
                launched += 1
                assert _beq(dw.get(), aj), unit
                res[("j", unit)] = (r1, r2)
            dw.free()
            dw = h.alloc(sz, p)
            ok, r1, r2 = h.rbsor4(du, dw, db, sz, idx, cf, 0, 1.3)
            if ok:
                launched += 1
                assert _beq(dw.get(), ar), unit
                res[("r", unit)] = (r1, r2)
            dw.free()
        for k in ("j", "r"):
            if (k, 1) in res:
                assert res[(k, 1)] == res[(k, 0)], k  # same values summed in the same order
        # one coefficient a single ULP above 1: the general form, and the oracle's result with THAT coefficient
        h.lib.czhip_set_unit_coef(1)
        cf2 = cf.copy()
        cf2[4] = np.nextafter(R(1), R(2))
        a2, w2 = p.copy(), np.zeros_like(p)
        for _ in range(2):
            ko.jacobi(a2, sz, idx, cf2, 0.9, b, w2, wide=np.zeros(1))
        dw = h.alloc(sz, p)
        ok, _, _ = h.jacobi2(du, dw, db, sz, idx, cf2, 0.9)
        if ok:
            assert _beq(dw.get(), a2)
        dw.free()
    finally:
        h.lib.czhip_set_unit_coef(1)
        h.lib.czhip_set_rb4(1, 0, 0)
    if nk + 4 >= 64:
        assert launched > 0


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", [(40, 36, 60), (33, 50, 123), (29, 31, 253), (70, 20, 126)], ids=lambda b: "x".join(map(str, b)))
@pytest.mark.parametrize("rb", [-1, 0, 1], ids=["jacobi_pair", "rb_ofst0", "rb_ofst1"])
def test_first_pass_of_a_preconditioner_solve_from_a_literal_zero(prec, box, rb, kwin):
    """czhip_jacobi2_from_zero_made_async: the start vector is not read (a literal zero) and the right-hand side is read (op 0) or made on the
    way from the operands of blas_triad_ (op 1) / blas_bicg_1_ (op 2) and stored.  == the vector update launched on its own, then the pass on a
    cleared start vector; output field and stored right-hand side bit for bit.  Rows that are no multiple of the vector width included."""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h = _hip(prec)
    R = np.float32 if prec == "f32" else np.float64
    rng = np.random.default_rng(ni + 3 * nj + 5 * nk + rb)
    shape = (nj + 4, ni + 4, nk + 4)
    cf = rng.uniform(0.5, 1.5, 7).astype(R)
    cf[6] = 6.1
    x, y, z, keep = (rng.uniform(-1, 1, shape).astype(R) for _ in range(4))
    zero = np.zeros(shape, dtype=R)
    dx, dy, dz = h.alloc(sz, x), h.alloc(sz, y), h.alloc(sz, z)
    for op in (0, 1, 2):
        # the reference sequence: the update as its own kernel, then the pass reading a cleared start vector
        db = h.alloc(sz, keep if op != 2 else z)
        if op == 1:
            h.blas_triad(db, dx, dy, -0.7, sz, idx)            # b = a*x + y
        if op == 2:
            h.blas_bicg_1(db, dx, dy, 0.6, 1.3, sz, idx)        # b = x + a*(b - bb*y), in place on a copy of z
        du, w1 = h.alloc(sz, zero), h.alloc(sz, zero)
        ok1 = h.jacobi2(du, w1, db, sz, idx, cf, 0.9)[0] if rb < 0 else h.rbsor2(du, w1, db, sz, idx, cf, rb, 1.2)[0]
        # the fused launch: b_out starts as what the update would have found there outside the inner box
        dbo, w2 = h.alloc(sz, keep if op != 2 else z), h.alloc(sz, zero)
        if op == 0:
            dbo.put(db.get())
        ok2 = h.pass_from_zero(w2, dbo, sz, idx, cf, 0.9 if rb < 0 else 1.2, op=op, x=dx, y=dy, z=dz, a=-0.7 if op == 1 else 0.6, bb=1.3, rb_ofst=rb)
        assert ok1 == ok2
        if ok2:
            assert _beq(w2.get(), w1.get()), (op, rb)
            assert _beq(dbo.get(), db.get()), (op, rb)
        elif nk + 4 >= 64:
            raise AssertionError("the pass refused a shape it is known to take")
        for d in (db, du, w1, dbo, w2):
            d.free()


def _coords(rng, n, R):
    return (np.cumsum(rng.uniform(0.5, 1.5, n + 4)).astype(R) * R(0.05)).astype(R)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_maf_golden_vectors(prec):
    """MAF flavour: reference-generated vectors on a stretched grid through the drop-in symbols."""
    g = np.load(os.path.join(GOLDEN, f"kernels_{prec}.npz"))
    h = _hip(prec)
    sz, idx, omg = list(g["sz"]), list(g["idx"]), float(g["omg"])
    p, b = g["in_p"], g["in_b"]
    xc, yc, zc = g["maf_x"], g["maf_y"], g["maf_z"]
    dp, db = h.alloc(sz, p), h.alloc(sz, b)
    pv = h.alloc(sz, g["maf_pvt_in"])
    h.search_pivot(pv, sz, idx, xc, yc, zc)
    assert _beq(pv.get(), g["maf_pvt"])
    pm, wm = h.alloc(sz, p), h.alloc(sz, np.zeros_like(p))
    res = h.jacobi_maf(pm, sz, idx, xc, yc, zc, omg, db, wm, res=0.5)
    assert _beq(pm.get(), g["maf_jacobi_p"]) and _beq(wm.get(), g["maf_jacobi_wk2"])
    assert h.last_flop == float(g["maf_jacobi_flop"])
    assert _rel(res, float(g["maf_jacobi_res"])) < _real_tol(prec)
    for ofst in (0, 1):
        ps, r = h.alloc(sz, p), 0.0
        for color in (0, 1):
            r = h.psor2sma_core_maf(ps, sz, idx, xc, yc, zc, ofst, color, omg, db, res=r)
            assert _beq(ps.get(), g[f"maf_rb{ofst}_p_c{color}"])
            assert _rel(r, float(g[f"maf_rb{ofst}_res_c{color}"])) < _real_tol(prec)
    a = h.alloc(sz, g["maf_ax_in"])
    h.calc_ax_maf(a, dp, sz, idx, xc, yc, zc, pv)
    assert _beq(a.get(), g["maf_ax"])
    a = h.alloc(sz, g["maf_rk_in"])
    h.calc_rk_maf(a, dp, db, sz, idx, xc, yc, zc, pv)
    assert _beq(a.get(), g["maf_rk"])


MAF_BOXES = [((9, 12, 7), None), ((16, 8, 32), None), ((40, 36, 60), None), ((33, 70, 124), None),
             ((24, 20, 28), (1, 24, 1, 20, 1, 28)), ((130, 20, 252), None)]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", MAF_BOXES, ids=[f"{b[0][0]}x{b[0][1]}x{b[0][2]}{'' if b[1] is None else '_idx'}" for b in MAF_BOXES])
def test_maf_random_boxes_vs_oracle(prec, box):
    (ni, nj, nk), idx = box
    sz = [ni, nj, nk]
    idx = list(idx) if idx else [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(17 * ni + 3 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    xc, yc, zc = _coords(rng, ni, R), _coords(rng, nj, R), _coords(rng, nk, R)
    p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    sentinel = rng.uniform(-1, 1, shape).astype(R)
    dp, db = h.alloc(sz, p), h.alloc(sz, b)
    pv1, pv2 = sentinel.copy(), h.alloc(sz, sentinel)
    ko.search_pivot(pv1, sz, idx, xc, yc, zc), h.search_pivot(pv2, sz, idx, xc, yc, zc)
    assert _beq(pv2.get(), pv1)
    a1, w1, wide = p.copy(), sentinel.copy(), np.zeros(1)
    ko.jacobi_maf(a1, sz, idx, xc, yc, zc, 0.9, b, w1, res=0.0, wide=wide)
    a2, w2 = h.alloc(sz, p), h.alloc(sz, sentinel)
    r2 = h.jacobi_maf(a2, sz, idx, xc, yc, zc, 0.9, db, w2, res=0.0)
    assert _beq(a2.get(), a1) and _beq(w2.get(), w1)
    assert _rel(r2, wide[0]) < RTOL_WIDE * 10
    for ofst in (0, 1):
        a1, a2, r2, wide = p.copy(), h.alloc(sz, p), 0.0, np.zeros(1)
        for color in (0, 1):
            ko.psor2sma_core_maf(a1, sz, idx, xc, yc, zc, ofst, color, 1.2, b, wide=wide)
            r2 = h.psor2sma_core_maf(a2, sz, idx, xc, yc, zc, ofst, color, 1.2, db, res=r2)
            assert _beq(a2.get(), a1), (ofst, color)
        assert _rel(r2, wide[0]) < RTOL_WIDE * 10
    o1, o2 = sentinel.copy(), h.alloc(sz, sentinel)
    ko.calc_ax_maf(o1, p, sz, idx, xc, yc, zc, pv1), h.calc_ax_maf(o2, dp, sz, idx, xc, yc, zc, pv2)
    assert _beq(o2.get(), o1)
    o1, o2 = sentinel.copy(), h.alloc(sz, sentinel)
    ko.calc_rk_maf(o1, p, b, sz, idx, xc, yc, zc, pv1), h.calc_rk_maf(o2, dp, db, sz, idx, xc, yc, zc, pv2)
    assert _beq(o2.get(), o1)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", T2_BOXES, ids=[f"{b[0][0]}x{b[0][1]}x{b[0][2]}{'' if b[1] is None else '_idx'}" for b in T2_BOXES])
def test_maf_two_stage_pass_equals_two_oracle_sweeps(prec, box, kwin):
    """czhip_pair_maf_async (VERDICT r1 "missing" 6: the MAF flavour of the two-stage pass, cz_maf.f90:131-438) on stretched grids ==
    two jacobi_maf sweeps / colour 0 + colour 1 of psor2sma_core_maf of the oracle, bit for bit, in every kernel shape."""
    (ni, nj, nk), idx = box
    sz = [ni, nj, nk]
    idx = list(idx) if idx else [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(5 * ni + 7 * nj + 3 * nk)
    shape = (nj + 4, ni + 4, nk + 4)
    xc, yc, zc = _coords(rng, ni, R), _coords(rng, nj, R), _coords(rng, nk, R)
    p, b = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    a1, w1, r = p.copy(), np.zeros_like(p), []
    for _ in range(2):
        wide = np.zeros(1)
        ko.jacobi_maf(a1, sz, idx, xc, yc, zc, 0.9, b, w1, res=0.0, wide=wide)
        r.append(wide[0])
    rb = {}
    for ofst in (0, 1):
        a, wide = p.copy(), np.zeros(1)
        for color in (0, 1):
            ko.psor2sma_core_maf(a, sz, idx, xc, yc, zc, ofst, color, 1.2, b, wide=wide)
        rb[ofst] = (a, wide[0])
    du, db = h.alloc(sz, p), h.alloc(sz, b)
    launched = 0
    try:
        for (tb, mv, tj) in T2_TUNINGS[2:6]:
            assert h.set_tuning2(tb, mv, tj, 1)
            dw = h.alloc(sz, p)
            ok, r1, r2 = h.pair_maf(du, dw, db, sz, idx, xc, yc, zc, 0.9)
            if ok:
                launched += 1
                assert _beq(dw.get(), a1), (tb, mv, tj)
                assert _beq(du.get(), p)
                assert _rel(r1, r[0]) < RTOL_WIDE * 10 and _rel(r2, r[1]) < RTOL_WIDE * 10
                for ofst in (0, 1):
                    ok2, s1, _ = h.pair_maf(du, dw, db, sz, idx, xc, yc, zc, 1.2, rb_ofst=ofst)
                    assert ok2 and _beq(dw.get(), rb[ofst][0]), (ofst, tb, mv, tj)
                    assert _rel(s1, rb[ofst][1]) < RTOL_WIDE * 10
            dw.free()
    finally:
        h.set_tuning2(-2, 2, 0, 1)
    if nk + 4 >= 64 and idx[0] >= 2:
        assert launched > 0


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_pcr_rb_golden_vectors(prec):
    """line SOR by parallel cyclic reduction: vectors from the reference's serial build."""
    g = np.load(os.path.join(GOLDEN, f"kernels_{prec}.npz"))
    h = _hip(prec)
    for (ni, nj, nk) in ((9, 8, 13), (12, 10, 37), (6, 7, 64)):
        sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
        tag = f"pcr_{ni}x{nj}x{nk}"
        x, rhs = h.alloc(sz, g[tag + "_x_in"]), h.alloc(sz, g[tag + "_rhs"])
        msk = h.alloc(sz, g[tag + "_x_in"])  # any content: imask_k_ overwrites the whole array
        h.imask_k(msk, sz, idx)
        assert _beq(msk.get(), g[tag + "_msk"])
        pn = O.get_num_stage(idx[5] - idx[4] + 1)
        r = 0.0
        for color in (0, 1):
            r = h.pcr_rb(sz, idx, pn, 0, color, x, msk, rhs, 1.1, res=r)
            assert _beq(x.get(), g[tag + f"_x_c{color}"]), (tag, color)
            assert _rel(r, float(g[tag + f"_res_c{color}"])) < 1e-12
        assert h.last_flop == float(g[tag + "_flop"])


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", [(40, 36, 60), (33, 21, 124), (20, 9, 252), (17, 12, 510), (6, 5, 1020)],
                         ids=lambda b: "x".join(map(str, b)))
def test_pcr_rb_random_boxes_vs_oracle(prec, box):
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    x1, dx, dm, dr = x0.copy(), h.alloc(sz, x0), h.alloc(sz, msk), h.alloc(sz, rhs)
    r1 = r2 = 0.0
    for it in range(2):
        for color in (0, 1):
            r1 = ko.pcr_rb(sz, idx, pn, 0, color, x1, msk, rhs, 1.3, res=r1)
            r2 = h.pcr_rb(sz, idx, pn, 0, color, dx, dm, dr, 1.3, res=r2)
            assert _beq(dx.get(), x1), (it, color)
    assert _rel(r2, r1) < 1e-12


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("kind", ["minus_zero", "fractions"])
def test_pcr_rb_mask_with_other_values_than_zero_and_one(prec, kind):
    """The mask is data, not a flag: -0.0 (same value, other bits: the sign reaches the field) and arbitrary factors, here and there in the
    box, vs the oracle.  (Written for a form of pcr_line_reg_k that kept a +0.0 / 1.0 mask as bits between the source term and the relaxation
    instead of reading it twice -- 17 % fewer bytes read, no time gained, removed: profiles/r03/pcr_rb_what_bounds_it.txt.)"""
    ni, nj, nk = 21, 14, 124
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(5)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pick = rng.uniform(0, 1, shape) < 0.02
    if kind == "minus_zero":
        msk[pick & (msk == 0)] = R(-0.0)
        msk[pick & (msk == 1)] = R(-0.0)
    else:
        msk[pick] = rng.uniform(-2, 2, shape).astype(R)[pick]
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    x1, dx, dm, dr = x0.copy(), h.alloc(sz, x0), h.alloc(sz, msk), h.alloc(sz, rhs)
    r1 = r2 = 0.0
    for color in (0, 1, 0, 1):
        r1 = ko.pcr_rb(sz, idx, pn, 0, color, x1, msk, rhs, 1.3, res=r1)
        r2 = h.pcr_rb(sz, idx, pn, 0, color, dx, dm, dr, 1.3, res=r2)
        assert _beq(dx.get(), x1), color
    assert _rel(r2, r1) < 1e-12


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("rb", [-1, 0, 1], ids=["jacobi_pair", "rb_ofst0", "rb_ofst1"])
def test_pair_split_equals_unsplit(prec, rb):
    """shell slabs + interior (what a decomposed brick launches so that the exchange overlaps the interior)
    == the unsplit fused pass, bit for bit, for every pattern of rank-internal faces."""
    h = _hip(prec)
    R = np.float32 if prec == "f32" else np.float64
    ni, nj, nk = 28, 22, 36
    sz = [ni, nj, nk]
    rng = np.random.default_rng(77 + rb)
    shape = (nj + 4, ni + 4, nk + 4)
    u0, b0 = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    w0 = rng.uniform(-1, 1, shape).astype(R)  # what must survive outside the output box
    cf = [1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3]
    du, db = h.alloc(sz, u0), h.alloc(sz, b0)
    patterns = [[0] * 6, [1] * 6, [0, 1, 1, 0, 0, 1], [1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 1], [1, 1, 0, 0, 0, 0]]
    n = [ni, ni, nj, nj, nk, nk]
    for pat in patterns:
        nID = [(3 if v else -1) for v in pat]
        idx = [(1 if v else 2) if f % 2 == 0 else (n[f] if v else n[f] - 1) for f, v in enumerate(pat)]
        idx1 = [idx[f] + ((1 if f % 2 else -1) if v else 0) for f, v in enumerate(pat)]
        dw1, dw2 = h.alloc(sz, w0), h.alloc(sz, w0)
        if rb < 0:
            ok1, r1a, r1b = h.jacobi2(du, dw1, db, sz, idx, cf, 0.8, idx1=idx1)
        else:
            ok1, r1a = h.rbsor2(du, dw1, db, sz, idx, cf, rb, 1.3, idx1=idx1)
            r1b = 0.0
        assert ok1
        ok2, r2a, r2b = h.pair_split(du, dw2, db, sz, idx, idx1, nID, cf, 0.8 if rb < 0 else 1.3, rb_ofst=rb)
        assert ok2 == any(pat)
        if not ok2:
            continue
        assert _beq(dw2.get(), dw1.get()), pat
        assert _rel(r2a, r1a) < 1e-12 and (rb >= 0 or _rel(r2b, r1b) < 1e-12)
        assert du.get().tobytes() == u0.tobytes()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_psor_golden_vectors(prec):
    """lexicographic point SOR: vectors from ONE thread of the reference."""
    g = np.load(os.path.join(GOLDEN, f"kernels_{prec}.npz"))
    h = _hip(prec)
    sz, idx, cf, omg = list(g["sz"]), list(g["idx"]), g["cf"], float(g["omg"])
    p, b = h.alloc(sz, g["in_p"]), h.alloc(sz, g["in_b"])
    r = h.psor(p, sz, idx, cf, omg, b, res=0.125)
    assert _beq(p.get(), g["psor_p"]) and _rel(r, float(g["psor_res"])) < (1e-3 if prec == "f32" else 1e-12)
    assert h.last_flop == float(g["psor_flop"])
    p = h.alloc(sz, g["in_p"])
    r = h.psor_maf(p, sz, idx, g["maf_x"], g["maf_y"], g["maf_z"], omg, b, res=0.125)
    assert _beq(p.get(), g["maf_psor_p"]) and _rel(r, float(g["maf_psor_res"])) < (1e-3 if prec == "f32" else 1e-12)
    assert h.last_flop == float(g["maf_psor_flop"])


@pytest.mark.parametrize("form", [1, 0], ids=["one_launch", "tile_hyperplanes"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", [(16, 16, 16), (18, 18, 18), (33, 17, 50), (70, 41, 90), (5, 64, 7), (36, 52, 20)], ids=lambda b: "x".join(map(str, b)))
def test_psor_random_boxes_vs_oracle(prec, box, form):
    """the wavefront of psor / psor_maf == the sequential loop, bit for bit, in both forms: the whole sweep in ONE launch (psor_col_k: columns of
    workgroups walking k, faces handed from column to column through memory; round 3) and a launch per tile hyperplane (psor_tile_k); tiles
    and columns that overhang the box, one and many of them, sub-boxes that do not start at 2."""
    ni, nj, nk = box
    sz = [ni, nj, nk]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    assert h.lib.czhip_set_psor(form, -1) == 0
    R = ko.real
    rng = np.random.default_rng(ni * 7 + nj * 3 + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    p0, b0 = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    cf = [1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3]
    xc, yc, zc = (np.cumsum(rng.uniform(0.5, 1.5, n + 4)).astype(R) for n in (ni, nj, nk))
    for idx in ([2, ni - 1, 2, nj - 1, 2, nk - 1], [1, ni, 1, nj, 1, nk], [3, ni - 2, 2, nj - 1, 4, nk - 1]):
        p1, dp, db = p0.copy(), h.alloc(sz, p0), h.alloc(sz, b0)
        w = np.zeros(1)
        ko.psor(p1, sz, idx, cf, 1.2, b0, wide=w)
        r = h.psor(dp, sz, idx, cf, 1.2, db)
        assert _beq(dp.get(), p1), idx
        assert _rel(r, float(w[0])) < 1e-12
        p1, dp = p0.copy(), h.alloc(sz, p0)
        w = np.zeros(1)
        ko.psor_maf(p1, sz, idx, xc, yc, zc, 1.2, b0, wide=w)
        r = h.psor_maf(dp, sz, idx, xc, yc, zc, 1.2, db)
        assert _beq(dp.get(), p1), idx
        assert _rel(r, float(w[0])) < 1e-12
        # a second sweep on the same context (the face words of the first carry another sweep number)
        ko.psor_maf(p1, sz, idx, xc, yc, zc, 1.2, b0)
        h.psor_maf(dp, sz, idx, xc, yc, zc, 1.2, db)
        assert _beq(dp.get(), p1), idx
    h.lib.czhip_set_psor(1, -1)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_psor_tiny_boxes_whose_line_streams_would_leave_the_array(prec):
    """the one-launch sweep streams every k-line with runs asked for up to two loop bodies ahead; on boxes of one or two lines
    with a handful of k those runs would end behind the array.  The launcher's guard is derived from the kernel's own constants (kPsorNS) and
    sends such boxes to the tile form -- same bits.  Arrays allocated exactly to size."""
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    assert h.lib.czhip_set_psor(1, -1) == 0
    R = ko.real
    cf = [1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3]
    for ni, nj, nk in ((1, 1, 9), (1, 1, 10), (2, 1, 12), (1, 3, 40), (3, 2, 70), (2, 2, 100)):
        sz, idx = [ni, nj, nk], [1, ni, 1, nj, 1, nk]
        rng = np.random.default_rng(ni * 100 + nj * 10 + nk)
        shape = (nj + 4, ni + 4, nk + 4)
        p0, b0 = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
        p1, dp, db = p0.copy(), h.alloc(sz, p0), h.alloc(sz, b0)
        w = np.zeros(1)
        ko.psor(p1, sz, idx, cf, 1.2, b0, wide=w)
        r = h.psor(dp, sz, idx, cf, 1.2, db)
        assert _beq(dp.get(), p1), (ni, nj, nk)
        assert _rel(r, float(w[0])) < 1e-12


def test_psor_one_launch_gives_up_instead_of_hanging():
    """Every wait of a column for the face words of the columns before it is bounded (the bound of czhip_set_pcr_lex_timeout): with the bound at
    zero the columns far from the corner give up, every workgroup leaves, the call returns with a NaN residual -- and the next sweep, with the
    normal bound, is right again."""
    import ctypes as C
    prec = "f32"
    ni, nj, nk = 200, 180, 60
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    rng = np.random.default_rng(5)
    shape = (nj + 4, ni + 4, nk + 4)
    p0, b0 = (rng.uniform(-1, 1, shape).astype(ko.real) for _ in range(2))
    cf = [1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3]
    h.lib.czhip_set_pcr_lex_timeout.restype = C.c_double
    h.lib.czhip_set_pcr_lex_timeout.argtypes = [C.c_double]
    before = h.lib.czhip_set_pcr_lex_timeout(-1.0)
    try:
        h.lib.czhip_set_psor(1, -1)
        h.lib.czhip_set_pcr_lex_timeout(0.0)
        r = h.psor(h.alloc(sz, p0), sz, idx, cf, 1.2, h.alloc(sz, b0))
        assert r != r, r  # NaN: the sweep is void
        h.lib.czhip_set_pcr_lex_timeout(before)
        p1, dp = p0.copy(), h.alloc(sz, p0)
        ko.psor(p1, sz, idx, cf, 1.2, b0)
        r = h.psor(dp, sz, idx, cf, 1.2, h.alloc(sz, b0))
        assert r == r and _beq(dp.get(), p1)
    finally:
        h.lib.czhip_set_pcr_lex_timeout(before)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", [(9, 8, 32), (20, 13, 64), (7, 11, 128), (12, 9, 40), (10, 6, 256), (5, 4, 512)],
                         ids=lambda b: "x".join(map(str, b)))
def test_pcr_variants_random_boxes_vs_oracle(prec, box):
    """pcr / pcr_esa (lexicographic, diagonal by diagonal), pcr_rb_esa (4x4 final stage), pcr_j_esa (all columns from the old
    field) == the oracle (itself pinned against the serial reference build), bit for bit.  (12,9,40): n = 38 < 3/4 * 64, where the
    reference's ESA arrays are too short and zeros are read instead."""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    for name in ("pcr", "pcr_esa", "pcr_eda", "pcr_rb_esa", "pcr_j_esa"):
        x1, dx = x0.copy(), h.alloc(sz, x0)
        for it in range(2):
            if name == "pcr_eda":
                r1, r2 = ko.pcr_eda(sz, idx, pn, x1, msk, rhs, 1.3), h.pcr_eda(sz, idx, pn, dx, dm, dr, 1.3)
            elif name == "pcr":
                r1, r2 = ko.pcr(sz, idx, pn, x1, msk, rhs, 1.3), h.pcr(sz, idx, pn, dx, dm, dr, 1.3)
            elif name == "pcr_esa":
                r1, r2 = ko.pcr_esa(sz, idx, pn, x1, msk, rhs, 1.3), h.pcr_esa(sz, idx, pn, dx, dm, dr, 1.3)
            elif name == "pcr_rb_esa":
                r1 = r2 = 0.0
                for color in (0, 1):
                    r1 = ko.pcr_rb_esa(sz, idx, pn, 0, color, x1, msk, rhs, 1.3, res=r1)
                    r2 = h.pcr_rb_esa(sz, idx, pn, 0, color, dx, dm, dr, 1.3, res=r2)
            else:
                src, wrk = np.zeros(shape, dtype=R), np.zeros(shape, dtype=R)
                r1 = ko.pcr_j_esa(sz, idx, pn, x1, msk, rhs, src, wrk, 1.3)
                r2 = h.pcr_j_esa(sz, idx, pn, dx, dm, dr, h.alloc(sz), h.alloc(sz), 1.3)
            assert _beq(dx.get(), x1), (name, it)
            assert _rel(r2, r1) < (2e-3 if prec == "f32" else 1e-11), (name, r1, r2)  # the oracle sums dp^2 in REAL here
            assert h.last_flop == ko.last_flop, name


LEX_SHAPES = [(0, 0, 1), (1, 0, 1), (1, 2, 1), (1, 0, 2), (1, 2, 2), (1, 3, 2)]  # (one launch?, groups per workgroup, rows per thread)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("lex", LEX_SHAPES, ids=lambda t: "diagonals" if not t[0] else f"one_launch_g{t[1]}_q{t[2]}")
@pytest.mark.parametrize("box", [(9, 8, 32), (20, 13, 64), (7, 11, 130), (12, 9, 40), (33, 37, 20), (5, 4, 512), (6, 1, 70), (1, 6, 70)],
                         ids=lambda b: "x".join(map(str, b)))
def test_lexicographic_line_sor_every_launch_shape_vs_oracle(prec, lex, box):
    """pcr / pcr_esa / pcr_eda (cz_solver.f90:666-878 and the _esa / _eda forms: lines in the order j outer, i inner, each seeing the new
    values of (i-1,j) and (i,j-1)) == the oracle bit for bit, whether the sweep runs as one launch per diagonal or as ONE launch with rows
    of lines handed from workgroup to workgroup (pcr_lex_wg_k) -- one or several groups of threads per workgroup, one or two rows per
    thread, strips that end in a partial one, a single row, a single line per row."""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    if ni < 3 or nj < 3:  # the inner range must exist: widen the single-row / single-line cases
        sz = [max(ni, 1) + 2, max(nj, 1) + 2, nk]
        idx = [2, ni + 1, 2, nj + 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (sz[1] + 4, sz[0] + 4, sz[2] + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    assert h.lib.czhip_set_pcr_lex(*lex) == 0
    try:
        for name in ("pcr", "pcr_esa", "pcr_eda"):
            x1, dx = x0.copy(), h.alloc(sz, x0)
            for it in range(3):
                r1, r2 = getattr(ko, name)(sz, idx, pn, x1, msk, rhs, 1.3), getattr(h, name)(sz, idx, pn, dx, dm, dr, 1.3)
                assert _beq(dx.get(), x1), (name, it)
                assert r2 == r2 and _rel(r2, r1) < (2e-3 if prec == "f32" else 1e-11), (name, r1, r2)
    finally:
        h.lib.czhip_set_pcr_lex(1, 0, 1)


def test_lexicographic_line_sor_gives_up_instead_of_hanging():
    """Every wait of a workgroup for the row above is bounded: with the bound at zero the rows far from the top give up long before their
    first line can arrive, every workgroup leaves, the call returns and the residual is NaN -- and the next sweep, with the normal bound,
    is right again (bit for bit the launch-per-diagonal result)."""
    import ctypes as C
    prec = "f32"
    ni, nj, nk = 300, 300, 66
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(7)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    h.lib.czhip_set_pcr_lex_timeout.restype = C.c_double
    h.lib.czhip_set_pcr_lex_timeout.argtypes = [C.c_double]
    before = h.lib.czhip_set_pcr_lex_timeout(-1.0)
    assert before > 0.1
    try:
        assert h.lib.czhip_set_pcr_lex(1, 0, 1) == 0
        h.lib.czhip_set_pcr_lex_timeout(0.0)
        r = h.pcr(sz, idx, pn, h.alloc(sz, x0), dm, dr, 1.3)
        assert r != r, r  # NaN: the sweep is void
        h.lib.czhip_set_pcr_lex_timeout(before)
        dx = h.alloc(sz, x0)
        r2 = h.pcr(sz, idx, pn, dx, dm, dr, 1.3)
        h.lib.czhip_set_pcr_lex(0, 0, 1)
        dy = h.alloc(sz, x0)
        r3 = h.pcr(sz, idx, pn, dy, dm, dr, 1.3)
        assert r2 == r2 and _beq(dx.get(), dy.get())
        assert _rel(r2, r3) < 1e-12
    finally:
        h.lib.czhip_set_pcr_lex_timeout(before)
        h.lib.czhip_set_pcr_lex(1, 0, 1)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", [(9, 8, 32), (12, 9, 40), (7, 11, 130), (5, 4, 700)], ids=lambda b: "x".join(map(str, b)))
def test_line_sor_literal_form_every_variant_vs_oracle(prec, box):
    """czhip_set_pcr_mode(0, .): every line-SOR variant through the literal per-line kernel (a, c and d of a line reduced in LDS, 2x2 or 4x4
    final systems from the line's own coefficients) -- the form that takes over when the coefficient table of a long line does not fit
    LDS (FP64 lines beyond ~640 unknowns with the 4x4 final stage) -- == the oracle, bit for bit."""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    assert h.lib.czhip_set_pcr_mode(0, 0) == 0
    try:
        for name in ("pcr", "pcr_esa", "pcr_eda", "pcr_rb_esa", "pcr_rb"):
            x1, dx = x0.copy(), h.alloc(sz, x0)
            for it in range(2):
                if name.startswith("pcr_rb"):
                    r1 = r2 = 0.0
                    for color in (0, 1):
                        r1 = getattr(ko, name)(sz, idx, pn, 0, color, x1, msk, rhs, 1.3, res=r1)
                        r2 = getattr(h, name)(sz, idx, pn, 0, color, dx, dm, dr, 1.3, res=r2)
                else:
                    r1, r2 = getattr(ko, name)(sz, idx, pn, x1, msk, rhs, 1.3), getattr(h, name)(sz, idx, pn, dx, dm, dr, 1.3)
                assert _beq(dx.get(), x1), (name, it)
                assert _rel(r2, r1) < (2e-3 if prec == "f32" else 1e-11), (name, r1, r2)
    finally:
        h.lib.czhip_set_pcr_mode(2, 0)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("nk", [3, 4, 5, 6, 7, 10])
def test_line_sor_short_lines(prec, nk):
    """k-lines of 1..8 unknowns (pn = 1..4): fewer reduction stages than the kernels are tuned for, none at all for n <= 3."""
    ni, nj = 6, 5
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    n = nk - 2
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(n)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    names = ["pcr_rb", "pcr_j_esa"] + (["pcr_rb_esa", "pcr"] if pn >= 2 else [])
    for name in names:
        x1, dx = x0.copy(), h.alloc(sz, x0)
        if name == "pcr_rb":
            for color in (0, 1):
                ko.pcr_rb(sz, idx, pn, 0, color, x1, msk, rhs, 1.1)
                h.pcr_rb(sz, idx, pn, 0, color, dx, dm, dr, 1.1)
        elif name == "pcr_rb_esa":
            for color in (0, 1):
                ko.pcr_rb_esa(sz, idx, pn, 0, color, x1, msk, rhs, 1.1)
                h.pcr_rb_esa(sz, idx, pn, 0, color, dx, dm, dr, 1.1)
        elif name == "pcr":
            ko.pcr(sz, idx, pn, x1, msk, rhs, 1.1)
            h.pcr(sz, idx, pn, dx, dm, dr, 1.1)
        else:
            ko.pcr_j_esa(sz, idx, pn, x1, msk, rhs, np.zeros(shape, dtype=R), np.zeros(shape, dtype=R), 1.1)
            h.pcr_j_esa(sz, idx, pn, dx, dm, dr, h.alloc(sz), h.alloc(sz), 1.1)
        assert _beq(dx.get(), x1), (name, n, pn)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_empty_index_ranges_are_no_ops(prec):
    """ragged / empty inputs: an index range with ied < ist touches nothing and adds nothing to res (psor, pcr*, fused pair)."""
    h = _hip(prec)
    R = np.float32 if prec == "f32" else np.float64
    sz = [8, 7, 12]
    rng = np.random.default_rng(9)
    x0 = rng.uniform(-1, 1, (sz[1] + 4, sz[0] + 4, sz[2] + 4)).astype(R)
    dx, db, dm = h.alloc(sz, x0), h.alloc(sz, x0), h.alloc(sz, x0)
    idx = [5, 4, 2, 6, 2, 11]
    cf = [1, 1, 1, 1, 1, 1, 6]
    assert h.psor(dx, sz, idx, cf, 1.1, db, res=0.5) == 0.5
    assert h.pcr_rb(sz, idx, 4, 0, 0, dx, dm, db, 1.1, res=0.25) == 0.25
    assert h.pcr(sz, idx, 4, dx, dm, db, 1.1, res=0.25) == 0.25
    assert h.psor2sma_core(dx, sz, idx, cf, 0, 1, 1.1, db, res=0.125) == 0.125
    assert dx.get().tobytes() == x0.tobytes()


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("box", [(9, 8, 32), (12, 9, 40), (7, 11, 128), (6, 5, 5), (5, 4, 512)], ids=lambda b: "x".join(map(str, b)))
def test_pcr_maf_variants_random_boxes_vs_oracle(prec, box):
    """the five MAF line solvers (cz_maf.f90:442-1560) on stretched grids == the oracle (pinned against the serial reference build)"""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    xc, yc, zc = (np.cumsum(rng.uniform(0.5, 1.5, n + 4)).astype(R) for n in (ni, nj, nk))
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    for name in ("pcr_rb_maf", "pcr_rb_esa_maf", "pcr_maf", "pcr_eda_maf", "pcr_esa_maf"):
        x1, dx = x0.copy(), h.alloc(sz, x0)
        for it in range(2):
            r1 = r2 = 0.0
            w = np.zeros(1)
            for color in ((0, 1) if "_rb" in name else (0,)):
                x1w = x1.copy()
                ko.pcr_maf(name, sz, idx, pn, color, x1w, msk, rhs, xc, yc, zc, 1.3, wide=w)
                r1 = ko.pcr_maf(name, sz, idx, pn, color, x1, msk, rhs, xc, yc, zc, 1.3, res=r1)
                r2 = h.pcr_maf(name, sz, idx, pn, color, dx, dm, dr, xc, yc, zc, 1.3, res=r2)
                assert _beq(x1w, x1)
            assert _beq(dx.get(), x1), (name, it)
            assert _rel(r2, float(w[0])) < 1e-11, (name, r2, w)
            assert h.last_flop == ko.last_flop, name


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("lex", [(0, 0, 1), (1, 0, 1), (1, 2, 1)], ids=lambda t: "diagonals" if not t[0] else f"one_launch_g{t[1]}")
@pytest.mark.parametrize("box", [(9, 8, 32), (20, 13, 70), (33, 37, 20), (5, 4, 512)], ids=lambda b: "x".join(map(str, b)))
def test_lexicographic_maf_line_sor_every_launch_shape_vs_oracle(prec, lex, box):
    """pcr_maf / pcr_eda_maf / pcr_esa_maf (cz_maf.f90:1036-1560, lexicographic order) on stretched grids == the oracle bit for bit, with a
    launch per diagonal and with the whole sweep in one launch (pcr_lex_wg_k<MAF=1>: a, c and d of every line reduced together in LDS)."""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    xc, yc, zc = (np.cumsum(rng.uniform(0.5, 1.5, n + 4)).astype(R) for n in (ni, nj, nk))
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    assert h.lib.czhip_set_pcr_lex(*lex) == 0
    try:
        for name in ("pcr_maf", "pcr_eda_maf", "pcr_esa_maf"):
            x1, dx = x0.copy(), h.alloc(sz, x0)
            for it in range(3):
                w = np.zeros(1)
                ko.pcr_maf(name, sz, idx, pn, 0, x1.copy(), msk, rhs, xc, yc, zc, 1.3, wide=w)
                ko.pcr_maf(name, sz, idx, pn, 0, x1, msk, rhs, xc, yc, zc, 1.3)
                r2 = h.pcr_maf(name, sz, idx, pn, 0, dx, dm, dr, xc, yc, zc, 1.3)
                assert _beq(dx.get(), x1), (name, it)
                assert r2 == r2 and _rel(r2, float(w[0])) < 1e-11, (name, r2, w)
    finally:
        h.lib.czhip_set_pcr_lex(1, 0, 1)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_hoisted_division_is_the_ieee_division(prec):
    """The two-stage pass divides by the diagonal coefficient with the divisor's share of the IEEE expansion done once per thread
    (cz_k_fastdiv.h).  Every one of the 2^32 float numerators (a structured sample of 2^32 doubles: all sign/exponent patterns x 2^20
    mantissas) must give the bits of the ordinary `n / d`, for the benchmark's 6.0, the tests' 6.2 / 6.3, a negative and two awkward
    magnitudes; divisors near the ends of the exponent range are refused (the launchers then take the single-sweep kernels)."""
    import ctypes as C
    h = _hip(prec)
    h.lib.czhip_selftest_fastdiv.restype = C.c_longlong
    h.lib.czhip_selftest_fastdiv.argtypes = [h.creal]
    for d in (6.0, 6.2, 6.3, -6.0, 7.3e-4, 1.9e7):
        assert h.lib.czhip_selftest_fastdiv(d) == 0, d
    assert h.lib.czhip_selftest_fastdiv(3.0e38 if prec == "f32" else 1e300) == -1


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_lexicographic_line_sor_with_few_resident_workgroups(prec):
    """the one-launch sweep must finish however few of its workgroups the device keeps resident.  Four workgroups for 60 rows:
    the launcher sizes the hand-off rings so that the strip at the head of the window can always end (ring >= the whole row when it may
    count on one resident workgroup) -- same bits as the launch-per-diagonal form.  The same four workgroups with a ring forced to four
    lines cannot finish: every wait is bounded, the sweep returns, its residual is NaN."""
    import ctypes as C
    ni, nj, nk = 70, 62, 66
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(11)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    h.lib.czhip_set_pcr_lex_timeout.restype = C.c_double
    h.lib.czhip_set_pcr_lex_timeout.argtypes = [C.c_double]
    before = h.lib.czhip_set_pcr_lex_timeout(-1.0)
    try:
        assert h.lib.czhip_set_pcr_lex(1, 0, 1) == 0
        h.lib.czhip_set_pcr_lex_limits(1, 4, 0)  # four workgroups, the launcher's ring
        x1, dx = x0.copy(), h.alloc(sz, x0)
        for it in range(2):
            r1, r2 = ko.pcr(sz, idx, pn, x1, msk, rhs, 1.3), h.pcr(sz, idx, pn, dx, dm, dr, 1.3)
            assert r2 == r2 and _beq(dx.get(), x1), it
        h.lib.czhip_set_pcr_lex_limits(1, 4, 4)  # ... and a ring of four lines: 4 x 4 < 68 lines per row
        h.lib.czhip_set_pcr_lex_timeout(0.3)
        r = h.pcr(sz, idx, pn, h.alloc(sz, x0), dm, dr, 1.3)
        assert r != r, r  # NaN: the sweep is void, and the call came back
    finally:
        h.lib.czhip_set_pcr_lex_limits(0, 0, 0)
        h.lib.czhip_set_pcr_lex_timeout(before)


def _line_variants(h, ko, sz, idx, pn, x0, msk, rhs, prec, names=("pcr", "pcr_esa", "pcr_eda", "pcr_rb_esa", "pcr_rb", "pcr_j_esa"), sweeps=2):
    R = ko.real
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    for name in names:
        x1, dx = x0.copy(), h.alloc(sz, x0)
        for it in range(sweeps):
            if name.startswith("pcr_rb"):
                r1 = r2 = 0.0
                for color in (0, 1):
                    r1 = getattr(ko, name)(sz, idx, pn, 0, color, x1, msk, rhs, 1.3, res=r1)
                    r2 = getattr(h, name)(sz, idx, pn, 0, color, dx, dm, dr, 1.3, res=r2)
            elif name == "pcr_j_esa":
                src, wrk = np.zeros(x0.shape, dtype=R), np.zeros(x0.shape, dtype=R)
                r1 = ko.pcr_j_esa(sz, idx, pn, x1, msk, rhs, src, wrk, 1.3)
                r2 = h.pcr_j_esa(sz, idx, pn, dx, dm, dr, h.alloc(sz), h.alloc(sz), 1.3)
            else:
                r1, r2 = getattr(ko, name)(sz, idx, pn, x1, msk, rhs, 1.3), getattr(h, name)(sz, idx, pn, dx, dm, dr, 1.3)
            assert _beq(dx.get(), x1), (name, it)
            assert _rel(r2, r1) < (2e-3 if prec == "f32" else 1e-10), (name, r1, r2)


LONG_LINES = [("f64", (9, 8, 1400), 2), ("f64", (9, 8, 3600), 2), ("f64", (9, 8, 3600), 0), ("f64", (6, 5, 5300), 2), ("f32", (6, 5, 10500), 2),
              ("f32", (9, 8, 2600), 2), ("f32", (7, 6, 7000), 0)]


@pytest.mark.parametrize("prec,box,mode", LONG_LINES, ids=[f"{p}_{'x'.join(map(str, b))}_mode{m}" for p, b, m in LONG_LINES])
def test_line_sor_has_no_length_limit(prec, box, mode):
    """VERDICT r2 "missing" 4: the reference allocates its work arrays by kx and takes k-lines of any length (cz_solver.f90:1473-1676,
    cz_Evaluate.cpp:257-262).  Lines beyond what LDS holds run in further forms of the same kernels, picked by the launcher: the coefficient
    table in global memory with the right-hand sides in LDS (FP64 beyond ~640 unknowns, FP32 beyond ~1 290), and a, c, d of the line in
    global scratch (beyond ~5 100 / ~10 200, where not even the table's own reduction fits LDS; with czhip_set_pcr_mode(0, .) beyond
    ~3 400 / ~6 800).  Every variant, pcr_j_esa included, == the oracle bit for bit."""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    assert h.lib.czhip_set_pcr_mode(mode, 0) == 0
    try:
        names = ("pcr", "pcr_esa", "pcr_eda", "pcr_rb_esa", "pcr_rb", "pcr_j_esa")
        _line_variants(h, ko, sz, idx, pn, x0, msk, rhs, prec, names)
    finally:
        h.lib.czhip_set_pcr_mode(2, 0)


@pytest.mark.parametrize("prec,box", [("f64", (9, 8, 3600)), ("f32", (7, 6, 7000))], ids=["f64_3600", "f32_7000"])
def test_maf_line_sor_has_no_length_limit(prec, box):
    """the MAF line solvers on lines whose a, c, d do not fit LDS: the literal kernel on global scratch, == the oracle bit for bit"""
    ni, nj, nk = box
    sz, idx = [ni, nj, nk], [2, ni - 1, 2, nj - 1, 2, nk - 1]
    h, ko = _hip(prec), O.Kernels("oracle", prec)
    R = ko.real
    rng = np.random.default_rng(ni + 31 * nj + nk)
    shape = (nj + 4, ni + 4, nk + 4)
    x0, rhs = (rng.uniform(-1, 1, shape).astype(R) for _ in range(2))
    msk = np.zeros(shape, dtype=R)
    ko.imask_k(msk, sz, idx)
    xc, yc, zc = (np.cumsum(rng.uniform(0.5, 1.5, n + 4)).astype(R) for n in (ni, nj, nk))
    pn = O.get_num_stage(idx[5] - idx[4] + 1)
    dm, dr = h.alloc(sz, msk), h.alloc(sz, rhs)
    for name in ("pcr_rb_maf", "pcr_maf", "pcr_eda_maf"):
        x1, dx = x0.copy(), h.alloc(sz, x0)
        for it in range(2):
            r1 = r2 = 0.0
            for color in ((0, 1) if "_rb" in name else (0,)):
                r1 = ko.pcr_maf(name, sz, idx, pn, color, x1, msk, rhs, xc, yc, zc, 1.3, res=r1)
                r2 = h.pcr_maf(name, sz, idx, pn, color, dx, dm, dr, xc, yc, zc, 1.3, res=r2)
            assert _beq(dx.get(), x1), (name, it)
            assert _rel(r2, r1) < (2e-3 if prec == "f32" else 1e-10), (name, r1, r2)
