"""GPU test (-m gpu) of the decomposed solver path on ONE GPU: the ranks of a Cartesian division run as host threads
of this process and exchange faces through the LOCAL transport of cz_comm.cpp (device-to-device copies in place of
RCCL send/recv; same pack/unpack kernels, same decomposition, same inner ranges, same colouring).

Contract: decomposed run == single-domain run, field bit-for-bit (Jacobi, RB-SOR), iteration count
equal, residual to summation-order tolerance; BiCGSTAB to 1e-9."""
import os
import sys
import threading

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _single(prec, gsz, solver, itmax, coef, pc=None):
    from cubez_amd import CZ
    cz = CZ(prec, quiet=True)
    a = list(gsz) + [solver, itmax, coef] + ([pc] if pc else [])
    assert cz.setup(a) == 1
    itr = cz.solve()
    out = (itr, cz.res, cz.history(), cz.field())
    cz.close()
    return out


def _decomposed(prec, gsz, solver, itmax, coef, div, pc=None, overlap=1, solves=1, env=None):
    import os
    from cubez_amd import CZ, load
    os.environ["CZ_OVERLAP"] = str(overlap)  # read by the driver when a CZ is created
    for k, v in (env or {}).items():  # kernel switches: read by each rank thread when its library context is created (the threads are new)
        os.environ[k] = v
    lib = load(prec)
    import ctypes as C
    lib.cz_comm_local_world.restype = C.c_void_p
    lib.cz_comm_bootstrap_local.argtypes = [C.c_void_p, C.c_int]
    lib.cz_comm_local_world_free.argtypes = [C.c_void_p]
    n = div[0] * div[1] * div[2]
    world = lib.cz_comm_local_world(n)
    results, errors = [None] * n, []

    def work(r):
        try:
            lib.cz_comm_bootstrap_local(world, r)
            for _ in range(solves):  # consecutive solves share the thread's library context (arrival ticket, partial sums, streams)
                cz = CZ(prec, quiet=True)
                a = list(gsz) + [solver, itmax, coef] + ([pc] if pc else []) + list(div)
                assert cz.setup(a) == 1
                cz.timing(True)
                itr = cz.solve()
                loc = cz.local()
                loc["fused_pairs"] = cz.timing_read("jacobi2")[0] + cz.timing_read("rbsor2")[0]
                loc["shell_launches"] = cz.timing_read("pair_shell")[0]
                loc["info"] = cz.info()
                cz.timing(False)
                out = (itr, cz.res, cz.history(), cz.field(), loc)
                if results[r] is not None:  # every solve must repeat the first one exactly
                    assert out[0] == results[r][0] and out[2] == results[r][2] and out[3].tobytes() == results[r][3].tobytes(), "solve differs from the previous one"
                results[r] = out
                cz.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join(timeout=90) for t in th]
    if any(t.is_alive() for t in th):
        # a rank thread is stuck in a collective: nothing can unblock it and the interpreter could not exit either -- fail the whole
        # run loudly instead of hanging it
        import sys
        sys.stderr.write(f"DEADLOCK: decomposed {solver} {gsz} {div} did not finish in 90 s\n")
        sys.stderr.flush()
        os._exit(3)
    assert not errors, errors
    assert all(r is not None for r in results)
    os.environ.pop("CZ_OVERLAP")
    for k in (env or {}):
        os.environ.pop(k)
    lib.cz_comm_local_world_free(world)
    # assemble the global field from the owned cells of every brick
    g = 2
    real = results[0][3].dtype
    G = np.zeros((gsz[1] + 4, gsz[0] + 4, gsz[2] + 4), dtype=real)
    for itr, res, hist, P, loc in results:
        (ni, nj, nk), (hi, hj, hk) = loc["size"], loc["head"]
        G[g + hj - 1:g + hj - 1 + nj, g + hi - 1:g + hi - 1 + ni, g + hk - 1:g + hk - 1 + nk] = P[g:g + nj, g:g + ni, g:g + nk]
    return results, G


CASES = [
    ("f32", (40, 36, 44), "jacobi", 25, 0.8, (1, 2, 1)),
    ("f32", (40, 36, 44), "jacobi", 25, 0.8, (2, 1, 1)),
    ("f32", (40, 36, 44), "jacobi", 24, 0.8, (1, 1, 2)),
    ("f32", (41, 37, 45), "jacobi", 17, 0.8, (2, 2, 2)),     # uneven bricks, odd sizes (rows of 27 / 26 values)
    ("f64", (36, 40, 44), "jacobi", 20, 0.9, (2, 2, 1)),
    ("f32", (40, 36, 44), "sor2sma", 20, 1.5, (1, 2, 1)),
    ("f32", (41, 37, 45), "sor2sma", 15, 1.5, (2, 2, 2)),    # odd heads: colour offset per brick
    ("f64", (36, 40, 44), "sor2sma", 20, 1.4, (2, 1, 2)),
    ("f32", (48, 36, 44), "jacobi", 12, 0.8, (3, 1, 1)),     # middle brick: both faces of an axis border a rank
    ("f32", (40, 48, 44), "sor2sma", 9, 1.5, (1, 3, 1)),
    ("f64", (36, 40, 54), "jacobi", 10, 0.8, (1, 1, 3)),
    ("f32", (40, 36, 41), "jacobi", 10, 0.8, (1, 1, 2)),      # k-extents 21 / 20: one brick with rows of 25 values, one of 24
    ("f32", (40, 36, 41), "sor2sma", 8, 1.5, (1, 1, 2)),
    ("f32", (40, 36, 32), "pcr_rb", 8, 1.2, (2, 1, 1)),       # line SOR: whole k-lines per brick, exchange after each colour
    ("f64", (41, 37, 32), "pcr_rb", 8, 1.2, (2, 2, 1)),       # odd heads: global colouring
    ("f32", (40, 39, 64), "pcr_rb_esa", 6, 1.2, (1, 3, 1)),
    ("f64", (36, 40, 32), "pcr_j_esa", 6, 0.9, (2, 2, 1)),
]


@pytest.mark.parametrize("overlap", [1, 0], ids=["overlap", "serial"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[2]}_{c[0]}_{'x'.join(map(str, c[5]))}" for c in CASES])
def test_decomposed_equals_single_domain(case, overlap):
    """overlap: shell slabs first, two-layer exchange on a second stream behind the interior sweep; serial: exchange after
    the whole sweep.  Both must reproduce the single-domain run."""
    prec, gsz, solver, itmax, coef, div = case
    itr1, res1, hist1, P1 = _single(prec, gsz, solver, itmax, coef)
    results, G = _decomposed(prec, gsz, solver, itmax, coef, div, overlap=overlap)
    g = 2
    inner = (slice(g, -g),) * 3
    assert G[inner].tobytes() == P1[inner].tobytes()
    for itr, res, hist, P, loc in results:
        assert itr == itr1
        assert np.allclose(hist, hist1, rtol=1e-12, atol=0)
        if solver in ("jacobi", "sor2sma"):
            # every brick takes the two-sweeps-per-pass kernel with the two-layer exchange -- since round 3 also bricks whose k extent is no
            # multiple of the vector width (rounds 1-2: such a brick made all of them agree on single sweeps)
            npass = itmax // 2 if solver == "jacobi" else itmax
            assert loc["fused_pairs"] == npass, loc
            assert loc["shell_launches"] == (npass if overlap else 0), loc


FORMS = [{"CZHIP_T2_KWIN": "3"}, {"CZHIP_T2_PRE": "0"}, {"CZHIP_T2_KWIN": "3", "CZHIP_T2_PRE": "0"}]


@pytest.mark.parametrize("form", FORMS, ids=["windows", "pipelined", "windows_pipelined"])
@pytest.mark.parametrize("case", [CASES[3], CASES[6], CASES[7], CASES[11], CASES[12]], ids=lambda c: f"{c[2]}_{c[0]}_{'x'.join(map(str, c[5]))}")
def test_decomposed_bricks_in_every_form_of_the_pass(case, form):
    """Round 4 gave the two-stage pass k windows and, on small boxes, a preloaded form: a decomposed brick applies the first stage to its
    ghost layer 1 and therefore reads rows and vectors a single-domain run never uses (the last vector of the last row, where rows are no
    multiple of the vector width, was shifted by a clamp in the first version of the windows).  Bricks with odd sizes and internal faces on
    every side, windows of three vectors, preloaded and pipelined form: all equal the single-domain run, bit for bit."""
    prec, gsz, solver, itmax, coef, div = case
    itr1, res1, hist1, P1 = _single(prec, gsz, solver, itmax, coef)
    for overlap in (1, 0):
        results, G = _decomposed(prec, gsz, solver, itmax, coef, div, overlap=overlap, env=form)
        inner = (slice(2, -2),) * 3
        assert G[inner].tobytes() == P1[inner].tobytes(), (form, overlap)
        assert all(r[0] == itr1 for r in results)


def test_rccl_one_rank_selftest():
    """RCCL plumbing on the one GPU we have: communicator from a unique id, all-reduce, grouped send/recv to self."""
    import torch  # noqa: F401  (same process set-up as bench.py: torch's bundled librccl is loaded first)
    from cubez_amd import CzHip
    h = CzHip("f32")
    assert h.lib.cz_comm_selftest() == 0


@pytest.mark.parametrize("lag", [1, 0], ids=["lagged_reduce", "inline_reduce"])
@pytest.mark.parametrize("gsz,coef", [((16, 16, 16), 0.8), ((16, 16, 16), 0.9), ((20, 16, 24), 0.85), ((16, 20, 16), 1.0)], ids=["a", "b", "c", "d"])
def test_decomposed_jacobi_converges_exactly_like_single_domain(gsz, coef, lag):
    """Jacobi to convergence, decomposed: fused pairs, overlapped exchange and (lagged) the residual all-reduce + test one pass behind on
    the exchange stream with three rotating buffers.  Iteration count, history and final field equal the single-domain run, whichever
    sweep of a pair converges (the four cases stop at iterations of both parities)."""
    import os
    prec = "f64"
    itr1, res1, hist1, P1 = _single(prec, gsz, "jacobi", 100000, coef)
    os.environ["CZ_LAG_REDUCE"] = str(lag)
    try:
        results, G = _decomposed(prec, gsz, "jacobi", 100000, coef, (2, 2, 1))
    finally:
        os.environ.pop("CZ_LAG_REDUCE")
    assert all(r[0] == itr1 for r in results), (itr1, [r[0] for r in results])
    assert all(len(r[2]) == len(hist1) for r in results)
    assert np.allclose(results[0][2], hist1, rtol=1e-12, atol=0)
    assert G[2:-2, 2:-2, 2:-2].tobytes() == P1[2:-2, 2:-2, 2:-2].tobytes()


@pytest.mark.parametrize("solver,coef", [("jacobi", 0.85), ("sor2sma", 1.5)])
def test_two_consecutive_lagged_solves_on_one_context(solver, coef):
    """a lagged pass overtaken by convergence used to run on part of its workgroups (each read the live flag), which left the
    arrival ticket of the in-kernel finalisation short for everything that followed on the same context.  Now a pass sees the flag as the
    test two passes earlier left it (all workgroups alike) and every solve starts from a zero ticket: the second solve on the same
    context repeats the first, and both equal the single-domain run."""
    prec, gsz = "f64", (20, 16, 24)
    itr1, res1, hist1, P1 = _single(prec, gsz, solver, 100000, coef)
    results, G = _decomposed(prec, gsz, solver, 100000, coef, (2, 2, 1), solves=2)
    assert all(r[4]["info"]["lagged_reduce"] == 1 for r in results)
    assert all(r[0] == itr1 for r in results), (itr1, [r[0] for r in results])
    assert np.allclose(results[0][2], hist1, rtol=1e-12, atol=0)
    assert G[2:-2, 2:-2, 2:-2].tobytes() == P1[2:-2, 2:-2, 2:-2].tobytes()


@pytest.mark.parametrize("solver,coef", [("jacobi", 0.8), ("sor2sma", 1.5)])
def test_rank_skew_cannot_change_what_a_rank_issues(solver, coef):
    """VERDICT r1 #5 (the hang of gpurun_out/decomp.log, commit 24dd087): every decision that gates a collective is a function of
    stream-ordered, all-reduced device state.  One rank is delayed 30 ms before each of its looks at the convergence flag; iteration count,
    history and field must not move (a rank that stopped issuing passes at another iteration would end the run in the bounded LOCAL
    barrier with exit code 3)."""
    import os
    prec, gsz = "f64", (16, 16, 16)
    itr1, res1, hist1, P1 = _single(prec, gsz, solver, 100000, coef)
    os.environ["CZ_TEST_SKEW"] = "1,30"
    try:
        results, G = _decomposed(prec, gsz, solver, 100000, coef, (2, 1, 2))
    finally:
        os.environ.pop("CZ_TEST_SKEW")
    assert all(r[0] == itr1 for r in results), (itr1, [r[0] for r in results])
    assert np.allclose(results[0][2], hist1, rtol=1e-12, atol=0)
    assert G[2:-2, 2:-2, 2:-2].tobytes() == P1[2:-2, 2:-2, 2:-2].tobytes()


@pytest.mark.parametrize("lag", [1, 0], ids=["lagged_reduce", "inline_reduce"])
def test_decomposed_converges_at_the_same_iteration(lag):
    import os
    prec, gsz = "f64", (32, 32, 32)
    itr1, res1, hist1, P1 = _single(prec, gsz, "sor2sma", 100000, 1.5)
    os.environ["CZ_LAG_REDUCE"] = str(lag)
    try:
        results, G = _decomposed(prec, gsz, "sor2sma", 100000, 1.5, (2, 2, 1))
    finally:
        os.environ.pop("CZ_LAG_REDUCE")
    assert all(r[0] == itr1 for r in results)
    assert G[2:-2, 2:-2, 2:-2].tobytes() == P1[2:-2, 2:-2, 2:-2].tobytes()


@pytest.mark.parametrize("pc", ["jacobi", "sor2sma"])
def test_decomposed_bicgstab(pc):
    prec, gsz = "f64", (32, 36, 40)
    itr1, res1, hist1, P1 = _single(prec, gsz, "pbicgstab", 200, 0.8 if pc == "jacobi" else 1.5, pc)
    results, G = _decomposed(prec, gsz, "pbicgstab", 200, 0.8 if pc == "jacobi" else 1.5, (2, 1, 2), pc)
    assert all(r[0] == itr1 for r in results)
    assert np.allclose(results[0][2], hist1, rtol=1e-6, atol=0)
    assert np.abs(G[2:-2, 2:-2, 2:-2] - P1[2:-2, 2:-2, 2:-2]).max() < 1e-9


BLOCK_LOCAL = [
    ("f32", (40, 36, 44), "psor", 8, 1.2, (2, 1, 1)),
    ("f64", (41, 37, 45), "psor", 6, 1.3, (2, 2, 2)),     # uneven bricks, cuts along all three axes
    ("f32", (40, 39, 32), "pcr", 6, 1.2, (1, 3, 1)),
    ("f64", (36, 40, 32), "pcr_esa", 5, 1.2, (2, 2, 1)),
    ("f32", (40, 36, 64), "pcr_eda", 5, 1.2, (2, 1, 1)),
    ("f32", (40, 36, 64), "pcr_rb", 6, 1.2, (1, 1, 2)),       # a cut along k: every brick solves its piece of a line
    ("f64", (41, 37, 70), "pcr_rb_esa", 5, 1.2, (2, 1, 2)),
    ("f64", (36, 40, 64), "pcr_j_esa", 5, 0.9, (1, 2, 2)),
    ("f32", (40, 36, 66), "pcr", 4, 1.2, (1, 1, 3)),
]


@pytest.mark.parametrize("case", BLOCK_LOCAL, ids=[f"{c[2]}_{c[0]}_{'x'.join(map(str, c[5]))}" for c in BLOCK_LOCAL])
def test_block_local_decompositions_like_the_reference(case):
    """VERDICT r1 "missing" 5: the reference accepts decomposed runs of its lexicographic solvers, and cuts along k for its line solvers,
    and sweeps every brick on its own with the ghost values of the last exchange (cz_Poisson.cpp:124, :586, :794).  Same here; the result
    is that loop's, not the single-domain iterate: compared bit for bit with the loop restated on the oracle's kernels
    (tests/blocklocal.py)."""
    from blocklocal import run as block_local_run
    prec, gsz, solver, nit, coef, div = case
    hist_o, G_o = block_local_run(gsz, div, solver, nit, coef, prec)
    results, G = _decomposed(prec, gsz, solver, nit, coef, div)
    g = 2
    assert G[g:-g, g:-g, g:-g].tobytes() == G_o[g:-g, g:-g, g:-g].tobytes()
    for itr, res, hist, P, loc in results:
        assert itr == nit + 1
        assert np.allclose(hist, hist_o, rtol=1e-11, atol=0)
    if div != (1, 1, 1):  # and it is NOT the single-domain run (documented semantics)
        itr1, res1, hist1, P1 = _single(prec, gsz, solver, nit, coef)
        assert G[g:-g, g:-g, g:-g].tobytes() != P1[g:-g, g:-g, g:-g].tobytes()


@pytest.mark.parametrize("solver,div", [("jacobi_maf", (2, 1, 1)), ("sor2sma_maf", (1, 2, 2)), ("pcr_rb_maf", (2, 2, 1))])
def test_decomposed_maf_flavour(solver, div):
    """The MAF kernels take the metrics from 1-D coordinate arrays that every brick fills from its LOCAL index (cz_Evaluate.cpp:342-363
    adds no brick origin).  On the uniform grid the metrics are differences of neighbouring coordinates, equal up to rounding on
    every brick: the decomposed run follows the single-domain run to rounding, not bit for bit."""
    prec, gsz = "f64", (40, 36, 32)
    coef = 0.8 if solver.startswith("jacobi") else 1.2
    itr1, res1, hist1, P1 = _single(prec, gsz, solver, 12, coef)
    results, G = _decomposed(prec, gsz, solver, 12, coef, div)
    assert all(r[0] == itr1 for r in results)
    assert np.allclose(results[0][2], hist1, rtol=1e-9, atol=0)
    assert np.abs(G[2:-2, 2:-2, 2:-2] - P1[2:-2, 2:-2, 2:-2]).max() < 1e-12
    if not solver.startswith("pcr"):
        # VERDICT r2 "missing" 5: the MAF flavour takes the fused pass in decomposed runs too (pair_shell_k<MAF = 1> for the shell slabs):
        # 6 pairs of jacobi_maf sweeps / 12 red-black iterations, each with its overlapped two-layer exchange
        npass = 6 if solver == "jacobi_maf" else 12
        assert all(r[4]["fused_pairs"] == npass and r[4]["shell_launches"] == npass for r in results), [(r[4]["fused_pairs"], r[4]["shell_launches"]) for r in results]
        assert all(r[4]["info"]["pass_kind"] == 2 and r[4]["info"]["exchange_depth"] == 2 for r in results), [r[4]["info"] for r in results]


def test_ranks_out_of_step_end_the_job_with_a_diagnostic_instead_of_hanging():
    """if ranks ever issue different sequences of collectives the job must end, not hang.  Here one of two ranks never
    shows up (its thread returns before set-up): the other rank's first collective waits in the LOCAL transport's bounded barrier and the
    process exits with code 3 and a line naming the rank and the barrier (with RCCL the watchdog of cz_comm.cpp does the same for the
    stream-ordered collectives).  Run in a child process: the exit is the process's."""
    import os
    import subprocess
    import sys
    code = r"""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.environ["CZ_ROOT"])
from cubez_amd import CZ, load
lib = load("f32")
lib.cz_comm_local_world.restype = C.c_void_p
lib.cz_comm_bootstrap_local.argtypes = [C.c_void_p, C.c_int]
world = lib.cz_comm_local_world(2)
def work(r):
    lib.cz_comm_bootstrap_local(world, r)
    if r == 1:
        return                      # this rank never issues anything
    cz = CZ("f32", quiet=True)
    cz.setup([32, 32, 32, "jacobi", 10, 0.8, 1, 2, 1])
    cz.solve()
th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
[t.start() for t in th]
[t.join() for t in th]
print("finished")                   # must not be reached
"""
    env = dict(os.environ, CZ_COMM_TIMEOUT="3", CZ_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 3, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    assert "finished" not in r.stdout
    assert "cz rank 0" in r.stderr and "barrier" in r.stderr and "ranks arrived" in r.stderr, r.stderr[-500:]
