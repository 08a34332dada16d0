"""GPU tests (-m gpu) at BASELINE.json's full single-GPU size, 512^3 (configs[1..3]).

The oracle is affordable for a few sweeps at this size (about 2 s per single-thread sweep), the rest is checked through
size-independent properties: two independent kernel paths must agree bit for bit, scaling the inputs by a power of two
must scale the outputs by exactly that power (the sweep is linear and the scaling is exact in binary floating point), and a
decomposed run must reproduce the single-domain field."""
import ctypes as C
import hashlib
import threading

import numpy as np
import pytest

from oracle import cz_oracle as O

pytestmark = pytest.mark.gpu
N = 512


def _driver(prec, solver, nit, coef, t2):
    from cubez_amd import CZ
    cz = CZ(prec, quiet=True)
    cz.lib.czhip_set_tuning2(0, 0, -1, 1 if t2 else 0)
    try:
        assert cz.setup([N, N, N, solver, nit, coef]) == 1
        itr = cz.solve()
        out = (itr, cz.history(), cz.field())
    finally:
        cz.lib.czhip_set_tuning2(0, 0, -1, 1)
        cz.close()
    return out


def test_jacobi_512_against_oracle_and_between_paths():
    """4 sweeps of `cz 512 512 512 jacobi`: fused-pair path == single-sweep path == oracle, bit for bit."""
    itr_a, hist_a, P_a = _driver("f32", "jacobi", 4, 0.8, t2=True)
    itr_b, hist_b, P_b = _driver("f32", "jacobi", 4, 0.8, t2=False)
    assert itr_a == itr_b == 5
    assert P_a.tobytes() == P_b.tobytes()
    assert np.allclose(hist_a, hist_b, rtol=1e-12, atol=0)
    o = O.run((N, N, N), "jacobi", 4, 0.8, kind="oracle", prec="f32", wide=True)
    assert o.P.tobytes() == P_a.tobytes()
    assert np.allclose(hist_a, [r for _, r in o.history], rtol=1e-11, atol=0)


def test_rbsor_512_against_oracle_and_between_paths():
    """configs[2]: 5 iterations of `cz 512 512 512 sor2sma ... 1.5`: fused red-black pass == two colour launches == oracle, bit for bit
    (field) and to the double-accumulation tolerance (history)."""
    itr_a, hist_a, P_a = _driver("f32", "sor2sma", 5, 1.5, t2=True)
    itr_b, hist_b, P_b = _driver("f32", "sor2sma", 5, 1.5, t2=False)
    assert itr_a == itr_b == 6
    assert P_a.tobytes() == P_b.tobytes()
    assert np.allclose(hist_a, hist_b, rtol=1e-12, atol=0)
    o = O.run((N, N, N), "sor2sma", 5, 1.5, kind="oracle", prec="f32", wide=True)
    assert o.itr == 6
    assert o.P.tobytes() == P_a.tobytes()
    assert np.allclose(hist_a, [r for _, r in o.history], rtol=1e-11, atol=0)


def _golden_hist(name):
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return [float(ln.split(",")[1]) for ln in open(os.path.join(g, name)).read().splitlines()[1:]]


def _large_case(tag):
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return json.load(open(os.path.join(g, "large_cases.json")))[tag]


def test_bicgstab_512_fp64_first_iterations_vs_reference():
    """configs[3] at full size: the first 4 iterations of `cz 512 512 512 pbicgstab ... 0.8 jacobi` (FP64) against the reference's own
    Fortran kernels run single-threaded in the build container (tests/golden/make_golden_large.py): residuals to 1e-6 (measured: 2e-12),
    solution to 1e-9."""
    from cubez_amd import CZ
    c = _large_case("pbicgstab_jacobi_512x512x512_f64_4it")
    ref = _golden_hist(c["hist"])
    cz = CZ("f64", quiet=True)
    assert cz.setup([N, N, N, "pbicgstab", c["itr_max"], 0.8, "jacobi"]) == 1
    itr = cz.solve()
    hist = cz.history()
    cz.close()
    assert itr == c["iter"] and len(hist) == len(ref) == 4
    assert np.allclose(hist, ref, rtol=1e-6, atol=0)
    assert np.allclose(hist, ref, rtol=1e-10, atol=0)  # what the double-accumulated dots actually give this early in the recurrence


def test_bicgstab_256_fp64_to_convergence_vs_reference():
    """`cz 256 256 256 pbicgstab 1000 0.8 jacobi` (FP64) to eps against the reference (73 iterations, Res 2.220299e-06).  The Krylov
    recurrence amplifies the rounding of the dot products: the reference's own history, re-run with the SAME dot products summed in
    another order (hist_*_permuted_dots.txt), leaves the original by 1e-3 at iteration 30 and by O(1) from iteration 40 on, and its
    residual then idles within a factor 2 of eps for a dozen iterations before it drops.  So: the first 20 iterations to 1e-6 (the stated
    bar, held while the recurrence is short), afterwards no further from the reference than a small multiple of what the permuted
    reference is, and convergence inside the window in which the reference's two histories idle next to eps."""
    from cubez_amd import CZ
    c = _large_case("pbicgstab_jacobi_256x256x256_f64")
    ref, perm = np.array(_golden_hist(c["hist"])), np.array(_golden_hist(c["permuted_dots"]["hist"]))
    cz = CZ("f64", quiet=True)
    assert cz.setup([256, 256, 256, "pbicgstab", 1000, 0.8, "jacobi"]) == 1
    itr = cz.solve()
    hist = np.array(cz.history())
    res = cz.res
    cz.close()
    assert np.allclose(hist[:20], ref[:20], rtol=1e-6, atol=0)
    m = min(len(hist), len(ref), len(perm))
    dev_gpu = np.abs(hist[:m] - ref[:m]) / ref[:m]
    dev_perm = np.abs(perm[:m] - ref[:m]) / ref[:m]
    run_gpu, run_perm = np.maximum.accumulate(dev_gpu), np.maximum.accumulate(dev_perm)
    assert np.all(run_gpu <= np.maximum(1e-6, 16.0 * run_perm)), (run_gpu[::8], run_perm[::8])
    eps = 1.0e-5
    first_near = 1 + int(np.argmax(np.minimum(ref[:m], perm[:m]) < 2.0 * eps))  # first iteration with a residual within 2x of eps
    last = max(c["iter"], c["permuted_dots"]["iter"])
    assert first_near <= itr <= last + (last - first_near), (itr, first_near, last)
    assert res < eps


def test_jacobi_512_fp64_paths_agree():
    itr_a, hist_a, P_a = _driver("f64", "jacobi", 6, 0.8, t2=True)
    itr_b, hist_b, P_b = _driver("f64", "jacobi", 6, 0.8, t2=False)
    assert P_a.tobytes() == P_b.tobytes() and np.allclose(hist_a, hist_b, rtol=1e-13, atol=0)


def test_sweep_is_exactly_linear_under_power_of_two_scaling():
    """jacobi_(2p, 2b) == 2 * jacobi_(p, b) bit for bit, res scales by 4 (random 512^3 inputs, drop-in symbol)."""
    from cubez_amd import CzHip
    h = CzHip("f32")
    sz, idx = [N, N, N], [2, N - 1, 2, N - 1, 2, N - 1]
    rng = np.random.default_rng(42)
    shape = (N + 4, N + 4, N + 4)
    p = rng.uniform(-1, 1, shape).astype(np.float32)
    b = rng.uniform(-1, 1, shape).astype(np.float32)
    cf = np.array([1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3], dtype=np.float32)
    dp, db, dw = h.alloc(sz, p), h.alloc(sz, b), h.alloc(sz)
    r1 = h.jacobi(dp, sz, idx, cf, 0.8, db, dw)
    out1 = dp.get()
    dp.put(p * np.float32(2)), db.put(b * np.float32(2))
    r2 = h.jacobi(dp, sz, idx, cf, 0.8, db, dw)
    out2 = dp.get()
    assert (out1 * np.float32(2)).tobytes() == out2.tobytes()
    assert abs(r2 - 4.0 * r1) <= 1e-12 * r2
    # a checksum of checksums: the per-plane sums of the field reproduce the total
    s_planes = out1.astype(np.float64).sum(axis=(1, 2))
    assert abs(s_planes.sum() - out1.astype(np.float64).sum()) <= 1e-9 * abs(s_planes).sum()
    for a in (dp, db, dw):
        a.free()


def test_residual_handoff_with_tens_of_thousands_of_tiny_workgroups():
    """The in-kernel finalisation of the residual (cz_k_common.h: arrive_and_test_last) under the load it was NOT tuned for: the two-stage
    pass cut into one-plane chunks, 44 880 workgroups of three plane steps each, several per CU at very different times, the last arriver
    with a warm L1.  Launches alternate between a field and its double (every partial then differs by exactly 4x between neighbours in
    time), so a partial read stale -- from L1, from another XCD's L2, from the previous launch -- cannot hide: every launch must return
    the bits of its own sums, and those must equal the sums of the ordinary geometry to rounding and the oracle's to 1e-11."""
    from cubez_amd import CzHip
    h = CzHip("f32")
    sz, idx = [N, N, N], [2, N - 1, 2, N - 1, 2, N - 1]
    rng = np.random.default_rng(7)
    shape = (N + 4, N + 4, N + 4)
    p = rng.uniform(-1, 1, shape).astype(np.float32)
    b = rng.uniform(-1, 1, shape).astype(np.float32)
    cf = np.array([1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3], dtype=np.float32)
    dA, bA, dB, bB = h.alloc(sz, p), h.alloc(sz, b), h.alloc(sz, p * np.float32(2)), h.alloc(sz, b * np.float32(2))
    dw = h.alloc(sz, p)
    try:
        assert h.set_tuning2(-2, 2, 0, 1)
        ok, a1, a2 = h.jacobi2(dA, dw, bA, sz, idx, cf, 0.8)
        assert ok
        assert h.set_tuning2(512, 2, 1, 1)  # one plane per chunk
        seen = []
        for it in range(24):
            src, rhs, scale = (dA, bA, 1.0) if it % 2 == 0 else (dB, bB, 4.0)
            ok, r1, r2 = h.jacobi2(src, dw, rhs, sz, idx, cf, 0.8)
            assert ok
            seen.append((r1 / scale, r2 / scale))
        assert all(v == seen[0] for v in seen), sorted(set(seen))  # bitwise: scaling by 4 is exact
        assert abs(seen[0][0] - a1) <= 1e-12 * a1 and abs(seen[0][1] - a2) <= 1e-12 * a2
    finally:
        h.set_tuning2(-2, 2, 0, 1)
    k = O.Kernels("oracle", "f32")
    a, w, r = p.copy(), np.zeros_like(p), []
    for _ in range(2):
        wide = np.zeros(1)
        k.jacobi(a, sz, idx, cf, 0.8, b, w, wide=wide)
        r.append(wide[0])
    assert abs(seen[0][0] - r[0]) <= 1e-11 * r[0] and abs(seen[0][1] - r[1]) <= 1e-11 * r[1]
    for x in (dA, bA, dB, bB, dw):
        x.free()


def test_decomposed_512_equals_single_domain():
    """`cz 512 512 512 jacobi 6 0.8 1 2 1` (two ranks as threads on this GPU, LOCAL transport) == the single-domain run."""
    from cubez_amd import CZ, load
    itr1, hist1, P1 = _driver("f32", "jacobi", 6, 0.8, t2=True)
    lib = load("f32")
    lib.cz_comm_local_world.restype = C.c_void_p
    lib.cz_comm_bootstrap_local.argtypes = [C.c_void_p, C.c_int]
    lib.cz_comm_local_world_free.argtypes = [C.c_void_p]
    world = lib.cz_comm_local_world(2)
    res, errs = [None, None], []

    def work(r):
        try:
            lib.cz_comm_bootstrap_local(world, r)
            cz = CZ("f32", quiet=True)
            assert cz.setup([N, N, N, "jacobi", 6, 0.8, 1, 2, 1]) == 1
            itr = cz.solve()
            res[r] = (itr, cz.history(), cz.field(), cz.local())
            cz.close()
        except BaseException as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errs, errs
    lib.cz_comm_local_world_free(world)
    g = 2
    h1 = hashlib.sha256()
    h2 = hashlib.sha256()
    for itr, hist, P, loc in res:
        assert itr == itr1 and np.allclose(hist, hist1, rtol=1e-12, atol=0)
        (ni, nj, nk), (hi, hj, hk) = loc["size"], loc["head"]
        own = P[g:g + nj, g:g + ni, g:g + nk]
        ref = P1[g + hj - 1:g + hj - 1 + nj, g + hi - 1:g + hi - 1 + ni, g + hk - 1:g + hk - 1 + nk]
        h1.update(np.ascontiguousarray(own).tobytes()), h2.update(np.ascontiguousarray(ref).tobytes())
    assert h1.hexdigest() == h2.hexdigest()


@pytest.mark.parametrize("solver,coef", [("jacobi", 0.8), ("sor2sma", 1.5)])
def test_configs4_geometry_1024_cube_2x2x2_bricks_equals_single_domain(solver, coef):
    """BASELINE configs[4] at full size, as far as one GPU can take it: `cz 1024 1024 1024 <solver> 6 <coef> 2 2 2` -- eight 512^3 bricks,
    every brick a corner brick with three rank-internal faces, shell slabs + interior, two-layer exchange with edges, lagged residual
    all-reduce -- with the eight ranks as host threads on this GPU (LOCAL transport: device copies where RCCL would send; same
    decomposition, kernels, pack/unpack and stream structure) against the single-domain 1024^3 run: same iteration count, history to
    1e-12, every owned cell bit for bit (compared through sha256 per brick)."""
    from cubez_amd import CZ, load
    n = 1024
    cz = CZ("f32", quiet=True)
    assert cz.setup([n, n, n, solver, 6, coef]) == 1
    itr1 = cz.solve()
    hist1, P1 = cz.history(), cz.field()
    cz.close()
    lib = load("f32")
    lib.cz_comm_local_world.restype = C.c_void_p
    lib.cz_comm_bootstrap_local.argtypes = [C.c_void_p, C.c_int]
    lib.cz_comm_local_world_free.argtypes = [C.c_void_p]
    world = lib.cz_comm_local_world(8)
    res, errs = [None] * 8, []
    g = 2

    def work(r):
        try:
            lib.cz_comm_bootstrap_local(world, r)
            c = CZ("f32", quiet=True)
            assert c.setup([n, n, n, solver, 6, coef, 2, 2, 2]) == 1
            itr = c.solve()
            loc, P = c.local(), c.field()
            (ni, nj, nk) = loc["size"]
            res[r] = (itr, c.history(), hashlib.sha256(np.ascontiguousarray(P[g:g + nj, g:g + ni, g:g + nk]).tobytes()).hexdigest(), loc, c.info())
            del P
            c.close()
        except BaseException as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(8)]
    [t.start() for t in th]
    [t.join(timeout=900) for t in th]
    if any(t.is_alive() for t in th):
        import os
        import sys
        sys.stderr.write("DEADLOCK: the 2x2x2 run did not finish\n")
        os._exit(3)
    assert not errs, errs
    lib.cz_comm_local_world_free(world)
    for itr, hist, digest, loc, info in res:
        assert itr == itr1 and np.allclose(hist, hist1, rtol=1e-12, atol=0)
        assert info["fused_pass"] == 1 and info["shell_slabs"] == 3 and info["lagged_reduce"] == 1, info
        (ni, nj, nk), (hi, hj, hk) = loc["size"], loc["head"]
        ref = P1[g + hj - 1:g + hj - 1 + nj, g + hi - 1:g + hi - 1 + ni, g + hk - 1:g + hk - 1 + nk]
        assert hashlib.sha256(np.ascontiguousarray(ref).tobytes()).hexdigest() == digest, loc


def test_psor_512_is_locally_consistent_with_the_sequential_order():
    """One lexicographic SOR sweep at 512^3.  In the order (j, i, k) every update reads the NEW values of its k-1, i-1, j-1
    neighbours and the OLD ones of k+1, i+1, j+1, so the result can be checked point by point (vectorised, exact) without
    running the sequential loop: new(p) == old(p) + ((ss - b)/dd - old(p))*omg with ss built from exactly those values."""
    from cubez_amd import CzHip
    h = CzHip("f32")
    R = np.float32
    sz, idx = [N, N, N], [2, N - 1, 2, N - 1, 2, N - 1]
    rng = np.random.default_rng(2024)
    old = rng.uniform(-1, 1, (N + 4, N + 4, N + 4)).astype(R)
    b = rng.uniform(-1, 1, (N + 4, N + 4, N + 4)).astype(R)
    cf = np.array([1.1, 0.9, 1.05, 0.95, 1.2, 0.8, 6.3], dtype=R)
    omg = R(1.2)
    dp_, db_ = h.alloc(sz, old), h.alloc(sz, b)
    res = h.psor(dp_, sz, idx, cf, omg, db_)
    new = dp_.get()
    c = (slice(3, N + 1),) * 3                      # inner box 2..N-1 (1-based) -> padded 3..N
    def sh(a, dj, di, dk):
        return a[3 + dj:N + 1 + dj, 3 + di:N + 1 + di, 3 + dk:N + 1 + dk]
    ss = cf[0] * sh(old, 0, 1, 0) + cf[1] * sh(new, 0, -1, 0)
    ss = ss + cf[2] * sh(old, 1, 0, 0)
    ss = ss + cf[3] * sh(new, -1, 0, 0)
    ss = ss + cf[4] * sh(old, 0, 0, 1)
    ss = ss + cf[5] * sh(new, 0, 0, -1)
    dp = ((ss - b[c]) / cf[6] - old[c]) * omg
    assert (old[c] + dp).tobytes() == new[c].tobytes()
    assert abs(res - float(np.sum(dp.astype(np.float64) ** 2))) <= 1e-10 * res
    mask = np.ones_like(old, dtype=bool)
    mask[c] = False
    assert np.array_equal(new[mask], old[mask])       # nothing outside the inner box is written


def test_line_sor_512_three_kernel_forms_agree():
    """pcr_rb at 512^3: the literal per-line kernel, the table kernel with the right-hand side in LDS and the register kernel are
    three implementations of the same arithmetic -- same field, bit for bit, after two iterations."""
    from cubez_amd import CZ
    outs = []
    for form in (0, 1, 2):
        cz = CZ("f32", quiet=True)
        cz.lib.czhip_set_pcr_mode(form, 0)
        try:
            assert cz.setup([N, N, N, "pcr_rb", 2, 1.2]) == 1
            cz.solve()
            outs.append((hashlib.sha256(cz.field().tobytes()).hexdigest(), cz.history()))
        finally:
            cz.lib.czhip_set_pcr_mode(2, 0)
            cz.close()
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert np.allclose(outs[0][1], outs[2][1], rtol=1e-12, atol=0) and np.allclose(outs[1][1], outs[2][1], rtol=1e-12, atol=0)


@pytest.mark.parametrize("solver,prec", [("pcr", "f32"), ("pcr_eda", "f64"), ("pcr_maf", "f32")])
def test_lexicographic_line_sor_512_one_launch_equals_diagonals(solver, prec):
    """pcr / pcr_eda / pcr_maf at 512^3: the whole sweep in one launch (rows of k-lines handed from workgroup to workgroup, 510 strips
    in flight) gives the field of the launch-per-diagonal form, bit for bit, after two iterations -- and neither residual is NaN."""
    from cubez_amd import CZ
    outs = []
    for one_launch in (1, 0):
        cz = CZ(prec, quiet=True)
        assert cz.lib.czhip_set_pcr_lex(one_launch, 0, 1) == 0
        try:
            assert cz.setup([N, N, N, solver, 2, 1.2]) == 1
            cz.solve()
            h = cz.history()
            assert all(v == v for v in h), h
            outs.append((hashlib.sha256(cz.field().tobytes()).hexdigest(), h))
        finally:
            cz.lib.czhip_set_pcr_lex(1, 0, 1)
            cz.close()
    assert outs[0][0] == outs[1][0]
    assert np.allclose(outs[0][1], outs[1][1], rtol=1e-12, atol=0)


def test_arrays_beyond_2_to_31_elements():
    """maximum sizes: 1300^3 cells = 2.2e9 elements (8.9 GB) per FP32 array -- every linear index needs 64 bits.  The fused-pair path and
    the single-sweep path agree bit for bit, the Dirichlet data sit where the 64-bit index says."""
    from cubez_amd import CZ
    n = 1300
    assert (n + 4) ** 3 > 2 ** 31
    out = []
    for t2 in (1, 0):
        cz = CZ("f32", quiet=True)
        cz.lib.czhip_set_tuning2(0, 0, -1, t2)
        try:
            assert cz.setup([n, n, n, "jacobi", 4, 0.8]) == 1
            itr = cz.solve()
            P = cz.field()
            out.append((itr, cz.history(), hashlib.sha256(P.tobytes()).hexdigest(), float(P[n // 2 + 2, n // 2 + 2, 2])))
            del P
        finally:
            cz.lib.czhip_set_tuning2(0, 0, -1, 1)
            cz.close()
    assert out[0][0] == out[1][0] == 5 and out[0][2] == out[1][2]
    assert np.allclose(out[0][1], out[1][1], rtol=1e-12, atol=0)
    x = (n // 2) / (n - 1)
    assert abs(out[0][3] - np.sin(np.pi * x) ** 2) < 1e-5
