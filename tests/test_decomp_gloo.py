"""CPU tests of the N>1 path: decomposition host logic (C functions of cz_comm.cpp through the C-ABI) and a
world_size-2 / world_size-4 torch.distributed (gloo) run of the decomposed Jacobi and RB-SOR loops -- one process per
brick, halo exchange with the face conventions of the GPU path, residual all-reduce -- with the CPU oracle doing the
sweeps, checked against the single-domain oracle run: field bit-for-bit, residual history to 1e-12."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cubez_amd import decomp
from oracle import cz_oracle as O


def test_decompose_covers_the_cube_once():
    for gsz, div in (((40, 36, 44), (2, 2, 2)), ((41, 37, 45), (2, 3, 1)), ((512, 1024, 512), (1, 2, 1)), ((9, 9, 9), (1, 1, 4))):
        n = div[0] * div[1] * div[2]
        cover = np.zeros(gsz, dtype=np.int32)
        bricks = [decomp.decompose(gsz, div, n, r) for r in range(n)]
        assert all(b is not None for b in bricks)
        for r, b in enumerate(bricks):
            (ni, nj, nk), (hi, hj, hk) = b["size"], b["head"]
            cover[hi - 1:hi - 1 + ni, hj - 1:hj - 1 + nj, hk - 1:hk - 1 + nk] += 1
            # neighbour tables are mutually consistent and -1 exactly on the physical boundary
            for f in range(6):
                nb = b["nID"][f]
                a = f // 2
                at_boundary = (b["head"][a] == 1) if f % 2 == 0 else (b["head"][a] + b["size"][a] - 1 == gsz[a])
                assert (nb < 0) == at_boundary
                if nb >= 0:
                    assert bricks[nb]["nID"][decomp.OPPOSITE[f]] == r
        assert (cover == 1).all()
    assert decomp.decompose((8, 8, 8), (2, 2, 2), 7, 0) is None       # division does not match the rank count
    assert decomp.decompose((3, 8, 8), (2, 1, 1), 2, 0) is None       # bricks thinner than 2 cells


def test_auto_division_prefers_contiguous_faces():
    assert decomp.auto_division(1, (512, 512, 512)) == [1, 1, 1]
    assert decomp.auto_division(2, (512, 512, 512)) == [1, 2, 1]      # J cut: faces are contiguous planes
    assert decomp.auto_division(8, (1024, 1024, 1024)) == [2, 2, 2]
    d = decomp.auto_division(4, (1024, 1024, 512))
    assert d[0] * d[1] * d[2] == 4 and d[2] == 1


def test_inner_range_single_domain_is_the_reference_rule():
    # cz_miscel.cpp:24-40: (2, N-1) on every axis when all faces are physical
    assert decomp.inner_range((128, 64, 32), [-1] * 6) == [2, 127, 2, 63, 2, 31]
    assert decomp.inner_range((16, 16, 16), [3, -1, -1, 5, 0, 1]) == [1, 15, 2, 16, 1, 16]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, gsz, div, solver, nit, coef, prec, q):
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k = O.Kernels("oracle", prec)
        R = k.real
        b = decomp.decompose(gsz, div, world, rank)
        size, head, nID = b["size"], b["head"], b["nID"]
        idx = decomp.inner_range(size, nID)
        pitch = R(1.0 / float(R(gsz[2] - 1)))
        origin = np.array([R(0) + R(head[a] - 1) * pitch for a in range(3)], dtype=R)
        cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=R)
        P, RHS, WRK = k.alloc(size), k.alloc(size), k.alloc(size)
        faces = decomp.face_slices(size)

        def halo(X):
            reqs, recvs = [], []
            for f in range(6):
                if nID[f] < 0:
                    continue
                own, ghost = faces[f]
                send = torch.from_numpy(np.ascontiguousarray(X[own]))
                recv = torch.empty_like(send)
                reqs.append(dist.isend(send, dst=nID[f], tag=f))
                reqs.append(dist.irecv(recv, src=nID[f], tag=decomp.OPPOSITE[f]))
                recvs.append((ghost, recv))
            for r in reqs:
                r.wait()
            for ghost, recv in recvs:
                X[ghost] = recv.numpy()

        # Dirichlet data from GLOBAL indices (what the GPU driver does: bit-identical faces on every decomposition);
        # built here by applying the oracle's bc_k to the global cube and slicing out this brick
        Pg = k.alloc(gsz)
        k.bc_k(gsz, Pg, pitch, np.zeros(3, dtype=R), [-1] * 6)
        (ni, nj, nk), (hi, hj, hk) = size, head
        own = (slice(2, 2 + nj), slice(2, 2 + ni), slice(2, 2 + nk))
        glob = (slice(hj + 1, hj + 1 + nj), slice(hi + 1, hi + 1 + ni), slice(hk + 1, hk + 1 + nk))
        P[own] = Pg[glob]
        RHS[own] = Pg[glob]
        halo(P)
        halo(RHS)
        del origin
        npts = torch.tensor([float(idx[1] - idx[0] + 1) * (idx[3] - idx[2] + 1) * (idx[5] - idx[4] + 1)], dtype=torch.float64)
        dist.all_reduce(npts)
        res_normal = 1.0 / float(npts[0])
        ofst = decomp.rb_offset(head, idx, world)
        hist = []
        for _ in range(nit):
            w = np.zeros(1)
            if solver == "jacobi":
                k.jacobi(P, size, idx, cf, coef, RHS, WRK, wide=w)
                halo(P)
            elif solver == "pcr_rb":
                # line SOR (whole k-lines per brick): GLOBAL colouring = the brick's rule shifted by its head, exchange per colour
                if _ == 0:
                    MSK = k.alloc(size)
                    k.imask_k(MSK, size, idx)
                    pn = O.get_num_stage(idx[5] - idx[4] + 1)
                for color in (0, 1):
                    k.pcr_sweep_wide("pcr_rb_2x2", size, idx, pn, (color + head[0] + head[1]) & 1, P, MSK, RHS, None, coef, w)
                    halo(P)
            else:
                for color in (0, 1):
                    k.psor2sma_core(P, size, idx, cf, ofst, color, coef, RHS, wide=w)
                    halo(P)
            t = torch.from_numpy(w)
            dist.all_reduce(t)
            hist.append(float(np.sqrt(w[0] * res_normal)))
        q.put((rank, size, head, hist, P))
    finally:
        dist.destroy_process_group()


DECOMP_CASES = [
    ("jacobi", (20, 18, 22), (1, 2, 1), 12, 0.8, "f32"),
    ("jacobi", (21, 19, 23), (2, 1, 1), 9, 0.8, "f64"),
    ("sor2sma", (21, 18, 23), (1, 1, 2), 10, 1.5, "f32"),
    ("sor2sma", (21, 19, 23), (2, 2, 1), 8, 1.5, "f32"),
    ("pcr_rb", (21, 19, 18), (2, 2, 1), 6, 1.2, "f64"),
]


@pytest.mark.parametrize("case", DECOMP_CASES, ids=[f"{c[0]}_{'x'.join(map(str, c[2]))}_{c[5]}" for c in DECOMP_CASES])
def test_gloo_decomposed_equals_single_domain(case):
    solver, gsz, div, nit, coef, prec = case
    world = div[0] * div[1] * div[2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, gsz, div, solver, nit, coef, prec, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)

    ref = O.run(gsz, solver, nit, coef, kind="oracle", prec=prec, wide=True)
    g = 2
    G = np.zeros_like(ref.P)
    for rank, size, head, hist, P in outs:
        (ni, nj, nk), (hi, hj, hk) = size, head
        G[g + hj - 1:g + hj - 1 + nj, g + hi - 1:g + hi - 1 + ni, g + hk - 1:g + hk - 1 + nk] = P[g:g + nj, g:g + ni, g:g + nk]
        assert np.allclose(hist, [r for _, r in ref.history], rtol=1e-12, atol=0)
    assert G[g:-g, g:-g, g:-g].tobytes() == ref.P[g:-g, g:-g, g:-g].tobytes()


def _rank_pairs(rank, world, port, gsz, div, solver, npairs, coef, prec, q, split=False):
    """the decomposed FUSED-PAIR algorithm of the GPU driver with the oracle doing the arithmetic: two ghost layers +
    edges exchanged once per pair (or per red-black iteration), first sweep / first colour also applied to ghost layer 1."""
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k = O.Kernels("oracle", prec)
        R = k.real
        b = decomp.decompose(gsz, div, world, rank)
        size, head, nID = b["size"], b["head"], b["nID"]
        idx = decomp.inner_range(size, nID)
        idx1 = decomp.first_sweep_range(idx, nID)
        pitch = R(1.0 / float(R(gsz[2] - 1)))
        cf = np.array([1, 1, 1, 1, 1, 1, 6], dtype=R)
        P, RHS, WRK = k.alloc(size), k.alloc(size), k.alloc(size)
        msgs = decomp.exchange_boxes(size, div, rank, depth=2, edges=True)

        def halo2_start(X):
            reqs, recvs = [], []
            for tag, m in enumerate(msgs):
                send = torch.from_numpy(np.ascontiguousarray(X[m["send"]]))
                recv = torch.empty_like(send)
                d = m["dir"]
                stag = (d[0] + 1) + 3 * (d[1] + 1) + 9 * (d[2] + 1)
                rtag = (-d[0] + 1) + 3 * (-d[1] + 1) + 9 * (-d[2] + 1)
                reqs.append(dist.isend(send, dst=m["peer"], tag=stag))
                reqs.append(dist.irecv(recv, src=m["peer"], tag=rtag))
                recvs.append((m["recv"], recv))
            return reqs, recvs

        def halo2_finish(X, pending):
            reqs, recvs = pending
            for r in reqs:
                r.wait()
            for sl, recv in recvs:
                X[sl] = recv.numpy()

        def halo2(X):
            halo2_finish(X, halo2_start(X))

        def sweep2(X, box, box1):
            """the fused pass on one index box (in place on X)"""
            if solver == "jacobi":
                k.jacobi(X, size, box1, cf, coef, RHS, WRK)   # sweep n+1 incl. ghost layer 1 (redundant with the neighbour)
                k.jacobi(X, size, box, cf, coef, RHS, WRK)    # sweep n+2 on the owned inner box
            else:
                # same global colouring whatever the box's kst is
                k.psor2sma_core(X, size, box1, cf, (ofst + idx[4] + box1[4]) % 2, 0, coef, RHS)
                k.psor2sma_core(X, size, box, cf, (ofst + idx[4] + box[4]) % 2, 1, coef, RHS)

        def view(box):
            ist, ied, jst, jed, kst, ked = box
            return (slice(jst + 1, jed + 2), slice(ist + 1, ied + 2), slice(kst + 1, ked + 2))

        Pg = k.alloc(gsz)
        k.bc_k(gsz, Pg, pitch, np.zeros(3, dtype=R), [-1] * 6)
        (ni, nj, nk), (hi, hj, hk) = size, head
        own = (slice(2, 2 + nj), slice(2, 2 + ni), slice(2, 2 + nk))
        glob = (slice(hj + 1, hj + 1 + nj), slice(hi + 1, hi + 1 + ni), slice(hk + 1, hk + 1 + nk))
        P[own] = Pg[glob]
        RHS[own] = Pg[glob]
        halo2(P), halo2(RHS)
        ofst = decomp.rb_offset(head, idx, world)
        shell, interior, interior1 = decomp.pair_plan(idx, nID) if split else ([], None, None)
        for _ in range(npairs):
            if not shell:
                sweep2(P, idx, idx1)
                halo2(P)
                continue
            # the overlapped form (CZ::pair_overlapped): shell slabs, exchange started, interior, exchange finished
            Pn = P.copy()
            Pn[view(idx)] = np.nan
            for box in shell:
                T = P.copy()
                sweep2(T, box, decomp.sub_first_sweep_range(box, idx1))
                Pn[view(box)] = T[view(box)]
            for m in msgs:
                assert not np.isnan(Pn[m["send"]]).any(), "a cell to be sent is not in the shell"
            pending = halo2_start(Pn)
            T = P.copy()
            sweep2(T, interior, interior1)
            Pn[view(interior)] = T[view(interior)]
            halo2_finish(Pn, pending)
            assert not np.isnan(Pn[view(idx)]).any(), "shell + interior do not cover the inner box"
            P[...] = Pn
        q.put((rank, size, head, P))
    finally:
        dist.destroy_process_group()


PAIR_CASES = [("jacobi", (20, 18, 22), (1, 2, 1), 5, 0.8, "f32"), ("jacobi", (21, 19, 23), (2, 2, 1), 4, 0.8, "f32"),
              ("sor2sma", (21, 19, 23), (2, 1, 2), 6, 1.5, "f32"), ("jacobi", (12, 13, 14), (2, 2, 2), 3, 0.8, "f64"),
              ("sor2sma", (24, 12, 13), (3, 1, 1), 4, 1.5, "f64")]


@pytest.mark.parametrize("split", [False, True], ids=["unsplit", "shell_first"])
@pytest.mark.parametrize("case", PAIR_CASES, ids=[f"{c[0]}_{'x'.join(map(str, c[2]))}_{c[5]}" for c in PAIR_CASES])
def test_gloo_fused_pairs_with_two_layer_exchange(case, split):
    solver, gsz, div, npairs, coef, prec = case
    world = div[0] * div[1] * div[2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_pairs, args=(r, world, port, gsz, div, solver, npairs, coef, prec, q, split)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    nsweeps = 2 * npairs if solver == "jacobi" else npairs
    ref = O.run(gsz, solver, nsweeps, coef, kind="oracle", prec=prec)
    g = 2
    G = np.zeros_like(ref.P)
    for rank, size, head, P in outs:
        (ni, nj, nk), (hi, hj, hk) = size, head
        G[g + hj - 1:g + hj - 1 + nj, g + hi - 1:g + hi - 1 + ni, g + hk - 1:g + hk - 1 + nk] = P[g:g + nj, g:g + ni, g:g + nk]
    assert G[g:-g, g:-g, g:-g].tobytes() == ref.P[g:-g, g:-g, g:-g].tobytes()
