"""Host-side decomposition logic shared by the launcher and the CPU (gloo) tests.

``auto_division`` / ``decompose`` call the C functions of cz_comm.cpp (pure host code, no GPU needed);
``inner_range``, ``rb_offset`` and ``face_slices`` mirror, in numpy terms, what CZ::range_inner_index, CZ::RBSOR and
the pack/unpack kernels of cz_comm.cpp do, so that the exchange pattern can be exercised end to end on CPUs with
torch.distributed/gloo (tests/test_decomp_gloo.py) -- one rank per brick, exactly as the RCCL path runs one rank per GPU.
Replaces CBrick's SubDomain/BrickComm of the reference (cz_Evaluate.cpp:103-159, cz_comm.cpp:23-38).
"""
from __future__ import annotations

import ctypes as C

from .lib import GUIDE, load

I_MINUS, I_PLUS, J_MINUS, J_PLUS, K_MINUS, K_PLUS = range(6)  # cz_fparam.fi:10-16
OPPOSITE = [1, 0, 3, 2, 5, 4]


def auto_division(nproc: int, gsz, prec: str = "f32"):
    lib = load(prec)
    g, d = (C.c_int * 3)(*gsz), (C.c_int * 3)()
    lib.cz_comm_auto_division(int(nproc), g, d)
    return list(d)


def decompose(gsz, div, nproc: int, rank: int, prec: str = "f32"):
    """-> dict(size, head, nID) of one rank, or None if the division is impossible."""
    lib = load(prec)
    g, d = (C.c_int * 3)(*gsz), (C.c_int * 3)(*div)
    size, head, nid = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 6)()
    if not lib.cz_comm_decompose(g, d, int(nproc), int(rank), size, head, nid):
        return None
    return dict(size=list(size), head=list(head), nID=list(nid))


def inner_range(size, nID):
    """CZ::range_inner_index of this build: physical faces exclude the Dirichlet layer, rank-internal faces do not."""
    idx = []
    for a, (lo, hi) in enumerate(((I_MINUS, I_PLUS), (J_MINUS, J_PLUS), (K_MINUS, K_PLUS))):
        idx += [2 if nID[lo] < 0 else 1, size[a] - 1 if nID[hi] < 0 else size[a]]
    return idx


def rb_offset(head, idx, nproc):
    """colour offset passed to psor2sma_core so that colour 0 = even GLOBAL i+j+k (CZ::RBSOR)."""
    return 0 if nproc == 1 else (head[0] + head[1] + head[2] + 1 + idx[4]) % 2


def face_slices(size):
    """face -> (owned boundary layer, ghost layer) as numpy index tuples into the padded [j, i, k] array.
    Only owned cells travel (no edges/corners), one layer (Comm_S(X, 1))."""
    g = GUIDE
    ni, nj, nk = size
    J, I_, K = slice(g, g + nj), slice(g, g + ni), slice(g, g + nk)
    return {
        I_MINUS: ((J, g, K), (J, g - 1, K)),
        I_PLUS: ((J, g + ni - 1, K), (J, g + ni, K)),
        J_MINUS: ((g, I_, K), (g - 1, I_, K)),
        J_PLUS: ((g + nj - 1, I_, K), (g + nj, I_, K)),
        K_MINUS: ((J, I_, g), (J, I_, g - 1)),
        K_PLUS: ((J, I_, g + nk - 1), (J, I_, g + nk)),
    }


def exchange_boxes(size, div, rank, depth=1, edges=False):
    """Mirror of build_pattern() in cz_comm.cpp: the messages of one single-phase exchange of one brick.
    -> list of dict(peer, dir, send, recv): numpy index tuples into the padded [j, i, k] array.
    depth-1 faces = Comm_S(X, 1); depth-2 faces + the 12 one-cell edges = what a fused pair of sweeps reads."""
    g = GUIDE
    coord = [rank % div[0], (rank // div[0]) % div[1], rank // (div[0] * div[1])]
    out = []
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                d = (di, dj, dk)
                nz = sum(1 for v in d if v)
                if nz == 0 or nz == 3 or (nz == 2 and not edges):
                    continue
                rc = [coord[a] + d[a] for a in range(3)]
                if any(rc[a] < 0 or rc[a] >= div[a] for a in range(3)):
                    continue
                dep = depth if nz == 1 else 1
                send, recv = [], []
                for a in range(3):
                    n = size[a]
                    if d[a] == 0:
                        s0, r0, ext = 1, 1, n
                    elif d[a] < 0:
                        s0, r0, ext = 1, 1 - dep, dep
                    else:
                        s0, r0, ext = n - dep + 1, n + 1, dep
                    send.append(slice(s0 + g - 1, s0 + g - 1 + ext))
                    recv.append(slice(r0 + g - 1, r0 + g - 1 + ext))
                # (i, j, k) order -> array order [j, i, k]
                out.append(dict(peer=rc[0] + div[0] * (rc[1] + div[1] * rc[2]), dir=d,
                                send=(send[1], send[0], send[2]), recv=(recv[1], recv[0], recv[2])))
    return out


def first_sweep_range(idx, nID):
    """index range of the FIRST sweep of a fused pair: one layer into the ghost cells across rank-internal faces
    (CZ::JACOBI / RBSOR, idx1)."""
    return [idx[f] + ((1 if f & 1 else -1) if nID[f] >= 0 else 0) for f in range(6)]


def pair_plan(idx, nID):
    """Mirror of pair_plan() in cz_kernels.hip: split of the inner box of a brick for overlapped exchanges.
    -> (shell boxes, interior, interior1): the cells within two layers of a rank-internal face as up to six disjoint
    (ist,ied,jst,jed,kst,ked) slabs (J faces, then I, then K), the rest of the inner box and its first-sweep range.
    ([], None, None) when there is no internal face or the brick is too thin."""
    interior = list(idx)
    for a in range(3):
        interior[2 * a] = idx[2 * a] + (2 if nID[2 * a] >= 0 else 0)
        interior[2 * a + 1] = idx[2 * a + 1] - (2 if nID[2 * a + 1] >= 0 else 0)
        if interior[2 * a + 1] - interior[2 * a] + 1 < 2:
            return [], None, None
    O, I = idx, interior
    boxes = []
    if nID[2] >= 0:
        boxes.append([O[0], O[1], O[2], O[2] + 1, O[4], O[5]])
    if nID[3] >= 0:
        boxes.append([O[0], O[1], O[3] - 1, O[3], O[4], O[5]])
    if nID[0] >= 0:
        boxes.append([O[0], O[0] + 1, I[2], I[3], O[4], O[5]])
    if nID[1] >= 0:
        boxes.append([O[1] - 1, O[1], I[2], I[3], O[4], O[5]])
    if nID[4] >= 0:
        boxes.append([I[0], I[1], I[2], I[3], O[4], O[4] + 1])
    if nID[5] >= 0:
        boxes.append([I[0], I[1], I[2], I[3], O[5] - 1, O[5]])
    if not boxes:
        return [], None, None
    return boxes, interior, first_sweep_range(interior, nID)


def sub_first_sweep_range(box, idx1):
    """first-sweep range of a sub-box of the brick: the box grown by one layer, inside the brick's own first-sweep range."""
    return [max(box[f] - 1, idx1[f]) if f % 2 == 0 else min(box[f] + 1, idx1[f]) for f in range(6)]
