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
