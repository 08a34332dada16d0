"""GPU-side cost of the exchange kernels (pack / unpack of the two-layer pattern) and of the shell / interior launches of a decomposed
512^3-per-rank run, measured on ONE GPU with two ranks as host threads (LOCAL transport): run under
`rocprofv3 --kernel-trace -- python3 tools/comm_kernels_cost.py <di> <dj> <dk>` and read box_copy_k / pair_shell_k / jacobi2_k."""
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubez_amd import CZ, load  # noqa: E402

div = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [1, 1, 2]
n = 512
lib = load("f32")
lib.cz_comm_local_world.restype = C.c_void_p
lib.cz_comm_bootstrap_local.argtypes = [C.c_void_p, C.c_int]
nr = div[0] * div[1] * div[2]
world = lib.cz_comm_local_world(nr)


def work(r):
    lib.cz_comm_bootstrap_local(world, r)
    cz = CZ("f32", quiet=True)
    assert cz.setup([n * div[0], n * div[1], n * div[2], "jacobi", 12, 0.8] + div) == 1
    cz.solve()
    cz.close()


th = [threading.Thread(target=work, args=(r,)) for r in range(nr)]
[t.start() for t in th]
[t.join() for t in th]
print("done", div)
