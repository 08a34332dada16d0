"""One rank of a multi-PROCESS run of the restated driver over the RCCL transport (tests/test_gpu_rccl.py, tools/rccl_overlap_timeline.sh).

    python rccl_rank_worker.py <rank> <world> <outdir> <prec> <json list: driver argv incl. the division>

The communicator id travels through <outdir>/id (rank 0 writes it under a temporary name and renames it; the others poll), as in
cz_main.cpp.  The rank writes <outdir>/field_<rank>.npy and <outdir>/rank_<rank>.json (iterations, residual, history, local box, what the
driver decided).  The environment that lets several ranks share one GPU (NCCL_HOSTID per rank, NCCL_SOCKET_IFNAME=lo) is set by the parent
before this process starts: RCCL reads it when the library is loaded."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

rank, world, outdir, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
argv = json.loads(sys.argv[5])
sweeps = int(os.environ.get("CZ_WORKER_SWEEPS", "0"))  # > 0: unchecked sweeps after set-up instead of a solve (timeline runs)

from cubez_amd import CZ, load  # noqa: E402

lib = load(prec)
assert lib.czhip_init(int(os.environ.get("CZ_WORKER_DEVICE", "0"))) == 0
nb = lib.cz_comm_unique_id_bytes()
buf = C.create_string_buffer(nb)
idf = os.path.join(outdir, "id")
if rank == 0:
    lib.cz_comm_get_unique_id(buf)
    with open(idf + ".tmp", "wb") as f:
        f.write(buf.raw)
    os.rename(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        if time.time() - t0 > 120:
            sys.stderr.write(f"rank {rank}: no communicator id after 120 s\n")
            sys.exit(4)
        time.sleep(0.05)
    buf.raw = open(idf, "rb").read()
lib.cz_comm_bootstrap(rank, world, buf.raw)  # returns when every rank has joined

cz = CZ(prec, quiet=True, device=int(os.environ.get("CZ_WORKER_DEVICE", "0")))
assert cz.setup(argv) == 1, "cz_setup failed"
cz.timing(True)
t0 = time.time()
if sweeps > 0:
    cz.sweeps(sweeps)
    itr = sweeps
else:
    itr = cz.solve()
lib.czhip_sync()
wall = time.time() - t0
loc = cz.local()
rec = dict(rank=rank, itr=itr, res=cz.res, history=cz.history(), local=loc, info=cz.info(), wall_s=wall,
           fused_pairs=cz.timing_read("jacobi2")[0] + cz.timing_read("rbsor2")[0], shell_launches=cz.timing_read("pair_shell")[0],
           pair_ms=cz.timing_read("jacobi2"), rb_ms=cz.timing_read("rbsor2"))
cz.timing(False)
np.save(os.path.join(outdir, f"field_{rank}.npy"), cz.field())
with open(os.path.join(outdir, f"rank_{rank}.json"), "w") as f:
    json.dump(rec, f)
cz.close()
lib.cz_comm_shutdown()
lib.czhip_finalize()
