"""GPU tests (-m gpu) of the RCCL transport with MORE THAN ONE RANK on the one GPU a test box has.

RCCL refuses two ranks on one device of one host ("Duplicate GPU detected").  The ranks here are separate processes that each
announce a host id of their own (NCCL_HOSTID), so RCCL takes them for single-GPU nodes and connects them through its socket transport
over the loopback interface: the very calls the multi-GPU run makes -- ncclCommInitRank from a broadcast id, one grouped
ncclSend/ncclRecv per exchange (faces sent from and received into the array itself, packed rows, packed k pairs, the 12 edges),
ncclAllReduce of the residual sums on the exchange stream -- run between distinct ranks, with RCCL's own kernels on the GPU.  What this
cannot show is xGMI bandwidth.  Contract as in test_gpu_decomp.py: decomposed == single domain, bit for bit."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "rccl_rank_worker.py")


def rank_env(r, extra=None):
    env = dict(os.environ)
    env.update({"NCCL_HOSTID": f"cz-one-gpu-rank-{r}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NCCL_DEBUG": env.get("NCCL_DEBUG", "WARN"),
                "HSA_ENABLE_IPC_MODE_LEGACY": "0", "CZ_COMM_DEBUG": "1", "CZ_COMM_TIMEOUT": "90", "OMP_NUM_THREADS": "1"})
    env.update(extra or {})
    return env


def run_ranks(prec, gsz, solver, itmax, coef, div, pc=None, extra_env=None, timeout=240):
    world = div[0] * div[1] * div[2]
    argv = list(gsz) + [solver, itmax, coef] + ([pc] if pc else []) + list(div)
    with tempfile.TemporaryDirectory(prefix="cz_rccl_") as out:
        procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), out, prec, json.dumps(argv)], env=rank_env(r, extra_env),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs, codes = [], []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q in procs:  # exactly the processes started above
                    if q.poll() is None:
                        q.kill()
                o, _ = p.communicate()
                o = (o or "") + "\n[killed: time-out]"
            logs.append(o)
            codes.append(p.returncode)
        assert all(c == 0 for c in codes), "\n".join(f"--- rank {r} (exit {c})\n{lg[-3000:]}" for r, (c, lg) in enumerate(zip(codes, logs)))
        recs = [json.load(open(os.path.join(out, f"rank_{r}.json"))) for r in range(world)]
        fields = [np.load(os.path.join(out, f"field_{r}.npy")) for r in range(world)]
    g = 2
    G = np.zeros((gsz[1] + 4, gsz[0] + 4, gsz[2] + 4), dtype=fields[0].dtype)
    for rec, P in zip(recs, fields):
        (ni, nj, nk), (hi, hj, hk) = rec["local"]["size"], rec["local"]["head"]
        G[g + hj - 1:g + hj - 1 + nj, g + hi - 1:g + hi - 1 + ni, g + hk - 1:g + hk - 1 + nk] = P[g:g + nj, g:g + ni, g:g + nk]
    return recs, G, logs


def single(prec, gsz, solver, itmax, coef, pc=None):
    from cubez_amd import CZ
    cz = CZ(prec, quiet=True)
    assert cz.setup(list(gsz) + [solver, itmax, coef] + ([pc] if pc else [])) == 1
    itr = cz.solve()
    out = (itr, cz.res, cz.history(), cz.field())
    cz.close()
    return out


CASES = [
    ("f32", (40, 36, 44), "jacobi", 25, 0.8, (1, 2, 1)),     # J faces: sent from / received into the array itself
    ("f32", (40, 36, 44), "jacobi", 25, 0.8, (2, 1, 1)),     # I faces: packed k-rows
    ("f32", (40, 36, 44), "jacobi", 24, 0.8, (1, 1, 2)),     # K faces: packed pairs
    ("f64", (36, 40, 44), "jacobi", 20, 0.9, (2, 2, 1)),     # four ranks: faces + edges to the diagonal neighbour
    ("f32", (40, 36, 44), "sor2sma", 20, 1.5, (1, 2, 2)),    # red-black iteration per pass, global colouring
    ("f32", (41, 37, 45), "jacobi", 12, 0.8, (2, 1, 2)),     # odd sizes: one-layer exchange, single sweeps
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[2]}_{c[0]}_{'x'.join(map(str, c[5]))}" for c in CASES])
def test_rccl_ranks_on_one_gpu_equal_single_domain(case):
    prec, gsz, solver, itmax, coef, div = case
    itr1, res1, hist1, P1 = single(prec, gsz, solver, itmax, coef)
    recs, G, logs = run_ranks(prec, gsz, solver, itmax, coef, div)
    world = div[0] * div[1] * div[2]
    assert G[2:-2, 2:-2, 2:-2].tobytes() == P1[2:-2, 2:-2, 2:-2].tobytes()
    for rec in recs:
        assert rec["info"]["rccl_ranks"] == world and rec["info"]["ranks"] == world, rec["info"]  # the RCCL communicator, not the LOCAL transport
        assert rec["itr"] == itr1
        assert np.allclose(rec["history"], hist1, rtol=1e-12, atol=0)
    vw = 4 if prec == "f32" else 2
    if all((rec["local"]["size"][2] + 4) % vw == 0 for rec in recs):  # aligned bricks: fused passes, overlapped two-layer exchange
        npass = itmax // 2 if solver == "jacobi" else itmax
        assert all(rec["fused_pairs"] == npass and rec["shell_launches"] == npass and rec["info"]["lagged_reduce"] == 1 for rec in recs), \
            [(r["fused_pairs"], r["shell_launches"], r["info"]) for r in recs]


@pytest.mark.parametrize("solver,coef,lag", [("jacobi", 0.85, 1), ("jacobi", 0.85, 0), ("sor2sma", 1.5, 1)])
def test_rccl_ranks_converge_at_the_same_iteration(solver, coef, lag):
    """To convergence: the residual all-reduce and the test run one pass behind on the exchange stream (lag 1), every rank stops issuing
    passes at the same one, count / history / field equal the single-domain run."""
    prec, gsz = "f64", (20, 16, 24)
    itr1, res1, hist1, P1 = single(prec, gsz, solver, 100000, coef)
    recs, G, logs = run_ranks(prec, gsz, solver, 100000, coef, (2, 2, 1), extra_env={"CZ_LAG_REDUCE": str(lag)})
    assert all(rec["itr"] == itr1 for rec in recs), (itr1, [rec["itr"] for rec in recs])
    assert all(rec["info"]["lagged_reduce"] == lag and rec["info"]["rccl_ranks"] == 4 for rec in recs)
    assert np.allclose(recs[0]["history"], hist1, rtol=1e-12, atol=0)
    assert G[2:-2, 2:-2, 2:-2].tobytes() == P1[2:-2, 2:-2, 2:-2].tobytes()


def test_rccl_ranks_bicgstab():
    prec, gsz = "f64", (32, 36, 40)
    itr1, res1, hist1, P1 = single(prec, gsz, "pbicgstab", 200, 0.8, "jacobi")
    recs, G, logs = run_ranks(prec, gsz, "pbicgstab", 200, 0.8, (2, 1, 2), pc="jacobi")
    assert all(rec["itr"] == itr1 and rec["info"]["rccl_ranks"] == 4 for rec in recs)
    assert np.allclose(recs[0]["history"], hist1, rtol=1e-6, atol=0)
    assert np.abs(G[2:-2, 2:-2, 2:-2] - P1[2:-2, 2:-2, 2:-2]).max() < 1e-9


@pytest.mark.parametrize("gpus,extra", [(2, []), (4, []), (2, ["--solver", "sor2sma"])], ids=["2_ranks", "4_ranks", "2_ranks_rbsor"])
def test_bench_line_of_a_multi_rank_run(gpus, extra):
    """The command the driver runs for N > 1 -- `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- rehearsed with
    the N ranks on the one GPU of the box (CZ_BENCH_ONE_GPU=1: every rank on device 0, one RCCL host id per rank): rendezvous, id broadcast,
    ncclCommInitRank, the timed loop with its barriers and max over ranks, the rccl_ranks check, exposed_ms_per_step, ONE JSON line from
    rank 0.  The numbers mean nothing (the ranks share the GPU); that the line exists and what it says about the path does."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update({"CZ_BENCH_ONE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "1", "CZ_COMM_TIMEOUT": "90"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", str(gpus), "--steps", "8", "--warmup", "2", "--repeats", "2", "--settle", "0", "--cells", "96"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 8 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "rehearsal" in d
    assert d["config"]["division"] == {2: [1, 2, 1], 4: [2, 2, 1]}[gpus]
    mg = d["multi_gpu"]
    assert mg["rccl_ranks"] == gpus and [x["rccl_ranks"] for x in mg["ranks"]] == [gpus] * gpus
    assert mg["fused_pass"] and mg["overlap"] and mg["shell_slabs_rank0"] > 0 and mg["comm_cus_per_xcd"] == 2
    assert mg["exposed_ms_per_step"] is not None and d["roofline"]["kernel_launches_timed"] > 0
