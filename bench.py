#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CubeZ hot path on MI355X (contract: see the task statement / DESIGN.md 6).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

(--solver sor2sma / pbicgstab time the other BASELINE configs the same way: step = one RB-SOR iteration / one BiCGSTAB
iteration incl. its 2 x 8 preconditioner sweeps; pbicgstab defaults to --prec f64 as in configs[3].)
A "step" is one relaxed-Jacobi sweep of the FP32 cube with everything the reference's checked loop does per
iteration (sweep, residual reduction, normalise + history + eps test; cz_Poisson.cpp:39-79), inputs resident in HBM.
N=1: BASELINE.json configs[1], `cz 512 512 512 jacobi K 0.8`.  N>1: weak scaling, 512^3 cells per GPU
(2: 1x2x1, 4: 2x2x1, 8: 2x2x2 = configs[4], the 1024^3 cube), halo exchange + residual all-reduce over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL; must be set before HIP starts

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--n", type=int, default=512, help="cells per GPU and axis")
ap.add_argument("--solver", default="jacobi", choices=["jacobi", "sor2sma", "pbicgstab", "pcr_rb", "psor", "pcr", "pcr_eda", "pcr_esa", "pcr_rb_esa", "pcr_j_esa", "jacobi_maf",
                                                     "sor2sma_maf", "psor_maf", "pcr_rb_maf", "pcr_maf"])
ap.add_argument("--precond", default="jacobi", choices=["none", "jacobi", "sor2sma"])
ap.add_argument("--prec", default="f32", choices=["f32", "f64"])
ap.add_argument("--div", default=None, help="Cartesian division of the ranks, e.g. 1,8,1 (default: 1x2x1, 2x2x1, 2x2x2 for 2, 4, 8 GPUs)")
ap.add_argument("--no-cpu-baseline", action="store_true")
ap.add_argument("--cpu-seconds", type=float, default=12.0)
args = ap.parse_args()
if args.solver == "pbicgstab" and "--prec" not in " ".join(sys.argv):
    args.prec = "f64"

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
if world != args.gpus and world > 1:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
if args.gpus > 1 and world == 1:
    raise SystemExit("launch multi-GPU runs with python -m torch.distributed.run --nproc-per-node N (one rank per GPU)")

import torch  # noqa: E402  (plumbing only: rendezvous, barrier, max-reduce of the timings)
import torch.distributed as dist  # noqa: E402

from cubez_amd import CZ  # noqa: E402

DIVS = {1: (1, 1, 1), 2: (1, 2, 1), 4: (2, 2, 1), 8: (2, 2, 2)}
if args.div:
    div = tuple(int(v) for v in args.div.replace("x", ",").split(","))
    if len(div) != 3 or div[0] * div[1] * div[2] != world:
        raise SystemExit(f"--div {args.div} does not multiply to {world} ranks")
elif world in DIVS:
    div = DIVS[world]
else:
    raise SystemExit("supported GPU counts without --div: 1, 2, 4, 8")
n = args.n
gsz = [n * div[0], n * div[1], n * div[2]]
coef = 0.9 if args.solver == "pcr_j_esa" else 1.2 if (args.solver.startswith("pcr") or args.solver.startswith("psor")) else 1.5 if (args.solver.startswith("sor2sma") or (args.solver == "pbicgstab" and args.precond == "sor2sma")) else 0.8

if torch.cuda.is_available():
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
cz = CZ(args.prec, quiet=True, device=local_rank)
lib = cz.lib
if world > 1:
    os.environ.setdefault("CZ_COMM_DEBUG", "1")  # one diagnostic line per rank on stderr (stdout carries the JSON line only)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    nb = lib.cz_comm_unique_id_bytes()
    buf = C.create_string_buffer(nb)
    if rank == 0:
        lib.cz_comm_get_unique_id(buf)
    t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    lib.cz_comm_bootstrap(rank, world, bytes(t.numpy().tobytes()))

bicg = args.solver == "pbicgstab"
argv = gsz + [args.solver, (args.warmup + 1) if bicg else (args.steps + args.warmup), coef] + ([args.precond] if bicg else [])
if world > 1:
    argv += list(div)
assert cz.setup(argv) == 1, "cz_setup failed"
loc = cz.local()
inner = loc["inner"]
my_points = (inner[1] - inner[0] + 1) * (inner[3] - inner[2] + 1) * (inner[5] - inner[4] + 1)


def barrier():
    lib.czhip_sync()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()


if bicg:
    # the Krylov loop has no "continue" entry point: warm up with one solve of W iterations, time a second one of K
    cz.solve()
    cz.close()
    cz = CZ(args.prec, quiet=True, device=local_rank)
    argv[4] = args.steps + 1
    assert cz.setup(argv) == 1
    barrier()
    cz.timing(True)
    t0 = time.perf_counter()
    cz.solve()
    barrier()
    dt = time.perf_counter() - t0
    assert len(cz.history()) == args.steps, "BiCGSTAB converged before the requested number of iterations"
else:
    cz.sweeps(args.warmup)
    barrier()
    cz.timing(True)
    t0 = time.perf_counter()
    cz.sweeps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
_jl = args.solver in ("jacobi", "jacobi_maf") or (args.solver == "pbicgstab" and args.precond == "jacobi")
_line = args.solver.startswith("pcr")
nk, kern_ms = cz.timing_read("jacobi" if _jl else "pcr_rb" if _line else "psor" if args.solver.startswith("psor") else "rbsor")
nk2, kern2_ms = cz.timing_read("jacobi2" if _jl else "rbsor2")  # fused: 2 sweeps / both colours per launch
cz_shell = cz.timing_read("pair_shell")
cz.timing(False)

tot_points = float(my_points)
if world > 1:
    tt = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt[0])
    tp = torch.tensor([tot_points], dtype=torch.float64)
    dist.all_reduce(tp, op=dist.ReduceOp.SUM)
    tot_points = float(tp[0])

if rank == 0:
    word = 4 if args.prec == "f32" else 8
    # algorithmic bytes per lattice update: Jacobi reads p and b once, writes p' once = 3 words;
    # one RB-SOR colour launch updates half the points of the box: 4 words per point and iteration = 2 per launch
    jac_like = args.solver in ("jacobi", "jacobi_maf") or (bicg and args.precond == "jacobi")
    alg_bytes_per_launch = my_points * word * (3 if jac_like else 2)
    kernel_name = "stencil_k<jacobi>" if jac_like else "stencil_k<rbsor colour>"
    tkey = f"{'jacobi' if jac_like else 'sor2sma'}_{n}_{args.prec}"
    if _line:
        # one iteration: every line reads x, rhs, msk and writes x (4 words), and is read once more as i/j neighbour (1 word);
        # pcr_rb / pcr_rb_esa: two colour launches per iteration, pcr / pcr_esa: one launch per (i+j) diagonal, pcr_j_esa: one
        per_iter = {"pcr_rb": 2, "pcr_rb_esa": 2, "pcr_rb_maf": 2, "pcr_j_esa": 1}.get(args.solver, nk // max(args.steps, 1))
        alg_bytes_per_launch = my_points * word * 5 // max(per_iter, 1)
        kernel_name, tkey = f"pcr_rb2_k ({per_iter} launches of k-line solves per iteration)", f"{args.solver}_{n}_{args.prec}"
    if args.solver.startswith("psor"):
        # one sweep (all tile-hyperplane launches together): p read and written in place, b read: 3 words per point
        alg_bytes_per_launch = my_points * word * 3
        kernel_name, tkey = "psor_tile_k (one lexicographic sweep = 3N/16-2 tile-hyperplane launches)", f"psor_{n}_{args.prec}"
    if nk2 > nk:  # the dominant kernel is the fused one: 2 Jacobi sweeps (2 x 3 words) or both RB colours (2 x 2 words)
        nk, kern_ms, alg_bytes_per_launch = nk2, kern2_ms, 2 * alg_bytes_per_launch
        if jac_like:
            kernel_name, tkey = "jacobi2_k<RB=0> (two fused sweeps per launch)", f"jacobi2_{n}_{args.prec}"
        else:
            kernel_name, tkey = "jacobi2_k<RB=1> (both colours of one iteration per launch)", f"rbsor2_{n}_{args.prec}"
    kern_avg_s = (kern_ms / nk) * 1e-3 if nk else float("nan")
    achieved = alg_bytes_per_launch / kern_avg_s / 1e9 if nk else None
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            rec = json.load(open(tfile))
            if tkey in rec:
                traffic = rec[tkey]["bytes_per_launch"]
        except Exception:
            traffic = None
    out = {
        "metric": "MLUPS (lattice updates/s), 512^3 FP32 Jacobi per GPU" if args.solver == "jacobi" and args.prec == "f32" and n == 512
        else (f"BiCGSTAB iterations/s, {n}^3 {args.prec}, preconditioner {args.precond}" if bicg else
              f"MLUPS (lattice updates/s), {n}^3 {args.prec} {args.solver} per GPU"),
        "value": (args.steps / dt) if bicg else tot_points * args.steps / dt / 1e6,
        "unit": "iterations/s" if bicg else "MLUPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.prec,
        "data": "synthetic (the reference problem: P=0, Dirichlet sin(pi x)sin(pi y) on z faces, RHS=0)",
        "config": {"workload": f"cz {gsz[0]} {gsz[1]} {gsz[2]} {args.solver} {args.steps} {coef}" + (f" {args.precond}" if bicg else "")
                   + (f" {div[0]} {div[1]} {div[2]}" if world > 1 else ""),
                   "cells_per_gpu": f"{n}^3", "division": list(div), "global_grid": gsz,
                   "step": "one sweep + residual reduction + convergence bookkeeping (cz_Poisson.cpp:39-79)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic,
                     "kernel": kernel_name,
                     "kernel_avg_ms": kern_avg_s * 1e3, "kernel_launches_timed": nk,
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch},
    }
    if world > 1 and nk2 > 0 and not bicg:
        # exposed (non-overlapped) communication per step = wall time per step minus the rank-0 kernel time per step
        # (shell slabs + interior of a fused pass cover two steps); the exchange itself runs on a second stream behind the interior
        try:
            n_sh, sh_ms = cz_shell
            per_step_kernel_ms = (kern2_ms + sh_ms) / nk2 / (2.0 if jac_like else 1.0)
            out["multi_gpu"] = {"kernel_ms_per_step_rank0": per_step_kernel_ms, "exposed_ms_per_step": dt / args.steps * 1e3 - per_step_kernel_ms,
                                "per_gpu_algorithmic_GBps": achieved, "overlap": os.environ.get("CZ_OVERLAP", "1") != "0",
                                "lagged_reduce": os.environ.get("CZ_LAG_REDUCE", "1") != "0" and args.solver == "jacobi"}
        except Exception as e:  # reporting only
            out["multi_gpu"] = {"error": repr(e)}
    if bicg:
        out["config"]["step"] = "one BiCGSTAB iteration: 2 x 8 preconditioner sweeps, 2 SpMV, 5 dots, 4 axpy-type updates (cz_Poisson.cpp:373-500)"
        # 76 words per point and iteration with the Jacobi preconditioner
        out["roofline"]["iteration_algorithmic_GBps"] = my_points * word * (76 if args.precond == "jacobi" else 92) * args.steps / dt / 1e9
    if world == 1 and not args.no_cpu_baseline and not bicg and args.solver in ("jacobi", "sor2sma"):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--n", str(n), "--solver", args.solver,
                                "--prec", args.prec, "--seconds", str(args.cpu_seconds)], capture_output=True, text=True, timeout=600)
            out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is reporting only; never lose the GPU line over it
            out["cpu_baseline"] = {"value": None, "unit": "MLUPS", "cores": None, "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(out))

cz.close()
if world > 1:
    lib.cz_comm_shutdown()
    dist.destroy_process_group()
