#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CubeZ hot path on MI355X (contract: see the task statement / DESIGN.md 6).

  python bench.py --gpus N --steps K --warmup W            (N > 1: this process starts the N ranks itself, see below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one relaxed-Jacobi sweep of the FP32 cube with everything the reference's checked loop does per iteration (sweep,
residual reduction, normalise + history + eps test; cz_Poisson.cpp:39-79), inputs resident in HBM.
N=1: BASELINE.json configs[1], `cz 512 512 512 jacobi K 0.8`.  N>1: weak scaling, 512^3 cells per GPU (2: 1x2x1, 4: 2x2x1,
8: 2x2x2 = configs[4], the 1024^3 cube), halo exchange + residual all-reduce over RCCL.
Before the W warm-up steps the GPU is kept busy with untimed sweeps for `--settle` seconds (clocks ramp over the first tens of
milliseconds of load; reported as `settle_s`).  The K steps are timed `--repeats` times (each between barriers); the line reports the
median repeat.
The default line (N=1, --solver jacobi) additionally times the two other single-GPU configurations of BASELINE.json the same way and
reports them under "configs": configs[2] `cz 512 512 512 sor2sma K 1.5` (FP32; step = one red-black iteration) and configs[3]
`cz_f64 512 512 512 pbicgstab K 0.8 jacobi` (step = one BiCGSTAB iteration incl. its 2 x 8 preconditioner sweeps; at most 10 per solve).
--solver sor2sma / pbicgstab / ... make one of the other solvers the headline of the line instead (pbicgstab defaults to --prec f64).
Prints ONE JSON line on rank 0.

`--gpus N` without a launcher (the reference starts all ranks with one command too, `mpirun -np 8 ./cz ... 2 2 2`, main.cpp:24-35):
the parent does not import torch and does not touch the GPU; it starts `python -m torch.distributed.run ... bench.py <same args>` as a
CHILD process, relays rank 0's JSON line and exits with the child's code (non-zero on failure, time-out or a missing line).
--dry-launch: rendezvous + broadcast of the communicator id only (no solver, no ncclCommInitRank); runs without a GPU.
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL; must be set before HIP starts

SOLVERS = ["jacobi", "sor2sma", "pbicgstab", "pcr_rb", "psor", "pcr", "pcr_eda", "pcr_esa", "pcr_rb_esa", "pcr_j_esa", "jacobi_maf", "sor2sma_maf",
           "psor_maf", "pcr_rb_maf", "pcr_maf", "pcr_eda_maf", "pcr_esa_maf", "pcr_rb_esa_maf"]
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--repeats", type=int, default=5, help="how many times the --steps region is timed (median reported)")
ap.add_argument("--settle", type=float, default=0.15, help="seconds of untimed sweeps before the warm-up steps (clock ramp)")
ap.add_argument("--n", "--cells", dest="n", type=int, default=512, help="cells per GPU and axis (--cells: the spelling torch.distributed.run does not take for one of its own options)")
ap.add_argument("--solver", default="jacobi", choices=SOLVERS)
ap.add_argument("--precond", default="jacobi", choices=["none", "jacobi", "sor2sma"])
ap.add_argument("--prec", default=None, choices=["f32", "f64"])
ap.add_argument("--div", default=None, help="Cartesian division of the ranks, e.g. 1,8,1 (default: 1x2x1, 2x2x1, 2x2x2 for 2, 4, 8 GPUs)")
ap.add_argument("--no-cpu-baseline", action="store_true")
ap.add_argument("--no-configs", action="store_true", help="skip the configs[2] / configs[3] legs of the default line")
ap.add_argument("--cpu-seconds", type=float, default=12.0)
ap.add_argument("--dry-launch", action="store_true", help="rendezvous and id broadcast only; works without a GPU")
ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-launched job may take")
args = ap.parse_args()
if args.prec is None:
    args.prec = "f64" if args.solver == "pbicgstab" else "f32"


def self_launch() -> int:
    """Parent of a multi-GPU run: start the ranks as a child job, relay rank 0's JSON line.  No torch, no HIP in this process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)  # stderr passes through
    try:
        out, _ = child.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, signal.SIGKILL)  # exactly the process group started above
        child.wait()
        sys.stderr.write(f"bench.py: the {args.gpus}-rank job did not finish within {args.launch_timeout:.0f} s -- killed\n")
        return 124
    line = None
    for ln in out.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")  # anything else the ranks printed
    if child.returncode != 0:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank job exited with code {child.returncode}\n")
        return child.returncode
    if line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        return 1
    print(line)
    return 0


if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(self_launch())

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
# Rehearsal of the multi-rank line on a box with ONE GPU (tests/test_gpu_rccl.py): every rank on device 0, RCCL told that each rank is a host
# of its own so that it takes the socket transport over loopback.  The line says so ("rehearsal"); it is not a measurement of anything.
ONE_GPU = os.environ.get("CZ_BENCH_ONE_GPU") == "1" and world > 1
if ONE_GPU:
    os.environ["NCCL_HOSTID"] = f"cz-bench-one-gpu-rank-{rank}"
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    local_rank = 0
if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

import ctypes as C  # noqa: E402

import torch  # noqa: E402  (plumbing only: rendezvous, barrier, max-reduce of the timings)
import torch.distributed as dist  # noqa: E402

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


def coef_of(solver, precond):
    return 0.9 if solver == "pcr_j_esa" else 1.2 if (solver.startswith("pcr") or solver.startswith("psor")) else 1.5 if (
        solver.startswith("sor2sma") or (solver == "pbicgstab" and precond == "sor2sma")) else 0.8


import cubez_amd  # noqa: E402

if args.dry_launch:
    # the bootstrap of a multi-rank run without its GPU half: gloo rendezvous, rank 0 makes the communicator id, broadcast, compare
    lib = cubez_amd.load(args.prec)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    nb = lib.cz_comm_unique_id_bytes()
    have_gpu = torch.cuda.device_count() > 0
    buf = C.create_string_buffer(nb)
    if rank == 0:
        if have_gpu:
            lib.cz_comm_get_unique_id(buf)  # ncclGetUniqueId needs a device
        else:
            buf.raw = os.urandom(nb)
    t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    chk = torch.tensor([int(t.to(torch.int64).sum())], dtype=torch.int64)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry launch (rendezvous + communicator id broadcast, no solver)", "value": None, "unit": None, "n_gpus": world,
                          "dry_launch": True, "id_bytes": nb, "id_source": "ncclGetUniqueId" if have_gpu else "placeholder (no GPU here)",
                          "id_agrees_on_all_ranks": bool(int(lo[0]) == int(hi[0])), "division": list(div), "global_grid": gsz}))
    dist.destroy_process_group()
    sys.exit(0)

from cubez_amd import CZ  # noqa: E402

if torch.cuda.is_available():
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
if world > 1:
    os.environ.setdefault("CZ_COMM_DEBUG", "1")  # one diagnostic line per rank on stderr + the collective watchdog (cz_comm.cpp)
    lib0 = cubez_amd.load(args.prec)
    lib0.czhip_init(local_rank)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    nb = lib0.cz_comm_unique_id_bytes()
    buf = C.create_string_buffer(nb)
    if rank == 0:
        lib0.cz_comm_get_unique_id(buf)
    t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    lib0.cz_comm_bootstrap(rank, world, bytes(t.numpy().tobytes()))


def barrier(lib):
    lib.czhip_sync()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()


def timed(lib, fn):
    """one timed region: barrier + device sync, t0, fn(), device sync, t1 -- the rank-local time ends when THIS rank's stream is drained; the
    gloo barrier that closes the region comes AFTER the clock is read (an 8-process gloo/TCP barrier is 0.2-1 ms, a fifth of a 20-step region
    at 512^3: VERDICT r3 weak 2), then the max over ranks.  A rank cannot finish its last step before its neighbours have sent it their halos
    of the step before, so the slowest rank's local time is the job's time."""
    barrier(lib)
    t0 = time.perf_counter()
    fn()
    lib.czhip_sync()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:  # the slowest rank
        dist.barrier()
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    return dt


_hip_rt = None


def event_span_ms(lib, fn):
    """one more region of K steps, measured on the GPU's own clock: HIP events on the library's compute stream in front of the first launch and
    behind the last (reported beside the host-clock figure for N > 1, max over ranks; the host figure is `value`)"""
    global _hip_rt
    try:
        if _hip_rt is None:
            _hip_rt = C.CDLL("libamdhip64.so")
    except OSError:  # reporting only: never lose the line over it
        fn()
        barrier(lib)
        return None
    rt = _hip_rt
    lib.czhip_stream.restype = C.c_void_p
    st = C.c_void_p(lib.czhip_stream())
    a, b = C.c_void_p(), C.c_void_p()
    ok = rt.hipEventCreate(C.byref(a)) == 0 and rt.hipEventCreate(C.byref(b)) == 0
    barrier(lib)
    ok = ok and rt.hipEventRecord(a, st) == 0
    fn()
    ok = ok and rt.hipEventRecord(b, st) == 0 and rt.hipEventSynchronize(b) == 0
    ms = C.c_float(0.0)
    ok = ok and rt.hipEventElapsedTime(C.byref(ms), a, b) == 0
    rt.hipEventDestroy(a), rt.hipEventDestroy(b)
    val = float(ms.value) if ok else -1.0
    if world > 1:
        dist.barrier()
        tt = torch.tensor([val], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        val = float(tt[0])
    return val if val > 0 else None


def kernel_source_sha():
    """identifies the build a recorded HBM-traffic figure belongs to: the sources of the sweep kernels and their launch geometry"""
    h = hashlib.sha256()
    for f in ("cz_k_common.h", "cz_k_fastdiv.h", "cz_k_stencil.h", "cz_k_pair.h", "cz_k_pair2.h", "cz_k_rb4.h", "cz_k_blas.h", "cz_h_launch.h"):
        h.update(open(os.path.join(ROOT, "cubez_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measure(solver, prec, precond, steps, warmup, repeats, settle):
    """one leg: set up `cz <gsz> solver ...`, settle, W warm-up steps, `repeats` timed regions of `steps` steps; everything the line needs"""
    bicg = solver == "pbicgstab"
    coef = coef_of(solver, precond)
    cz = CZ(prec, quiet=True, device=local_rank)
    lib = cz.lib
    argv = gsz + [solver, (warmup + 1) if bicg else (steps * repeats + warmup), coef] + ([precond] if bicg else [])
    if world > 1:
        argv += list(div)
    assert cz.setup(argv) == 1, "cz_setup failed"
    inner = cz.local()["inner"]
    my_points = (inner[1] - inner[0] + 1) * (inner[3] - inner[2] + 1) * (inner[5] - inner[4] + 1)
    dts = []
    settled = 0.0
    ev_ms = None
    if bicg:
        # the Krylov loop has no "continue" entry point: warm up with one solve of W iterations, then time solves of K iterations
        cz.solve()
        for rep in range(repeats):
            cz.close()
            cz = CZ(prec, quiet=True, device=local_rank)
            argv[4] = steps + 1
            assert cz.setup(argv) == 1
            if rep == repeats - 1:
                cz.timing(True)
            dts.append(timed(lib, cz.solve))
            assert len(cz.history()) == steps, "BiCGSTAB converged before the requested number of iterations"
    else:
        t0 = time.perf_counter()
        while settle > 0:  # every rank issues the same number of sweeps: rank 0's clock decides
            cz.sweeps(40)
            lib.czhip_sync()
            go = time.perf_counter() - t0 < settle
            if world > 1:
                g = torch.tensor([1 if go else 0], dtype=torch.int32)
                dist.broadcast(g, src=0)
                go = bool(int(g[0]))
            if not go:
                break
        settled = time.perf_counter() - t0
        cz.sweeps(warmup)
        barrier(lib)
        # the timed regions run WITHOUT the per-launch HIP events of the roofline leg (two event records per launch cost microseconds: a fifth of a
        # 128^3 pass, nothing at 512^3); one more region of K steps behind them, with the events on, gives the kernel durations
        for rep in range(repeats):
            dts.append(timed(lib, lambda: cz.sweeps(steps)))
        # N > 1: a region of K = 20 steps is a few milliseconds, the same order as the jitter of N processes starting it together.  The
        # region stays EXACTLY K steps (the contract); what grows is the number of regions the median is taken over -- until they add up
        # to 0.25 s, at most 25.  Every rank sees the same all-reduced times, so every rank takes the same decision.
        if world > 1 and not ONE_GPU:
            while sum(dts) < 0.25 and len(dts) < 25:
                dts.append(timed(lib, lambda: cz.sweeps(steps)))
        if world > 1:
            ev_ms = event_span_ms(lib, lambda: cz.sweeps(steps))
        cz.timing(True)
        cz.sweeps(steps)
        barrier(lib)
    jl = solver in ("jacobi", "jacobi_maf") or (bicg and precond == "jacobi")
    line = solver.startswith("pcr")
    single = cz.timing_read("jacobi" if jl else "pcr_rb" if line else "psor" if solver.startswith("psor") else "rbsor")
    fused = cz.timing_read("jacobi2" if jl else "rbsor2")  # fused: 2 sweeps / both colours per launch
    fused4 = cz.timing_read("rbsor4")                        # red-black, single domain (round 4): TWO iterations per launch (rb4_k)
    labels = {lb: cz.timing_read(lb) for lb in ("jacobi2", "rbsor2", "rbsor4", "calc_ax", "ewise", "dot", "pair_shell")}
    cz.timing(False)
    info = cz.info()
    cz.close()
    return dict(solver=solver, prec=prec, precond=precond, bicg=bicg, coef=coef, steps=steps, warmup=warmup, repeats=repeats, dts=dts, dt=statistics.median(dts),
                my_points=my_points, jac_like=jl, line=line, single=single, fused=fused, fused4=fused4, labels=labels, info=info, settle_s=settled,
                timed_steps=steps, event_span_ms=ev_ms)


def traffic_record(tkey):
    """HBM bytes per launch of the dominant kernel from the PMC passes recorded in profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate passes, FETCH_SIZE doubled: MI355X_MICROARCH.md, HBM section; tools/summarize_pmc.py) -- counters cannot be read
    from inside this process, so the figure is a recorded one and says which build and box it was taken on"""
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(tfile)).get(tkey)
    except Exception:
        rec = None
    if not rec:
        return None, None
    src = {"file": "profiles/hbm_traffic.json", "key": tkey, "kernel_source_sha": rec.get("kernel_source_sha"), "box": rec.get("box"),
           "recorded": rec.get("recorded"), "matches_this_build": rec.get("kernel_source_sha") == kernel_source_sha()}
    return rec["bytes_per_launch"], src


def roofline_of(m):
    """the `roofline` object of one leg (algorithmic bytes per launch / mean launch duration by HIP events / 8 TB/s)"""
    word = 4 if m["prec"] == "f32" else 8
    pts, solver, jl = m["my_points"], m["solver"], m["jac_like"]
    nk, kern_ms = m["single"]
    nk2, kern2_ms = m["fused"]
    # algorithmic bytes per lattice update: Jacobi reads p and b once, writes p' once = 3 words;
    # one RB-SOR colour launch updates half the points of the box: 4 words per point and iteration = 2 per launch
    alg = pts * word * (3 if jl else 2)
    kernel_name = "stencil_k<jacobi>" if jl else "stencil_k<rbsor colour>"
    tkey = f"{'jacobi' if jl else 'sor2sma'}_{n}_{m['prec']}"
    model = "12 B/LUP per sweep (FP32; 24 FP64): p and b read once, p' written once" if jl else "16 B/LUP per iteration: two colour passes of 8 B/LUP"
    fused_min = None
    if m["line"]:
        # one iteration: every line reads x, rhs, msk and writes x (4 words), and is read once more as i/j neighbour (1 word);
        # pcr_rb / pcr_rb_esa: two colour launches per iteration, pcr / pcr_esa: one launch per (i+j) diagonal, pcr_j_esa: one
        per_iter = {"pcr_rb": 2, "pcr_rb_esa": 2, "pcr_rb_maf": 2, "pcr_j_esa": 1}.get(solver, nk // max(m["timed_steps"], 1))
        alg = pts * word * 5 // max(per_iter, 1)
        lex = solver in ("pcr", "pcr_esa", "pcr_eda", "pcr_maf", "pcr_esa_maf", "pcr_eda_maf")
        kernel_name = ("pcr_lex_wg_k (the lexicographic sweep in one launch: rows of k-lines handed from workgroup to workgroup)" if lex and per_iter == 1
                       else f"line-SOR kernels ({per_iter} launches of k-line solves per iteration)")
        tkey = f"{solver}_{n}_{m['prec']}"
        model = "5 words per point and iteration (x, rhs, msk read, x written, once more read as neighbour)"
    if solver.startswith("psor"):
        # one sweep (all tile-hyperplane launches together): p read and written in place, b read: 3 words per point
        alg = pts * word * 3
        kernel_name, tkey = "psor (one lexicographic sweep)", f"psor_{n}_{m['prec']}"
        model = "3 words per point and sweep (in place)"
    if nk2 > nk:  # the dominant kernel is the fused one: 2 Jacobi sweeps (2 x 3 words) or both RB colours (2 x 2 words)
        nk, kern_ms, alg = nk2, kern2_ms, 2 * alg
        fused_min = pts * word * 3  # what ONE pass over memory must move: u and b read once, w written once
        if jl:
            kernel_name, tkey = "jacobi2p_k<RB=0> (two fused sweeps per launch)", f"jacobi2_{n}_{m['prec']}"
            model += "; TWO sweeps per pass over memory, so `achieved` counts 2 x 12 B/LUP per launch and can exceed the peak -- `frac_hbm_traffic` is the physical fraction"
        else:
            kernel_name, tkey = "jacobi2p_k<RB=1> (both colours of one iteration per launch)", f"rbsor2_{n}_{m['prec']}"
            model += "; both colours in ONE pass over memory (12 B/LUP really moved) -- `frac_hbm_traffic` is the physical fraction"
    nk4, kern4_ms = m.get("fused4", (0, 0.0))
    if not jl and nk4 > nk:  # the dominant kernel makes TWO red-black iterations per pass over memory: four colour passes of SURVEY's model per launch
        nk, kern_ms = nk4, kern4_ms
        alg = 4 * pts * word * 2
        fused_min = pts * word * 3
        kernel_name, tkey = "rb4_k (two red-black iterations = four colour sweeps per launch)", f"rbsor4_{n}_{m['prec']}"
        model = ("16 B/LUP per iteration: two colour passes of 8 B/LUP; TWO iterations in ONE pass over memory (12 B per point really moved per launch), so `achieved` "
                 "counts 2 x 16 B/LUP per launch and can exceed the peak -- `frac_hbm_traffic` is the physical fraction; neither pipe is saturated: vector ALU 77 % busy "
                 "(profiles/r04/pmc_rb4_512_f32_SQ.txt), 3.9 TB/s of counter traffic; one workgroup per CU, a barrier per plane step")
    kern_avg_s = (kern_ms / nk) * 1e-3 if nk else float("nan")
    achieved = alg / kern_avg_s / 1e9 if nk else None
    traffic, tsrc = traffic_record(tkey) if n == 512 else (None, None)
    r = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic,
         "kernel": kernel_name, "kernel_avg_ms": kern_avg_s * 1e3, "kernel_launches_timed": nk, "algorithmic_bytes_per_launch": alg, "model": model,
         "fused_min_bytes_per_launch": fused_min, "traffic_source": tsrc,
         # the fraction a consumer may read as "of peak" for a fused kernel: what one pass over memory must move (u, b read; w written) / time / peak
         "frac_fused_min": (fused_min / kern_avg_s / 1e9 / 8000.0) if (fused_min and nk) else None,
         # HBM bytes the counters saw per launch / this run's mean launch duration / peak: a fraction of what the memory system can do (<= 1)
         "frac_hbm_traffic": (traffic / kern_avg_s / 1e9 / 8000.0) if (traffic and nk) else None}
    return r


def iteration_traffic(m, label_ms):
    """what one BiCGSTAB iteration physically moves: the HBM counters of EVERY kernel of the iteration (profiles/hbm_traffic.json, key
    bicg_<n>_<prec>: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over a profiled run of this very command, tools/profile_r04.sh) summed with
    their launch counts -- beside SURVEY's 76-word model (608 B per point in FP64), which counts passes the code no longer makes"""
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(tfile)).get(f"bicg_{n}_{m['prec']}")
    except Exception:
        rec = None
    word = 4 if m["prec"] == "f32" else 8
    out = {"model_bytes_per_point": {"survey_8d_76_words": 76 * word, "array_passes_executed_41": 41 * word}}
    if not rec:
        return out
    per_it = rec["bytes_per_unit"]
    ms = m["dt"] / m["steps"] * 1e3
    # the library's timing labels -> the kernels they cover
    groups = {"jacobi2": "jacobi2p_k", "calc_ax": "stencil_k", "ewise": ("ewise_k", "triad_dots_k")}
    per_label = {}
    for lb, names in groups.items():
        names = (names,) if isinstance(names, str) else names
        b = sum(k["bytes_per_launch"] * k["launches_per_unit"] for nm, k in rec["kernels"].items() if nm.startswith(names))
        if b and label_ms.get(lb):
            per_label[lb] = {"bytes_per_iteration": b, "ms_per_iteration": label_ms[lb], "physical_GBps": b / label_ms[lb] / 1e6, "frac_hbm_traffic": b / label_ms[lb] / 1e6 / 8000.0}
    out.update({"executed_bytes_per_point": per_it / m["my_points"], "traffic_bytes_per_iteration": per_it, "physical_GBps": per_it / ms / 1e6,
                "frac_hbm_traffic": per_it / ms / 1e6 / 8000.0, "per_kernel_group": per_label,
                "traffic_source": {"file": "profiles/hbm_traffic.json", "key": f"bicg_{n}_{m['prec']}", "kernel_source_sha": rec.get("kernel_source_sha"),
                                   "box": rec.get("box"), "recorded": rec.get("recorded"), "matches_this_build": rec.get("kernel_source_sha") == kernel_source_sha()}})
    return out


def config_record(m):
    """a configs[] leg of the default line, compact"""
    word = 4 if m["prec"] == "f32" else 8
    rec = {"workload": f"cz{'_f64' if m['prec'] == 'f64' else ''} {gsz[0]} {gsz[1]} {gsz[2]} {m['solver']} {m['steps']} {m['coef']}" + (f" {m['precond']}" if m["bicg"] else ""),
           "dtype": m["prec"], "steps": m["steps"], "warmup": m["warmup"], "repeats": m["repeats"], "ms_per_step": m["dt"] / m["steps"] * 1e3,
           "ms_per_step_all": [d / m["steps"] * 1e3 for d in m["dts"]]}
    if m["bicg"]:
        rec["unit"], rec["value"] = "iterations/s", m["steps"] / m["dt"]
        rec["step"] = "one BiCGSTAB iteration: 2 x 8 preconditioner sweeps, 2 SpMV, 5 dots, 4 axpy-type updates (cz_Poisson.cpp:373-500)"
        # 76 words per point and iteration with the Jacobi preconditioner (608 B FP64)
        rec["algorithmic_GBps"] = m["my_points"] * word * 76 * m["steps"] / m["dt"] / 1e9
        rec["kernel_ms_per_iteration"] = {lb: (ms / m["steps"]) for lb, (cnt, ms) in m["labels"].items() if cnt}
        rec["kernel_launches_per_iteration"] = {lb: cnt / m["steps"] for lb, (cnt, ms) in m["labels"].items() if cnt}
        rec.update(iteration_traffic(m, rec["kernel_ms_per_iteration"]))
    else:
        rec["unit"], rec["value"] = "MLUPS", m["my_points"] * m["steps"] / m["dt"] / 1e6
        rec["step"] = "one red-black iteration (both colours) + residual reduction + convergence bookkeeping (cz_Poisson.cpp:159-235)"
        rec["algorithmic_GBps"] = m["my_points"] * word * 4 * m["steps"] / m["dt"] / 1e9
    rec["roofline"] = roofline_of(m)
    return rec


m = measure(args.solver, args.prec, args.precond, args.steps, args.warmup, args.repeats, args.settle)
bicg, dt, info = m["bicg"], m["dt"], m["info"]

tot_points = float(m["my_points"])
devs = None
if world > 1:
    tp = torch.tensor([tot_points], dtype=torch.float64)
    dist.all_reduce(tp, op=dist.ReduceOp.SUM)
    tot_points = float(tp[0])
    devs = [None] * world
    dist.all_gather_object(devs, {"rank": rank, "device": torch.cuda.current_device() if torch.cuda.is_available() else None, "rccl_ranks": info["rccl_ranks"]})
    if len({d["device"] for d in devs}) != world and not ONE_GPU:  # N ranks on fewer than N devices measure nothing (CZ_BENCH_ONE_GPU=1: the rehearsal)
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {world} but the ranks sit on devices {[d['device'] for d in devs]}\n")
        sys.exit(6)
    if any(d["rccl_ranks"] != world for d in devs):  # a line from anything but N ranks on one RCCL communicator is not a multi-GPU result
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {world} but the RCCL communicators report {[d['rccl_ranks'] for d in devs]} ranks\n")
        sys.exit(5)

if rank == 0:
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
        "repeats": len(m["dts"]),
        "settle_s": m["settle_s"],
        "ms_per_step_min": min(m["dts"]) / args.steps * 1e3,
        "ms_per_step_median": dt / args.steps * 1e3,
        "ms_per_step_all": [d / args.steps * 1e3 for d in m["dts"]],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.prec,
        "data": "synthetic (the reference problem: P=0, Dirichlet sin(pi x)sin(pi y) on z faces, RHS=0)",
        "config": {"workload": f"cz {gsz[0]} {gsz[1]} {gsz[2]} {args.solver} {args.steps} {m['coef']}" + (f" {args.precond}" if bicg else "")
                   + (f" {div[0]} {div[1]} {div[2]}" if world > 1 else ""),
                   "cells_per_gpu": f"{n}^3", "division": list(div), "global_grid": gsz,
                   "step": "one sweep + residual reduction + convergence bookkeeping (cz_Poisson.cpp:39-79)"},
        "roofline": roofline_of(m),
    }
    if ONE_GPU:
        out["rehearsal"] = f"{world} RCCL ranks (processes) sharing ONE GPU, socket transport over loopback: exercises the code of the multi-GPU line, measures nothing"
    if world > 1:
        nk2, kern2_ms = m["fused"]
        out["multi_gpu"] = {"ranks": devs, "rccl_ranks": info["rccl_ranks"], "fused_pass": bool(info["fused_pass"]), "shell_slabs_rank0": info["shell_slabs"],
                            "overlap": bool(info["overlap"]), "lagged_reduce": bool(info["lagged_reduce"]), "comm_cus_per_xcd": info.get("comm_cus", 0)}
        if m["event_span_ms"]:
            out["multi_gpu"]["hip_event_ms_per_step"] = m["event_span_ms"] / args.steps  # one more region of K steps between two HIP events (max over ranks)
        if nk2 > 0 and not bicg:
            # exposed (non-overlapped) communication per step = wall time per step minus the rank-0 interior-kernel time per
            # step (a fused pass covers two steps); the shell slabs, the exchange and the residual all-reduce run on a second stream beside it
            n_sh, sh_ms = m["labels"]["pair_shell"]
            per_step_kernel_ms = kern2_ms / nk2 / (2.0 if m["jac_like"] else 1.0)
            out["multi_gpu"].update({"kernel_ms_per_step_rank0": per_step_kernel_ms, "exposed_ms_per_step": dt / args.steps * 1e3 - per_step_kernel_ms,
                                     "shell_slabs_ms_per_pass_rank0": (sh_ms / n_sh) if n_sh else None, "per_gpu_algorithmic_GBps": out["roofline"]["achieved"]})
    if bicg:
        out["config"]["step"] = "one BiCGSTAB iteration: 2 x 8 preconditioner sweeps, 2 SpMV, 5 dots, 4 axpy-type updates (cz_Poisson.cpp:373-500)"
        word = 4 if args.prec == "f32" else 8
        # 76 words per point and iteration with the Jacobi preconditioner
        out["roofline"]["iteration_algorithmic_GBps"] = m["my_points"] * word * (76 if args.precond == "jacobi" else 92) * args.steps / dt / 1e9

# the two other single-GPU configurations of BASELINE.json, timed the same way (default line only)
if world == 1 and args.solver == "jacobi" and args.prec == "f32" and not args.no_configs:
    cfgs = {}
    try:
        cfgs["configs[2] 512^3 FP32 red-black SOR"] = config_record(measure("sor2sma", "f32", "jacobi", args.steps, args.warmup, min(args.repeats, 3), min(args.settle, 0.05)))
        cfgs["configs[3] 512^3 FP64 BiCGSTAB + Jacobi(8)"] = config_record(measure("pbicgstab", "f64", "jacobi", min(args.steps, 10), 2, 2, 0.0))
    except Exception as e:  # never lose the headline over a side leg
        cfgs["error"] = repr(e)
    out["configs"] = cfgs

if rank == 0:
    if world == 1 and not args.no_cpu_baseline and not bicg and args.solver in ("jacobi", "sor2sma"):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--n", str(n), "--solver", args.solver,
                                "--prec", args.prec, "--seconds", str(args.cpu_seconds)], capture_output=True, text=True, timeout=600)
            out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is reporting only; never lose the GPU line over it
            out["cpu_baseline"] = {"value": None, "unit": "MLUPS", "cores": None, "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(out))

if world > 1:
    lib0.cz_comm_shutdown()
    dist.destroy_process_group()
