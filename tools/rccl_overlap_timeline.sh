#!/bin/bash
# Kernel timeline of a decomposed run with REAL RCCL ranks on ONE GPU (ranks = processes, each under rocprofv3 --kernel-trace):
#   tools/rccl_overlap_timeline.sh <tag> <CZ_COMM_CUS> <prec> <di> <dj> <dk> [n]
# The ranks take each other for single-GPU nodes (NCCL_HOSTID) and talk over RCCL's socket transport on the loopback interface
# (tests/test_gpu_rccl.py); both sweep an n^3 brick, so the two interiors share the GPU -- what the timeline shows is whether RCCL's
# kernels, the packs and the shell slabs run INSIDE an interior sweep or wait for its end.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; K=$2; PREC=$3; DI=$4; DJ=$5; DK=$6; N=${7:-512}
W=$((DI * DJ * DK))
O=gpurun_out/rccl_tl_$TAG
rm -rf $O; mkdir -p $O/x
ARGV="[$((N * DI)), $((N * DJ)), $((N * DK)), \"jacobi\", 200, 0.8, $DI, $DJ, $DK]"
pids=()
for r in $(seq 0 $((W - 1))); do
  NCCL_HOSTID=cz-one-gpu-rank-$r NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 CZ_COMM_DEBUG=1 CZ_COMM_TIMEOUT=120 \
  CZ_COMM_CUS=$K CZ_WORKER_SWEEPS=40 OMP_NUM_THREADS=1 \
    timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r$r -- python3 tests/rccl_rank_worker.py $r $W $O/x $PREC "$ARGV" > $O/rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
if [ $rc -ne 0 ]; then tail -20 $O/rank0.log; exit 1; fi
dbs=$(for r in $(seq 0 $((W - 1))); do ls $O/r$r/*/*.db | head -1; done)
python3 tools/rccl_overlap.py $dbs > $O/overlap.txt 2>&1 || { cat $O/overlap.txt; exit 1; }
grep -h "comm_cus" $O/rank0.log | head -2 >> $O/overlap.txt
python3 - "$O" <<'PY' >> $O/overlap.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/x/rank_*.json")):
    d = json.load(open(f))
    n, ms = d["pair_ms"]
    print(f"rank {d['rank']}: {d['itr']} sweeps in {d['wall_s'] * 1e3:.1f} ms wall ({d['wall_s'] * 1e3 / d['itr']:.3f} ms per sweep), interior by HIP events {ms / max(n, 1):.3f} ms per pass, info {d['info']}")
PY
cat $O/overlap.txt
rm -rf $O/x $O/r[0-9]*   # (the traces are tens of MB: gpurun merges at most 64 MiB back)
