#!/bin/bash
# where the two-stage pass loses to single sweeps: small grids, long k-lines (FP64), MAF; default against CZHIP_T2=0
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_aa; rm -rf $O; mkdir -p $O
run() {  # prec n solver
  for t2 in 1 0; do
    CZHIP_T2=$t2 timeout -k 10 200 python3 bench.py --cells $2 --solver $3 --prec $1 --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; return 1; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%5s^3 $1 %-10s CZHIP_T2=$t2 %9.0f MLUPS  %.4f ms/step  %s' % ('$2', '$3', d['value'], d['ms_per_step'], d['roofline']['kernel'][:30]))" | tee -a $O/times.txt
  done
}
for n in 64 96 128 160 192 256; do for s in jacobi sor2sma jacobi_maf; do run f32 $n $s || exit 1; done; done
for n in 64 128 192; do for s in jacobi sor2sma; do run f64 $n $s || exit 1; done; done
for n in 640 768 896 1000; do run f64 $n jacobi || exit 1; done
for n in 768 1024; do run f32 $n jacobi || exit 1; done
